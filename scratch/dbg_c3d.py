import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import unet_rir_amd as U
ops = U.ops
dev = "cuda:0"
torch.manual_seed(0)
B, H, W, C, N = 2, 32, 64, 64, 128
x = ((torch.rand(B, H, W, C, device=dev) - 0.5) * 2).to(torch.bfloat16)
g = ops.geom(B, H, W, C, N, 3, 2)
def ref(w):   # w [N][9][C]
    wt = w.float().view(N, 3, 3, C).permute(0, 3, 1, 2)
    xp = F.pad(x.float().permute(0, 3, 1, 2), (0, 1, 0, 1))
    return F.conv2d(xp, wt, stride=2).permute(0, 2, 3, 1)
for tap in list(range(9)) + [-1]:
    w = torch.zeros(N, 9, C, device=dev)
    if tap >= 0:
        w[:, tap, :] = (torch.rand(N, C, device=dev) - 0.5)
    else:
        w = (torch.rand(N, 9, C, device=dev) - 0.5)
    w = w.to(torch.bfloat16)
    y = ops.Act(torch.zeros(B, H // 2, W // 2, N, dtype=torch.bfloat16, device=dev))
    ops.conv2d_fwd(g, ops.Act(x), w, None, y)
    torch.cuda.synchronize()
    r = ref(w)
    d = (y.base.float() - r).abs()
    print("tap", tap, "max err", float(d.max()), "scale", float(r.abs().max()))
    if float(d.max()) > 0.05 * float(r.abs().max()):
        bad = d > 0.05 * float(r.abs().max())
        print("  bad frac", float(bad.float().mean()), "by img", bad.float().mean((1, 2, 3)).tolist())
        print("  by oy", [round(v, 2) for v in bad.float().mean((0, 2, 3)).tolist()])
        print("  by ox", [round(v, 2) for v in bad.float().mean((0, 1, 3)).tolist()])
        print("  by n ", [round(v, 2) for v in bad.float().mean((0, 1, 2)).tolist()])
