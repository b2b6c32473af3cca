import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import unet_rir_amd as U
ops = U.ops
dev = "cuda:0"
torch.manual_seed(0)
B, H, W, C, N = 2, 32, 64, 64, 128
g = ops.geom(B, H, W, C, N, 3, 2)
w = (torch.rand(N, 9, C, device=dev) - 0.5).to(torch.bfloat16)
def ref(x, w):
    wt = w.float().view(N, 3, 3, C).permute(0, 3, 1, 2)
    xp = F.pad(x.float().permute(0, 3, 1, 2), (0, 1, 0, 1))
    return F.conv2d(xp, wt, stride=2).permute(0, 2, 3, 1)
for name, xoff, use_bias, use_add, ypad in [("plain", 0, 0, 0, 0), ("xoff", 8, 0, 0, 0), ("bias", 0, 1, 0, 0), ("add", 0, 0, 1, 0), ("ypad", 0, 0, 0, 8), ("all", 8, 1, 1, 8)]:
    xb = torch.full((B, H, W, C + xoff), 768.0, dtype=torch.bfloat16, device=dev)
    xb[..., xoff:] = ((torch.rand(B, H, W, C, device=dev) - 0.5) * 2).to(torch.bfloat16)
    xa = ops.Act(xb, xoff, C)
    bias = (torch.rand(N, device=dev) - 0.5) * 2 if use_bias else None
    add = ((torch.rand(B, H // 2, W // 2, N, device=dev) - 0.5) * 2).to(torch.bfloat16) if use_add else None
    yb = torch.full((B, H // 2, W // 2, N + ypad), 512.0, dtype=torch.bfloat16, device=dev)
    ya = ops.Act(yb, 0, N)
    ops.conv2d_fwd(g, xa, w, bias, ya, ops.Act(add) if use_add else None)
    torch.cuda.synchronize()
    r = ref(xb[..., xoff:], w)
    if use_bias: r = r + bias
    if use_add: r = r + add.float()
    d = (yb[..., :N].float() - r).abs()
    print(name, "max err", float(d.max()), "scale", float(r.abs().max()))
    if float(d.max()) > 0.02 * float(r.abs().max()):
        bad = d > 0.02 * float(r.abs().max())
        print("  bad frac", float(bad.float().mean()), "by img", bad.float().mean((1, 2, 3)).tolist())
        print("  by oy", [round(v, 2) for v in bad.float().mean((0, 2, 3)).tolist()])
        print("  by ox", [round(v, 2) for v in bad.float().mean((0, 1, 3)).tolist()])
        print("  by n ", [round(v, 2) for v in bad.float().mean((0, 1, 2)).tolist()])
