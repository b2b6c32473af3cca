import os, sys
sys.path.insert(0, "/root/repo")
import torch
import unet_rir_amd as U
ops = U.ops
dev, dt, B = "cuda:0", torch.bfloat16, 32
def timed(fn, K=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3
for (C, N, HW, s) in ((32, 32, 128, 1), (64, 64, 64, 1), (128, 128, 32, 1), (256, 256, 16, 1), (32, 64, 128, 2), (8, 32, 256, 2)):
    g = ops.geom(B, HW, HW, C, N, 1, s)
    Ho = HW // s
    x = ops.Act((torch.rand((B, HW, HW, C), device=dev) - 0.5).to(dt))
    w = ((torch.rand((N, 1, C), device=dev) - 0.5) * 0.1).to(dt)
    wt = ((torch.rand((C, 1, N), device=dev) - 0.5) * 0.1).to(dt)
    bias = torch.zeros(N, device=dev)
    y = ops.Act(torch.empty((B, Ho, Ho, N), device=dev, dtype=dt))
    dx = ops.Act(torch.empty((B, HW, HW, C), device=dev, dtype=dt))
    rows = ops.conv2d_colstat_rows(g, 0, x)
    cst = torch.empty((rows, N, 2), device=dev)
    t_f = timed(lambda: ops.conv2d_fwd_colstat(g, x, w, bias, y, cst))
    t_d = timed(lambda: ops.conv2d_dgrad(g, y, wt, dx)) if C >= 32 else 0
    by = B * (HW * HW * C + Ho * Ho * N) * 2
    print(f"1x1 s{s} {C}->{N}@{HW}: fwd+stat {t_f:6.1f} us ({by / t_f / 1e6:5.2f} TB/s) rows {rows} | dgrad {t_d:6.1f} us", flush=True)
