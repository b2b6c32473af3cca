"""Dense(8192 -> 4096) of the information-vector branch at batch 32: forward, data gradient through the transposed copy
(transpose + forward kernel) and the direct data gradient: python scripts/micro_dense.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
ops = U.ops
dev = "cuda:0"
B, K, N = 32, 8192, 4096
w = torch.randn((N, K), device=dev) * 0.01
wt = torch.empty((K, N), device=dev)
x = ops.Act(torch.randn((B, 1, 1, K), device=dev)); y = ops.Act(torch.empty((B, 1, 1, N), device=dev))
dy = ops.Act(torch.randn((B, 1, 1, N), device=dev)); dx = ops.Act(torch.empty((B, 1, 1, K), device=dev))
bias = torch.zeros(N, device=dev)
ws = ops.Workspace(dev)


def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


mb = N * K * 4 / 1e6
t = timeit(lambda: ops.dense_fwd(x, w, bias, y, ws)); print(f"forward                 {t:7.1f} us  {mb / t:.2f} TB/s of weight bytes")
t1 = timeit(lambda: ops.transpose_weight(w, wt, N, 1, K)); t2 = timeit(lambda: ops.dense_fwd(dy, wt, None, dx, ws))
print(f"dgrad via transposed    {t1:7.1f} + {t2:.1f} us")
ref = dx.base.clone()
t = timeit(lambda: ops.dense_dgrad(dy, w, dx, ws)); print(f"dgrad direct            {t:7.1f} us  {mb / t:.2f} TB/s of weight bytes")
print("max |direct - transposed| =", float((dx.base - ref).abs().max()), " scale", float(ref.abs().max()))
