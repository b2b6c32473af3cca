"""Micro-benchmark of one 3x3 stride-1 conv weight gradient (bf16) through the C ABI: python scripts/micro_wgrad.py C N HW"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
ops = U.ops
C, N, HW = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B, dev, dt = 32, "cuda:0", torch.bfloat16
x = ops.Act((torch.rand((B, HW, HW, C), device=dev) - 0.5).to(dt))
dy = ops.Act((torch.rand((B, HW, HW, N), device=dev) - 0.5).to(dt))
dw = torch.zeros((N, 9, C), device=dev)
g = ops.geom(B, HW, HW, C, N, 3, 1)
ws = ops.Workspace(dev)
for _ in range(3): ops.conv2d_wgrad(g, x, dy, dw, ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 20
e0.record()
for _ in range(K): ops.conv2d_wgrad(g, x, dy, dw, ws)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
fl = 2.0 * B * HW * HW * C * N * 9
print(f"wgrad C={C} N={N} {HW}x{HW}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s (incl. split-K reduce)")
