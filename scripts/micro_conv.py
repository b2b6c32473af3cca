"""Micro-benchmark of one 3x3 stride-1 conv layer (bf16 or f32) through the C ABI: python scripts/micro_conv.py C N HW [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
ops = U.ops
C, N, HW = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = torch.bfloat16 if (len(sys.argv) < 5 or sys.argv[4] == "bf16") else torch.float32
B = 32
dev = "cuda:0"
x = ops.Act((torch.rand((B, HW, HW, C), device=dev) - 0.5).to(dt))
w = ((torch.rand((N, 9, C), device=dev) - 0.5) * 0.1).to(dt)
bias = torch.zeros(N, device=dev)
y = ops.Act(torch.empty((B, HW, HW, N), device=dev, dtype=dt))
g = ops.geom(B, HW, HW, C, N, 3, 1)
for _ in range(3): ops.conv2d_fwd(g, x, w, bias, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 20
e0.record()
for _ in range(K): ops.conv2d_fwd(g, x, w, bias, y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
fl = 2.0 * B * HW * HW * C * N * 9
print(f"C={C} N={N} {HW}x{HW} {sys.argv[4] if len(sys.argv)>4 else 'bf16'} nostore={os.environ.get('UNETRIR_CONV3X3_NOSTORE','0')} v2={os.environ.get('UNETRIR_CONV3X3','1')}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TFLOP/s")
