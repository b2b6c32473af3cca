"""Timing ablations of one 3x3 layer on the ablation build (results of the ablated runs are invalid by construction):
python scripts/abl_conv.py C N HW "0,1,2,4,7" [fwd|dgrad] [stride]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
U.build.build_ablations()
U._lib.use_library(U.build.ABL_LIB)
import ctypes as C
L = U._lib.lib()
aset = C.CDLL(U.build.ABL_LIB).unetrir_abl_set
ops = U.ops
Cc, N, HW = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
modes = [int(v) for v in sys.argv[4].split(",")]
which = sys.argv[5] if len(sys.argv) > 5 else "fwd"
stride = int(sys.argv[6]) if len(sys.argv) > 6 else 1
B, dev, dt = 32, "cuda:0", torch.bfloat16
x = ops.Act((torch.rand((B, HW, HW, Cc), device=dev) - 0.5).to(dt))
w = ((torch.rand((N, 9, Cc), device=dev) - 0.5) * 0.1).to(dt)
wt = ((torch.rand((Cc, 9, N), device=dev) - 0.5) * 0.1).to(dt)
bias = torch.zeros(N, device=dev)
y = ops.Act(torch.empty((B, HW // stride, HW // stride, N), device=dev, dtype=dt))
dx = ops.Act(torch.empty((B, HW, HW, Cc), device=dev, dtype=dt))
g = ops.geom(B, HW, HW, Cc, N, 3, stride)
fn = (lambda: ops.conv2d_fwd(g, x, w, bias, y)) if which == "fwd" else (lambda: ops.conv2d_dgrad(g, y, wt, dx))
for rnd in range(2):
    for m in modes:
        aset(m)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"round {rnd} abl={m}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
aset(0)
