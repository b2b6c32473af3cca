"""In-kernel shader clock of the two dominant kernels on random data (ablation build, stamps around the loop; after >= 2 s of
back-to-back launches): clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups.
python scripts/inkernel_clock.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import unet_rir_amd as U
U.build.build_ablations()
U._lib.use_library(U.build.ABL_LIB)
U._lib.lib()
L = C.CDLL(U.build.ABL_LIB)
ops = U.ops
B, dev, dt = 32, "cuda:0", torch.bfloat16


def clock(fn, reader, nwg, what, flops, bits=0):
    L.unetrir_abl_set(2048 | bits)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.5:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    buf = (C.c_ulonglong * (2 * nwg))()
    assert reader(buf, nwg) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 2).astype(np.float64)
    a = a[a[:, 1] > 0]
    ghz = a[:, 0] / a[:, 1] * 0.1
    span_us = a[:, 1] / 100.0
    L.unetrir_abl_set(0)
    peak_at_clock = 256 * 4 * 1024 * np.median(ghz) * 1e9 / 1e12        # TFLOP/s of the bf16 matrix pipes at the held clock
    print(f"{what}: launch {us:.1f} us = {flops / us / 1e6:.0f} TFLOP/s; stamped span median {np.median(span_us):.1f} us; in-kernel clock median "
          f"{np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f}) -> bf16 matrix peak at that clock {peak_at_clock:.0f} TFLOP/s, "
          f"launch = {flops / us / 1e6 / peak_at_clock:.2f} of it", flush=True)


for Cc, N, HW in ((128, 128, 128), (256, 256, 64), (512, 512, 32)):
    x = ops.Act((torch.rand((B, HW, HW, Cc), device=dev) - 0.5).to(dt))
    dy = ops.Act((torch.rand((B, HW, HW, N), device=dev) - 0.5).to(dt))
    y = ops.Act(torch.empty((B, HW, HW, N), device=dev, dtype=dt))
    w = (torch.rand((N, 3, 3, Cc), device=dev) - 0.5).to(dt)
    bias = torch.zeros(N, device=dev)
    dw = torch.zeros((N, 9, Cc), device=dev)
    g = ops.geom(B, HW, HW, Cc, N, 3, 1)
    ws = ops.Workspace(dev)
    fl = 2.0 * B * HW * HW * Cc * N * 9
    if B * (HW // 16) * (HW // 32) * (N // 128) >= 512:      # else the layer runs on conv3x3g, which carries no stamps
        clock(lambda: ops.conv2d_fwd(g, x, w, bias, y), L.unetrir_abl_stamps_conv3x3p, 256, f"conv3x3p  {Cc}->{N} @{HW}", fl)
    clock(lambda: ops.conv2d_wgrad(g, x, dy, dw, ws), L.unetrir_abl_stamps_wgrad3x3g, 256, f"wgrad3x3g {Cc}->{N} @{HW} (incl. reduce launch)", fl)
    if "--fake16" in sys.argv:
        clock(lambda: ops.conv2d_wgrad(g, x, dy, dw, ws), L.unetrir_abl_stamps_wgrad3x3g, 256, f"wgrad3x3g {Cc}->{N} @{HW} FAKE 16x16x32 (wrong results)", fl, 4096)
