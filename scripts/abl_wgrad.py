"""wgrad3x3g timing ablations (ablation build; bits: 512 no XCD remap, 16 / 32 / 64 DMA placement 2-2-2-2-2 / 4-3-3 / all up front, 128 staggered quartets, 1 no DMA after the first patch, 2 no partial stores,
4 no MFMA): python scripts/abl_wgrad.py "128,128,128;256,256,64;128,64,256" [modes, default 0,512] """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
U.build.build_ablations()
U._lib.use_library(U.build.ABL_LIB)
import ctypes as C
U._lib.lib()
aset = C.CDLL(U.build.ABL_LIB).unetrir_abl_set
ops = U.ops
B, dev, dt = 32, "cuda:0", torch.bfloat16
for sh in sys.argv[1].split(";"):
    Cc, N, HW = (int(v) for v in sh.split(","))
    x = ops.Act((torch.rand((B, HW, HW, Cc), device=dev) - 0.5).to(dt))
    dy = ops.Act((torch.rand((B, HW, HW, N), device=dev) - 0.5).to(dt))
    dw = torch.zeros((N, 9, Cc), device=dev)
    g = ops.geom(B, HW, HW, Cc, N, 3, 1)
    ws = ops.Workspace(dev)
    for rnd in range(3):
        for m in ([int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 512)):
            aset(m)
            for _ in range(3): ops.conv2d_wgrad(g, x, dy, dw, ws)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.conv2d_wgrad(g, x, dy, dw, ws)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            if m == 0:
                ref = dw.clone()
            elif not (m & 7) and 'ref' in dir():      # variants that keep the arithmetic: bit-identical to the product kernel
                assert torch.equal(dw, ref), f"abl={m}: result differs from abl=0 (max {float((dw - ref).abs().max())})"
            print(f"C={Cc} N={N} {HW}: abl={m:3d} {ms*1e3:7.1f} us {2.0*B*HW*HW*Cc*N*9/ms/1e9:6.0f} TF/s", flush=True)
aset(0)
