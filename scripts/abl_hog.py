"""What a long-running kernel on a few CUs (a collective beside the backward pass) costs the 3x3 kernels (ablation build):
python scripts/abl_hog.py [n_hog_workgroups=32] [lds_bytes=32768]
A stand-in kernel holds n workgroups x lds bytes for ~50 ms on a side stream; the same layer is timed with the persistent kernel
(one 158 KB workgroup per CU) and with the per-tile kernel."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
U.build.build_ablations()
U._lib.use_library(U.build.ABL_LIB)
U._lib.lib()
abl = C.CDLL(U.build.ABL_LIB)
abl.unetrir_abl_hog.argtypes = [C.c_int, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]
ops = U.ops
n_hog = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lds = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dev, dt, B = "cuda:0", torch.bfloat16, 32
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()


def timed(fn, hog):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    if hog:
        abl.unetrir_abl_hog(n_hog, 100_000_000 * 5 // 10, lds, sink.data_ptr(), C.c_void_p(side.cuda_stream))     # ~0.5 s at 100 MHz wall clock
        import time; time.sleep(0.05)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    torch.cuda.synchronize()
    return ms * 1e3


for (Cc, N, HW) in ((128, 128, 128), (256, 256, 64)):
    x = ops.Act((torch.rand((B, HW, HW, Cc), device=dev) - 0.5).to(dt))
    w = ((torch.rand((N, 9, Cc), device=dev) - 0.5) * 0.1).to(dt)
    y = ops.Act(torch.empty((B, HW, HW, N), device=dev, dtype=dt))
    g = ops.geom(B, HW, HW, Cc, N, 3, 1)
    fn = lambda: ops.conv2d_fwd(g, x, w, None, y)
    for p_on in (1, 0):
        ops.set_config(conv3x3p=p_on)
        t0, t1 = timed(fn, False), timed(fn, True)
        print(f"{Cc}->{N}@{HW} {'persistent' if p_on else 'per-tile  '}: alone {t0:7.1f} us, beside {n_hog} x {lds} B workgroups {t1:7.1f} us ({t1/t0:.2f}x)", flush=True)


# the stride-2 pair: conv3x3d (down direction) and upconv3x3q (up direction) against the kernels they replaced
for (Cc, N, HW) in ((64, 128, 256), (128, 256, 128)):
    x = ops.Act((torch.rand((B, HW, HW, Cc), device=dev) - 0.5).to(dt))
    w = ((torch.rand((N, 9, Cc), device=dev) - 0.5) * 0.1).to(dt)
    wt = ((torch.rand((Cc, 9, N), device=dev) - 0.5) * 0.1).to(dt)
    y = ops.Act(torch.empty((B, HW // 2, HW // 2, N), device=dev, dtype=dt))
    g = ops.geom(B, HW, HW, Cc, N, 3, 2)
    for name, sw, fn in (("conv3x3d  ", "conv3x3d", lambda: ops.conv2d_fwd(g, x, w, None, y)), ("upconv3x3q", "upconv3x3q", lambda: ops.conv2d_dgrad(g, y, wt, x))):
        for on in (1, 0):
            ops.set_config(**{sw: on})
            t0, t1 = timed(fn, False), timed(fn, True)
            print(f"{Cc}<->{N}@{HW} {name} {'on ' if on else 'off'}: alone {t0:7.1f} us, beside {n_hog} x {lds} B workgroups {t1:7.1f} us ({t1/t0:.2f}x)", flush=True)
        ops.set_config(**{sw: 1})


# the 64 -> 64 strip kernel (fixed assignment of column strips) against conv3x3h
x = ops.Act((torch.rand((B, 256, 256, 64), device=dev) - 0.5).to(dt))
w = ((torch.rand((64, 9, 64), device=dev) - 0.5) * 0.1).to(dt)
y = ops.Act(torch.empty((B, 256, 256, 64), device=dev, dtype=dt))
g = ops.geom(B, 256, 256, 64, 64, 3, 1)
fn = lambda: ops.conv2d_fwd(g, x, w, None, y)
for on in (1, 0):
    ops.set_config(conv3x3s=on)
    t0, t1 = timed(fn, False), timed(fn, True)
    print(f"64->64@256 conv3x3s {'on ' if on else 'off'}: alone {t0:7.1f} us, beside {n_hog} x {lds} B workgroups {t1:7.1f} us ({t1/t0:.2f}x)", flush=True)
