"""Micro-benchmark of the tap-table implicit-GEMM path (bf16): the layers no patch-staged kernel takes - small grids (the reference
geometry's 36 x 40 / 18 x 20 / 9 x 10 levels, main_training.py:27), 6 x 6 kernels (the constructor default), stride-2 layers at 16 x 16.
python scripts/micro_igemm.py ["C,N,H,W,k,s;..."] [--lib path] [--switch=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
if "--lib" in sys.argv:
    U._lib.use_library(os.path.abspath(sys.argv[sys.argv.index("--lib") + 1]))
    del sys.argv[sys.argv.index("--lib"):sys.argv.index("--lib") + 2]
ops = U.ops
dev, dt, B = "cuda:0", torch.bfloat16, 32
for a_ in [v for v in sys.argv[1:] if v.startswith("--")]:
    k_, v_ = a_[2:].split("=")
    ops.set_config(**{k_: int(v_)})
args = [v for v in sys.argv[1:] if not v.startswith("--")]
shapes = args[0] if args else ("128,128,36,40,3,1;256,256,18,20,3,1;512,512,9,10,3,1;256,128,36,40,3,1;512,256,18,20,3,1;64,128,72,80,3,2;128,256,36,40,3,2;"
                               "256,512,18,20,3,2;512,1024,32,32,3,2;64,64,64,64,6,1;64,128,64,64,6,2;1024,1024,16,16,3,1")


def timed(fn, K=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3


for sh in shapes.split(";"):
    C, N, H, W, k, s = (int(v) for v in sh.split(","))
    Ho, Wo = -(-H // s), -(-W // s)
    g = ops.geom(B, H, W, C, N, k, s)
    x = ops.Act((torch.rand((B, H, W, C), device=dev) - 0.5).to(dt))
    w = ((torch.rand((N, k * k, C), device=dev) - 0.5) * 0.1).to(dt)
    wt = ((torch.rand((C, k * k, N), device=dev) - 0.5) * 0.1).to(dt)
    bias = torch.zeros(N, device=dev)
    y = ops.Act(torch.empty((B, Ho, Wo, N), device=dev, dtype=dt))
    dx = ops.Act(torch.empty((B, H, W, C), device=dev, dtype=dt))
    fl = 2.0 * B * Ho * Wo * C * N * k * k
    t_f = timed(lambda: ops.conv2d_fwd(g, x, w, bias, y))
    t_d = timed(lambda: ops.conv2d_dgrad(g, y, wt, dx))
    kern = ops.conv3x3_kernel(g, 0, x) if (k == 3 and s == 1) else "-"
    print(f"C={C:4d} N={N:4d} {H}x{W} k{k} s{s} [{kern}]: fwd {t_f:7.1f} us {fl / t_f / 1e6:6.0f} TF/s | dgrad {t_d:7.1f} us {fl / t_d / 1e6:6.0f} TF/s", flush=True)
