"""Micro-benchmark of the streaming BatchNorm kernels (bf16): python scripts/micro_bn.py
Times bn_stats, bn_bwd (reduce + apply) and colsum at the layer shapes of configs[1]; the slab assignment is switched with
UNETRIR_CHAN_SWEEP=0/1 (read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
if "--lib" in sys.argv:      # another build of the library (A/B of two builds, one process each)
    U._lib.use_library(os.path.abspath(sys.argv[sys.argv.index("--lib") + 1]))
ops = U.ops
for a_ in [v for v in sys.argv[1:] if v.startswith("--") and "=" in v]:       # --bn_fused=0: kernel-selection switches
    k_, v_ = a_[2:].split("=")
    ops.set_config(**{k_: int(v_)})
dev = "cuda:0"
B = 32
ws = ops.Workspace(dev)


def timeit(f, K=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3


SHAPES = ((128, 32), (64, 64), (32, 128), (16, 256)) if "--resae" in sys.argv else ((256, 64), (128, 128), (64, 256), (32, 512))
for HW, C in SHAPES:
    x = ops.Act((torch.rand((B, HW, HW, C), device=dev) - 0.5).to(torch.bfloat16))
    da = ops.Act((torch.rand((B, HW, HW, C), device=dev) - 0.5).to(torch.bfloat16))
    dx = ops.Act(torch.empty((B, HW, HW, C), device=dev, dtype=torch.bfloat16))
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    affine = torch.empty(2 * C, device=dev); saved = torch.empty(2 * C, device=dev)
    dg = torch.empty(C, device=dev); db = torch.empty(C, device=dev)
    nbytes = x.base.numel() * 2
    t_s = timeit(lambda: ops.bn_stats(x, gamma, beta, affine, saved, ws))
    t_b = timeit(lambda: ops.bn_bwd(da, x, gamma, affine, saved, dx, dg, db, ws))
    t_c = timeit(lambda: ops.colsum(x, db, ws))
    t_a = timeit(lambda: ops.bn_apply(x, affine, dx, relu=True))
    rows = 256
    cst = torch.rand((rows, C, 2), device=dev)
    t_f = timeit(lambda: ops.bn_colstat_act_add(cst, rows, x, gamma, beta, affine, saved, dx, 2, None))
    print(f"sweep={os.environ.get('UNETRIR_CHAN_SWEEP','1')} {HW}x{HW}x{C}: bn_stats {t_s:.1f} us ({nbytes/t_s/1e6:.2f} TB/s)  "
          f"bn_bwd {t_b:.1f} us ({5*nbytes/t_b/1e6:.2f} TB/s over 5 passes)  colsum {t_c:.1f} us ({nbytes/t_c/1e6:.2f} TB/s)  bn_apply {t_a:.1f} us ({2*nbytes/t_a/1e6:.2f} TB/s)  colstat->apply {t_f:.1f} us nt={os.environ.get('UNETRIR_BN_NT','0')}")
