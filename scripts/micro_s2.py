"""Micro-benchmark of the stride-2 layer pair at every level of cfg 2 (bf16, batch 32) through the C ABI:
strided Conv2D (C -> 2C, HW -> HW/2): forward, data gradient, weight gradient; Conv2DTranspose (2C -> C, HW/2 -> HW): the same."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
ops = U.ops
dev, dt, B = "cuda:0", torch.bfloat16, 32


def timeit(fn, K=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K


def r(shape, s=1.0):
    return ((torch.rand(shape, device=dev) - 0.5) * s).to(dt)


args = [a for a in sys.argv[1:] if not a.startswith("--")]
for a in sys.argv[1:]:                     # --conv3x3d=0: kernel-selection switches for an A/B in one process order
    if a.startswith("--"):
        k, v = a[2:].split("=")
        ops.set_config(**{k: int(v)})
levels = [(64, 256), (128, 128), (256, 64), (512, 32)] if not args else [tuple(int(v) for v in a.split(",")) for a in args]
ws = ops.Workspace(dev)
for C, HW in levels:
    N, ho = 2 * C, HW // 2
    fl = 2.0 * B * ho * ho * C * N * 9
    g = ops.geom(B, HW, HW, C, N, 3, 2)
    x, y = ops.Act(r((B, HW, HW, C))), ops.Act(r((B, ho, ho, N)))
    w, wt = r((N, 9, C), 0.1), r((C, 9, N), 0.1)
    w32 = w.float()
    bias = torch.zeros(N, device=dev)
    dw = torch.zeros((N, 9, C), device=dev)
    def packed(w16, n_, c_):        # the packed copy cast_weights_batched makes for the stride-2 forward kernel (None: not defined)
        ne = ops.conv3x3s2_packed_elems(n_, c_)
        if not ne:
            return None
        pk = torch.zeros(ne, dtype=dt, device=dev)
        ops.cast_weights_batched(ops.make_cast_table([(w16.float(), torch.empty_like(w16), torch.empty((c_, 9, n_), dtype=dt, device=dev), n_, 9, c_, c_, n_, pk)], dev))
        return pk
    wpk = packed(w, N, C)
    res = [("conv  fwd", timeit(lambda: ops.conv2d_fwd(g, x, w, bias, y, w_packed=wpk))),
           ("conv  dgrad", timeit(lambda: ops.conv2d_dgrad(g, y, wt, x))),
           ("conv  wgrad", timeit(lambda: ops.conv2d_wgrad(g, x, y, dw, ws, reg=0.002, w=w32)))]
    # transposed: 2C -> C, coarse ho -> HW
    gt = ops.geom(B, ho, ho, N, C, 3, 2)
    bias_c = torch.zeros(C, device=dev)
    dwt = torch.zeros((N, 9, C), device=dev)
    wt_t, w_t = r((C, 9, N), 0.1), r((N, 9, C), 0.1)
    wpk_t = packed(w_t, N, C)
    res += [("convT fwd", timeit(lambda: ops.conv2d_transpose_fwd(gt, y, wt_t, bias_c, x))),
            ("convT dgrad", timeit(lambda: ops.conv2d_transpose_dgrad(gt, x, w_t, y, w_packed=wpk_t))),
            ("convT wgrad", timeit(lambda: ops.conv2d_transpose_wgrad(gt, y, x, dwt, ws, reg=0.002, w=w32)))]
    print(f"C={C} <-> {N}, {HW}^2 <-> {ho}^2, {fl/1e9:.1f} GFLOP: " + " | ".join(f"{n} {ms*1e3:6.1f} us {fl/ms/1e9:5.0f} TF" for n, ms in res), flush=True)
