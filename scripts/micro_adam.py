"""Micro-benchmark of the optimizer launch over the flat variable buffer (configs[1]: 31 M variables, 7 fp32 streams):
does the relative placement of theta / g / m / v matter (equal offsets of four 124 MB arrays meet in the same HBM channel)?
python scripts/micro_adam.py [n] [--lib path]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
if "--lib" in sys.argv:
    U._lib.use_library(os.path.abspath(sys.argv[sys.argv.index("--lib") + 1]))
    del sys.argv[sys.argv.index("--lib"):sys.argv.index("--lib") + 2]
ops = U.ops
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 31_040_000


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


for skew in (0, 64, 256, 1024, 4096, 16384, 65536 + 4096):          # floats between the starts of consecutive arrays, beyond n
    pool = torch.zeros(4 * n + 4 * skew + 1024, device=dev)
    arrs = []
    for j in range(4):
        o = j * (n + skew)
        o -= o % 4
        arrs.append(pool[o:o + n])
    theta, g, m, v = arrs
    g.uniform_(-1e-3, 1e-3)
    t = timed(lambda: ops.adam(theta, g, m, v, 1e-4))
    print(f"n {n} skew {skew:6d} floats: adam {t:7.1f} us  {7 * 4 * n / t / 1e6:.2f} TB/s", flush=True)
sep = [torch.zeros(n, device=dev) for _ in range(4)]
sep[1].uniform_(-1e-3, 1e-3)
t = timed(lambda: ops.adam(sep[0], sep[1], sep[2], sep[3], 1e-4))
print(f"separate allocations: adam {t:7.1f} us  {7 * 4 * n / t / 1e6:.2f} TB/s; ptr mod 2^21: {[a.data_ptr() % (1 << 21) for a in sep]}")
t = timed(lambda: ops.sgd(sep[0], sep[1], 1e-4))
print(f"sgd {t:7.1f} us  {3 * 4 * n / t / 1e6:.2f} TB/s")
