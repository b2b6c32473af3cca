"""A/B of one ablation-build variant bit on the FULL configs[1] train step, alternating in one process on one engine (the ablation
library, scripts only): python scripts/ab_abl_step.py <bits> [rounds] [--no-overlap]
e.g. 32768 = conv3x3p output stores with sc1 (device scope: written through the XCD's L2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import unet_rir_amd as U
U.build.build_ablations()
U._lib.use_library(U.build.ABL_LIB)
aset = C.CDLL(U.build.ABL_LIB).unetrir_abl_set
import bench
bits = int(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 8
overlap = "--no-overlap" not in sys.argv
dev = torch.device("cuda:0")
eng = U.UNetEngine(256, 256, 32, F0=64, k=3, device=dev, dtype="bf16", overlap_wgrad=overlap)
g = torch.Generator(); g.manual_seed(0); eng.reset_parameters(g)
tr = U.Trainer(eng, lr=5e-7)
batch = bench.synthetic_batch(32, 256, 256, dev, 1234)
res = {0: [], bits: []}
for r in range(rounds):
    for v in (0, bits) if r % 2 == 0 else (bits, 0):
        aset(v)
        for _ in range(3): tr.step(*batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(12): tr.step(*batch)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 12 * 1e3)
aset(0)
for v in (0, bits):
    t = sorted(res[v]); print(f"abl={v}: median {t[len(t)//2]:.3f} min {t[0]:.3f} mean {sum(t)/len(t):.3f} ms/step ({'overlap' if overlap else 'single stream'})")
