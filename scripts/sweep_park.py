"""Sweep of the two ReduceBatch knobs (largest slab set that is parked, arena size = bytes parked before a flush) on the ResAE step
(BASELINE.json configs[4]) and the U-Net step (configs[1]), engines alternating in one process:
python scripts/sweep_park.py [--unet] [--no-overlap]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
import bench

dev = torch.device("cuda:0")
unet = "--unet" in sys.argv
overlap = "--no-overlap" not in sys.argv
combos = [(0, 96), (4, 32), (16, 32), (16, 96), (64, 96), (64, 256), (None, 1024)]      # (park_max MB, arena MB); 0: nothing parked
trainers = []
for pm, ar in combos:
    if unet:
        eng = U.UNetEngine(256, 256, 32, F0=64, k=3, device=dev, dtype="bf16", overlap_wgrad=overlap)
    else:
        eng = U.ResAEEngine(256, 256, 32, (32, 64, 128, 256), (3, 3, 3, 3), (2, 2, 2, 2), 32, 1024, device=dev, dtype="bf16", overlap_wgrad=overlap)
    g = torch.Generator(); g.manual_seed(0)
    eng.reset_parameters(g)
    if pm == 0:
        eng.park_reduces = False
    else:
        eng._rb = U.ops.ReduceBatch(dev, ar << 20, park_max_bytes=None if pm is None else pm << 20)
    trainers.append(U.Trainer(eng, lr=5e-7))
batch = bench.synthetic_batch(32, 256, 256, dev, 1234)
res = [[] for _ in combos]
for r in range(5):
    for i, tr in enumerate(trainers):
        for _ in range(3):
            tr.step(*batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(12):
            tr.step(*batch)
        torch.cuda.synchronize()
        res[i].append((time.perf_counter() - t0) / 12 * 1e3)
for (pm, ar), v in zip(combos, res):
    v = sorted(v)
    print(f"park_max {pm} MB arena {ar} MB: median {v[len(v)//2]:.3f} min {v[0]:.3f} ms/step", flush=True)
