"""A/B of one kernel-selection switch on the full train step, alternating two engines in ONE process (cfg 2, bf16):
python scripts/ab_switch.py conv3x3s [--no-overlap] [--resae] [rounds]        (attr:park_reduces flips an ENGINE attribute instead of a
library switch; --resae: BASELINE.json configs[4] instead of configs[1])"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
import bench

name = sys.argv[1]
attr = name[5:] if name.startswith("attr:") else None
overlap = "--no-overlap" not in sys.argv
resae = "--resae" in sys.argv
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 6
dev = torch.device("cuda:0")
trainers = {}
def flip(val, eng=None):
    if attr is None:
        U.ops.set_config(**{name: val})
    elif eng is not None:
        assert getattr(eng, attr) in (True, False) and getattr(eng, attr + "_supported", True)
        setattr(eng, attr, bool(val))


for val in (1, 0):
    flip(val)
    if resae:
        eng = U.ResAEEngine(256, 256, 32, (32, 64, 128, 256), (3, 3, 3, 3), (2, 2, 2, 2), 32, 1024, device=dev, dtype="bf16", overlap_wgrad=overlap)
    else:
        eng = U.UNetEngine(256, 256, 32, F0=64, k=3, device=dev, dtype="bf16", overlap_wgrad=overlap)
    flip(val, eng)
    g = torch.Generator(); g.manual_seed(0)
    eng.reset_parameters(g)
    trainers[val] = U.Trainer(eng, lr=5e-7)
batch = bench.synthetic_batch(32, 256, 256, dev, 1234)
res = {0: [], 1: []}
for r in range(rounds):
    for val in (1, 0):
        flip(val)
        tr = trainers[val]
        for _ in range(2):
            tr.step(*batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.step(*batch)
        torch.cuda.synchronize()
        res[val].append((time.perf_counter() - t0) * 100)
        print(f"round {r} {name}={val}: {res[val][-1]:.3f} ms/step", flush=True)
for val in (1, 0):
    v = sorted(res[val])
    print(f"{name}={val}: median {v[len(v)//2]:.3f} min {v[0]:.3f} ms/step ({'overlap' if overlap else 'single stream'})")
