"""Full configs[1] step with two builds of the library in two processes run alternately: python scripts/ab_two_libs.py libA.so libB.so [rounds]
(each process: 3 warm-up + 12 timed steps; prints per-build medians)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
import unet_rir_amd as U
U._lib.use_library(sys.argv[1])
import bench
dev = torch.device("cuda:0")
eng = U.UNetEngine(256, 256, 32, F0=64, k=3, device=dev, dtype="bf16", overlap_wgrad=(sys.argv[2] == "1"))
g = torch.Generator(); g.manual_seed(0); eng.reset_parameters(g)
tr = U.Trainer(eng, lr=5e-7)
batch = bench.synthetic_batch(32, 256, 256, dev, 1234)
out = []
for r in range(4):
    for _ in range(3): tr.step(*batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(12): tr.step(*batch)
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 12 * 1e3)
print(min(out), sorted(out)[len(out) // 2])
''' % ROOT
libs = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for overlap in ("1", "0"):
    res = {l: [] for l in libs}
    for r in range(rounds):
        for l in (libs if r % 2 == 0 else libs[::-1]):
            o = subprocess.run([sys.executable, "-c", CHILD, l, overlap], capture_output=True, text=True).stdout.strip().splitlines()[-1].split()
            res[l].append(float(o[1]))
    for l in libs:
        v = sorted(res[l]); print(f"{os.path.basename(l)} {'overlap' if overlap == '1' else 'single stream'}: medians {['%.3f' % x for x in v]}", flush=True)
