"""Full configs[1] step with two or more builds of the library, one process per build and round, the order rotated from round to round:
python scripts/ab_two_libs.py libA.so libB.so [libC.so ...] [--resae | --refgeom] [rounds]      (each process: 4 x (3 warm-up + 12 timed steps); prints per-build medians)."""
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
if sys.argv[3] == "resae":      # BASELINE.json configs[4]
    eng = U.ResAEEngine(256, 256, 32, (32, 64, 128, 256), (3, 3, 3, 3), (2, 2, 2, 2), 32, 1024, device=dev, dtype="bf16", overlap_wgrad=(sys.argv[2] == "1"))
elif sys.argv[3] == "refgeom":  # the reference's own default: 144 x 160, F0 = 32, batch 16
    eng = U.UNetEngine(144, 160, 16, F0=32, k=3, device=dev, dtype="bf16", overlap_wgrad=(sys.argv[2] == "1"))
else:
    eng = U.UNetEngine(256, 256, 32, F0=64, k=3, device=dev, dtype="bf16", overlap_wgrad=(sys.argv[2] == "1"))
Bn, Hn, Wn = (16, 144, 160) if sys.argv[3] == "refgeom" else (32, 256, 256)
g = torch.Generator(); g.manual_seed(0); eng.reset_parameters(g)
tr = U.Trainer(eng, lr=5e-7)
batch = bench.synthetic_batch(Bn, Hn, Wn, dev, 1234)
out = []
for r in range(4):
    for _ in range(3): tr.step(*batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(12): tr.step(*batch)
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 12 * 1e3)
print(min(out), sorted(out)[len(out) // 2])
''' % ROOT
model = "resae" if "--resae" in sys.argv else "refgeom" if "--refgeom" in sys.argv else "unet"
libs = [os.path.abspath(a) for a in sys.argv[1:] if not a.isdigit() and not a.startswith("--")]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else len(libs)
for overlap in ("1", "0"):
    res = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs[r % len(libs):] + libs[:r % len(libs)]:
            o = subprocess.run([sys.executable, "-c", CHILD, l, overlap, model], capture_output=True, text=True).stdout.strip().splitlines()[-1].split()
            res[l].append(float(o[1]))
    for l in libs:
        v = sorted(res[l]); print(f"{os.path.basename(l)} {'overlap' if overlap == '1' else 'single stream'}: medians {['%.3f' % x for x in v]}", flush=True)
