"""One train-step workload on a given build of the library, round by round (A/B of two builds in alternating processes inside one
gpurun call; a kernel that is faster in a warm micro-benchmark loop can be slower inside the step, so this is the measurement that
decides):  python scripts/ab_step.py cfg2|refgeom|resae [--lib path/to/libunetrir.so] [--no-overlap] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
if "--lib" in sys.argv:
    U._lib.use_library(os.path.abspath(sys.argv[sys.argv.index("--lib") + 1]))
import bench
what = sys.argv[1]
overlap = "--no-overlap" not in sys.argv
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 5
dev = torch.device("cuda:0")
if what == "resae":
    model = U.ResAE((256, 256, 2), (2, 16), (32, 64, 128, 256), (3, 3, 3, 3), (2, 2, 2, 2), 32, 16 * 64, name="resae", batch_size=32, device=dev,
                    dtype="bf16", overlap=overlap)
    tr = U.Trainer(model, lr=5e-7, alpha=0.9)
    batch = bench.synthetic_batch(32, 256, 256, dev, 1234)
else:
    H, W, F0 = (144, 160, 32) if what == "refgeom" else (256, 256, 64)
    eng = U.UNetEngine(H, W, 32, F0=F0, k=3, device=dev, dtype="bf16", overlap_wgrad=overlap)
    g = torch.Generator(); g.manual_seed(0)
    eng.reset_parameters(g)
    tr = U.Trainer(eng, lr=5e-7)
    batch = bench.synthetic_batch(32, H, W, dev, 1234)
res = []
for r in range(rounds):
    for _ in range(3):
        tr.step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step(*batch)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) * 50)
print(f"{what} {'base' if '--lib' in sys.argv else 'new '} {'overlap' if overlap else 'single'}: " + " ".join(f"{v:.3f}" for v in res) + f"  min {min(res):.3f} ms/step", flush=True)
