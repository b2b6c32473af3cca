"""Full train step (cfg 2, bf16, side-stream schedule) on a given build of the library, for A/B of two builds in alternating
processes inside one gpurun call: python scripts/ab_lib.py [--lib path/to/libunetrir.so] [--no-overlap] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
if "--lib" in sys.argv:
    U._lib.use_library(os.path.abspath(sys.argv[sys.argv.index("--lib") + 1]))
import bench
overlap = "--no-overlap" not in sys.argv
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 4
dev = torch.device("cuda:0")
eng = U.UNetEngine(256, 256, 32, F0=64, k=3, device=dev, dtype="bf16", overlap_wgrad=overlap)
g = torch.Generator(); g.manual_seed(0)
eng.reset_parameters(g)
tr = U.Trainer(eng, lr=5e-7)
batch = bench.synthetic_batch(32, 256, 256, dev, 1234)
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
print(f"{'base' if '--lib' in sys.argv else 'new '} {'overlap' if overlap else 'single'}: " + " ".join(f"{v:.3f}" for v in res) + f"  min {min(res):.3f} ms/step", flush=True)
