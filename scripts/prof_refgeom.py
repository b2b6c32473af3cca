"""The train step at the reference's own geometry (main_training.py:27, :154-161: 144 x 160, number_filters_0 = 32, batch 32, bf16) for
rocprofv3: rocprofv3 --kernel-trace --stats -d OUT -o s --output-format csv -- python3 scripts/prof_refgeom.py [F0] [steps] [--overlap]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import unet_rir_amd as U
F0 = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 10
dev = torch.device("cuda", 0)
eng = U.UNetEngine(144, 160, 32, F0=F0, k=3, device=dev, dtype="bf16", overlap_wgrad="--overlap" in sys.argv)
g = torch.Generator(); g.manual_seed(0)
eng.reset_parameters(g)
tr = U.Trainer(eng, lr=5e-7)
batch = bench.synthetic_batch(32, 144, 160, dev, 1234)
for _ in range(2):
    tr.step(*batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(*batch)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"reference geometry F0={F0}: {32 * steps / dt:.1f} spectrograms/s, {dt / steps * 1e3:.2f} ms/step", flush=True)
