"""ResAE (BASELINE.json configs[4]) train steps through the boundary class, for rocprofv3:
rocprofv3 --kernel-trace --stats -d OUT -o s --output-format csv -- python3 scripts/prof_resae.py [bf16|f32] [steps] [--overlap]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import unet_rir_amd as U
if "--lib" in sys.argv:      # another build of the library
    U._lib.use_library(os.path.abspath(sys.argv[sys.argv.index("--lib") + 1]))
    del sys.argv[sys.argv.index("--lib"):sys.argv.index("--lib") + 2]
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 10
overlap = "--overlap" in sys.argv
dev = torch.device("cuda", 0)
model = U.ResAE((256, 256, 2), (2, 16), (32, 64, 128, 256), (3, 3, 3, 3), (2, 2, 2, 2), 32, 16 * 64, name="resae",
                batch_size=32, device=dev, dtype=dtype, overlap=overlap)
tr = U.Trainer(model, lr=5e-7, alpha=0.9)
batch = bench.synthetic_batch(32, 256, 256, dev, 1234)
for _ in range(2):
    tr.step(*batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(*batch)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"ResAE cfg5 {dtype} overlap={overlap}: {32 * steps / dt:.1f} spectrograms/s, {dt / steps * 1e3:.2f} ms/step (host enqueue {t_host / steps * 1e3:.2f} ms/step)", flush=True)
