"""Train-step rate with the batches coming from HOST memory through DeviceBatchPipeline (pinned staging + copy stream), i.e.
the PCIe-inclusive figure beside bench.py's HBM-resident one: python scripts/pcie_inclusive.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import unet_rir_amd as U
dev = "cuda:0"
B, H = 32, 256
eng = U.UNetEngine(H, H, B, F0=64, dtype="bf16", device=dev, overlap_wgrad=True)
g = torch.Generator(); g.manual_seed(0); eng.reset_parameters(g)
tr = U.Trainer(eng, lr=5e-7)
rng = np.random.default_rng(0)
host = [(rng.random((B, H, H, 2), dtype=np.float32), rng.integers(26, 1282, (B, 2, 16)).astype(np.int32),
         rng.random((B, H, H, 2), dtype=np.float32)) for _ in range(3)]          # NHWC, as DataGenerator.__getitem__ yields them
N = 60


def batches():
    for i in range(N):
        yield host[i % 3]


t0 = None
for i, (spec_in, emb, spec_out) in enumerate(U.DeviceBatchPipeline(batches(), dev, nhwc=True, stream=(eng.opt_stream if os.environ.get('PIPE_OPT_STREAM', '0') == '1' else None))):
    if i == 10:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(spec_in, emb, spec_out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (N - 10)
mb = sum(a.nbytes for a in host[0]) / 1e6
print(f"host-fed (PCIe-inclusive): {dt*1e3:.2f} ms/step, {B/dt:.0f} spectrograms/s, {mb:.1f} MB per batch over PCIe ({mb/dt/1e3:.1f} GB/s sustained)")
