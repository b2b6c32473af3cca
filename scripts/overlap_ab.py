"""A/B of the side-stream weight-gradient schedule (UNetEngine(overlap_wgrad=True)) at BASELINE configs[1], bf16:
alternating fresh engines in one process, 30 timed steps each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
dev = "cuda:0"
B, H = 32, 256
spec_in, emb, spec_out = next(U.synthetic_batches(1, B, H, H, dev))
for rep in range(4):
    for ov in ((False, True) if rep % 2 == 0 else (True, False)):
        eng = U.UNetEngine(H, H, B, F0=64, dtype="bf16", device=dev, overlap_wgrad=ov)
        g = torch.Generator(); g.manual_seed(0); eng.reset_parameters(g)
        tr = U.Trainer(eng, lr=5e-7)
        for _ in range(5): tr.step(spec_in, emb, spec_out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): tr.step(spec_in, emb, spec_out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        print(f"overlap_wgrad={ov}: {dt*1e3:.3f} ms/step", flush=True)
        del eng, tr
        torch.cuda.empty_cache()
