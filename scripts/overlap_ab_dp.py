"""As overlap_ab.py but through the bucketed all-reduce path on a 1-rank RCCL group (UNETRIR_FORCE_DP=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["UNETRIR_FORCE_DP"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
import torch
import torch.distributed as dist
import unet_rir_amd as U
dev = "cuda:0"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
B, H = 32, 256
spec_in, emb, spec_out = next(U.synthetic_batches(1, B, H, H, dev))
for rep in range(4):
    for ov in ((False, True) if rep % 2 == 0 else (True, False)):
        eng = U.UNetEngine(H, H, B, F0=64, dtype="bf16", device=dev, overlap_wgrad=ov)
        g = torch.Generator(); g.manual_seed(0); eng.reset_parameters(g)
        tr = U.Trainer(eng, lr=5e-7)
        assert tr.bucketer is not None
        for _ in range(5): tr.step(spec_in, emb, spec_out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): tr.step(spec_in, emb, spec_out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        print(f"DP overlap_wgrad={ov}: {dt*1e3:.3f} ms/step", flush=True)
        del eng, tr
        torch.cuda.empty_cache()
dist.destroy_process_group()
