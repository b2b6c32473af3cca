"""Host-side enqueue time per train step vs GPU step time (is the Python launch path keeping ahead of the GPU?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import unet_rir_amd as U
dev = "cuda:0"
dp = os.environ.get("UNETRIR_FORCE_DP") == "1"
if dp:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
B, H = 32, 256
spec_in, emb, spec_out = next(U.synthetic_batches(1, B, H, H, dev))
for ov in (False, True):
    eng = U.UNetEngine(H, H, B, F0=64, dtype="bf16", device=dev, overlap_wgrad=ov)
    g = torch.Generator(); g.manual_seed(0); eng.reset_parameters(g)
    tr = U.Trainer(eng, lr=5e-7)
    for _ in range(3): tr.step(spec_in, emb, spec_out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tr.step(spec_in, emb, spec_out)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"dp={dp} overlap={ov}: host enqueue {t_host/20*1e3:.2f} ms/step, wall {t_all/20*1e3:.2f} ms/step")
    del eng, tr; torch.cuda.empty_cache()
if dp: dist.destroy_process_group()
