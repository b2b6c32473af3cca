"""Time the ResAE train step (BASELINE.json configs[4]: filters 32..256, batch 32, 256x256, fp32) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
B, H = 32, 256
eng = U.ResAEEngine(H, H, B, device="cuda:0")
g = torch.Generator(); g.manual_seed(0)
eng.reset_parameters(g)
dev = "cuda:0"
spec_in = torch.rand((B, 2, H, H), device=dev); spec_out = torch.rand((B, 2, H, H), device=dev)
emb = torch.randint(26, 1282, (B, 2, 16), device=dev)
def step():
    ml, md = eng.make_dropout_masks()
    eng.forward(spec_in, emb, ml, md, target=spec_out, global_batch=B)
    eng.backward()
    eng.adam_step(5e-7)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 10
for _ in range(K): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"ResAE cfg5 fp32: {B*K/dt:.1f} spectrograms/s, {dt/K*1e3:.2f} ms/step, params {eng.n_params():,}, loss {float(eng.loss_out[0]):.5f}")
