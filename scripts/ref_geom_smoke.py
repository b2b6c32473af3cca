import sys, math
sys.path.insert(0, "/root/repo")
import torch
import unet_rir_amd as U
dev = "cuda:0"
for (H, W, F0, B) in [(144, 160, 32, 16), (96, 224, 16, 3), (256, 256, 16, 4)]:
    eng = U.UNetEngine(H, W, B, F0=F0, dtype="bf16", device=dev, overlap_wgrad=True)
    g = torch.Generator(); g.manual_seed(0); eng.reset_parameters(g)
    tr = U.Trainer(eng, lr=1e-4, dropout=True)
    si, e, so = next(U.synthetic_batches(1, B, H, W, dev))
    l0 = tr.step(si, e, so, return_loss=True)
    for _ in range(10): l1 = tr.step(si, e, so, return_loss=True)
    print(H, W, F0, B, "loss", l0, "->", l1, "ok" if math.isfinite(l1) and l1 < l0 else "BAD")
