import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import unet_rir_amd as U
from oracle import torch_ref as R
DEV = "cuda:0"
cfg = R.Config(32, 32, 8, 3)
Pn = R.init_params(cfg, randomize_all=True, dtype=np.float64)
spec_in, emb, spec_out = R.synthetic_batch(cfg, 2)
t = lambda a: torch.tensor(a).to(DEV)
for nsteps in (1, 2):
    res = []
    for ov in (False, True):
        eng = U.UNetEngine(32, 32, 2, F0=8, device=DEV, dtype="f32", overlap_wgrad=ov)
        eng.load_keras_params(Pn)
        tr = U.Trainer(eng, lr=1e-3, dropout=False, bucket_bytes=16 << 10)
        for _ in range(nsteps):
            tr.step(t(spec_in), t(emb), t(spec_out))
        torch.cuda.synchronize()
        res.append((eng.theta.clone(), eng.grad.clone(), eng))
    (ta, ga, ea), (tb, gb, eb) = res
    print("steps", nsteps, "grad equal", torch.equal(ga, gb), "theta equal", torch.equal(ta, tb))
    for n, s_ in ea.specs.items():
        d = (ta[s_.offset:s_.offset + s_.numel] - tb[s_.offset:s_.offset + s_.numel]).abs().max().item()
        dg = (ga[s_.offset:s_.offset + s_.numel] - gb[s_.offset:s_.offset + s_.numel]).abs().max().item()
        if d > 0 or dg > 0:
            print("   ", n, "theta diff", d, "grad diff", dg)
