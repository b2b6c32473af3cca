import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np, torch
import unet_rir_amd as U
from oracle import torch_ref as R
H, W, F0, B = [int(a) for a in sys.argv[1:5]]
cfg = R.Config(H, W, F0, 3)
Pn = R.init_params(cfg, randomize_all=True, dtype=np.float64)
spec_in, emb, spec_out = R.synthetic_batch(cfg, B)
loss, dl, pred, grads = R.loss_and_grads(Pn, spec_in, emb, spec_out, cfg, dtype=torch.float64, storage=(None if len(sys.argv) > 5 else 'bf16'))
eng = U.UNetEngine(H, W, B, F0=F0, device="cuda:0", dtype="bf16")
eng.load_keras_params(Pn)
t = lambda a: torch.tensor(a).cuda()
eng.forward(t(spec_in), t(emb), target=t(spec_out), global_batch=B)
eng.backward(); eng.reg_loss()
torch.cuda.synchronize()
print("pred maxerr", float((eng.pred.double().cpu() - pred).abs().max()), "loss", float(eng.loss_out[0]) + float(eng.reg_out[0]), loss)
kg = eng.export_keras_grads()
worst = 0
for n, g_ref in grads.items():
    g = kg[n].double(); s = float(g_ref.abs().max())
    l2 = float((g - g_ref).norm() / (g_ref.norm() + 1e-30))
    if not n.endswith(("cb1.bias", "cb1a.bias", "cb1b.bias")):
        worst = max(worst, l2)
    print(f"{n:24s} scale {s:.3e} maxrel {float((g-g_ref).abs().max())/(s+1e-30):.3e} l2rel {l2:.3e}")
print("worst l2rel", worst)
