import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import unet_rir_amd as U
from oracle import torch_ref as R
H, W, F0, B = [int(a) for a in sys.argv[1:5]]
cfg = R.Config(H, W, F0, 3)
Pn = R.init_params(cfg, randomize_all=True, dtype=np.float64)
spec_in, emb, spec_out = R.synthetic_batch(cfg, B)
P = R.to_torch(Pn, torch.float64, True)
inter = {}
pred = R.forward(P, torch.tensor(spec_in).double(), torch.tensor(emb), cfg, True, None, None, inter)
for v in inter.values():
    v.retain_grad()
loss = R.data_loss(torch.tensor(spec_out).double(), pred, 0.9, B) + R.reg_loss(P, cfg, 1)
loss.backward()
eng = U.UNetEngine(H, W, B, F0=F0, device="cuda:0")
eng.load_keras_params(Pn)
t_in = torch.tensor(spec_in).cuda(); t_out = torch.tensor(spec_out).cuda(); t_emb = torch.tensor(emb).cuda()
eng.forward(t_in, t_emb, target=t_out, global_batch=B)
eng.backward()
torch.cuda.synchronize()
def cmp(name, act):
    e = inter[name].grad
    a = act.dense().permute(0, 3, 1, 2).double().cpu()
    d = (a - e).abs()
    idx = np.unravel_index(int(d.argmax()), d.shape)
    print(f"{name:20s} scale {float(e.abs().max()):.3e} relerr {float(d.max())/float(e.abs().max()):.3e} at {idx}  nbad {(d > 1e-4*e.abs().max()).sum().item()}")
D = cfg.depth
for l in range(1, D + 1):
    cmp(f"dec{l}.cb1b.out", eng.g_ab[l]); cmp(f"dec{l}.cb1b.conv", eng.g_yb[l]); cmp(f"dec{l}.cb1a.out", eng.g_aa[l])
    cmp(f"dec{l}.cb1a.conv", eng.g_ya[l]); cmp(f"dec{l}.up", eng.g_cat[l].slice(eng.ch[l-1], eng.ch[l-1]))
cmp("bottleneck", eng.g_z)
for l in range(D + 1, 0, -1):
    if l <= D: cmp(f"enc{l}.cb1.out", eng.g_cat[l].slice(0, eng.ch[l-1]))
    cmp(f"enc{l}.cb1.conv", eng.g_y[l]); cmp(f"enc{l}.down", eng.g_down[l])
