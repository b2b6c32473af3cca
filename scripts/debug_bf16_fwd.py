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
P = R.to_torch(Pn, torch.float64, True)
inter = {}
pred = R.forward(P, torch.tensor(spec_in).double(), torch.tensor(emb), cfg, True, None, None, inter, "bf16")
eng = U.UNetEngine(H, W, B, F0=F0, device="cuda:0", dtype="bf16")
eng.load_keras_params(Pn)
t = lambda a: torch.tensor(a).cuda()
eng.forward(t(spec_in), t(emb), target=t(spec_out), global_batch=B)
torch.cuda.synchronize()
def cmp(name, act):
    e = inter[name].detach()
    a = act.dense().permute(0, 3, 1, 2).double().cpu()
    d = (a - e).abs()
    print(f"{name:20s} scale {float(e.abs().max()):.3e} maxerr {float(d.max()):.3e} nbad(>1% scale) {(d > 1e-2*e.abs().max()).sum().item()} / {d.numel()}")
D = cfg.depth
for l in range(1, D + 2):
    cmp(f"enc{l}.down", eng.down[l]); cmp(f"enc{l}.cb1.conv", eng.y[l]); cmp(f"enc{l}.cb1.out", eng.a[l])
cmp("bottleneck", eng.z)
for l in range(D, 0, -1):
    cmp(f"dec{l}.up", eng.cat[l].slice(eng.ch[l-1], eng.ch[l-1])); cmp(f"dec{l}.cb1a.conv", eng.ya[l]); cmp(f"dec{l}.cb1a.out", eng.aa[l])
    cmp(f"dec{l}.cb1b.conv", eng.yb[l]); cmp(f"dec{l}.cb1b.out", eng.ab[l])
e = inter["head.logits"].detach(); a = eng.logits.dense()[..., :2].permute(0, 3, 1, 2).double().cpu()
print("logits maxerr", float((a - e).abs().max()), "scale", float(e.abs().max()))
