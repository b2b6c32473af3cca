import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import unet_rir_amd as U
from oracle import torch_ref as R
import test_model_gpu as T
H, W, F0, B = [int(a) for a in sys.argv[1:5]]
cfg, Pn, eng, tr, ref, _ = T.run_case(U, H, W, F0, B)
loss, dl, pred, grads, inter = ref
print("pred err", float((eng.pred.double().cpu() - pred).abs().max()))
kg = eng.export_keras_grads()
for n, g_ref in grads.items():
    g = kg[n].double(); s = float(g_ref.abs().max()); e = float((g - g_ref).abs().max())
    print(f"{n:24s} scale {s:.3e} relerr {e/(s+1e-30):.3e}")
