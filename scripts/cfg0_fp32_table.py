"""BASELINE.json configs[0] at its exact size (F0 = 16, batch 4, 256 x 256), fp32 storage: per-tensor distance of the HIP engine's gradients
from the fp64 oracle, beside the distance of the SAME oracle evaluated in fp32 (torch CPU) - the table behind the criterion of
tests/test_cfg0_gpu.py.  python scripts/cfg0_fp32_table.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import unet_rir_amd as U
from oracle import torch_ref as R

H = W = 256; F0, B = 16, 4
cfg = R.Config(H, W, F0, 3)
Pn = R.init_params(cfg, randomize_all=True, dtype=np.float64)
spec_in, emb, spec_out = R.synthetic_batch(cfg, B, seed_name="cfg0")
g64 = R.loss_and_grads(Pn, spec_in, emb, spec_out, cfg, 0.9, B, 1, None, torch.float64, None)[3]
g32 = R.loss_and_grads(Pn, spec_in, emb, spec_out, cfg, 0.9, B, 1, None, torch.float32, None)[3]
eng = U.UNetEngine(H, W, B, F0=F0, k=3, device="cuda:0", dtype="f32")
eng.load_keras_params(Pn)
t = lambda a: torch.tensor(a).to("cuda:0")
eng.training = True
eng.forward(t(spec_in), t(emb), target=t(spec_out), global_batch=B, alpha=0.9)
eng.backward()
torch.cuda.synchronize()
kg = eng.export_keras_grads()
rows = []
for n, r in g64.items():
    nrm, scale = float(r.norm()), float(r.abs().max())
    if nrm < 1e-12:
        continue
    h, c = kg[n].double(), g32[n].double()
    eh, ec = (h - r).abs(), (c - r).abs()
    rows.append({"tensor": n, "numel": r.numel(), "scale": scale, "hip_l2": float((h - r).norm()) / nrm, "cpu32_l2": float((c - r).norm()) / nrm,
                 "hip_max": float(eh.max()) / scale, "cpu32_max": float(ec.max()) / scale,
                 "hip_entries_over_1e-3": int((eh > 1e-3 * scale).sum()), "cpu32_entries_over_1e-3": int((ec > 1e-3 * scale).sum())})
rows.sort(key=lambda x: -x["hip_l2"])
for x in rows[:25]:
    print(f"{x['tensor']:22s} L2 hip {x['hip_l2']:.2e} cpu32 {x['cpu32_l2']:.2e} | max hip {x['hip_max']:.2e} cpu32 {x['cpu32_max']:.2e} | entries > 1e-3: hip {x['hip_entries_over_1e-3']} cpu32 {x['cpu32_entries_over_1e-3']} of {x['numel']}")
names = [x["tensor"] for x in rows]
ref = torch.cat([g64[n].flatten() for n in names]); hip = torch.cat([kg[n].double().flatten() for n in names]); c32 = torch.cat([g32[n].double().flatten() for n in names])
whole = {"hip": float((hip - ref).norm() / ref.norm()), "cpu32": float((c32 - ref).norm() / ref.norm())}
print("whole gradient", whole)
if len(sys.argv) > 1:
    json.dump({"rows": rows, "whole": whole}, open(sys.argv[1], "w"), indent=1)
