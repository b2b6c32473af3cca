"""BASELINE.json configs[4] (ResAE, batch 32, 256 x 256): how far is the bf16-storage train step from the fp32-storage one, and how
much does EACH of them move under a permutation of the batch (the noise floor the permutation bound of
tests/test_fullsize_resae_gpu.py has to be read against)?  Same initial variables, same batch, same dropout masks.
python scripts/resae_fidelity.py [steps] [lr] [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
out_path = sys.argv[3] if len(sys.argv) > 3 else None
B = 32


def build(dtype):
    eng = U.ResAEEngine(256, 256, B, (32, 64, 128, 256), (3, 3, 3, 3), (2, 2, 2, 2), 32, 1024, device=dev, dtype=dtype)
    g = torch.Generator(); g.manual_seed(0)
    eng.reset_parameters(g)
    eng.dropout_seed = 9
    return eng


spec_in, emb, spec_out = next(U.synthetic_batches(1, B, 256, 256, dev))
e32, e16 = build("f32"), build("bf16")
e16.load_keras_params(e32.export_keras_params())
res = {"loss": {}, "perm": {}}
grads, preds = {}, {}
pg = torch.Generator(device=dev); pg.manual_seed(3)
perm = torch.randperm(B, device=dev, generator=pg)
for name, eng in (("f32", e32), ("bf16", e16)):
    eng.training = True
    eng.forward(spec_in, emb, target=spec_out, global_batch=B)          # no dropout: a permuted batch would need permuted masks
    eng.backward()
    torch.cuda.synchronize()
    res["loss"][name] = float(eng.loss_out[0])
    grads[name] = {k: v.double().clone() for k, v in eng.export_keras_grads().items()}
    preds[name] = eng.pred.clone()
    whole = eng.grad.double().clone()
    eng.forward(spec_in[perm].contiguous(), emb[perm].contiguous(), target=spec_out[perm].contiguous(), global_batch=B)
    eng.backward()
    torch.cuda.synchronize()
    d = (eng.pred - preds[name][perm]).abs().double()
    res["perm"][name] = {"pred_max": float(d.max()), "pred_rms": float(d.pow(2).mean().sqrt()),
                         "loss_rel": abs(float(eng.loss_out[0]) - res["loss"][name]) / res["loss"][name],
                         "grad_rel_l2": float((eng.grad.double() - whole).norm() / whole.norm())}
d = (preds["bf16"] - preds["f32"]).abs().double()
res["pred_bf16_vs_f32"] = {"max": float(d.max()), "rms": float(d.pow(2).mean().sqrt())}
rows = []
for n, g32 in grads["f32"].items():
    g16 = grads["bf16"][n]
    n32 = float(g32.norm())
    rel = float((g16 - g32).norm()) / (n32 + 1e-300)
    cos = float((g16 * g32).sum()) / (n32 * float(g16.norm()) + 1e-300)
    rows.append((n, rel, cos, n32))
rows.sort(key=lambda r: -r[1])
nonzero = [r for r in rows if r[3] > 1e-12]
w32 = torch.cat([g.flatten() for g in grads["f32"].values()])
w16 = torch.cat([g.flatten() for g in grads["bf16"].values()])
res["grad"] = {"tensors": len(rows), "zero_fp32": len(rows) - len(nonzero), "worst": rows[:12],
               "median_rel": sorted(r[1] for r in nonzero)[len(nonzero) // 2], "min_cos": min(r[2] for r in nonzero),
               "whole_rel": float((w16 - w32).norm() / w32.norm()), "whole_cos": float((w16 * w32).sum() / (w16.norm() * w32.norm()))}
print(f"loss fp32 {res['loss']['f32']:.8f} bf16 {res['loss']['bf16']:.8f}")
print("prediction bf16 vs fp32:", res["pred_bf16_vs_f32"])
print("batch permutation:", res["perm"])
for n, rel, cos, n32 in rows[:12]:
    print(f"  {n:36s} rel L2 {rel:.4f}  cosine {cos:.6f}  |g| {n32:.3e}")
print({k: v for k, v in res["grad"].items() if k != "worst"})
traj = {}
for name, eng in (("f32", e32), ("bf16", e16)):
    eng._shared["dropout_step"] = 0
    tr = U.Trainer(eng, lr=lr, dropout=True)
    ls = []
    for _ in range(steps):
        tr.step(spec_in, emb, spec_out)
        ls.append(eng.loss_out[0].clone())
    torch.cuda.synchronize()
    traj[name] = [float(v) for v in ls]
gap = [abs(a - b) / a for a, b in zip(traj["f32"], traj["bf16"])]
res["traj"] = {"f32": traj["f32"][::5] + traj["f32"][-1:], "bf16": traj["bf16"][::5] + traj["bf16"][-1:], "max_gap": max(gap), "final_gap": gap[-1]}
print(f"{steps}-step trajectory: fp32 {traj['f32'][0]:.6f} -> {traj['f32'][-1]:.6f}; bf16 {traj['bf16'][0]:.6f} -> {traj['bf16'][-1]:.6f}; max gap {max(gap):.3e}")
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
