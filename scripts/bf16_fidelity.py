"""How far is the bf16-storage train step (the benchmarked mode) from the fp32-storage one (the fp32-tolerance parity mode) at
BASELINE.json configs[1] size (batch 32, 256 x 256, number_filters_0 = 64)?  Same initial variables, same batch, same dropout
masks: per-tensor relative L2 distance and cosine of every gradient, loss difference, and the loss trajectory of N Adam steps.
python scripts/bf16_fidelity.py [steps] [lr]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
import bench

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
B, H, F0 = 32, 256, 64


def build(dtype):
    eng = U.UNetEngine(H, H, B, F0=F0, k=3, device=dev, dtype=dtype)
    g = torch.Generator(); g.manual_seed(0)
    eng.reset_parameters(g)
    eng.dropout_seed = 4321
    return eng


spec_in, emb, spec_out = bench.synthetic_batch(B, H, H, dev, 1234)
e32, e16 = build("f32"), build("bf16")
e16.load_keras_params(e32.export_keras_params())          # identical initial variables (the flat layouts differ by channel padding)
out = {}
grads = {}
for name, eng in (("f32", e32), ("bf16", e16)):
    eng.training = True
    mask = eng.make_dropout_mask()
    eng.forward(spec_in, emb, dropout_mask=mask, target=spec_out, global_batch=B)
    eng.backward()
    torch.cuda.synchronize()
    out[name] = {"loss": float(eng.loss_out[0])}
    grads[name] = {k: v.double().clone() for k, v in eng.export_keras_grads().items()}
    eng._shared["dropout_step"] = 0
rows = []
for n, g32 in grads["f32"].items():
    g16 = grads["bf16"][n]
    n32 = float(g32.norm())
    rel = float((g16 - g32).norm()) / (n32 + 1e-300)
    cos = float((g16 * g32).sum()) / (n32 * float(g16.norm()) + 1e-300)
    rows.append((n, rel, cos, n32))
rows.sort(key=lambda r: -r[1])
print(f"loss fp32 {out['f32']['loss']:.8f}  bf16 {out['bf16']['loss']:.8f}  rel diff {abs(out['f32']['loss'] - out['bf16']['loss']) / out['f32']['loss']:.2e}")
print(f"{len(rows)} gradient tensors; worst by relative L2:")
for n, rel, cos, n32 in rows[:12]:
    print(f"  {n:28s} rel L2 {rel:.4f}  cosine {cos:.6f}  |g| {n32:.3e}")
nonzero = [r for r in rows if r[3] > 1e-12]
print(f"  median rel L2 {sorted(r[1] for r in nonzero)[len(nonzero) // 2]:.4f}; min cosine {min(r[2] for r in nonzero):.6f}; "
      f"tensors with an (analytically) zero fp32 gradient: {len(rows) - len(nonzero)}")
whole32 = torch.cat([g.flatten() for g in grads["f32"].values()])
whole16 = torch.cat([g.flatten() for g in grads["bf16"].values()])
print(f"  whole gradient: rel L2 {float((whole16 - whole32).norm() / whole32.norm()):.4f}, cosine {float((whole16 * whole32).sum() / (whole16.norm() * whole32.norm())):.6f}")

traj = {}
for name, eng in (("f32", e32), ("bf16", e16)):
    tr = U.Trainer(eng, lr=lr, dropout=True)
    ls = []
    for _ in range(steps):
        tr.step(spec_in, emb, spec_out)
        ls.append(float(eng.loss_out[0]))
    traj[name] = ls
d = [abs(a - b) / a for a, b in zip(traj["f32"], traj["bf16"])]
print(f"{steps}-step trajectory (lr {lr}): fp32 {traj['f32'][0]:.6f} -> {traj['f32'][-1]:.6f}; bf16 {traj['bf16'][0]:.6f} -> {traj['bf16'][-1]:.6f}; "
      f"max rel loss gap {max(d):.3e}, final {d[-1]:.3e}")
print(json.dumps({"loss": out, "worst": rows[:8], "traj_f32": traj["f32"][::5], "traj_bf16": traj["bf16"][::5], "max_gap": max(d)}))
