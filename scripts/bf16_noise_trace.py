"""Where does the bf16-storage step leave the fp32-storage step?  BASELINE.json configs[1] (batch 32, 256 x 256, number_filters_0 = 64),
same variables / batch / dropout mask in both engines: relative L2 distance of every stored tensor of the forward pass and of every
stored gradient tensor of the backward pass, in execution order, plus the information-vector branch's gradients.  (Round 4 ran it with
dL/dz of the bottleneck handed to that branch before its rounding to bf16 - an fp32 side output of the data-gradient kernel - and from
the bf16 tensor: 0.2865 / 0.2906 / 0.2899 against 0.2865 / 0.2906 / 0.2899; the side output was removed again)
python scripts/bf16_noise_trace.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
import bench

dev = torch.device("cuda:0")
B, H, F0 = 32, 256, 64
out_path = sys.argv[1] if len(sys.argv) > 1 else None


def build(dtype):
    eng = U.UNetEngine(H, H, B, F0=F0, k=3, device=dev, dtype=dtype)
    g = torch.Generator(); g.manual_seed(0)
    eng.reset_parameters(g)
    eng.dropout_seed = 4321
    return eng


def rel(a, b):
    a, b = a.dense().float(), b.dense().float()
    c = min(a.shape[-1], b.shape[-1])          # the 2-channel ends are padded to 4 (fp32) / 8 (bf16) stored channels
    a, b = a[..., :c], b[..., :c]
    return float((a - b).norm() / (b.norm() + 1e-30))


spec_in, emb, spec_out = bench.synthetic_batch(B, H, H, dev, 1234)
e32, e16 = build("f32"), build("bf16")
e16.load_keras_params(e32.export_keras_params())
res = {"forward": [], "backward": [], "vec": {}}
for flag in (False,):
    for eng in (e32, e16):
        eng._shared["dropout_step"] = 0
        eng.training = True
        mask = eng.make_dropout_mask()
        eng.forward(spec_in, emb, dropout_mask=mask, target=spec_out, global_batch=B)
        eng.backward()
    torch.cuda.synchronize()
    g32, g16 = e32.export_keras_grads(), e16.export_keras_grads()
    res["vec"]["fp32 dL/dz" if flag else "bf16 dL/dz"] = {
        n: [float((g16[n].double() - g32[n].double()).norm() / g32[n].double().norm()),
            float((g16[n].double() * g32[n].double()).sum() / (g16[n].double().norm() * g32[n].double().norm()))]
        for n in g32 if n.startswith("vec.")}
D, L = e32.depth, e32.L
fw = [("x4", e32.x4, e16.x4)]
for l in range(1, L + 1):
    fw += [(f"enc{l}.down", e32.down[l], e16.down[l]), (f"enc{l}.cb1 conv", e32.y[l], e16.y[l]), (f"enc{l}.cb1 out", e32.a[l], e16.a[l])]
fw.append(("z (bottleneck)", e32.z, e16.z))
for l in range(D, 0, -1):
    c = e32.ch[l - 1]
    fw += [(f"dec{l}.up", e32.cat[l].slice(c, c), e16.cat[l].slice(c, c)), (f"dec{l}.cb1a conv", e32.ya[l], e16.ya[l]),
           (f"dec{l}.cb1a out", e32.aa[l], e16.aa[l]), (f"dec{l}.cb1b conv", e32.yb[l], e16.yb[l]), (f"dec{l}.cb1b out", e32.ab[l], e16.ab[l])]
fw.append(("logits", e32.logits, e16.logits))
for n, a, b in fw:
    res["forward"].append([n, rel(b, a)])
res["pred"] = {"max": float((e16.pred - e32.pred).abs().max()), "rms": float((e16.pred - e32.pred).pow(2).mean().sqrt())}
bw = [("dL/dlogits", e32.g_logits, e16.g_logits)]
for l in range(1, D + 1):
    c = e32.ch[l - 1]
    bw += [(f"d dec{l}.cb1b out", e32.g_ab[l], e16.g_ab[l]), (f"d dec{l}.cb1b conv", e32.g_yb[l], e16.g_yb[l]),
           (f"d dec{l}.cb1a out", e32.g_aa[l], e16.g_aa[l]), (f"d dec{l}.cb1a conv", e32.g_ya[l], e16.g_ya[l]),
           (f"d dec{l}.up", e32.g_cat[l].slice(c, c), e16.g_cat[l].slice(c, c))]
bw.append(("d z", e32.g_z, e16.g_z))
for l in range(L, 0, -1):
    bw += [(f"d enc{l}.cb1 conv", e32.g_y[l], e16.g_y[l]), (f"d enc{l}.down", e32.g_down[l], e16.g_down[l])]
    if l > 1:
        c = e32.ch[l - 2]
        bw.append((f"d enc{l - 1}.cb1 out (skip + down)", e32.g_cat[l - 1].slice(0, c), e16.g_cat[l - 1].slice(0, c)))
for n, a, b in bw:
    res["backward"].append([n, rel(b, a)])
for k in ("forward", "backward"):
    print(k)
    for n, r in res[k]:
        print(f"  {n:36s} {r:.4f}")
print("prediction", res["pred"])
print(json.dumps(res["vec"], indent=1))
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
