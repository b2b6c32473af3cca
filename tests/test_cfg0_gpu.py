"""BASELINE.json configs[0] at its EXACT size on the HIP path: UNet 4 down / 4 up, number_filters_0 = 16, batch 4 of [2,256,256]
(the reference's own CPU-runnable case; bench.py times the oracle on it as `cpu_baseline.cfg1`).  The fixtures under tests/golden are
a 64 x 64 cut of it; at 256 x 256 the 16 / 32 / 64-channel levels run on kernels and grids no other test launches (bf16: the register-
staged conv3x3r where C % 32 != 0, conv3x3s<32>, the C % 32 != 0 fall-backs of the stride-2 pair on 256^2 ... 16^2 grids).  The fp64
oracle does this step in well under a minute, so the WHOLE network is compared: prediction, loss, all 77 gradients.

fp32 storage: prediction atol 1e-4, loss rtol 1e-5; gradients: the small-size bounds (max-norm 1e-3, relative L2 1e-4) or - where fp32
arithmetic itself is further from fp64 at this size - within a small factor of what the fp32 evaluation of the same oracle loses (see
the assertion; the table is profiles/r04_cfg0_fp32_table.json).  bf16 storage: the storage-model criterion of test_bf16_gradients_as_accurate_as_the_storage_model_allows."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H = W = 256
F0, B = 16, 4


@pytest.fixture(scope="module")
def U():
    import unet_rir_amd
    unet_rir_amd._lib.lib()
    return unet_rir_amd


@pytest.fixture(scope="module")
def case():
    cfg = R.Config(H, W, F0, 3)
    Pn = R.init_params(cfg, randomize_all=True, dtype=np.float64)
    spec_in, emb, spec_out = R.synthetic_batch(cfg, B, seed_name="cfg0")
    inter = {}
    loss, dl, pred, grads = R.loss_and_grads(Pn, spec_in, emb, spec_out, cfg, 0.9, B, 1, None, torch.float64, inter)
    pre = {n: v.detach() for n, v in inter.items() if n.endswith(".pre")}
    return cfg, Pn, (spec_in, emb, spec_out), (loss, dl, pred.detach(), grads), pre


def _run(U, Pn, batch, dtype):
    spec_in, emb, spec_out = batch
    eng = U.UNetEngine(H, W, B, F0=F0, k=3, device=DEV, dtype=dtype)
    eng.load_keras_params(Pn)
    t = lambda a: torch.tensor(a).to(DEV)
    eng.training = True
    eng.forward(t(spec_in), t(emb), target=t(spec_out), global_batch=B, alpha=0.9)
    eng.backward()
    eng.reg_loss()
    torch.cuda.synchronize()
    return eng


def test_configs0_exact_size_fp32_against_the_oracle(U, case):
    cfg, Pn, batch, (loss, dl, pred, grads), pre = case
    assert (cfg.H, cfg.W, cfg.F0, cfg.depth) == (256, 256, 16, 4) and batch[0].shape == (4, 2, 256, 256)
    eng = _run(U, Pn, batch, "f32")
    assert float((eng.pred.double().cpu() - pred).abs().max()) <= 1e-4
    assert abs(float(eng.loss_out[0]) - dl) <= 1e-5 * abs(dl)
    assert abs(float(eng.loss_out[0]) + float(eng.reg_out[0]) - loss) <= 1e-5 * abs(loss)
    # What does fp32 arithmetic cost at this size?  Measured, not assumed (scripts/cfg0_fp32_table.py, profiles/r04_cfg0_fp32_table.json):
    # the SAME oracle evaluated in fp32 (torch CPU) is 8.9e-5 from its fp64 evaluation over the whole gradient, but 2e-3 ... 3.5e-3 in
    # relative L2 (up to 1.2e-2 in max-norm) on the deep tensors (enc4 / enc5 / dec4 / vec.*: BatchNorm backward at 16 x 16 and 32 x 32
    # subtracts nearly equal sums, and 46 of the 1.7e7 ReLU inputs lie within 2e-6 of zero).  The HIP engine (sequential fp32 MFMA
    # chains, fp64 BatchNorm sums) is a second fp32 evaluation of the same step: observed 1.2e-4 over the whole gradient and, per
    # tensor, 1.0 ... 2.0 x the fp32 oracle's relative L2 (worst: enc5.cb1.kernel 4.9e-3 against 3.4e-3) and up to 5.4 x its max-norm.
    # Criterion per tensor: the small-size bounds (max-norm 1e-3, relative L2 1e-4), or within 3 x (L2) / 8 x (max-norm) of what the
    # fp32 oracle itself loses; whole gradient within 2 x; nothing beyond 1e-2.
    spec_in, emb, spec_out = batch
    _, _, _, g_cpu32 = R.loss_and_grads(Pn, spec_in, emb, spec_out, cfg, 0.9, B, 1, None, torch.float32, None)
    kg = eng.export_keras_grads()
    floor = 1e-6 * max(float(g_.abs().max()) for g_ in grads.values())
    assert len(grads) == 77
    worst, problems = (0.0, None), []
    for n, g_ref in grads.items():
        g = kg[n].double()
        scale, nrm = float(g_ref.abs().max()), float(g_ref.norm()) + 1e-30
        e = float((g - g_ref).abs().max())
        if e <= floor:
            continue                              # (numerically) zero gradient: biases in front of a BatchNorm
        l2 = float((g - g_ref).norm()) / nrm
        c32 = g_cpu32[n].double()
        e32, l2_32 = float((c32 - g_ref).abs().max()), float((c32 - g_ref).norm()) / nrm
        if e > max(1e-3 * scale, 8.0 * e32) + floor:
            problems.append(f"{n}: max err {e / scale:.3e} of its scale, fp32 oracle {e32 / scale:.3e}")
        if l2 > max(1e-4, 3.0 * l2_32):
            problems.append(f"{n}: relative L2 {l2:.3e}, fp32 oracle {l2_32:.3e}")
        if l2 > worst[0]:
            worst = (l2, n)
    assert not problems, "; ".join(problems)
    assert worst[0] <= 1e-2, worst
    names = [n for n in grads if float(grads[n].norm()) > 1e-12]
    ref = torch.cat([grads[n].flatten() for n in names])
    hip = torch.cat([kg[n].double().flatten() for n in names])
    c32 = torch.cat([g_cpu32[n].double().flatten() for n in names])
    assert float((hip - ref).norm()) <= 2.0 * float((c32 - ref).norm()), (float((hip - ref).norm() / ref.norm()), float((c32 - ref).norm() / ref.norm()))


def test_configs0_exact_size_bf16_against_the_storage_model(U, case):
    cfg, Pn, batch, (loss, dl, pred, g_true), _ = case
    spec_in, emb, spec_out = batch
    loss_q, dl_q, pred_q, g_q = R.loss_and_grads(Pn, spec_in, emb, spec_out, cfg, 0.9, B, 1, None, torch.float64, None, storage="bf16")
    eng = _run(U, Pn, batch, "bf16")
    # the kernels that serve this size
    ops = U.ops
    served = {n: ops.conv3x3_kernel(eng.geo[n], 0, x) for n, x in (("enc1.cb1", eng.down[1]), ("enc2.cb1", eng.down[2]), ("enc3.cb1", eng.down[3]),
                                                                   ("dec1.cb1a", eng.cat[1]), ("dec2.cb1a", eng.cat[2]))}
    assert served["enc1.cb1"] == "conv3x3r" and served["enc2.cb1"] == "conv3x3s", served       # 16 -> 16 @256^2: C % 32 != 0; 32 -> 32 @128^2
    assert float((eng.pred.double().cpu() - pred_q.detach()).abs().max()) <= 2e-2
    assert abs(float(eng.loss_out[0]) - dl_q) <= 1e-3 * abs(dl_q), (float(eng.loss_out[0]), dl_q)
    assert abs(float(eng.loss_out[0]) - dl) <= 2e-3 * abs(dl), (float(eng.loss_out[0]), dl)
    kg = eng.export_keras_grads()
    checked, worst = 0, (None, 0.0)
    for n, gt in g_true.items():
        if n.endswith(("cb1.bias", "cb1a.bias", "cb1b.bias")):
            assert float(kg[n].abs().max()) == 0.0, n          # analytically zero: exact zeros
            continue
        nt = float(gt.norm()) + 1e-30
        e_hip = float((kg[n].double() - gt).norm()) / nt
        e_orc = float((g_q[n] - gt).norm()) / nt
        assert e_hip <= 2.0 * e_orc + 0.02, (n, e_hip, e_orc)
        if e_hip > worst[1]:
            worst = (n, e_hip)
        checked += 1
    assert checked == 77 - 13
