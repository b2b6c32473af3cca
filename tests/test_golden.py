"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the fp64 oracle).
CPU: the fp32 oracle reproduces them (pins the oracle).  GPU: the HIP engine reproduces them through the C ABI.
PARITY UNPINNED w.r.t. the TensorFlow reference (it cannot run here and has no fixtures of its own)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as MG  # noqa: E402
from oracle import torch_ref as R  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


@pytest.mark.parametrize("name", list(MG.CASES))
def test_oracle_fp32_matches_golden(name):
    gold = load(name)
    got = MG.compute(name, torch.float32)
    assert abs(got["loss"] - gold["loss"]) <= 1e-5 * abs(gold["loss"])
    assert float(np.abs(got["pred"] - gold["pred"]).max()) <= 1e-4
    gmax = max(float(v) for k, v in gold.items() if k.startswith("gnorm/"))
    for k, v in gold.items():
        if k.startswith("gnorm/"):      # biases in front of BatchNorm have an analytically zero gradient: absolute floor
            assert abs(got[k] - v) <= 1e-3 * v + 1e-6 * gmax, k


@pytest.mark.parametrize("name", list(MG.CASES))
def test_oracle_fp64_is_bit_stable(name):
    """Regenerating with the generating script gives the stored numbers (detrand inputs are platform independent)."""
    gold = load(name)
    got = MG.compute(name, torch.float64)
    assert abs(got["loss"] - gold["loss"]) <= 1e-12
    for k, v in gold.items():
        if k.startswith("ghead/"):
            np.testing.assert_allclose(got[k], v, rtol=1e-9, atol=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(MG.CASES))
def test_hip_engine_matches_golden(name):
    import unet_rir_amd as U
    gold = load(name)
    cfg, params, spec_in, emb, spec_out, mask, gb, nrep = MG.case_inputs(name)
    H, W, F0, B = MG.CASES[name][:4]
    eng = U.UNetEngine(H, W, B, F0=F0, k=3, device="cuda:0", n_replicas=nrep)
    eng.load_keras_params(params)
    dev = "cuda:0"
    t_mask = None if mask is None else torch.tensor(mask, dtype=torch.float32).to(dev)
    eng.forward(torch.tensor(spec_in).to(dev), torch.tensor(emb).to(dev), dropout_mask=t_mask,
                target=torch.tensor(spec_out).to(dev), global_batch=gb, alpha=0.9)
    eng.backward()
    eng.reg_loss()
    torch.cuda.synchronize()
    loss = float(eng.loss_out[0]) + float(eng.reg_out[0])
    assert abs(loss - gold["loss"]) <= 1e-5 * abs(gold["loss"])
    assert abs(float(eng.loss_out[0]) - gold["data_loss"]) <= 1e-5 * abs(gold["data_loss"])
    assert float(np.abs(eng.pred.cpu().numpy() - gold["pred"]).max()) <= 1e-4
    kg = eng.export_keras_grads()
    gmax = max(float(v) for k, v in gold.items() if k.startswith("gnorm/"))
    for n, g in kg.items():
        g = g.double()
        # 1e-2 (not 1e-3) in the max norm: a ReLU input within an fp32 ulp of zero may take the other branch than in
        # the fp64 oracle and shift individual entries upstream of it (see tests/test_model_gpu.py docstring)
        assert abs(float(g.norm()) - gold[f"gnorm/{n}"]) <= 1e-2 * gold[f"gnorm/{n}"] + 1e-6 * gmax, n
        head = g.flatten()[:8].numpy()
        assert float(np.abs(head - gold[f"ghead/{n}"]).max()) <= 1e-2 * float(g.abs().max()) + 1e-6 * gmax, n
    eng.adam_step(MG.LR)
    torch.cuda.synchronize()
    kp = eng.export_keras_params()
    for n, p in kp.items():
        # parameters whose gradient is rounding noise (biases in front of BatchNorm) move by +-lr at random sign
        if n.endswith(("cb1.bias", "cb1a.bias", "cb1b.bias")):
            continue
        # the first Adam step moves every weight by ~lr*sign(g): where g is rounding noise the sign (hence 2*lr of the sum)
        # is not reproducible, so allow 0.1% of the entries to differ that way
        tol = 2 * MG.LR * max(4.0, 1e-3 * p.numel()) + 1e-4 * abs(gold[f"psum/{n}"])
        assert abs(float(p.double().sum()) - gold[f"psum/{n}"]) <= tol, n
