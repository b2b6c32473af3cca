"""bf16 storage on the graph engines (ResAE = BASELINE.json configs[4] "on the same HIP conv kernels", Autoencoder, U-Net
feature-block modes 1-3).  The storage model is emulated exactly by running the SAME product graph on the simulated
runtime with the oracle-backed CPU kernels (tests/cpu_ops.py): bf16 buffers are rounded where the product stores them,
bf16 work copies of the kernels are what the matrix cores see, arithmetic is fp64.  The HIP path must agree with that
emulation on the forward pass and be as close to the EXACT (fp64, unrounded) gradient as the emulation is - the criterion of
tests/test_model_gpu.py::test_bf16_gradients_as_accurate_as_the_storage_model_allows."""
import os
import sys

import numpy as np
import pytest
import torch
from _pytest.monkeypatch import MonkeyPatch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import torch_ae as AE, torch_ref as R, torch_resae as RA  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H = W = 64          # bottleneck 4 x 4: BatchNorm at the deepest level still averages over 64 samples
B = 4


def _case(kind):
    if kind == "resae":
        cfg = RA.ResAEConfig(H, W, (8, 16, 16, 32), (3, 3, 3, 3), (2, 2, 2, 2), 8, 16)
        params = RA.init_params(cfg, randomize_all=True, dtype=np.float64)
        exact = lambda a, e, b: RA.loss_and_grads(params, a, e, b, cfg, 0.9, B, 1)
        make = lambda U, **kw: U.ResAEEngine(H, W, B, cfg.conv_filters, cfg.conv_kernels, cfg.conv_strides, cfg.latent_space_dim,
                                             cfg.n_neurons, dtype="bf16", **kw)
    elif kind == "ae":
        cfg = AE.AEConfig(H, W, (8, 16, 16, 32), (3, 3, 3, 3), (2, 2, 2, 2), 8, 16)
        params = AE.init_params(cfg, randomize_all=True, dtype=np.float64)
        exact = lambda a, e, b: AE.loss_and_grads(params, a, e, b, cfg, 0.9, B, 1)
        make = lambda U, **kw: U.AutoencoderEngine(H, W, B, cfg.conv_filters, cfg.conv_kernels, cfg.conv_strides, cfg.latent_space_dim,
                                                   cfg.n_neurons, dtype="bf16", **kw)
    else:
        mode = int(kind[-1])
        cfg = R.Config(H, W, 8, 3, mode=mode)
        params = R.init_params(cfg, randomize_all=True, dtype=np.float64)
        exact = lambda a, e, b: R.loss_and_grads(params, a, e, b, cfg, dtype=torch.float64)
        make = lambda U, **kw: U.UNetGraphEngine(H, W, B, F0=8, k=3, mode=mode, dtype="bf16", **kw)
    return params, exact, make


def _run(eng, params, batch, dev):
    eng.load_keras_params(params)
    t = lambda a: torch.tensor(a).to(dev)
    eng.forward(t(batch[0]), t(batch[1]), target=t(batch[2]), global_batch=B)
    eng.backward()
    eng.reg_loss()
    return (eng.pred.double().cpu().clone(), float(eng.loss_out[0]) + float(eng.reg_out[0]),
            {k: v.double() for k, v in eng.export_keras_grads().items()})


@pytest.mark.parametrize("kind", ["resae", "ae", "graph1", "graph3"])
def test_bf16_graph_engine_against_storage_emulation_and_exact_gradients(kind):
    import unet_rir_amd as U
    import cpu_ops
    from sim_runtime import SimRuntime
    U._lib.lib()
    params, exact, make = _case(kind)
    batch = R.synthetic_batch(R.Config(H, W), B)
    # the storage emulation: the product graph on CPU tensors, fp64 arithmetic, bf16 buffers
    mp = MonkeyPatch()
    try:
        rt = SimRuntime()
        cpu_ops.install(mp, rt)
        pred_q, loss_q, g_q = _run(make(U, device="cpu", runtime=rt), params, batch, "cpu")
    finally:
        mp.undo()
    pred_h, loss_h, g_h = _run(make(U, device=DEV), params, batch, DEV)
    torch.cuda.synchronize()
    loss_x, _, pred_x, g_x = exact(*batch)
    # forward: single 1-ulp rounding decisions (fp32 vs fp64 accumulation landing on either side of a bf16 boundary) propagate
    # through ~50 BatchNorm layers, so the HIP prediction is compared with the EXACT one and must be as close to it as the
    # storage emulation is (relative L2 error <= 2x + 1 %), with a loose element-wise sanity bound
    nx = float(pred_x.norm())
    e_h, e_q = float((pred_h - pred_x).norm()) / nx, float((pred_q - pred_x).norm()) / nx
    print(f"{kind}: pred rel L2 error hip {e_h:.2e} emulation {e_q:.2e}; max |hip - emulation| {float((pred_h - pred_q).abs().max()):.2e}; "
          f"loss hip {loss_h:.6f} emulation {loss_q:.6f} exact {loss_x:.6f}")
    assert e_h <= 2.0 * e_q + 0.01, (e_h, e_q)
    assert float((pred_h - pred_q).abs().max()) <= 0.15
    assert abs(loss_h - loss_x) <= 2.0 * abs(loss_q - loss_x) + 1e-2 * abs(loss_x), (loss_h, loss_q, loss_x)
    # gradients: as accurate as the storage model allows
    checked = 0
    gmax = max(float(g.abs().max()) for g in g_x.values())
    for n, gx in g_x.items():
        if float(gx.abs().max()) < 1e-6 * gmax:
            continue                               # analytically zero (biases in front of a BatchNorm)
        nx = float(gx.norm()) + 1e-30
        e_h = float((g_h[n] - gx).norm()) / nx
        e_q = float((g_q[n] - gx).norm()) / nx
        assert e_h <= 2.0 * e_q + 0.03, (n, e_h, e_q)
        checked += 1
    assert checked > 20
