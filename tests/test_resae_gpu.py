"""ResAE (dl_models/res_ae.py, BASELINE.json configs[4]) on the HIP kernels against the CPU oracle (fp64): prediction, loss and
every gradient.  Tolerances as for the U-Net (tests/test_model_gpu.py)."""
import numpy as np
import pytest
import torch

from oracle import detrand, torch_ref as R, torch_resae as RA

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def U():
    import unet_rir_amd
    unet_rir_amd._lib.lib()
    return unet_rir_amd


def run(U, H, W, filters, B, latent, n_neurons, dropout):
    cfg = RA.ResAEConfig(H, W, filters, (3,) * len(filters), (2,) * len(filters), latent, n_neurons)
    Pn = RA.init_params(cfg, randomize_all=True, dtype=np.float64)
    spec_in, emb, spec_out = R.synthetic_batch(R.Config(H, W), B)
    h, w, c = cfg.bottleneck_shape()
    ml = md = None
    if dropout:
        ml = (detrand.uniform("ml", (B, latent)) >= 0.3).astype(np.float64) / 0.7
        md = (detrand.uniform("md", (B, h * w * c)) >= 0.3).astype(np.float64) / 0.7
    loss, dl, pred, grads = RA.loss_and_grads(Pn, spec_in, emb, spec_out, cfg, 0.9, B, 1, ml, md)
    eng = U.ResAEEngine(H, W, B, filters, (3,) * len(filters), (2,) * len(filters), latent, n_neurons, device=DEV)
    eng.load_keras_params(Pn)
    t = lambda a, dt=None: None if a is None else torch.tensor(a, dtype=dt).to(DEV)
    eng.forward(t(spec_in), t(emb), t(ml, torch.float32), t(md, torch.float32), target=t(spec_out), global_batch=B)
    eng.backward()
    eng.reg_loss()
    torch.cuda.synchronize()
    return cfg, eng, (loss, dl, pred, grads)


@pytest.mark.parametrize("H,W,filters,B,latent,nn,do", [(32, 32, (8, 8, 16, 16), 2, 8, 16, False),
                                                         (64, 48, (8, 16, 32, 64), 2, 32, 64, True)])
def test_resae_forward_backward_vs_oracle(U, H, W, filters, B, latent, nn, do):
    cfg, eng, (loss, dl, pred, grads) = run(U, H, W, filters, B, latent, nn, do)
    assert float((eng.pred.double().cpu() - pred).abs().max()) <= 1e-4
    got = float(eng.loss_out[0]) + float(eng.reg_out[0])
    assert abs(float(eng.loss_out[0]) - dl) <= 1e-5 * abs(dl)
    assert abs(got - loss) <= 1e-5 * abs(loss), (got, loss)
    kg = eng.export_keras_grads()
    assert set(kg) == set(grads)
    floor = 1e-6 * max(float(g.abs().max()) for g in grads.values())
    for n, g_ref in grads.items():
        g = kg[n].double()
        e = float((g - g_ref).abs().max())
        # biases in front of a BatchNorm: analytically zero gradient (the product writes an exact 0)
        assert e <= 1e-3 * float(g_ref.abs().max()) + floor, (n, e, float(g_ref.abs().max()))
    # a step of Adam moves the loss down and keeps the padded weights at zero
    l0 = got
    for _ in range(10):
        eng.adam_step(1e-3)
        eng.forward(*[torch.tensor(a).to(DEV) for a in R.synthetic_batch(R.Config(H, W), B)[:2]],
                    target=torch.tensor(R.synthetic_batch(R.Config(H, W), B)[2]).to(DEV), global_batch=B)
        eng.backward()
    torch.cuda.synchronize()
    assert float(eng.loss_out[0]) < l0
    assert float(eng.p["e_res_1_conv.1.kernel"][..., 2:].abs().max()) == 0.0
    assert float(eng.p["d_out.kernel"][..., 2:].abs().max()) == 0.0


def test_resae_param_count_cfg5(U):
    """main_training.py:132-141 configuration at 256x256: ~17.2 M parameters."""
    cfg = RA.ResAEConfig(256, 256)
    assert sum(int(np.prod(s)) for s in RA.param_shapes(cfg).values()) == 17_173_922


@pytest.mark.parametrize("kind,dtype", [("resae", "bf16"), ("resae", "f32"), ("unet3", "bf16"), ("ae", "f32")])
def test_graph_engine_side_stream_schedule_is_bit_identical(U, kind, dtype):
    """overlap_wgrad=True (weight gradients on a side stream, buckets handed over from it, Adam bucket by bucket on a third)
    runs the SAME launches in another order across streams: three Trainer steps give bit-identical parameters, moving
    statistics and loss as the single-stream schedule (the race screen of the schedule itself is tests/test_schedule_sim.py)."""
    H = W = 64
    B = 4

    def build(overlap):
        if kind == "resae":
            eng = U.ResAEEngine(H, W, B, (8, 16, 32, 64), (3, 3, 3, 3), (2, 2, 2, 2), 32, 64, device=DEV, dtype=dtype, overlap_wgrad=overlap)
        elif kind == "ae":
            eng = U.AutoencoderEngine(H, W, B, (8, 16, 32, 64), (3, 3, 3, 3), (2, 2, 2, 2), 32, 64, device=DEV, dtype=dtype, overlap_wgrad=overlap)
        else:
            eng = U.UNetGraphEngine(H, W, B, F0=8, k=3, mode=3, device=DEV, dtype=dtype, overlap_wgrad=overlap)
        g = torch.Generator(); g.manual_seed(3)
        eng.reset_parameters(g)
        eng.dropout_seed = 77
        return eng, U.Trainer(eng, lr=1e-3, bucket_bytes=16 << 10)

    gen = torch.Generator(); gen.manual_seed(5)
    spec_in = torch.rand((B, 2, H, W), generator=gen).to(DEV)
    spec_out = torch.rand((B, 2, H, W), generator=gen).to(DEV)
    emb = torch.randint(26, 1282, (B, 2, 16), generator=gen).to(DEV)
    res = []
    for overlap in (False, True):
        eng, tr = build(overlap)
        assert (eng.wg_stream is not None) == overlap
        if overlap:
            assert tr.adam_stream is not None and len(tr.bucketer.bounds) > 3
        losses = [tr.step(spec_in, emb, spec_out, return_loss=True) for _ in range(3)]
        torch.cuda.synchronize()
        res.append((losses, eng.theta.clone(), eng.adam_m.clone(), {k: v.clone() for k, v in eng.moving.items()}))
    (l0, t0, m0, mv0), (l1, t1, m1, mv1) = res
    assert l0 == l1
    assert torch.equal(t0, t1) and torch.equal(m0, m1)
    for k in mv0:
        assert torch.equal(mv0[k], mv1[k]), k
