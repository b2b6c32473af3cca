"""The drop-in boundary (unet_rir_amd.UNet / ResAE, the classes main_training.py:130-161 constructs) against the oracle.

Every test runs twice: on the simulated runtime with oracle-backed kernels (CPU, `sim`: checks the module / trainer / engine
plumbing and its stream dependencies) and on the MI355X through libunetrir.so (`hip`, marked gpu: the same assertions on
the HIP kernels)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import torch_ae as AE, torch_ref as R, torch_resae as RA  # noqa: E402

H = W = 16
F0 = 8


class Env:
    def __init__(self, kind, monkeypatch):
        import unet_rir_amd as U
        self.U, self.kind = U, kind
        if kind == "sim":
            import cpu_ops
            from sim_runtime import SimRuntime
            self.rt = SimRuntime()
            cpu_ops.install(monkeypatch, self.rt)
            self.device = "cpu"
        else:
            U._lib.lib()
            self.rt, self.device = None, "cuda:0"

    def t(self, a, dtype=None):
        return torch.tensor(np.asarray(a), dtype=dtype).to(self.device)

    def unet(self, B, **kw):
        return self.U.UNet((H, W, 2), (2, 16), number_filters_0=F0, kernels=3, batch_size=B, device=self.device, runtime=self.rt, **kw)

    def sync(self):
        if self.kind == "hip":
            torch.cuda.synchronize()


@pytest.fixture(params=["sim", pytest.param("hip", marks=pytest.mark.gpu)])
def env(request, monkeypatch):
    return Env(request.param, monkeypatch)


def _load(model, cfg, mode=0):
    params = R.init_params(cfg, randomize_all=True, dtype=np.float64)
    model.engine.load_keras_params(params)
    return params


def test_reference_shaped_training_loop_equals_trainer_step_and_oracle(env):
    """main_training.py:253-290 written against the boundary: model.model([spec_in, emb], training=True) -> compute_loss ->
    backward -> apply_gradients, NHWC tensors as the reference's generator yields them."""
    B = 2
    cfg = R.Config(H, W, F0, 3)
    spec_in, emb, spec_out = R.synthetic_batch(cfg, B)
    model = env.unet(B, dropout=False)
    params = _load(model, cfg)
    tr = env.U.Trainer(model, lr=1e-3, dropout=False)
    nhwc = lambda a: env.t(a).permute(0, 2, 3, 1).contiguous()
    pred = model.model([nhwc(spec_in), env.t(emb)], training=True)
    assert tuple(pred.shape) == (B, H, W, 2)
    loss = tr.compute_loss(nhwc(spec_out), pred)
    loss.backward()
    tr.apply_gradients()
    env.sync()
    want_loss, _, want_pred, grads = R.loss_and_grads(params, spec_in, emb, spec_out, cfg, dtype=torch.float64)
    assert abs(float(loss) - want_loss) <= 1e-5 * abs(want_loss)
    assert float((pred.permute(0, 3, 1, 2).double().cpu() - want_pred).abs().max()) <= 1e-4
    got = model.engine.export_keras_grads()
    for n, g in grads.items():
        e = float((got[n].double() - g).abs().max())
        assert e <= 1e-3 * float(g.abs().max()) + 1e-9, (n, e)
    # the parameters moved exactly as one Adam step of the oracle moves them
    for n, g in grads.items():
        new, _, _ = R.adam_update(torch.tensor(params[n]), g, torch.zeros_like(g), torch.zeros_like(g), 1, 1e-3)
        e = float((model.engine.export_keras_params()[n].double() - new).abs().max())
        assert e <= 2e-5, (n, e)


def test_torch_loss_and_torch_optimizer_drive_the_engine(env):
    """Any torch loss on the prediction + any torch optimizer: gradients reach `.grad` as views of the engine's flat gradient
    buffer (no copies), the l2 terms are folded by the engine, the optimizer's in-place update is seen by the next forward."""
    B = 2
    cfg = R.Config(H, W, F0, 3)
    spec_in, emb, spec_out = R.synthetic_batch(cfg, B)
    model = env.unet(B, dropout=False)
    params = _load(model, cfg)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    plist = list(model.parameters())
    assert len(plist) == len(model.engine.specs)
    pred = model(env.t(spec_in), env.t(emb))
    loss = ((pred - env.t(spec_out)) ** 2).mean() + sum(model.model.losses)
    opt.zero_grad()
    loss.backward()
    env.sync()
    # oracle: the same loss through autograd
    P = R.to_torch(params, torch.float64, True)
    p_or = R.forward(P, torch.tensor(spec_in, dtype=torch.float64), torch.tensor(emb), cfg, True)
    l_or = ((p_or - torch.tensor(spec_out, dtype=torch.float64)) ** 2).mean() + R.reg_loss(P, cfg, 1)
    l_or.backward()
    assert abs(float(loss) - float(l_or)) <= 1e-5 * abs(float(l_or))
    got = model.engine.export_keras_grads()
    for n, p in P.items():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        e = float((got[n].double() - g).abs().max())
        assert e <= 1e-3 * float(g.abs().max()) + 1e-9, (n, e)
    named = model.named_engine_parameters()
    assert all(named[n].grad.data_ptr() == model.engine.g[n].data_ptr() for n in named)      # views, not clones
    theta0 = model.engine.theta.clone()
    opt.step()
    assert not torch.equal(theta0, model.engine.theta)
    # an in-place optimizer step must invalidate the transposed work copies (Conv2DTranspose forward reads them)
    model.eval()
    with torch.no_grad():
        p_eval = model(env.t(spec_in), env.t(emb)).clone()
    fresh = env.unet(B, dropout=False)
    fresh.load_state_dict(model.state_dict())
    fresh.eval()
    with torch.no_grad():
        p_fresh = fresh(env.t(spec_in), env.t(emb))
    env.sync()
    assert float((p_eval - p_fresh).abs().max()) <= 1e-6


def test_batch_size_change_keeps_parameters_and_optimizer_alive(env):
    """train at B=2, predict at B=1, train at B=2 again - same Parameter objects, the optimizer still moves the
    live model, the B=1 engine aliases the same variables."""
    cfg = R.Config(H, W, F0, 3)
    model = env.unet(2, dropout=False)
    _load(model, cfg)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    ids = [id(p) for p in model.parameters()]
    spec_in, emb, spec_out = R.synthetic_batch(cfg, 2)

    def train_once():
        model.train()
        opt.zero_grad()
        pred = model(env.t(spec_in), env.t(emb))
        ((pred - env.t(spec_out)) ** 2).mean().backward()
        before = model.engine.theta.clone()
        opt.step()
        env.sync()
        return float((model.engine.theta - before).abs().max())

    assert train_once() > 0
    out1 = model.predict_stft([env.t(spec_in[:1]).permute(0, 2, 3, 1).contiguous(), env.t(emb[:1])])
    assert tuple(out1.shape) == (1, H, W, 2)
    assert [id(p) for p in model.parameters()] == ids
    assert model._engines[1].theta.data_ptr() == model.engine.theta.data_ptr()
    assert train_once() > 0
    # and the B=1 prediction equals image 0 of a B=2 eval forward (inference BatchNorm is per-pixel)
    model.eval()
    with torch.no_grad():
        out2 = model(env.t(spec_in), env.t(emb)).clone()
        out1b = model(env.t(spec_in[:1]), env.t(emb[:1]))
    env.sync()
    assert float((out2[:1] - out1b).abs().max()) <= 1e-5


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_feature_block_modes_through_the_boundary_class(env, mode):
    """UNet(mode=1|2|3) (dl_models/u_net.py:280-287) routes to the graph engine; Trainer drives it."""
    B = 2
    cfg = R.Config(H, W, F0, 3, mode=mode)
    spec_in, emb, spec_out = R.synthetic_batch(cfg, B)
    model = env.unet(B, mode=mode, dropout=False)
    params = _load(model, cfg)
    tr = env.U.Trainer(model, lr=1e-3, dropout=False)
    got_loss = tr.step(env.t(spec_in), env.t(emb), env.t(spec_out), return_loss=True)
    env.sync()
    want_loss, _, _, grads = R.loss_and_grads(params, spec_in, emb, spec_out, cfg, dtype=torch.float64)
    assert abs(got_loss - want_loss) <= 1e-5 * abs(want_loss)
    got = model.engine.export_keras_grads()
    for n, g in grads.items():
        e = float((got[n].double() - g).abs().max())
        assert e <= 1e-3 * float(g.abs().max()) + 1e-9, (n, e)


def test_resae_boundary_class(env):
    """ResAE(input_shape, inf_vector_shape, conv_filters, conv_kernels, conv_strides, latent_space_dim, n_neurons, name)
    (dl_models/res_ae.py:41-50): .model, .encoder / .decoder, Trainer."""
    B = 2
    cfg = RA.ResAEConfig(H, W, (8, 8, 8, 16), (3, 3, 3, 3), (2, 2, 2, 2), 8, 16)
    params = RA.init_params(cfg, randomize_all=True, dtype=np.float64)
    spec_in, emb, spec_out = R.synthetic_batch(R.Config(H, W), B)
    model = env.U.ResAE((H, W, 2), (2, 16), cfg.conv_filters, cfg.conv_kernels, cfg.conv_strides, cfg.latent_space_dim,
                        cfg.n_neurons, name="resae", batch_size=B, device=env.device, runtime=env.rt, dropout=False)
    model.engine.load_keras_params(params)
    tr = env.U.Trainer(model, lr=1e-3, dropout=False)
    nhwc = lambda a: env.t(a).permute(0, 2, 3, 1).contiguous()
    pred = model.model([nhwc(spec_in), env.t(emb)], training=True)
    loss = tr.compute_loss(nhwc(spec_out), pred)
    want_loss, _, want_pred, grads = RA.loss_and_grads(params, spec_in, emb, spec_out, cfg, 0.9, B, 1)
    env.sync()
    assert abs(float(loss) - want_loss) <= 1e-5 * abs(want_loss)
    assert float((pred.permute(0, 3, 1, 2).double().cpu() - want_pred).abs().max()) <= 1e-4
    loss.backward()
    env.sync()
    got = model.engine.export_keras_grads()
    floor = 1e-6 * max(float(g.abs().max()) for g in grads.values())
    for n, g in grads.items():
        e = float((got[n].double() - g).abs().max())
        assert e <= 1e-3 * float(g.abs().max()) + floor, (n, e)
    tr.apply_gradients()
    # decoder(encoder(x)) == model(x) with training=False (dl_models/res_ae.py:150-152)
    z = model.encoder([nhwc(spec_in), env.t(emb)])
    assert tuple(z.shape) == (B, cfg.latent_space_dim)
    rec = model.decoder(z).clone()
    full = model.predict_stft([nhwc(spec_in), env.t(emb)])
    env.sync()
    assert float((rec - full).abs().max()) <= 1e-6


def test_autoencoder_boundary_class(env, tmp_path):
    """Autoencoder(input_shape, inf_vector_shape, conv_filters, conv_kernels, conv_strides, latent_space_dim, n_neurons, name)
    (dl_models/autoencoder.py:41-46; main_training.py:118-129): .model, .encoder / .decoder, Trainer, save / load."""
    B = 2
    cfg = AE.AEConfig(H, W, (8, 8, 16, 16), (3, 3, 3, 3), (2, 2, 2, 2), 8, 16)
    params = AE.init_params(cfg, randomize_all=True, dtype=np.float64)
    spec_in, emb, spec_out = R.synthetic_batch(R.Config(H, W), B)
    model = env.U.Autoencoder((H, W, 2), (2, 16), cfg.conv_filters, cfg.conv_kernels, cfg.conv_strides, cfg.latent_space_dim,
                              cfg.n_neurons, name="ae", batch_size=B, device=env.device, runtime=env.rt, dropout=False)
    model.engine.load_keras_params(params)
    tr = env.U.Trainer(model, lr=1e-3, dropout=False)
    got_loss = tr.step(env.t(spec_in), env.t(emb), env.t(spec_out), return_loss=True)
    env.sync()
    want_loss, _, want_pred, grads = AE.loss_and_grads(params, spec_in, emb, spec_out, cfg, 0.9, B, 1)
    assert abs(got_loss - want_loss) <= 1e-5 * abs(want_loss)
    assert float((model.engine.pred.double().cpu() - want_pred).abs().max()) <= 1e-4
    got = model.engine.export_keras_grads()
    assert set(got) == set(grads)
    floor = 1e-6 * max(float(g.abs().max()) for g in grads.values())
    for n, g in grads.items():
        e = float((got[n].double() - g).abs().max())
        assert e <= 1e-3 * float(g.abs().max()) + floor, (n, e)
    nhwc = lambda a: env.t(a).permute(0, 2, 3, 1).contiguous()
    z = model.encoder([nhwc(spec_in), env.t(emb)])
    assert tuple(z.shape) == (B, cfg.latent_space_dim)
    rec = model.decoder(z).clone()
    full = model.predict_stft([nhwc(spec_in), env.t(emb)])
    env.sync()
    assert float((rec - full).abs().max()) <= 1e-6
    model.save(str(tmp_path))
    again = env.U.Autoencoder.load(str(tmp_path), batch_size=B, device=env.device, runtime=env.rt)
    assert torch.equal(again.predict_stft([nhwc(spec_in), env.t(emb)]), full)
    # main_training.py:118-129 configuration at the reference's 144 x 160: parameter count of the Keras model
    assert sum(int(np.prod(s)) for s in AE.param_shapes(AE.AEConfig(144, 160)).values()) == 28_830_658


def test_save_load_round_trip(env, tmp_path):
    cfg = R.Config(H, W, F0, 3)
    model = env.unet(1, dropout=False)
    _load(model, cfg)
    spec_in, emb, _ = R.synthetic_batch(cfg, 1)
    x = [env.t(spec_in).permute(0, 2, 3, 1).contiguous(), env.t(emb)]
    a = model.predict_stft(x)
    model.save(str(tmp_path))
    again = env.U.UNet.load(str(tmp_path), device=env.device, runtime=env.rt)
    b = again.predict_stft(x)
    env.sync()
    assert torch.equal(a, b)


def test_compile_and_fit_follows_the_keras_recipe_of_the_reference(env, tmp_path, monkeypatch):
    """dl_models/u_net.py:72-118: compile(Adam(InverseTimeDecay(lr, steps_per_epoch * 100, 1)), MSE) + fit(batch_size, epochs,
    shuffle=False, [CSVLogger(name.log), EarlyStopping(val_loss, 20)]).  Against the same recipe written on the oracle in fp64: per-epoch
    `loss` (batch-size weighted mean of MSE + l2 terms, training-mode BatchNorm) and `val_loss` (moving statistics), a last batch smaller
    than batch_size, the rate decaying with the optimizer's step count; the CSV log holds the history."""
    import math
    monkeypatch.chdir(tmp_path)
    cfg = R.Config(H, W, F0, 3)
    n_train, n_val, bs, epochs, spe, lr0 = 6, 3, 4, 3, 2, 1e-3
    spec_in, emb, spec_out = R.synthetic_batch(cfg, n_train, seed_name="fit")
    v_in, v_emb, v_out = R.synthetic_batch(cfg, n_val, seed_name="fitval")
    model = env.unet(bs, dropout=False, learning_rate=lr0, name="fit-test")
    params = _load(model, cfg)
    nhwc = lambda a: np.ascontiguousarray(np.transpose(np.asarray(a), (0, 2, 3, 1)))
    hist = model.compile_and_fit(nhwc(spec_in), emb, nhwc(spec_out), nhwc(v_in), v_emb, nhwc(v_out), bs, epochs, spe)
    env.sync()
    # the oracle's trajectory
    st = R.TrainState(cfg, params, dtype=torch.float64)
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    want, it = {"loss": [], "val_loss": []}, 0
    for _ in range(epochs):
        tot = 0.0
        for i in range(0, n_train, bs):
            sl = slice(i, min(i + bs, n_train))
            for p in st.P.values():
                p.grad = None
            pred = R.forward(st.P, t64(spec_in[sl]), torch.tensor(emb[sl]), cfg, True, None, st.bn_state)
            loss = ((pred - t64(spec_out[sl])) ** 2).mean() + R.reg_loss(st.P, cfg, 1)
            loss.backward()
            st.t += 1
            with torch.no_grad():
                for k, p in st.P.items():
                    g = p.grad if p.grad is not None else torch.zeros_like(p)
                    new, st.m[k], st.v[k] = R.adam_update(p, g, st.m[k], st.v[k], st.t, lr0 / (1.0 + it / (spe * 100.0)))
                    p.copy_(new)
            it += 1
            tot += float(loss.detach()) * (sl.stop - sl.start)
        want["loss"].append(tot / n_train)
        with torch.no_grad():
            pred = R.forward(st.P, t64(v_in), torch.tensor(v_emb), cfg, False, None, st.bn_state)
            want["val_loss"].append(float(((pred - t64(v_out)) ** 2).mean() + R.reg_loss(st.P, cfg, 1)))
    assert set(hist) == {"loss", "val_loss"} and len(hist["loss"]) == len(hist["val_loss"]) == epochs
    for k in want:
        for a, b in zip(hist[k], want[k]):
            assert math.isfinite(a) and abs(a - b) <= 2e-4 * abs(b), (k, hist[k], want[k])
    assert hist["loss"][-1] < hist["loss"][0]
    rows = (tmp_path / "fit-test.log").read_text().strip().splitlines()
    assert rows[0] == "epoch,loss,val_loss" and len(rows) == 1 + epochs and rows[1].startswith("0,")
    assert abs(float(rows[-1].split(",")[1]) - hist["loss"][-1]) <= 1e-12


def test_early_stopping_and_csv_logger_callbacks():
    """Keras' EarlyStopping(monitor='val_loss', patience) semantics: stop after `patience` epochs without a new minimum."""
    from unet_rir_amd import callbacks

    class M:
        stop_training = False
    m, cb = M(), callbacks.EarlyStopping(monitor="val_loss", patience=3)
    cb.set_model(m)
    cb.on_train_begin()
    seq = [1.0, 0.9, 0.95, 0.91, 0.9, 0.7]
    for e, v in enumerate(seq):
        cb.on_epoch_end(e, {"val_loss": v, "loss": 0.0})
        if m.stop_training:
            break
    assert e == 4 and cb.stopped_epoch == 4 and cb.best == 0.9 and cb.best_epoch == 1       # 0.9 again is not an improvement


def test_compile_and_fit_on_the_residual_autoencoder_class(env, tmp_path, monkeypatch):
    """dl_models/res_ae.py:78-127 is the same recipe on the ResAE class: it runs through the shared implementation (engines for the full
    and the last, smaller batch over one parameter set), losses are finite and fall, the log is written under the model's name, and a
    callback's stop request ends the loop."""
    import math
    monkeypatch.chdir(tmp_path)
    cfg = RA.ResAEConfig(H, W, (8, 8, 8, 16), (3, 3, 3, 3), (2, 2, 2, 2), 8, 16)
    spec_in, emb, spec_out = R.synthetic_batch(R.Config(H, W), 3, seed_name="rfit")
    model = env.U.ResAE((H, W, 2), (2, 16), cfg.conv_filters, cfg.conv_kernels, cfg.conv_strides, cfg.latent_space_dim,
                        cfg.n_neurons, name="resae-fit", batch_size=2, device=env.device, runtime=env.rt, dropout=False)
    model.engine.load_keras_params(RA.init_params(cfg, randomize_all=True, dtype=np.float64))
    nhwc = lambda a: np.ascontiguousarray(np.transpose(np.asarray(a), (0, 2, 3, 1)))
    hist = model.compile_and_fit(nhwc(spec_in), emb, nhwc(spec_out), nhwc(spec_in[:2]), emb[:2], nhwc(spec_out[:2]), 2, 3, 2, learning_rate=2e-3)
    env.sync()
    assert len(hist["loss"]) == len(hist["val_loss"]) == 3 and all(math.isfinite(v) for v in hist["loss"] + hist["val_loss"])
    assert hist["loss"][-1] < hist["loss"][0]
    assert len((tmp_path / "resae-fit.log").read_text().strip().splitlines()) == 4
    # patience 1 on a loss that cannot improve (rate 0): Keras stops at the end of the second epoch
    from unet_rir_amd import callbacks
    model.get_callbacks = lambda: [callbacks.EarlyStopping(monitor="loss", patience=1)]
    hist = model.compile_and_fit(nhwc(spec_in), emb, nhwc(spec_out), nhwc(spec_in[:2]), emb[:2], nhwc(spec_out[:2]), 2, 4, 2, learning_rate=0.0)
    assert len(hist["loss"]) == 2
