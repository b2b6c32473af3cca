"""The train step as a HIP graph (trainer.Trainer(graph=True)), the per-step scalars in device memory behind it, and the
host-side guards around the step: the replayed graph must be the SAME step - identical variables, moving statistics, loss and
dropout stream - as the launches issued one by one.

Reference anchors: train_step (main_training.py:253-290) is a @tf.function, i.e. the reference itself runs the step as one
compiled graph; Adam's bias-corrected rate (main_training.py:168-169), Dropout(.3) (dl_models/u_net.py:260), the lr
schedule (main_training.py:342-344) and tf.train.CheckpointManager (:171-172) are what changes from step to step.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def U():
    import unet_rir_amd
    unet_rir_amd._lib.lib()
    return unet_rir_amd


def _engine(U, kind, dtype, overlap):
    H = W = 64
    B = 4
    if kind == "unet":
        eng = U.UNetEngine(H, W, B, F0=8, k=3, device=DEV, dtype=dtype, overlap_wgrad=overlap)
    elif kind == "resae":
        eng = U.ResAEEngine(H, W, B, (8, 16, 32, 64), (3, 3, 3, 3), (2, 2, 2, 2), 32, 64, device=DEV, dtype=dtype, overlap_wgrad=overlap)
    else:
        eng = U.UNetGraphEngine(H, W, B, F0=8, k=3, mode=3, device=DEV, dtype=dtype, overlap_wgrad=overlap)
    g = torch.Generator(); g.manual_seed(3)
    eng.reset_parameters(g)
    eng.dropout_seed = 77
    return eng


def _batches(n, B=4, H=64, W=64):
    gen = torch.Generator(); gen.manual_seed(5)
    out = []
    for _ in range(n):
        out.append((torch.rand((B, 2, H, W), generator=gen).to(DEV), torch.randint(26, 1282, (B, 2, 16), generator=gen).to(DEV),
                    torch.rand((B, 2, H, W), generator=gen).to(DEV)))
    return out


def _state(eng):
    return (eng.theta.clone(), eng.adam_m.clone(), eng.adam_v.clone(), {k: v.clone() for k, v in eng.moving.items()}, eng.adam_t,
            eng._shared["dropout_step"])


def _same(a, b):
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k
    assert a[4] == b[4] and a[5] == b[5]


@pytest.mark.parametrize("kind,dtype,overlap", [("unet", "bf16", True), ("unet", "f32", False), ("resae", "bf16", True),
                                                 ("resae", "f32", False), ("unet3", "bf16", True)])
def test_graph_replay_is_the_same_step(U, kind, dtype, overlap):
    """Five steps on changing inputs with a rate that changes on the way (the epoch schedule), dropout on: the graph trainer and a
    trainer that issues the same launches one by one (counters in device memory for both, so the arithmetic is the same by
    construction) end bit-identical; against the host-argument path the only possible difference is one rounding of lr_t."""
    data = _batches(5)
    lrs = [1e-3, 1e-3, 9e-4, 9e-4, 5e-4]
    res = []
    for mode in ("graph", "eager_dev", "eager_host"):
        eng = _engine(U, kind, dtype, overlap)
        tr = U.Trainer(eng, lr=1e-3, bucket_bytes=16 << 10, graph=(mode == "graph"))
        if mode == "eager_dev":
            eng.use_device_counters(True)
        losses = []
        for (a, e, b), lr in zip(data, lrs):
            losses.append(tr.step(a, e, b, lr=lr, return_loss=True))
        torch.cuda.synchronize()
        if mode == "graph":
            assert set(tr._graphs) == {True}
        res.append((losses, _state(eng)))
    (lg, sg), (ld, sd), (lh, sh) = res
    assert lg == ld
    _same(sg, sd)
    # one fp32 rounding of lr_t moves a parameter by ~1e-10; in bf16 storage that can flip the rounding of a work copy, so the
    # two trajectories are compared to the size of such flips, not bit for bit
    assert abs(lg[0] - lh[0]) <= 1e-6 * abs(lh[0])
    assert all(abs(x - y) <= 2e-3 * abs(y) for x, y in zip(lg, lh))
    # (Adam moves an entry by up to lr per step whatever its gradient's size: a near-zero gradient entry whose sign differs between
    # the two runs separates them by 2 lr per step)
    assert float((sg[0] - sh[0]).abs().max()) <= 2 * sum(lrs)
    assert sg[4] == sh[4] == 5 and sg[5] == sh[5]
    assert lg[-1] < lg[0] or kind != "unet"          # the toy problem trains


def test_graph_step_without_the_reported_loss_and_with_it_are_two_graphs_over_one_state(U):
    """return_loss decides whether the l2 terms of the REPORTED loss are evaluated: two captured variants, one model state."""
    data = _batches(4)
    ref_eng = _engine(U, "resae", "bf16", True)
    ref_eng.use_device_counters(True)
    ref = U.Trainer(ref_eng, lr=1e-3)
    eng = _engine(U, "resae", "bf16", True)
    tr = U.Trainer(eng, lr=1e-3, graph=True)
    for i, (a, e, b) in enumerate(data):
        want = ref.step(a, e, b, return_loss=True)
        got = tr.step(a, e, b, return_loss=bool(i % 2))
        if i % 2:
            assert got == want
    torch.cuda.synchronize()
    assert set(tr._graphs) == {False, True}
    _same(_state(eng), _state(ref_eng))


def test_graph_step_falls_back_for_an_external_mask_and_resumes(U):
    """A caller-supplied dropout mask cannot be part of the captured step: that call runs the ordinary way (launch arguments from
    the host), and the next graph step picks the counters up again."""
    data = _batches(3)
    ea, eb = _engine(U, "unet", "bf16", False), _engine(U, "unet", "bf16", False)
    ta, tb = U.Trainer(ea, lr=1e-3, graph=True), U.Trainer(eb, lr=1e-3)
    mask = (torch.rand((4, ea.vec_dim), device=DEV) >= 0.3).float() / 0.7
    for i, (a, e, b) in enumerate(data):
        m = mask if i == 1 else None
        ta.step(a, e, b, dropout_mask=m)
        tb.step(a, e, b, dropout_mask=m)
    torch.cuda.synchronize()
    assert ea.adam_t == eb.adam_t == 3 and ea._shared["dropout_step"] == eb._shared["dropout_step"] == 2
    assert float((ea.theta - eb.theta).abs().max()) <= 2 * 3 * 1e-3          # up to one rounding of lr_t apart (see above)


def test_checkpoint_restores_the_dropout_stream_and_the_graph_continues(U, tmp_path):
    """CheckpointManager holds the Dropout stream position (seed, draw number) beside the Adam state: a restored run draws the
    masks the uninterrupted run draws, also when the step is a replayed graph."""
    data = _batches(6)
    e0 = _engine(U, "unet", "bf16", True)
    t0 = U.Trainer(e0, lr=1e-3, graph=True)
    for a, e, b in data:
        t0.step(a, e, b)
    torch.cuda.synchronize()
    e1 = _engine(U, "unet", "bf16", True)
    t1 = U.Trainer(e1, lr=1e-3, graph=True)
    for a, e, b in data[:3]:
        t1.step(a, e, b)
    path = U.CheckpointManager(t1, str(tmp_path)).save(epoch=0)
    e2 = _engine(U, "unet", "bf16", True)
    e2.dropout_seed = 5          # overwritten by the checkpoint
    t2 = U.Trainer(e2, lr=1e-3, graph=True)
    U.CheckpointManager(t2, str(tmp_path)).restore(path)
    assert e2._shared["dropout_step"] == 3 and e2.dropout_seed == 77
    for a, e, b in data[3:]:
        t2.step(a, e, b)
    torch.cuda.synchronize()
    _same(_state(e0), _state(e2))


def test_host_information_vector_is_copied_not_dereferenced(U):
    """DataGenerator.__getitem__ yields host arrays (datageneratorv2.py:88-102): an `emb` that lives on the host must reach the
    embedding kernel as a device copy, and a wrong-device index tensor handed to the kernel wrapper itself must raise."""
    (a, e, b), = _batches(1)
    eng = _engine(U, "unet", "f32", False)
    p0 = eng.forward(a, e).clone()
    p1 = eng.forward(a, e.cpu()).clone()
    torch.cuda.synchronize()
    assert torch.equal(p0, p1)
    ge = _engine(U, "resae", "f32", False)
    q0 = ge.forward(a, e).clone()
    q1 = ge.forward(a, e.cpu().to(torch.int32)).clone()
    torch.cuda.synchronize()
    assert torch.equal(q0, q1)
    with pytest.raises(ValueError):
        U.ops.index_to_i32(e.cpu().contiguous(), eng.emb_idx)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("diff,sig", [(True, False), (False, True), (True, True)])
def test_compute_loss_switches_against_the_oracle(U, dtype, diff, sig):
    """sigmoid_loss / diff_loss of compute_loss (main_training.py:38-39, :214-222): value, metric sums and dL/dlogits of the fused
    kernel against the fp64 restatement differentiated by autograd."""
    from oracle import torch_ref as R
    ops = U.ops
    B, H, W, gb = 3, 20, 28, 6
    gen = torch.Generator(); gen.manual_seed(11)
    logits = torch.randn((B, H, W, 4), generator=gen)
    target = torch.rand((B, 2, H, W), generator=gen)
    x_in = torch.rand((B, 2, H, W), generator=gen)
    wgt = R.sigmoid_weight(0.5, W)
    z = logits[..., :2].double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    loss = R.data_loss(target.double(), torch.sigmoid(z), 0.9, gb, x_in.double() if diff else None, wgt if sig else None)
    loss.backward()
    la = ops.Act(logits.to(DEV).contiguous())
    PAD = 4 if dtype == "f32" else 8
    dl = ops.new_act(B, H, W, PAD, DEV, dtype=torch.float32 if dtype == "f32" else torch.bfloat16)
    pr = torch.empty((B, 2, H, W), device=DEV)
    out = torch.zeros(4, device=DEV)
    ws = ops.Workspace(DEV)
    from unet_rir_amd.trainer import phase_weight
    pw = phase_weight(0.5, W)
    assert float((pw.double() - wgt).abs().max()) <= 1e-7
    ops.sigmoid_loss(la, target.to(DEV), 0.9, 1.0 / (2 * H * W * gb), pr, dl, out, ws,
                     phase_ref=x_in.to(DEV) if diff else None, phase_weight=pw.to(DEV) if sig else None)
    torch.cuda.synchronize()
    assert abs(float(out[0]) - float(loss)) <= 1e-5 * abs(float(loss))
    # the metric sums stay unweighted (main_training.py:278-284)
    t1 = target[:, 1].double() - (x_in[:, 1].double() if diff else 0.0)
    ph = torch.remainder((t1 - torch.sigmoid(z[:, 1].detach())) * 2 * math.pi + math.pi, 2 * math.pi) - math.pi
    assert abs(float(out[2]) - float((1 - torch.cos(ph)).sum())) <= 1e-4 * float((1 - torch.cos(ph)).sum())
    g = dl.base[..., :2].double().cpu().permute(0, 3, 1, 2)
    tol = 1e-5 if dtype == "f32" else 1e-2
    assert float((g - z.grad).abs().max()) <= tol * float(z.grad.abs().max())
    assert float(dl.base[..., 2:].float().abs().max()) == 0.0


def test_trainer_loss_switches_reach_the_engine_step(U):
    """Trainer(sigmoid_loss=True, diff_loss=True, beta=...) on the whole model: the loss of a step equals the oracle's
    compute_loss on the oracle's prediction (fp32 engine, no dropout)."""
    import numpy as np
    from oracle import torch_ref as R
    H, W, F0, B = 32, 32, 8, 2
    cfg = R.Config(H, W, F0, 3)
    params = R.init_params(cfg, randomize_all=True, dtype=np.float64)
    spec_in, emb, spec_out = R.synthetic_batch(cfg, B)
    P = R.to_torch(params, torch.float64)
    pred = R.forward(P, torch.tensor(spec_in).double(), torch.tensor(emb), cfg, training=True)
    pred = pred[0] if isinstance(pred, tuple) else pred
    want = float(R.data_loss(torch.tensor(spec_out).double(), pred, 0.9, B, torch.tensor(spec_in).double(), R.sigmoid_weight(0.7, W)) +
                 R.reg_loss(P, cfg, 1))
    eng = U.UNetEngine(H, W, B, F0=F0, k=3, device=DEV)
    eng.load_keras_params(params)
    tr = U.Trainer(eng, lr=1e-3, dropout=False, sigmoid_loss=True, diff_loss=True, beta=0.7)
    t = lambda a: torch.tensor(a).to(DEV)
    got = tr.step(t(spec_in), t(emb), t(spec_out), return_loss=True)
    assert abs(got - want) <= 1e-5 * abs(want), (got, want)
