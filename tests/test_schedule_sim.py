"""The PRODUCT's step schedule on CPU: UNetEngine.forward / backward, Trainer.step, the gradient-bucket hand-over, the
bucket-wise Adam on its own stream and the data-parallel SUM all-reduce run unmodified; only the kernels (tests/cpu_ops.py,
oracle-backed) and the stream runtime (tests/sim_runtime.py, vector clocks + race check) are stand-ins.

  * single process: a full train step equals the oracle's (loss, every gradient, every parameter after Adam) in the plain
    schedule and in the side-stream schedule, and the side-stream schedule has no unordered cross-stream access;
  * the race check is live: removing one event wait from the schedule is detected;
  * world_size 2 on gloo: Trainer.step on two ranks == the oracle's single-process step over the concatenated batch with
    per-replica BatchNorm, global-batch loss normalisation and l2 / replicas (MirroredStrategy semantics,
    main_training.py:56, :230-233, :323-327), for two consecutive steps, with buckets small enough that several all-reduces
    and several bucket-wise Adam launches interleave with the backward pass.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

H = W = 16
F0 = 4
KINDS = ("unet", "graph3", "resae", "ae")     # UNetEngine (mode 0, hand schedule), UNetGraphEngine mode 3, ResAEEngine, AutoencoderEngine


def _oracle(kind):
    """(cfg, initial parameters, loss_and_grads(params, spec_in, emb, spec_out, global_batch, n_replicas)) of one model kind."""
    from oracle import torch_ref as R, torch_resae as RA, torch_ae as AE
    if kind == "ae":
        cfg = AE.AEConfig(H, W, (4, 8, 8, 8), (3, 3, 3, 3), (2, 2, 2, 2), 8, 16)
        lg = lambda p, a, e, b, gb, nr: AE.loss_and_grads(p, a, e, b, cfg, 0.9, gb, nr, None, None)
        return cfg, AE.init_params(cfg, randomize_all=True, dtype=np.float64), lg
    if kind == "resae":
        cfg = RA.ResAEConfig(H, W, (4, 8, 8, 8), (3, 3, 3, 3), (2, 2, 2, 2), 8, 16)
        lg = lambda p, a, e, b, gb, nr: RA.loss_and_grads(p, a, e, b, cfg, 0.9, gb, nr, None, None)
        return cfg, RA.init_params(cfg, randomize_all=True, dtype=np.float64), lg
    cfg = R.Config(H, W, F0, 3, mode=3 if kind == "graph3" else 0)
    lg = lambda p, a, e, b, gb, nr: R.loss_and_grads(p, a, e, b, cfg, 0.9, gb, nr, None, torch.float64)
    return cfg, R.init_params(cfg, randomize_all=True, dtype=np.float64), lg


def _oracle_steps(world, B, n_steps, lr, kind="unet", opt="adam"):
    """The reference semantics on one process: per-replica gradients summed, one Adam per step."""
    from oracle import torch_ref as R
    cfg, params, loss_and_grads = _oracle(kind)
    params = {k: np.asarray(v, np.float64) for k, v in params.items()}
    m = {k: torch.zeros(v.shape, dtype=torch.float64) for k, v in params.items()}
    v_ = {k: torch.zeros(v.shape, dtype=torch.float64) for k, v in params.items()}
    spec_in, emb, spec_out = R.synthetic_batch(R.Config(H, W), B * world)
    losses, grads_first = [], None
    ms = 1.0
    for t in range(1, n_steps + 1):
        total, loss = None, 0.0
        for r in range(world):
            sl = slice(r * B, (r + 1) * B)
            l, _, _, g = loss_and_grads(params, spec_in[sl], emb[sl], spec_out[sl], B * world, world)
            loss += l
            total = g if total is None else {k: total[k] + g[k] for k in g}
        if grads_first is None:
            grads_first = total
        losses.append(loss)
        ms_next = ms
        for k in params:
            if opt == "sgd":
                new = R.sgd_update(torch.tensor(params[k]), total[k], lr)
            elif opt == "nadam":
                new, m[k], v_[k], ms_next = R.nadam_update(torch.tensor(params[k]), total[k], m[k], v_[k], t, lr, ms)
            else:
                new, m[k], v_[k] = R.adam_update(torch.tensor(params[k]), total[k], m[k], v_[k], t, lr)
            params[k] = new.numpy()
        ms = ms_next
    return params, grads_first, losses


def _build(rt, B, overlap, world=1, bucket_bytes=32 << 20, kind="unet", **kw):
    import unet_rir_amd as U
    from oracle import torch_ref as R
    cfg, params, _ = _oracle(kind)
    if kind == "unet":
        eng = U.UNetEngine(H, W, B, F0=F0, k=3, device="cpu", runtime=rt, overlap_wgrad=overlap, n_replicas=world, **kw)
    elif kind == "graph3":
        eng = U.UNetGraphEngine(H, W, B, F0=F0, k=3, mode=3, device="cpu", runtime=rt, n_replicas=world, overlap_wgrad=overlap)
    elif kind == "ae":
        eng = U.AutoencoderEngine(H, W, B, cfg.conv_filters, cfg.conv_kernels, cfg.conv_strides, cfg.latent_space_dim, cfg.n_neurons,
                                  device="cpu", runtime=rt, n_replicas=world, overlap_wgrad=overlap)
    else:
        eng = U.ResAEEngine(H, W, B, cfg.conv_filters, cfg.conv_kernels, cfg.conv_strides, cfg.latent_space_dim, cfg.n_neurons,
                            device="cpu", runtime=rt, n_replicas=world, overlap_wgrad=overlap)
    eng.load_keras_params(params)
    tr = U.Trainer(eng, lr=1e-3, dropout=False, world_size=world, bucket_bytes=bucket_bytes)
    return R.Config(H, W), eng, tr


LR, N_STEPS = 1e-3, 2
# Adam moves every entry by ~lr per step whatever its gradient's magnitude, so fp32 rounding of a near-zero gradient entry
# shows up as a fraction of lr: parameters are compared to 2 % of the distance they can have moved (gradients are held
# to 1e-4 of their tensor's scale separately)
P_ATOL = 0.02 * LR * N_STEPS


def _check_params(eng, want):
    got = eng.export_keras_params()
    for n, w in want.items():
        err = float(np.abs(got[n].double().numpy() - w).max())
        assert err <= P_ATOL, (n, err)


@pytest.mark.parametrize("overlap,kind", [(False, "unet"), (True, "unet"), (False, "graph3"), (False, "resae"), (False, "ae"),
                                          (True, "graph3"), (True, "resae"), (True, "ae")])
def test_full_step_on_the_product_schedule_matches_oracle(monkeypatch, overlap, kind):
    import cpu_ops
    from sim_runtime import SimRuntime
    from oracle import torch_ref as R
    rt = SimRuntime()
    cpu_ops.install(monkeypatch, rt)
    B = 2
    cfg, eng, tr = _build(rt, B, overlap, bucket_bytes=8192, kind=kind)
    spec_in, emb, spec_out = R.synthetic_batch(cfg, B)
    t = torch.tensor
    want_p, want_g, want_l = _oracle_steps(1, B, 2, 1e-3, kind)
    loss = tr.step(t(spec_in), t(emb), t(spec_out), return_loss=True)
    assert abs(loss - want_l[0]) <= 1e-5 * abs(want_l[0])
    got_g = eng.export_keras_grads()
    floor = 1e-6 * max(float(g.abs().max()) for g in want_g.values())     # tensors whose exact gradient is ~0 (behind a 2-sample BatchNorm)
    for n, g in want_g.items():
        e = float((got_g[n].double() - g).abs().max())
        assert e <= 1e-4 * float(g.abs().max()) + floor, (n, e)
    loss2 = tr.step(t(spec_in), t(emb), t(spec_out), return_loss=True)
    assert abs(loss2 - want_l[1]) <= 1e-5 * abs(want_l[1])
    _check_params(eng, want_p)
    if overlap:
        assert len(tr.bucketer.bounds) > 3            # several buckets: Adam ran bucket by bucket beside the backward pass
        assert rt.n_cross_stream > 50                 # ... and the schedule's event edges were what ordered it
    else:
        assert rt.n_cross_stream == 0


@pytest.mark.parametrize("kind", ["unet", "resae"])
def test_race_check_detects_a_missing_event_wait(monkeypatch, kind):
    """Negative control: take the 'side stream waits for the main stream' edge out of the schedule (hand schedule and graph engine)."""
    import cpu_ops
    from sim_runtime import RaceError, SimRuntime
    from oracle import torch_ref as R
    import unet_rir_amd.engine as E
    rt = SimRuntime()
    cpu_ops.install(monkeypatch, rt)
    cfg, eng, tr = _build(rt, 2, True, kind=kind)
    spec_in, emb, spec_out = R.synthetic_batch(cfg, 2)
    monkeypatch.setattr(rt, "wait", lambda stream, ev: None if stream is eng.wg_stream else stream.merge(ev.clock))
    with pytest.raises(RaceError):
        tr.step(torch.tensor(spec_in), torch.tensor(emb), torch.tensor(spec_out))


def test_race_check_detects_a_missing_optimizer_join(monkeypatch):
    """Negative control: the next forward pass must wait for the bucket-wise Adam on the optimizer stream."""
    import cpu_ops
    from sim_runtime import RaceError, SimRuntime
    from oracle import torch_ref as R
    rt = SimRuntime()
    cpu_ops.install(monkeypatch, rt)
    cfg, eng, tr = _build(rt, 2, True)
    spec_in, emb, spec_out = (torch.tensor(a) for a in R.synthetic_batch(cfg, 2))
    tr.step(spec_in, emb, spec_out)
    monkeypatch.setattr(rt, "wait", lambda stream, ev: None if stream is rt.main and "s2" in ev.clock and ev.clock.get("s2", 0) > rt.main.clock.get("s2", 0)
                        else stream.merge(ev.clock))
    with pytest.raises(RaceError):
        tr.step(spec_in, emb, spec_out)
        tr.step(spec_in, emb, spec_out)


@pytest.mark.parametrize("kind,overlap", [("unet", True), ("unet", False), ("resae", True)])
def test_parked_split_k_reductions_are_flushed_before_their_first_reader(monkeypatch, kind, overlap):
    """The engines park the split-K reductions of their weight gradients (ops.ReduceBatch) and the trainer's bucketer runs them
    before a bucket's gradients are first read.  In the simulated operators a parked gradient holds NaNs until the flush: the steps
    above equal the oracle WITH the parking in place; here the flush the bucketer asks for is taken away (negative control) and the
    optimizer must be seen reading gradients that were never reduced - in the buckets handed over during the backward pass, the
    end-of-backward flush comes too late for them."""
    import cpu_ops
    from sim_runtime import SimRuntime
    from oracle import torch_ref as R
    rt = SimRuntime()
    cpu_ops.install(monkeypatch, rt)
    cfg, eng, tr = _build(rt, 2, overlap, bucket_bytes=8192, kind=kind)
    spec_in, emb, spec_out = (torch.tensor(a) for a in R.synthetic_batch(cfg, 2))
    assert eng._rb is not None
    if tr.bucketer is None:                # no side stream, one replica: the optimizer runs after backward()'s own final flush
        tr.step(spec_in, emb, spec_out)
        assert bool(torch.isfinite(eng.theta).all())
        return
    assert tr.bucketer.before_bucket is not None and len(tr.bucketer.bounds) > 3
    tr.step(spec_in, emb, spec_out)
    assert bool(torch.isfinite(eng.theta).all()) and len(eng._rb) == 0
    tr.bucketer.before_bucket = lambda: None          # a hook that claims to run the parked reductions and does not
    from sim_runtime import RaceError
    try:
        tr.step(spec_in, emb, spec_out)
        seen = not bool(torch.isfinite(eng.theta).all())         # the optimizer consumed NaNs ...
    except RaceError:
        seen = True                                               # ... or the late flush wrote what the optimizer had already read
    assert seen


def _dp_worker(rank, world, port, out_path, overlap, kind, B=2):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 2:
        torch.set_num_threads(1)                      # eight ranks on the build container's eight cores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from _pytest.monkeypatch import MonkeyPatch
        import cpu_ops
        from sim_runtime import SimRuntime
        from oracle import torch_ref as R
        mpatch = MonkeyPatch()
        rt = SimRuntime()
        cpu_ops.install(mpatch, rt)
        cfg, eng, tr = _build(rt, B, overlap, world=world, bucket_bytes=8192, kind=kind)
        if rank != 0:
            eng.theta.mul_(0.5)                       # replicas must end up with rank 0's variables
        tr.broadcast_parameters(0)
        spec_in, emb, spec_out = R.synthetic_batch(cfg, B * world)
        sl = slice(rank * B, (rank + 1) * B)
        t = torch.tensor
        losses = [tr.step(t(spec_in[sl]), t(emb[sl]), t(spec_out[sl]), return_loss=True) for _ in range(2)]
        lt = torch.tensor(losses, dtype=torch.float64)
        dist.all_reduce(lt)                           # strategy.reduce(SUM, per_replica_losses) (main_training.py:326)
        torch.save({"params": {k: v.double() for k, v in eng.export_keras_params().items()}, "losses": lt,
                    "n_all_reduce": len(rt.collectives), "n_buckets": len(tr.bucketer.bounds), "cross": rt.n_cross_stream},
                   f"{out_path}.{rank}")
        mpatch.undo()
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap,kind", [(False, "unet"), (True, "unet"), (False, "graph3"), (False, "resae"), (False, "ae"),
                                          (True, "resae"), (True, "graph3")])
def test_two_rank_trainer_step_equals_single_process_oracle(tmp_path, overlap, kind):
    world = 2
    out = str(tmp_path / "dp")
    port = 29600 + (os.getpid() % 1500) + (7 if overlap else 0) + 11 * KINDS.index(kind)
    mp.spawn(_dp_worker, args=(world, port, out, overlap, kind), nprocs=world, join=True)
    want_p, _, want_l = _oracle_steps(world, 2, 2, 1e-3, kind)
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    for r in range(world):
        assert res[r]["n_buckets"] > 3 and res[r]["n_all_reduce"] == 2 * res[r]["n_buckets"]      # one collective per bucket per step
        for n, w in want_p.items():
            err = float(np.abs(res[r]["params"][n].numpy() - w).max())
            assert err <= P_ATOL, (r, n, err)
        np.testing.assert_allclose(res[r]["losses"].numpy(), np.array(want_l), rtol=1e-5)
    for n in want_p:                                  # replicas stay bit-identical
        assert torch.equal(res[0]["params"][n], res[1]["params"][n]), n


def test_eight_rank_trainer_step_equals_single_process_oracle(tmp_path):
    """The real rank count of BASELINE.json configs[2] (global batch split over 8 replicas, main_training.py:56, :114-117): the
    product Trainer.step - side-stream schedule, bucket hand-over, bucket-wise Adam, SUM all-reduce, loss normalised by the global
    batch, l2 terms by the replica count - on eight gloo ranks equals the oracle's single-process training on the concatenated
    batch; bucket boundaries and hand-over order are the ones the 8-GPU run will see."""
    world, B = 8, 2
    out = str(tmp_path / "dp8")
    port = 31200 + (os.getpid() % 1500)
    mp.spawn(_dp_worker, args=(world, port, out, True, "unet", B), nprocs=world, join=True)
    want_p, _, want_l = _oracle_steps(world, B, 2, 1e-3, "unet")
    res = [torch.load(f"{out}.{r}") for r in range(world)]
    for r in range(world):
        assert res[r]["n_buckets"] > 3 and res[r]["n_all_reduce"] == 2 * res[r]["n_buckets"]
        for n, w in want_p.items():
            err = float(np.abs(res[r]["params"][n].numpy() - w).max())
            assert err <= P_ATOL, (r, n, err)
        np.testing.assert_allclose(res[r]["losses"].numpy(), np.array(want_l), rtol=1e-5)
    for r in range(1, world):
        for n in want_p:
            assert torch.equal(res[0]["params"][n], res[r]["params"][n]), (r, n)


def _dropout_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from _pytest.monkeypatch import MonkeyPatch
        import cpu_ops
        import unet_rir_amd as U
        from sim_runtime import SimRuntime
        mpatch = MonkeyPatch()
        rt = SimRuntime()
        cpu_ops.install(mpatch, rt)
        torch.manual_seed(1234)                       # every process seeds alike, as a launcher script would
        eng = U.UNetEngine(H, W, 2, F0=F0, k=3, device="cpu", runtime=rt, n_replicas=world)
        base = eng.dropout_seed
        tr = U.Trainer(eng, lr=1e-3, world_size=world, dropout=True)
        tr2 = U.Trainer(eng, lr=1e-3, world_size=world, dropout=True)          # a second trainer over the same engine must not offset again
        m0 = eng.make_dropout_mask().clone()
        m1 = eng.make_dropout_mask().clone()
        # resume: rank 0 writes ONE checkpoint, every rank restores it (trainer.CheckpointManager) - each replica must come back on
        # ITS stream (base + rank), at the saved position
        mgr = U.CheckpointManager(tr, out_path + ".ckpt")
        if rank == 0:
            mgr.save(epoch=0)
        dist.barrier()
        eng.dropout_seed = 0                          # whatever it was: the checkpoint decides
        eng._shared["dropout_step"] = 99
        mgr.restore()
        seed_r, step_r = eng.dropout_seed, eng._shared["dropout_step"]
        m2 = eng.make_dropout_mask().clone()
        torch.save({"base": base, "seed": eng.dropout_seed, "m0": m0, "m1": m1, "step": 2, "seed_r": seed_r, "step_r": step_r, "m2": m2},
                   f"{out_path}.{rank}")
        mpatch.undo()
        del tr, tr2
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_replicas_draw_independent_dropout_masks(tmp_path):
    """Dropout(.3) under MirroredStrategy draws per replica (dl_models/u_net.py:260, main_training.py:56): with the process-wide
    default seed identical in every rank, the trainer offsets each replica's stream by its rank - different masks per replica,
    different masks per step, the same masks when the run is repeated."""
    world = 2
    out = str(tmp_path / "do")
    mp.spawn(_dropout_worker, args=(world, 30100 + (os.getpid() % 1500), out), nprocs=world, join=True)
    r0, r1 = (torch.load(f"{out}.{r}") for r in range(world))
    assert r0["base"] == r1["base"] and r0["seed"] == r0["base"] and r1["seed"] == r0["base"] + 1
    assert not torch.equal(r0["m0"], r1["m0"]) and not torch.equal(r0["m0"], r0["m1"])
    assert r0["step"] == r1["step"] == 2
    assert abs(float((r0["m0"] > 0).float().mean()) - 0.7) < 0.1
    # after every rank restored rank 0's checkpoint: own stream again (base + rank), saved position, masks differ across replicas
    assert r0["seed_r"] == r0["base"] and r1["seed_r"] == r0["base"] + 1 and r0["step_r"] == r1["step_r"] == 2
    assert not torch.equal(r0["m2"], r1["m2"]) and not torch.equal(r0["m2"], r0["m1"])


@pytest.mark.parametrize("opt", ["sgd", "nadam"])
def test_other_optimizers_of_the_reference_on_the_product_schedule(monkeypatch, opt):
    """main_training.py:164-169 selects Nadam / SGD / Adam by name: the product step with optimizer="nadam" / "sgd" (side-stream
    schedule, bucket-wise optimizer) against the oracle's restatement, two steps."""
    import cpu_ops
    import unet_rir_amd as U
    from sim_runtime import SimRuntime
    from oracle import torch_ref as R
    rt = SimRuntime()
    cpu_ops.install(monkeypatch, rt)
    B = 2
    cfg, eng, _ = _build(rt, B, True, bucket_bytes=8192)
    tr = U.Trainer(eng, lr=LR, dropout=False, bucket_bytes=8192, optimizer=opt)
    spec_in, emb, spec_out = (torch.tensor(a) for a in R.synthetic_batch(cfg, B))
    want_p, _, want_l = _oracle_steps(1, B, 2, LR, "unet", opt)
    losses = [tr.step(spec_in, emb, spec_out, return_loss=True) for _ in range(2)]
    np.testing.assert_allclose(np.array(losses), np.array(want_l), rtol=1e-5)
    _check_params(eng, want_p)
