"""Train step and data parallelism.

``Trainer.step`` is the counterpart of ``train_step`` in main_training.py:253-290 (and of the legacy
``Trainer.step`` of trainer.py:133-144): forward, compute_loss, gradients of every trainable variable, gradient
all-reduce across replicas, Adam apply.  Data parallelism follows tf.distribute.MirroredStrategy
(main_training.py:56, :114-117, :323-327): one process per GPU, the global batch is split evenly, every replica
holds the full model, BatchNorm statistics stay per replica, the loss is normalised by the GLOBAL batch so
gradients are SUM-reduced.  The collective is RCCL (torch.distributed backend "nccl") over xGMI, issued per
bucket while the backward pass is still running.
"""
import os

import torch
import torch.distributed as dist

from .engine import ALIGN


class GradBucketer:
    """Cuts a flat gradient buffer, laid out in backward-completion order, into contiguous buckets and all-reduces
    (SUM) each bucket as soon as the backward pass has produced it.  Device agnostic: the same object drives RCCL
    on GPUs and gloo in the CPU tests."""

    def __init__(self, flat_grad, boundaries, bucket_bytes=32 << 20, group=None, reduce=True, on_bucket=None, runtime=None,
                 before_bucket=None):
        """boundaries: increasing element offsets at which a gradient range [0, off) can become final.
        reduce=False: no collective (single replica), only the bucket schedule.  on_bucket(lo, hi, work) is called for
        every bucket right after its all-reduce has been launched (work is None without a collective): the hook the
        trainer uses to start the optimizer on finished buckets while the backward pass is still running."""
        self.flat = flat_grad
        self.group = group
        self.rt = runtime          # device.HipRuntime (or the tests' simulated one); None: plain torch.distributed calls
        self.reduce = reduce
        self.on_bucket = on_bucket
        # called (on the stream the engine hands buckets over from) before a bucket's gradients are first read: the engines park the
        # split-K reductions of their weight gradients until then and run them in one launch (engine.flush_reduces)
        self.before_bucket = before_bucket
        self.bounds = []
        last, limit = 0, max(bucket_bytes // 4, 1)
        for off in boundaries:
            if off - last >= limit:
                self.bounds.append(off)
                last = off
        n = flat_grad.numel()
        if not self.bounds or self.bounds[-1] != n:
            self.bounds.append(n)
        self.reset()

    def reset(self):
        self.next = 0
        self.sent = 0
        self.works = []

    def buckets(self):
        lo = 0
        for hi in self.bounds:
            yield lo, hi
            lo = hi

    def mark_ready(self, offset_end):
        """Everything in [0, offset_end) is final (the launches that write it are enqueued)."""
        while self.next < len(self.bounds) and self.bounds[self.next] <= offset_end:
            hi = self.bounds[self.next]
            work = None
            if self.before_bucket is not None:
                self.before_bucket()
            if self.reduce:
                if self.rt is not None:
                    work = self.rt.all_reduce_sum(self.flat[self.sent:hi], self.group)
                else:
                    work = dist.all_reduce(self.flat[self.sent:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.works.append(work)
            if self.on_bucket is not None:
                self.on_bucket(self.sent, hi, work)
            self.sent = hi
            self.next += 1

    def finish(self):
        self.mark_ready(self.flat.numel())
        for w in self.works:
            if self.rt is not None:
                self.rt.wait_work(w)
            else:
                w.wait()
        self.works = []
        self.next = 0
        self.sent = 0


def phase_weight(beta, W):
    """preprocess.sigmoid(beta, (H, W)) (preprocess.py:116-121): every row of the mask the `sigmoid_loss` switch multiplies into
    the phase term is z = 1 / (1 + exp(-(x + 5) beta)) on x = linspace(-10, 10, W), flipped.  fp32 [W]."""
    x = torch.linspace(-10.0, 10.0, W, dtype=torch.float64)
    return torch.flip(1.0 / (1.0 + torch.exp(-(x + 5.0) * beta)), dims=(0,)).to(torch.float32)


def lr_schedule(lr0, epoch, decay=(True, 80)):
    """main_training.py:342-344: from epoch >= 80 the rate is lr0 * 0.9 ** (epoch / 80)."""
    if decay and decay[0] and epoch >= decay[1]:
        return lr0 * 0.9 ** (epoch / decay[1])
    return lr0


class _LossFunction(torch.autograd.Function):
    """compute_loss of main_training.py:203-235 as an autograd node over the engine's fused sigmoid + loss kernel: the value is
    the per-replica scalar, backward() runs the engine's whole backward pass (and, through the trainer, the gradient
    all-reduce and the optimizer of finished buckets)."""

    @staticmethod
    def forward(ctx, trainer, anchor, loss_value):
        ctx.trainer = trainer
        return loss_value.view(())

    @staticmethod
    def backward(ctx, g):
        ctx.trainer._backward_and_reduce()           # loss.backward() passes d(loss)/d(loss) = 1
        return None, None, None


class Trainer:
    """One call of ``step`` = one ``distributed_train_step`` of main_training.py:323-327 on this replica.

    `model` is an engine (UNetEngine, UNetGraphEngine, ResAEEngine) or a boundary module holding one (`UNet`, `ResAE`: then
    the engine built for the module's `batch_size` is driven).  Two ways to run a step:

      trainer.step(spec_in, emb, spec_out)                       # everything in one call (NCHW shards)

      pred = model.model([spec_in, emb], training=True)          # the reference's own loop shape, main_training.py:253-290
      loss = trainer.compute_loss(spec_out, pred)                #   compute_loss(spec_out, spec_generated, model.losses)
      loss.backward()                                            #   tape.gradient  (+ all-reduce + Adam of finished buckets)
      trainer.apply_gradients()                                  #   optimizer.apply_gradients
    """

    def __init__(self, model, lr=5e-7, alpha=0.9, world_size=1, group=None, bucket_bytes=32 << 20, dropout=True, force_dp=False,
                 sigmoid_loss=False, diff_loss=False, beta=0.5, graph=False, dropout_seed=None, optimizer="adam"):
        """force_dp: run the bucketed all-reduce path even at world_size 1 (needs an initialised process group; rehearsal).
        sigmoid_loss / diff_loss / beta: the switches of compute_loss (main_training.py:38-40, :214-222; both False in the
        reference's live configuration): column weights sigmoid(beta, ...) of preprocess.py:116-121 on the phase term, and the
        phase target taken relative to the input's phase.
        graph: capture `step` once into a HIP graph (after one ordinary step that is undone) and replay it - one host call per
        step instead of several hundred launches; single-replica steps without an externally supplied dropout mask only
        (anything else runs the ordinary way).
        optimizer: "adam" (the live configuration), "nadam" or "sgd" - main_training.py:164-169 selects by substring, so does this.
        dropout_seed: base seed of the Dropout streams; replica r draws from stream seed + r (independent masks per replica, as
        under MirroredStrategy).  Default: the engine's own seed (torch.initial_seed())."""
        self.module = None
        engine = model
        if not hasattr(model, "specs") and hasattr(model, "engine"):
            self.module = model
            engine = model.engine
            if engine is None:
                raise RuntimeError("build the model first (batch_size=...): the trainer drives the engine of one batch size")
        self.engine = engine
        self.rt = engine.rt
        self.lr, self.alpha = lr, alpha
        self.world_size = world_size
        self.group = group
        self.dropout = dropout
        self.bucketer = None
        engine.n_replicas = world_size
        dp = world_size > 1 or (force_dp and dist.is_initialized())
        # per-replica Dropout streams: the engine's seed is process-wide (torch.initial_seed()), identical in every rank
        if dropout_seed is not None:
            engine.dropout_seed = int(dropout_seed) & 0xFFFFFFFF
        self.rank = dist.get_rank(group) if (world_size > 1 and dist.is_initialized()) else 0
        if world_size > 1 and not getattr(engine, "_seed_offset_by_rank", False):
            engine.dropout_seed = (engine.dropout_seed + self.rank) & 0xFFFFFFFF
            engine._seed_offset_by_rank = True
        engine.loss_diff = bool(diff_loss)
        engine.loss_phase_weight = phase_weight(beta, engine.W).to(engine.device) if sigmoid_loss else None
        opt = str(optimizer).lower()
        engine.optimizer = "nadam" if "nadam" in opt else ("sgd" if "sgd" in opt else "adam")
        if "adam" not in opt and "sgd" not in opt:
            raise ValueError("optimizer must name adam, nadam or sgd (main_training.py:164-169)")
        self.use_graph = bool(graph) and not dp and engine.optimizer == "adam"
        self._graphs, self._g_in = {}, None
        # With a side stream in the engine (overlap_wgrad) the optimizer also leaves the critical path: Adam runs bucket by
        # bucket on a third stream as soon as a bucket's gradients are final (and, data-parallel, all-reduced), while the
        # backward pass continues.  A finished bucket's parameters are never read again by that backward pass.
        self.adam_stream = getattr(engine, "opt_stream", None) if getattr(engine, "wg_stream", None) is not None else None
        self._adam_args = None
        self._lr_now = lr
        self._pending = False
        self._total_on_device = False
        self._anchor = torch.zeros((), device=engine.device, requires_grad=True)     # makes compute_loss() a graph leaf's consumer
        if dp or self.adam_stream is not None:
            bounds = [s_.offset + (-(-s_.numel // ALIGN) * ALIGN) for s_ in engine.specs.values()]
            self.bucketer = GradBucketer(engine.grad, bounds, bucket_bytes, group, reduce=dp,
                                         on_bucket=self._adam_bucket if self.adam_stream is not None else None, runtime=self.rt,
                                         before_bucket=getattr(engine, "flush_reduces", None))

    def broadcast_parameters(self, src=0):
        """Replicas start from identical variables (MirroredStrategy mirrors them at creation)."""
        if self.world_size > 1:
            self.rt.broadcast(self.engine.theta, src, self.group)
            for b in self.engine.moving.values():
                self.rt.broadcast(b, src, self.group)
            self.engine.t_dirty = True

    def _adam_bucket(self, lo, hi, work):
        """Adam on parameters [lo, hi) once everything queued so far on the calling stream (the engine hands buckets over
        from its side stream, which has waited for the main stream) and the bucket's all-reduce are done."""
        rt = self.rt
        ev = rt.record()
        with rt.on(self.adam_stream):
            rt.wait(self.adam_stream, ev)
            if work is not None:
                rt.wait_work(work)
            self.engine.adam_range(lo, hi, *self._adam_args)

    # ---- the three phases of a step after the forward pass
    def _backward_and_reduce(self):
        """tape.gradient + the cross-replica SUM (+ Adam of every bucket that is final, when the optimizer has its own stream)."""
        eng = self.engine
        if self.adam_stream is not None:
            self._adam_args = eng.adam_begin(self._lr_now)
        if self.bucketer is not None:
            self.bucketer.reset()
            eng.backward(on_ready=self.bucketer.mark_ready)
        else:
            eng.backward()
        self._pending = True

    def apply_gradients(self):
        """optimizer.apply_gradients (main_training.py:268): waits for the all-reduce, applies Adam (the part of it that has not
        already run bucket by bucket), joins the optimizer stream."""
        if not self._pending:
            raise RuntimeError("apply_gradients() without a backward pass")
        eng, rt = self.engine, self.rt
        if self.bucketer is not None:
            self.bucketer.finish()              # hands over the last bucket(s); every all-reduce is waited for
        if self.adam_stream is not None:
            rt.wait(rt.current_stream(), rt.record(self.adam_stream))
            eng.t_dirty = True
        else:
            eng.adam_step(self._lr_now)
        self._pending = False

    def _make_mask(self):
        eng = self.engine
        side = getattr(eng, "wg_stream", None) if getattr(eng, "mask_on_side_stream", False) else None
        if side is not None:       # its only consumers (the information-vector branch, forward and backward) run on that stream
            if eng.device_counters is not None:          # the draw number comes from the step_advance launch of the main stream
                self.rt.wait(side, self.rt.record())
            with self.rt.on(side):
                return eng.make_dropout_mask()
        return eng.make_dropout_mask()

    def _step_body(self, spec_in, emb, spec_out, dropout_mask, with_reg):
        """The launches of one step (what a HIP graph captures)."""
        eng = self.engine
        eng.begin_step(self._lr_now, n_draws=eng.n_dropout_draws if (dropout_mask is None and self.dropout) else 0)     # device counters only
        if dropout_mask is None and self.dropout:
            dropout_mask = self._make_mask()
        gb = eng.B * self.world_size
        eng.forward(spec_in, emb, dropout_mask=dropout_mask, target=spec_out, global_batch=gb, alpha=self.alpha)
        if with_reg:
            eng.reg_loss()                      # on the pre-update weights, as compute_loss sees them
        self._backward_and_reduce()
        self.apply_gradients()

    def step(self, spec_in, emb, spec_out, dropout_mask=None, lr=None, return_loss=False):
        """inputs as DataGenerator.__getitem__ yields them (datageneratorv2.py:101-102), NCHW, per-replica shard."""
        eng = self.engine
        eng.training = True
        self._total_on_device = False
        self._lr_now = self.lr if lr is None else lr
        if self.use_graph and dropout_mask is None:
            self._graph_step(spec_in, emb, spec_out, bool(return_loss))
        else:
            if self.use_graph:
                eng.use_device_counters(False)     # e.g. an externally supplied mask: launch arguments from the host again
            self._step_body(spec_in, emb, spec_out, dropout_mask, return_loss)
        if return_loss:
            return self.last_loss()
        return None

    # ---- the step as a HIP graph
    def _snapshot(self):
        eng = self.engine
        return {"theta": eng.theta.clone(), "m": eng.adam_m.clone(), "v": eng.adam_v.clone(), "adam_t": eng.adam_t,
                "moving": {k: v.clone() for k, v in eng.moving.items()}, "dropout_step": eng._shared["dropout_step"]}

    def _restore(self, snap):
        eng = self.engine
        eng.theta.copy_(snap["theta"]); eng.adam_m.copy_(snap["m"]); eng.adam_v.copy_(snap["v"])
        for k, v in snap["moving"].items():
            eng.moving[k].copy_(v)
        eng.adam_t = snap["adam_t"]
        eng._shared["dropout_step"] = snap["dropout_step"]
        eng.t_dirty = True
        eng.sync_device_counters()

    def _graph_step(self, spec_in, emb, spec_out, with_reg):
        """Replay the captured step on copies of the inputs.  First call (per variant: with / without the l2 terms of the reported
        loss): one ordinary step with the counters in device memory - every lazily created buffer, table and workspace then
        exists - undone from a snapshot, then the capture."""
        eng, rt = self.engine, self.rt
        eng.use_device_counters(True)          # (re-)synchronises the device counters when another path ran in between
        if self._g_in is None:
            if emb.dtype not in (torch.int32, torch.int64):
                emb = emb.to(torch.int64)
            self._g_in = (spec_in.to(eng.device, torch.float32).contiguous().clone(), emb.to(eng.device).contiguous().clone(),
                          spec_out.to(eng.device, torch.float32).contiguous().clone())
        for dst, src in zip(self._g_in, (spec_in, emb, spec_out)):
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"graph step: input shape {tuple(src.shape)} differs from the captured {tuple(dst.shape)}")
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        eng.set_step_cfg(self._lr_now)
        g = self._graphs.get(with_reg)
        if g is None:
            snap = self._snapshot()
            self._step_body(*self._g_in, None, with_reg)
            rt.synchronize()
            self._restore(snap)
            rt.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._step_body(*self._g_in, None, with_reg)
            # the capture pass advanced the host mirrors as a step does, but ran nothing: undo it on the host only
            eng.adam_t = snap["adam_t"]
            eng._shared["dropout_step"] = snap["dropout_step"]
            self._graphs[with_reg] = g
        g.replay()
        eng.adam_t += 1                          # host mirrors of the device counters
        eng._shared["dropout_step"] += eng.n_dropout_draws if self.dropout else 0
        eng.t_dirty = True

    def compute_loss(self, y_true, y_pred, lr=None):
        """compute_loss(spec_out, spec_generated, model.model.losses) of main_training.py:203-235 for the prediction the model has
        just produced (`y_pred` must be the output of the last forward call of this trainer's model: NHWC from `model.model(...)`
        or NCHW from `model(...)`).  Returns a differentiable 0-dim tensor: data term / global batch + l2 terms / replicas."""
        eng = self.engine
        eng.use_device_counters(False)             # the module path passes its per-step scalars as launch arguments
        if y_pred.data_ptr() != eng.pred.data_ptr():
            raise ValueError("compute_loss needs the prediction of the model's last forward pass (its own output buffer)")
        tgt = y_true
        if tgt.dim() == 4 and tgt.shape[1] != 2 and tgt.shape[-1] == 2:
            tgt = tgt.permute(0, 3, 1, 2)
        tgt = tgt.to(eng.device, torch.float32).contiguous()
        self._lr_now = self.lr if lr is None else lr
        eng.loss_from_logits(tgt, eng.B * self.world_size, self.alpha)       # fused sigmoid + loss + dL/dlogits (seeds backward)
        self._total_on_device = True
        return _LossFunction.apply(self, self._anchor, eng.loss_total())

    def last_loss(self):
        """Scalar loss of the last step incl. the l2 term (host sync).  Per-replica share: SUM over replicas gives the
        value strategy.reduce(SUM, ...) returns (main_training.py:326)."""
        eng = self.engine
        if self._total_on_device:                # compute_loss() path: data + l2 terms were summed on the device
            return float(eng.loss_tot[0])
        return float(eng.loss_out[0]) + float(eng.reg_out[0])


class CheckpointManager:
    """tf.train.CheckpointManager(checkpoint, directory, max_to_keep=2) of main_training.py:171-172: numbered checkpoints
    holding the model variables (incl. the BatchNorm moving statistics) and the optimizer state, oldest ones deleted."""

    def __init__(self, trainer: "Trainer", directory, max_to_keep=2):
        self.trainer, self.directory, self.max_to_keep = trainer, directory, max_to_keep
        os.makedirs(directory, exist_ok=True)

    def _paths(self):
        out = []
        for f in os.listdir(self.directory):
            if f.startswith("ckpt-") and f.endswith(".pt"):
                try:
                    out.append((int(f[5:-3]), os.path.join(self.directory, f)))
                except ValueError:
                    pass
        return sorted(out)

    @property
    def latest_checkpoint(self):
        p = self._paths()
        return p[-1][1] if p else None

    def save(self, epoch=None):
        """manager.save() (main_training.py:364-365).  Rank 0 only writes in a multi-process job."""
        if dist.is_initialized() and dist.get_rank() != 0:
            return None
        eng = self.trainer.engine
        p = self._paths()
        n = (p[-1][0] + 1) if p else 1
        path = os.path.join(self.directory, f"ckpt-{n}.pt")
        state = {
            "format": 1, "epoch": epoch, "lr": self.trainer.lr, "adam_t": eng.adam_t,
            # the BASE seed of the Dropout streams: replica r draws from stream base + r (Trainer.__init__), and every rank restores
            # this one file (rank 0 writes it)
            "dropout_step": int(eng._shared["dropout_step"]),
            "dropout_seed": int((eng.dropout_seed - (self.trainer.rank if getattr(eng, "_seed_offset_by_rank", False) else 0)) & 0xFFFFFFFF),
            "optimizer": eng.optimizer, "m_schedule": float(eng._shared.get("m_schedule", 1.0)),
            "layout": [(k, tuple(s_.shape), int(s_.offset)) for k, s_ in eng.specs.items()],
            "theta": eng.theta.detach().cpu(), "adam_m": eng.adam_m.detach().cpu(), "adam_v": eng.adam_v.detach().cpu(),
            "moving": {k: v.detach().cpu() for k, v in eng.moving.items()},
        }
        tmp = path + ".tmp"
        torch.save(state, tmp)
        os.replace(tmp, path)
        for _, old in self._paths()[:-self.max_to_keep]:
            os.remove(old)
        return path

    def restore(self, path=None):
        """checkpoint.restore(manager.latest_checkpoint): variables, moving statistics and Adam slots / step count."""
        path = path or self.latest_checkpoint
        if path is None:
            return None
        eng = self.trainer.engine
        state = torch.load(path, map_location="cpu", weights_only=True)
        layout = [(k, tuple(s_.shape), int(s_.offset)) for k, s_ in eng.specs.items()]
        if [tuple(x) for x in state["layout"]] != layout:
            raise ValueError("checkpoint was written for a different model configuration")
        eng.theta.copy_(state["theta"]); eng.adam_m.copy_(state["adam_m"]); eng.adam_v.copy_(state["adam_v"])
        for k, v in state["moving"].items():
            eng.moving[k].copy_(v)
        eng.adam_t = int(state["adam_t"])
        if "dropout_step" in state:              # the Dropout stream continues where it stopped (older checkpoints: from draw 0)
            eng._shared["dropout_step"] = int(state["dropout_step"])
            # base seed + this replica's rank: the replicas keep independent masks after a resume (MirroredStrategy draws per replica)
            by_rank = self.trainer.world_size > 1
            eng.dropout_seed = (int(state["dropout_seed"]) + (self.trainer.rank if by_rank else 0)) & 0xFFFFFFFF
            eng._seed_offset_by_rank = by_rank
            self.trainer._graphs = {}            # the seed is a launch argument frozen into a captured step: capture again
        if "m_schedule" in state:
            eng._shared["m_schedule"] = float(state["m_schedule"])
        eng.t_dirty = True
        eng.sync_device_counters()
        return state.get("epoch")


def fit(trainer: "Trainer", train_batches, n_epochs, val_batches=None, manager: CheckpointManager = None, lr0=None,
        lr_exp_decay=(True, 80), start_epoch=0, log=print, val_updates_moving=True):
    """The epoch loop of main_training.py:336-390: exponential rate from epoch 80, running means of the amplitude / phase
    loss terms, a checkpoint every second epoch.  `train_batches` / `val_batches` are callables returning an iterable of
    (spec_in, emb, spec_out) per-replica shards for the epoch.

    Reported values are the reference's: `train_loss` = mean over steps of the cross-replica SUM of the per-replica loss
    (data term / global batch + l2 terms / replicas: strategy.reduce(SUM, ...), main_training.py:323-327, :345-352);
    `*_amp` / `*_phase` = tf.keras.metrics.Mean over every (b, h, w) element of (a - a^)^2 and 1 - cos(...) across steps and
    replicas (main_training.py:239-244, :279-284).  The validation pass calls the model with training=True as the reference's
    test_step does (main_training.py:297-300): batch statistics, dropout active - and, as in Keras, the BatchNorm moving
    statistics move during it (`val_updates_moving=False` keeps them at their post-training values instead)."""
    eng = trainer.engine
    lr0 = trainer.lr if lr0 is None else lr0
    per_elem = 1.0 / (eng.H * eng.W * eng.B * trainer.world_size)       # loss_out[1:3] are raw per-replica sums of the two terms
    history = []

    def reduce_(t):
        if trainer.world_size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=trainer.group)
        return t.cpu()

    for epoch in range(start_epoch, n_epochs):
        lr = lr_schedule(lr0, epoch, lr_exp_decay)
        tot = torch.zeros(4, dtype=torch.float64, device=eng.device)      # data loss, amplitude sum, phase sum, l2 terms
        nb = 0
        eng.reg_loss()
        reg0 = eng.reg_out[0].double()           # l2 terms on the weights the first step of the epoch sees
        for spec_in, emb, spec_out in train_batches(epoch):
            trainer.step(spec_in, emb, spec_out, lr=lr)
            tot[:3] += eng.loss_out[:3].double()     # total data loss, amplitude term, phase term of this step (raw sums ~1e5: fp64)
            nb += 1
        # the l2 terms drift over an epoch (Adam moves every weight by ~lr per step whatever the gradient's scale): the reported
        # mean takes the trapezoid of their value before the first and after the last step instead of nine reductions per step
        eng.reg_loss()
        tot[3] = 0.5 * (reg0 + eng.reg_out[0].double()) * max(nb, 1)
        tot = reduce_(tot)
        n = max(nb, 1)
        rec = {"epoch": epoch + 1, "lr": lr, "train_loss": float(tot[0] + tot[3]) / n,
               "train_amp": float(tot[1]) * per_elem / n, "train_phase": float(tot[2]) * per_elem / n}
        if val_batches is not None:
            vt = torch.zeros(3, dtype=torch.float64, device=eng.device)
            vb = 0
            saved_moving = None if val_updates_moving else {k: v.clone() for k, v in eng.moving.items()}
            for spec_in, emb, spec_out in val_batches(epoch):
                eng.training = True
                eng.begin_step(lr, n_draws=eng.n_dropout_draws if trainer.dropout else 0, forward_only=True)     # device counters only
                mask = eng.make_dropout_mask() if trainer.dropout else None
                eng.forward(spec_in, emb, dropout_mask=mask, target=spec_out, global_batch=eng.B * trainer.world_size, alpha=trainer.alpha)
                vt += eng.loss_out[:3].double()
                vb += 1
            if saved_moving is not None:
                for k, v in saved_moving.items():
                    eng.moving[k].copy_(v)
            vt = reduce_(vt)
            m = max(vb, 1)
            rec.update(val_loss=float(vt[0]) / m, val_amp=float(vt[1]) * per_elem / m, val_phase=float(vt[2]) * per_elem / m)
        if manager is not None and epoch % 2 == 0:
            rec["checkpoint"] = manager.save(epoch=epoch)
        history.append(rec)
        if log is not None:
            log(rec)
    return history
