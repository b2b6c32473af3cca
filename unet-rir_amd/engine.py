"""Execution plan of the U-Net train step on one MI355X: buffers, forward schedule, backward
schedule.  Mirrors the Keras graph UNet._build creates (dl_models/u_net.py:201-251, mode 0) and
the tape.gradient pass of main_training.py:256-267, as an explicit list of HIP kernel launches.

Memory layout in HBM (all fp32):
  * activations NHWC, one buffer per layer output, allocated once for a fixed batch size;
    the skip concat of level l is ONE buffer [B,Hl,Wl,2*Cl]: the encoder's BN+ReLU output is written
    into channels [0,Cl), the Conv2DTranspose output into [Cl,2Cl) (dl_models/u_net.py:308 costs no copy);
  * parameters, gradients and Adam moments are four flat buffers with identical layout, ordered by
    backward completion (head first, enc1 last) so gradient buckets for the all-reduce are contiguous
    slices that become final in order;
  * Conv2D kernels [Cout][k][k][Cin], Conv2DTranspose kernels [Cin][k][k][Cout], Dense [out][in];
    the first conv's Cin and the head's Cout are zero-padded 2 -> 4 (pad weights stay exactly zero:
    their gradient is identically zero).
"""
import math
import struct
from collections import OrderedDict

import torch

from . import _lib, ops
from .device import HipRuntime
from .ops import Act

BN_EPS = 1e-3          # keras BatchNormalization default (dl_models/u_net.py:368)
BN_MOMENTUM = 0.99
L2_COEF = 1e-3         # l2(0.001) on strided Conv2D / Conv2DTranspose kernels (dl_models/u_net.py:274, :302)
VOCAB, EMB_DIM, VEC_CH = 2000, 256, 16   # dl_models/u_net.py:255-257
DROPOUT_P = 0.3        # dl_models/u_net.py:260
ALIGN = 64             # parameter offsets are multiples of 64 floats (256 B)


def same_out(n, s):
    return -(-n // s)


def _as_f32(v):
    """v rounded to fp32, as a Python float."""
    return struct.unpack("f", struct.pack("f", float(v)))[0]


class ParamSpec:
    __slots__ = ("name", "shape", "offset", "numel", "kind", "keras_shape")

    def __init__(self, name, shape, kind, keras_shape):
        self.name, self.shape, self.kind, self.keras_shape = name, tuple(shape), kind, tuple(keras_shape)
        self.numel = int(math.prod(shape))
        self.offset = -1


def pick_concurrent_streams(device, n, candidates=12):
    """n HIP streams that really run beside the current stream and beside each other.  Streams are multiplexed onto a few
    hardware queues, and two streams that share a queue execute in order: the side-stream schedule then gains nothing (or
    loses: measured 14.1 vs 15.0-17.3 ms per step depending on which pool streams an engine happened to get).  So probe:
    queue ~2 ms of copies on stream A, then a tiny kernel on candidate B; B is concurrent with A if its kernel finishes
    while A is still busy."""
    dev = torch.device(device)
    main = torch.cuda.current_stream(dev)
    a = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    tiny = torch.zeros(64, device=dev)

    def runs_beside(busy, cand):
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(busy):
            for _ in range(24):
                b.copy_(a, non_blocking=True)
            end_busy = torch.cuda.Event()
            end_busy.record(busy)
        with torch.cuda.stream(cand):
            tiny.add_(1.0)
            end_cand = torch.cuda.Event()
            end_cand.record(cand)
        end_cand.synchronize()
        ok = not end_busy.query()
        torch.cuda.synchronize(dev)
        return ok

    chosen, pool = [], [torch.cuda.Stream(device=dev) for _ in range(candidates)]
    for c in pool:
        if len(chosen) == n:
            break
        if all(runs_beside(o, c) and runs_beside(c, o) for o in [main] + chosen):
            chosen.append(c)
    for c in pool:                      # fewer independent queues than asked for: fill up (correct, just less overlap)
        if len(chosen) == n:
            break
        if c not in chosen:
            chosen.append(c)
    return chosen


class _SideStream:
    """`with engine._wg() as ws:` - enqueue on the weight-gradient stream after everything the main stream has queued."""

    def __init__(self, eng):
        self.eng = eng
        self.ctx = None

    def __enter__(self):
        eng = self.eng
        if eng.wg_stream is None:
            return eng.ws
        eng.rt.wait(eng.wg_stream, eng.rt.record())
        self.ctx = eng.rt.on(eng.wg_stream)
        self.ctx.__enter__()
        eng._flush_ready()         # buckets whose hand-over was deferred to this event (see UNetEngine.backward.ready)
        return eng.ws_w

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class DeviceCounters:
    """Mixin of the engines: the per-step scalars (Adam's bias-corrected rate, the dropout draw number) either travel as launch
    arguments computed on the host (default) or live in DEVICE memory, advanced by one tiny kernel at the start of every step
    (`use_device_counters()`): a step then consists of the same launches with the same arguments every time and can be captured
    once into a HIP graph and replayed (trainer.Trainer(graph=True)).  The host mirrors (`adam_t`, `dropout_step`) are kept in
    step either way; checkpoints hold the host values."""
    n_dropout_draws = 1          # dropout masks drawn per step
    # compute_loss switches (main_training.py:38-39, :214-222); set through trainer.Trainer(sigmoid_loss=, diff_loss=, beta=)
    loss_diff = False            # diff_loss: the phase target is phase_true - phase of the network input
    loss_phase_weight = None     # sigmoid_loss: fp32 [W] column weights of the phase term (device tensor)
    _last_spec = None            # the input of the last forward pass (diff_loss through loss_from_logits)

    def _loss_extras(self, spec=None):
        if spec is not None:
            self._last_spec = spec
        ref = self._last_spec if self.loss_diff else None
        if self.loss_diff and ref is None:
            raise RuntimeError("diff_loss needs the network input of the last forward pass")
        return {"phase_ref": ref, "phase_weight": self.loss_phase_weight}

    def use_device_counters(self, on=True):
        dev = self._shared.get("dev")
        if not on:
            if dev is not None:
                dev["on"] = False          # the tensors stay alive: a captured graph may still reference them
            return
        if dev is None:
            dev = self._shared["dev"] = {"state": torch.zeros(3, dtype=torch.int64, device=self.device),
                                         "cfg": torch.zeros(8, dtype=torch.float32, device=self.device),
                                         "hyper": torch.zeros(8, dtype=torch.float32, device=self.device), "cfg_host": None, "offset": 0,
                                         "on": False}
        if not dev["on"]:
            dev["on"] = True
            self.sync_device_counters()

    def _dev(self):
        dev = self._shared.get("dev")
        return dev if (dev is not None and dev["on"]) else None

    def sync_device_counters(self):
        """Host counters -> device (when the mode is switched on, after reset_parameters, after a restored checkpoint)."""
        dev = self._dev()
        if dev is not None:
            dev["state"].copy_(torch.tensor([self.adam_t, self._shared["dropout_step"], self._shared["dropout_step"]], dtype=torch.int64))

    @property
    def device_counters(self):
        return self._dev()

    def begin_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0, n_draws=None, forward_only=False):
        """Device counters only: the launch that opens a step (before the dropout masks are drawn; n_draws of them, default
        n_dropout_draws).  forward_only: a pass without an optimizer step (validation) - the Adam step count stays."""
        dev = self._dev()
        if dev is None:
            return
        self.set_step_cfg(lr, beta1, beta2, eps, grad_scale)
        ops.step_advance(dev["state"], dev["cfg"], dev["hyper"], self.n_dropout_draws if n_draws is None else n_draws, not forward_only)
        dev["offset"] = 0

    def set_step_cfg(self, lr, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        """Device counters only: the optimizer's scalars as the next step_advance launch reads them.  Copies only when a value
        changed (the rate changes once per epoch, main_training.py:342-344): a blocking 20-byte transfer, outside any graph."""
        dev = self._dev()
        cfg = (float(lr), beta1, beta2, eps, float(grad_scale))
        if dev is not None and dev["cfg_host"] != cfg:
            dev["cfg"][:5].copy_(torch.tensor(cfg, dtype=torch.float32))
            dev["cfg_host"] = cfg

    def _draw_mask(self, buf):
        """Fill `buf` with the next keep mask of this engine's dropout stream."""
        dev = self._dev()
        if dev is None:
            ops.dropout_mask(buf, DROPOUT_P, self.dropout_seed, self._shared["dropout_step"])
        else:
            ops.dropout_mask_dev(buf, DROPOUT_P, self.dropout_seed, dev["state"], dev["offset"])
            dev["offset"] += 1
        self._shared["dropout_step"] += 1
        return buf

    optimizer = "adam"           # "adam" | "nadam" | "sgd" (main_training.py:164-169); set through trainer.Trainer(optimizer=)

    def _adam(self, lo, hi, *args):
        dev = self._dev()
        th, g, m, v = (self.theta, self.grad, self.adam_m, self.adam_v) if lo is None else \
            (self.theta[lo:hi], self.grad[lo:hi], self.adam_m[lo:hi], self.adam_v[lo:hi])
        if args[0] == "sgd":
            ops.sgd(th, g, args[1], args[2])
        elif args[0] == "nadam":
            ops.nadam(th, g, m, v, *args[1:])
        elif dev is None:
            ops.adam(th, g, m, v, *args)
        else:
            ops.adam_dev(th, g, m, v, dev["hyper"])
        self.t_dirty = True

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        """One optimizer step over the whole flat parameter buffer in one launch (optimizer.apply_gradients, main_training.py:268)."""
        self._adam(None, None, *self.adam_begin(lr, beta1, beta2, eps, grad_scale))

    def adam_begin(self, lr, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        """Advance the step count once and return the arguments of adam_range for this step (bucket-wise optimizer).  Adam: the
        bias-corrected rate; Nadam: the momentum-schedule coefficients of this step (the running product of the schedule lives
        beside the step count); SGD: the rate."""
        self.adam_t += 1
        t = self.adam_t
        if self.optimizer == "sgd":
            return ("sgd", lr, grad_scale)
        if self.optimizer == "nadam":
            mu_t = beta1 * (1.0 - 0.5 * 0.96 ** (0.004 * t))
            mu_t1 = beta1 * (1.0 - 0.5 * 0.96 ** (0.004 * (t + 1)))
            ms_new = self._shared.get("m_schedule", 1.0) * mu_t
            self._shared["m_schedule"] = ms_new
            return ("nadam", lr, beta1, beta2, eps, (1.0 - mu_t) / (1.0 - ms_new), mu_t1 / (1.0 - ms_new * mu_t1), 1.0 / (1.0 - beta2 ** t), grad_scale)
        # the betas as the kernels see them (fp32), so that the launched step and the device-counter step (step_advance_kernel computes
        # the same expression from its fp32 cfg in fp64) produce the same bias-corrected rate bit for bit
        b1, b2 = _as_f32(beta1), _as_f32(beta2)
        return (lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t), beta1, beta2, eps, grad_scale)

    def adam_range(self, lo, hi, *args):
        """The optimizer on the flat parameter range [lo, hi) (element offsets, multiples of the 64-float alignment)."""
        self._adam(lo, hi, *args)


class UNetEngine(DeviceCounters):
    """One replica of the model for a fixed per-replica batch size B on one device."""
    mask_on_side_stream = True       # the dropout mask's only consumers (information-vector branch) run on the side stream

    def __init__(self, H, W, B, F0=32, k=3, depth=4, batchnorm=True, inf_vector_shape=(2, 16), s0=1, s=2,
                 device="cuda:0", n_replicas=1, dtype="f32", overlap_wgrad=False, runtime=None, share=None,
                 fused_stats=True, defer_ready=True):
        """runtime: stream / event provider (device.HipRuntime by default; the CPU tests pass a simulated one).
        share: another UNetEngine of the same configuration whose parameters, gradients, Adam moments, work copies and
        BatchNorm moving statistics this engine aliases (only the activation buffers depend on the batch size)."""
        self.rt = runtime if runtime is not None else HipRuntime(device)
        if s0 != 1 or s != 2:
            raise NotImplementedError("HIP path implements resize_factor_0=[1,1], res_factor=[2,2] (the reference defaults)")
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        # storage type of activations and their gradients; parameters, statistics and weight gradients are always fp32
        self.dtype = dtype
        self.fused_stats = bool(fused_stats)     # conv-epilogue BN statistics / bias gradients (bf16); False: separate reduction passes
        self._cst_rows, self._cst_buf, self._cst_gen = {}, None, ops.config_generation()
        self._cast_table = None
        self.adt = torch.float32 if dtype == "f32" else torch.bfloat16
        self.PAD = 4 if dtype == "f32" else 8          # channel granule = 16 bytes
        if F0 % self.PAD:
            raise ValueError(f"number_filters_0 must be a multiple of {self.PAD} for dtype {dtype}")
        # kernels != 3 in bf16 storage (kernels=6 is the reference's constructor default, dl_models/u_net.py:40-45): forward and data
        # gradients on the tap-table kernels, weight gradients on the tap-table weight-gradient kernel with bf16 operand loads
        if k < 1 or k > 6:
            raise ValueError("kernels must be in 1..6")
        self.H, self.W, self.B, self.F0, self.k, self.depth = H, W, B, F0, k, depth
        self.batchnorm = batchnorm
        self.inf_vector_shape = tuple(inf_vector_shape)
        self.device = torch.device(device)
        self.n_replicas = n_replicas
        self.L = depth + 1
        self.ch = [F0 * 2 ** l for l in range(self.L)]
        self.hw = [(H, W)]
        for _ in range(depth):
            h, w = self.hw[-1]
            self.hw.append((same_out(h, 2), same_out(w, 2)))
        for l in range(depth):
            if self.hw[l][0] != 2 * self.hw[l + 1][0] or self.hw[l][1] != 2 * self.hw[l + 1][1]:
                raise ValueError("spatial size must halve exactly at every level that feeds a skip concat")
        self.h5, self.w5 = self.hw[-1]
        self.n_idx = int(math.prod(self.inf_vector_shape))
        self.vec_in = self.n_idx * EMB_DIM
        self.vec_dim = self.h5 * self.w5 * VEC_CH
        self._shared = share._shared if share is not None else {"adam_t": 0, "t_dirty": True, "dropout_step": 0}
        self.dropout_seed = share.dropout_seed if share is not None else (torch.initial_seed() & 0xFFFFFFFF)
        self._mask_buf = None
        if share is not None and (share.H, share.W, share.F0, share.k, share.depth, share.batchnorm, share.dtype, share.inf_vector_shape) != \
                (H, W, F0, k, depth, batchnorm, dtype, self.inf_vector_shape):
            raise ValueError("share= needs an engine of the same configuration (only the batch size may differ)")
        self._build_params(share)
        self._alloc()
        self.ws = ops.Workspace(self.device)
        self._reserve_workspace()
        # bf16 storage: the split-K reductions of weight gradients with SMALL slab sets (<= 16 MB: small images, narrow layers) are parked
        # and run together (ops.ReduceBatch); the 37.7 MB slab sets of configs[1] reduce at once, while they are still in the
        # Infinity Cache (parking them measured 0.06-0.09 ms per step slower)
        self._rb = ops.ReduceBatch(self.device, 96 << 20, park_max_bytes=16 << 20) if ops.wgrad_defer_supported(self.dtype) else None
        self.park_reduces = self._rb is not None          # False: every weight gradient reduces its slabs at once (A/B, scripts/ab_switch.py)
        self.training = True
        self._pending_ready = []
        self._defer_ready = bool(defer_ready)    # park bucket hand-overs until the side stream next waits for the main stream
        # overlap_wgrad: weight gradients depend only on tensors the main stream has already produced, so they can run on a
        # side HIP stream (own scratch buffer) beside the dgrad -> BatchNorm-backward chain.  Measured -0.5 ms per step
        # (3 %) in bf16 at cfg 2 (scripts/overlap_ab.py, alternating engines in one process), 0 % in fp32.  bench.py turns
        # it on; the default stays off because overlapping launches make per-kernel event brackets (tests, roofline of the
        # backward kernels) ill-defined.
        # two probed streams: weight gradients, and the trainer's bucket-wise optimizer (trainer.py)
        if share is not None and overlap_wgrad and share.wg_stream is not None:
            self.wg_stream, self.opt_stream = share.wg_stream, share.opt_stream
        else:
            self.wg_stream, self.opt_stream = self.rt.concurrent_streams(2) if overlap_wgrad else (None, None)
        self.ws_w = ops.Workspace(self.device, self.ws.nbytes) if overlap_wgrad else self.ws
        self.head_direct = ops.head6x6_supported(self.ch[0])
        if self.head_direct:
            self.ws.reserve(512 * 2 * 36 * self.ch[0] * 4)
            self.ws_w.reserve(512 * 2 * 36 * self.ch[0] * 4)
        elif self.dtype == "bf16":
            raise ValueError("the bf16 path needs number_filters_0 % 8 == 0 (direct head kernels)")

    # state shared by every engine built over one parameter set (UNet keeps one engine per batch size)
    @property
    def adam_t(self):
        return self._shared["adam_t"]

    @adam_t.setter
    def adam_t(self, v):
        self._shared["adam_t"] = v

    @property
    def t_dirty(self):
        """The work copies (transposed / bf16 kernels) are older than the master parameters."""
        return self._shared["t_dirty"]

    @t_dirty.setter
    def t_dirty(self, v):
        self._shared["t_dirty"] = v

    # ------------------------------------------------------------------ parameters
    def _build_params(self, share=None):
        k, ch, L = self.k, self.ch, self.L
        specs = []

        def add(name, shape, kind, keras_shape):
            specs.append(ParamSpec(name, shape, kind, keras_shape))

        def bn(prefix, c):
            if self.batchnorm:
                add(prefix + ".gamma", (c,), "gamma", (c,))
                add(prefix + ".beta", (c,), "beta", (c,))

        # backward completion order: head, dec1..decD, vec, encL..enc1
        PAD = self.PAD
        add("head.kernel", (PAD, 6, 6, ch[0]), "conv_padout", (6, 6, ch[0], 2))
        add("head.bias", (PAD,), "bias_pad", (2,))
        for l in range(1, self.depth + 1):
            c = ch[l - 1]
            add(f"dec{l}.cb1b.kernel", (c, 3, 3, c), "conv", (3, 3, c, c))
            add(f"dec{l}.cb1b.bias", (c,), "bias", (c,))
            bn(f"dec{l}.cb1b", c)
            add(f"dec{l}.cb1a.kernel", (c, k, k, 2 * c), "conv", (k, k, 2 * c, c))
            add(f"dec{l}.cb1a.bias", (c,), "bias", (c,))
            bn(f"dec{l}.cb1a", c)
            add(f"dec{l}.up.kernel", (ch[l], k, k, c), "convT", (k, k, c, ch[l]))
            add(f"dec{l}.up.bias", (c,), "bias", (c,))
        add("vec.conv.kernel", (ch[-1], 1, 1, VEC_CH), "conv", (1, 1, VEC_CH, ch[-1]))
        add("vec.conv.bias", (ch[-1],), "bias", (ch[-1],))
        add("vec.dense.kernel", (self.vec_dim, self.vec_in), "dense", (self.vec_in, self.vec_dim))
        add("vec.dense.bias", (self.vec_dim,), "bias", (self.vec_dim,))
        add("vec.embedding", (VOCAB, EMB_DIM), "embedding", (VOCAB, EMB_DIM))
        for l in range(L, 0, -1):
            c = ch[l - 1]
            cin = ch[l - 2] if l > 1 else 2
            add(f"enc{l}.cb1.kernel", (c, 3, 3, c), "conv", (3, 3, c, c))
            add(f"enc{l}.cb1.bias", (c,), "bias", (c,))
            bn(f"enc{l}.cb1", c)
            if l > 1:
                add(f"enc{l}.down.kernel", (c, k, k, cin), "conv", (k, k, cin, c))
            else:
                add(f"enc{l}.down.kernel", (c, k, k, PAD), "conv_padin", (k, k, 2, c))
            add(f"enc{l}.down.bias", (c,), "bias", (c,))
        off = 0
        for s_ in specs:
            s_.offset = off
            off += -(-s_.numel // ALIGN) * ALIGN
        self.specs = OrderedDict((s_.name, s_) for s_ in specs)
        self.n_flat = off
        dev = self.device
        if share is not None:
            self.theta, self.grad, self.adam_m, self.adam_v = share.theta, share.grad, share.adam_m, share.adam_v
        else:
            self.theta = torch.zeros(off, dtype=torch.float32, device=dev)
            self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
            self.adam_m = torch.zeros(off, dtype=torch.float32, device=dev)
            self.adam_v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.p = {n: self.theta[s_.offset:s_.offset + s_.numel].view(s_.shape) for n, s_ in self.specs.items()}
        self.g = {n: self.grad[s_.offset:s_.offset + s_.numel].view(s_.shape) for n, s_ in self.specs.items()}
        # transposed work copies: Conv2D kernels for their data gradient, Conv2DTranspose kernels for their forward
        toff, self.tspec = 0, {}
        # the Dense kernel (49 % of the parameters) needs no transposed copy when its data gradient can read it as stored
        self.dense_direct = ops.dense_dgrad_supported(self.B, self.vec_in, self.vec_dim) if share is None else share.dense_direct
        if share is not None and self.dense_direct and not ops.dense_dgrad_supported(self.B, self.vec_in, self.vec_dim):
            raise ValueError("this batch size needs a transposed Dense kernel copy the shared parameter set does not hold")
        for n, s_ in self.specs.items():
            if s_.kind == "dense" and self.dense_direct:
                continue
            if s_.kind in ("conv", "convT", "dense", "conv_padout"):
                self.tspec[n] = toff
                toff += -(-s_.numel // ALIGN) * ALIGN
        self.theta_t = share.theta_t if share is not None else torch.zeros(max(toff, 4), dtype=torch.float32, device=dev)
        self.pt = {n: self.theta_t[o:o + self.specs[n].numel] for n, o in self.tspec.items()}
        # bf16 mode: bf16 work copies of every trunk kernel in both orientations (the information-vector branch stays fp32)
        self.ph, self.pth, self.ppk = {}, {}, {}
        if self.dtype == "bf16":
            hoff, hspec = 0, []
            for n, s_ in self.specs.items():
                if s_.kind in ("conv", "convT", "conv_padin", "conv_padout") and not n.startswith("vec."):
                    hspec.append((n, hoff)); hoff += -(-s_.numel // ALIGN) * ALIGN
            self.theta_h = share.theta_h if share is not None else torch.zeros(max(hoff, 8), dtype=torch.bfloat16, device=dev)
            self.theta_th = share.theta_th if share is not None else torch.zeros(max(hoff, 8), dtype=torch.bfloat16, device=dev)
            for n, o in hspec:
                k_ = self.specs[n].numel
                self.ph[n] = self.theta_h[o:o + k_]
                self.pth[n] = self.theta_th[o:o + k_]
            # stride-2 3x3 kernels: a third copy in the order the stride-2 forward kernel's LDS-DMA reads it (csrc/conv3x3d.hip)
            self.ppk = share.ppk if share is not None else {}
            for n, _ in hspec if share is None else ():
                s_ = self.specs[n]
                strided = n.endswith(".up.kernel") or (n.endswith(".down.kernel") and not n.startswith("enc1."))
                if strided and s_.shape[1] == 3:
                    ne = ops.conv3x3s2_packed_elems(s_.shape[0], s_.shape[3])
                    if ne:
                        self.ppk[n] = torch.zeros(ne, dtype=torch.bfloat16, device=dev)
        # BatchNorm moving statistics (non-trainable)
        self.bn_names = [n[:-len(".gamma")] for n in self.specs if n.endswith(".gamma")]
        self.moving = share.moving if share is not None else {}
        for b in self.bn_names if share is None else ():
            c = self.specs[b + ".gamma"].numel
            self.moving[b + ".moving_mean"] = torch.zeros(c, dtype=torch.float32, device=dev)
            self.moving[b + ".moving_variance"] = torch.ones(c, dtype=torch.float32, device=dev)
        self.l2_names = [f"enc{l}.down.kernel" for l in range(1, L + 1)] + [f"dec{l}.up.kernel" for l in range(1, self.depth + 1)]

    def reset_parameters(self, generator=None):
        """Keras default initialisers (no initialiser argument anywhere in dl_models/u_net.py): glorot_uniform
        kernels, zero biases, gamma 1, beta 0, Embedding U(-0.05, 0.05)."""
        with torch.no_grad():
            for n, s_ in self.specs.items():
                t = self.p[n]
                ks = s_.keras_shape
                if s_.kind == "embedding":
                    t.copy_((torch.rand(s_.shape, generator=generator) * 0.1 - 0.05).to(self.device))
                elif s_.kind in ("conv", "convT", "conv_padin", "conv_padout", "dense"):
                    if len(ks) == 4:
                        rf = ks[0] * ks[1]
                        fan_in, fan_out = ks[2] * rf, ks[3] * rf
                    else:
                        fan_in, fan_out = ks
                    lim = math.sqrt(6.0 / (fan_in + fan_out))
                    w = ((torch.rand(s_.shape, generator=generator) * 2 - 1) * lim).to(self.device)
                    if s_.kind == "conv_padin":
                        w[..., 2:] = 0
                    if s_.kind == "conv_padout":
                        w[2:] = 0
                    t.copy_(w)
                elif s_.kind == "gamma":
                    t.fill_(1.0)
                else:
                    t.zero_()
            for n, b in self.moving.items():
                b.fill_(1.0 if n.endswith("variance") else 0.0)
            self.adam_m.zero_(); self.adam_v.zero_(); self.adam_t = 0
            self._shared["m_schedule"] = 1.0
        self.t_dirty = True

    # ---- conversion to / from the reference's own (Keras) layouts -----------------------------
    def load_keras_params(self, params):
        """params: name -> array in Keras layout (HWIO Conv2D, HWOI Conv2DTranspose, [in,out] Dense)."""
        with torch.no_grad():
            for n, s_ in self.specs.items():
                a = torch.as_tensor(params[n]).to(torch.float32)
                if tuple(a.shape) != s_.keras_shape:
                    raise ValueError(f"{n}: expected Keras shape {s_.keras_shape}, got {tuple(a.shape)}")
                t = self.p[n]
                if s_.kind in ("conv", "convT"):
                    t.copy_(a.permute(3, 0, 1, 2).to(self.device))
                elif s_.kind == "conv_padin":
                    t.zero_(); t[..., :2].copy_(a.permute(3, 0, 1, 2).to(self.device))
                elif s_.kind == "conv_padout":
                    t.zero_(); t[:2].copy_(a.permute(3, 0, 1, 2).to(self.device))
                elif s_.kind == "dense":
                    t.copy_(a.t().to(self.device))
                elif s_.kind == "bias_pad":
                    t.zero_(); t[:2].copy_(a.to(self.device))
                else:
                    t.copy_(a.to(self.device))
        self.t_dirty = True

    def _to_keras(self, views):
        out = {}
        for n, s_ in self.specs.items():
            t = views[n].detach()
            if s_.kind in ("conv", "convT"):
                a = t.permute(1, 2, 3, 0)
            elif s_.kind == "conv_padin":
                a = t[..., :2].permute(1, 2, 3, 0)
            elif s_.kind == "conv_padout":
                a = t[:2].permute(1, 2, 3, 0)
            elif s_.kind == "dense":
                a = t.t()
            elif s_.kind == "bias_pad":
                a = t[:2]
            else:
                a = t
            out[n] = a.contiguous().cpu()
        return out

    def export_keras_params(self):
        return self._to_keras(self.p)

    def export_keras_grads(self):
        return self._to_keras(self.g)

    # ------------------------------------------------------------------ buffers
    def _alloc(self):
        B, dev, ch, hw, D = self.B, self.device, self.ch, self.hw, self.depth
        A = lambda h, w, c: ops.new_act(B, h, w, c, dev, dtype=self.adt)      # trunk activations (fp32 or bf16)
        F = lambda h, w, c: ops.new_act(B, h, w, c, dev)                      # always fp32
        PAD = self.PAD
        self.x4, self.down, self.y, self.a = A(self.H, self.W, PAD), {}, {}, {}
        self.cat, self.g_cat = {}, {}
        self.g_down, self.g_y = {}, {}
        for l in range(1, self.L + 1):
            h, w = hw[l - 1]
            c = ch[l - 1]
            self.down[l], self.y[l] = A(h, w, c), A(h, w, c)
            self.g_down[l], self.g_y[l] = A(h, w, c), A(h, w, c)
            if l <= D:
                self.cat[l], self.g_cat[l] = A(h, w, 2 * c), A(h, w, 2 * c)
                self.a[l] = self.cat[l].slice(0, c)
            else:
                self.a[l] = A(h, w, c)
        cL = ch[-1]
        self.emb_out = torch.empty((B * self.n_idx, EMB_DIM), dtype=torch.float32, device=dev)
        self.g_emb_out = torch.empty_like(self.emb_out)
        self.flat = Act(self.emb_out.view(B, 1, 1, self.vec_in))
        self.g_flat = Act(self.g_emb_out.view(B, 1, 1, self.vec_in))
        self.v = ops.new_act(B, 1, 1, self.vec_dim, dev)
        self.vd = ops.new_act(B, 1, 1, self.vec_dim, dev)
        self.g_v = ops.new_act(B, 1, 1, self.vec_dim, dev)
        self.g_vd = ops.new_act(B, 1, 1, self.vec_dim, dev)
        self.vd_sp = Act(self.vd.base.view(B, self.h5, self.w5, VEC_CH))       # Reshape((h5,w5,16)) is NHWC
        self.g_vd_sp = Act(self.g_vd.base.view(B, self.h5, self.w5, VEC_CH))
        self.z, self.g_z = A(self.h5, self.w5, cL), A(self.h5, self.w5, cL)
        self.ya, self.aa, self.yb, self.ab = {}, {}, {}, {}
        self.g_ya, self.g_aa, self.g_yb, self.g_ab = {}, {}, {}, {}
        for l in range(1, D + 1):
            h, w = hw[l - 1]
            c = ch[l - 1]
            for d_ in (self.ya, self.aa, self.yb, self.ab, self.g_ya, self.g_aa, self.g_yb, self.g_ab):
                d_[l] = A(h, w, c)
        self.logits, self.g_logits = F(self.H, self.W, 4), A(self.H, self.W, PAD)      # logits stay fp32 for sigmoid + loss
        if self.dtype == "bf16":    # glue to the fp32 information-vector branch
            self.v1x1, self.g_z32 = F(self.h5, self.w5, cL), F(self.h5, self.w5, cL)
        self.pred = torch.empty((B, 2, self.H, self.W), dtype=torch.float32, device=dev)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=dev)
        self.reg_out = torch.zeros(1, dtype=torch.float32, device=dev)
        self.loss_tot = torch.zeros(1, dtype=torch.float32, device=dev)
        self.bn_affine = {b: torch.empty(2 * self.specs[b + ".gamma"].numel, dtype=torch.float32, device=dev) for b in self.bn_names}
        self.bn_saved = {b: torch.empty(2 * self.specs[b + ".gamma"].numel, dtype=torch.float32, device=dev) for b in self.bn_names}
        self.dropout_mask = None
        self.emb_idx = torch.zeros(B * self.n_idx, dtype=torch.int32, device=dev)
        # geometry descriptors
        G, k = ops.geom, self.k
        self.geo = {}
        for l in range(1, self.L + 1):
            h, w = hw[l - 1]
            c = ch[l - 1]
            if l == 1:
                self.geo["enc1.down"] = G(B, h, w, PAD, c, k, 1)
            else:
                hi, wi = hw[l - 2]
                self.geo[f"enc{l}.down"] = G(B, hi, wi, ch[l - 2], c, k, 2)
            self.geo[f"enc{l}.cb1"] = G(B, h, w, c, c, 3, 1)
        self.geo["vec.dense"] = G(B, 1, 1, self.vec_in, self.vec_dim, 1, 1)
        self.geo["vec.conv"] = G(B, self.h5, self.w5, VEC_CH, cL, 1, 1)
        for l in range(1, D + 1):
            h, w = hw[l - 1]
            c = ch[l - 1]
            hl, wl = hw[l]
            self.geo[f"dec{l}.up"] = G(B, hl, wl, ch[l], c, k, 2)
            self.geo[f"dec{l}.cb1a"] = G(B, h, w, 2 * c, c, k, 1)
            self.geo[f"dec{l}.cb1b"] = G(B, h, w, c, c, 3, 1)
        self.geo["head"] = G(B, self.H, self.W, ch[0], PAD, 6, 1)

    def _reserve_workspace(self):
        need = 1 << 16
        for n, g in self.geo.items():
            if n.endswith(".up"):
                need = max(need, ops.conv2d_transpose_wgrad_ws_bytes(g))
            else:
                need = max(need, ops.conv2d_wgrad_ws_bytes(g))
        self._wgrad_ws_max = need
        P0 = self.B * self.H * self.W
        need = max(need, ops.bn_ws_bytes(P0, max(self.ch[0], 4)), ops.bn_ws_bytes(self.B * self.h5 * self.w5, self.ch[-1]),
                   ops.bn_ws_bytes(self.B, self.vec_dim))
        for l in range(1, self.L + 1):
            h, w = self.hw[l - 1]
            need = max(need, ops.bn_ws_bytes(self.B * h * w, 2 * self.ch[l - 1]))
        if self.dense_direct:
            need = max(need, _lib.lib().unetrir_dense_dgrad_ws_bytes(self.B, self.vec_in, self.vec_dim))
        self.ws.reserve(need)

    # ------------------------------------------------------------------ helpers
    def refresh_transposed(self):
        """Conv2D kernels -> [Cin][T][Cout] for dgrad; Conv2DTranspose kernels -> [Cout][T][Cin] for forward."""
        for n in self.tspec:
            s_ = self.specs[n]
            N = s_.shape[0]
            if s_.kind == "dense":
                T, C_ = 1, s_.shape[1]
            else:
                T, C_ = s_.shape[1] * s_.shape[2], s_.shape[3]
            if self.dtype == "bf16" and n in self.ph:
                continue                                    # trunk kernels: bf16 copies below
            ops.transpose_weight(self.p[n], self.pt[n], N, T, C_)
        if self.ph:
            if self._cast_table is None:
                ent = []
                for n in self.ph:
                    s_ = self.specs[n]
                    N, T, C_ = s_.shape[0], s_.shape[1] * s_.shape[2], s_.shape[3]
                    ent.append((self.p[n], self.ph[n], self.pth[n], N, T, C_, C_, N, self.ppk.get(n)))
                self._cast_table = ops.make_cast_table(ent, self.device)
            ops.cast_weights_batched(self._cast_table)
        self.t_dirty = False

    def wf(self, name):
        """Trunk kernel in its stored orientation, in the storage type of the trunk (fp32 master or bf16 work copy)."""
        return self.ph[name] if self.dtype == "bf16" else self.p[name]

    def wb(self, name):
        """Trunk kernel with the channel roles swapped ([C][T][N]) in the storage type of the trunk."""
        return self.pth[name] if self.dtype == "bf16" else self.pt[name]

    def _bn_relu_fwd(self, name, y: Act, out: Act):
        if self.batchnorm:
            if self.training:
                ops.bn_stats(y, self.p[name + ".gamma"], self.p[name + ".beta"], self.bn_affine[name], self.bn_saved[name],
                             self.ws, self.moving[name + ".moving_mean"], self.moving[name + ".moving_variance"],
                             BN_EPS, BN_MOMENTUM)
            else:
                ops.bn_inference_affine(self.p[name + ".gamma"], self.p[name + ".beta"], self.moving[name + ".moving_mean"],
                                        self.moving[name + ".moving_variance"], BN_EPS, self.bn_affine[name])
            ops.bn_apply(y, self.bn_affine[name], out, relu=True)
        else:
            ops.relu_fwd(y, out)

    def _colstat(self, name, dgrad, x: Act, n_out):
        """(rows, buffer) of fused column statistics for conv `name` (forward or data gradient), or (0, None)."""
        key = (name, dgrad)
        if self._cst_gen != ops.config_generation():     # the switches changed: another kernel may serve the layer, with other row counts
            self._cst_rows, self._cst_gen = {}, ops.config_generation()
        if key not in self._cst_rows:
            self._cst_rows[key] = ops.conv2d_colstat_rows(self.geo[name], dgrad, x) if self.dtype == "bf16" and self.fused_stats else 0
        rows = self._cst_rows[key]
        if rows == 0:
            return 0, None
        need = rows * n_out * 2
        if self._cst_buf is None or self._cst_buf.numel() < need:
            self._cst_buf = torch.empty(need, device=self.device, dtype=torch.float32)
        return rows, self._cst_buf

    def _conv_bn_relu_fwd(self, name, x: Act, y: Act, out: Act):
        """Conv2D -> BatchNormalization -> ReLU (conv_block_1, dl_models/u_net.py:364-371); in bf16 training the batch
        statistics come from the convolution's own epilogue where the serving kernel provides them."""
        p = self.p
        rows, buf = self._colstat(name, 0, x, y.C) if (self.batchnorm and self.training) else (0, None)
        if rows == 0:
            ops.conv2d_fwd(self.geo[name], x, self.wf(name + ".kernel"), p[name + ".bias"], y)
            self._bn_relu_fwd(name, y, out)
            return
        ops.conv2d_fwd_colstat(self.geo[name], x, self.wf(name + ".kernel"), p[name + ".bias"], y, buf)
        # statistics rows -> affine / moving statistics -> BatchNorm + ReLU: one call
        ops.bn_colstat_act_add(buf, rows, y, p[name + ".gamma"], p[name + ".beta"], self.bn_affine[name], self.bn_saved[name], out, 1, None,
                               self.moving[name + ".moving_mean"], self.moving[name + ".moving_variance"], BN_EPS, BN_MOMENTUM)

    def _dgrad_colsum(self, name, dy: Act, dx: Act, bias_grad, c0, c_n):
        """Data gradient of conv `name` plus the bias gradient of the layer that produced its input (channels
        [c0, c0+c_n) of dx summed over pixels), fused into the dgrad epilogue where the serving kernel allows."""
        rows, buf = self._colstat(name, 1, dy, dx.C)
        if rows == 0:
            ops.conv2d_dgrad(self.geo[name], dy, self.wb(name + ".kernel"), dx)
            return False
        ops.conv2d_dgrad_colstat(self.geo[name], dy, self.wb(name + ".kernel"), dx, buf)
        ops.colsum_colstat(buf, rows, dx.C, c0, c_n, bias_grad)
        return True

    def _bn_relu_bwd(self, name, da: Act, y: Act, dy: Act):
        if self.batchnorm:
            ops.bn_bwd(da, y, self.p[name + ".gamma"], self.bn_affine[name], self.bn_saved[name], dy,
                       self.g[name + ".gamma"], self.g[name + ".beta"], self.ws, relu=True)
        else:
            ops.relu_bwd(da, y, dy)

    # ------------------------------------------------------------------ forward
    def forward(self, spec, emb, dropout_mask=None, target=None, global_batch=None, alpha=0.9):
        """spec f32 [B,2,H,W] NCHW, emb int [B,2,16].  With `target` also evaluates compute_loss
        (main_training.py:203-235) and seeds the backward pass.  Returns the NCHW prediction buffer."""
        B, D, p = self.B, self.depth, self.p
        if tuple(spec.shape) != (B, 2, self.H, self.W) or spec.dtype != torch.float32 or not spec.is_contiguous():
            raise ValueError(f"spec must be a contiguous float32 [{B},2,{self.H},{self.W}] tensor, got {tuple(spec.shape)} {spec.dtype}")
        if tuple(emb.shape) != (B,) + self.inf_vector_shape:
            raise ValueError(f"emb must be [{B},{self.inf_vector_shape}]")
        if spec.device != self.device:
            raise ValueError("inputs must live on the engine's device")
        if self.t_dirty or self.training:
            self.refresh_transposed()
        if emb.dtype not in (torch.int32, torch.int64):
            emb = emb.to(torch.int64)
        if emb.device != self.device:      # DataGenerator.__getitem__ hands over host arrays: a small copy, never a host pointer
            emb = emb.to(self.device)
        ops.index_to_i32(emb.contiguous(), self.emb_idx)
        self.dropout_mask = dropout_mask

        def vec_branch(ws_):
            """information vector branch (dl_models/u_net.py:253-263) up to (bf16: including) its 1x1 conv"""
            ops.embedding_fwd(self.emb_idx, p["vec.embedding"], self.emb_out)
            ops.dense_fwd(self.flat, p["vec.dense.kernel"], p["vec.dense.bias"], self.v, ws_)
            if dropout_mask is not None:
                ops.mul(self.v.base, dropout_mask, self.vd.base)
                vsp_ = self.vd_sp
            else:
                vsp_ = Act(self.v.base.view(B, self.h5, self.w5, VEC_CH))
            if self.dtype != "f32":   # the branch is fp32; its 1x1 conv output joins the bf16 trunk through the Add()
                ops.conv2d_fwd(self.geo["vec.conv"], vsp_, p["vec.conv.kernel"], p["vec.conv.bias"], self.v1x1)
            return vsp_

        vsp = None
        if self.wg_stream is not None:      # independent of the encoder until the Add(): side stream, joined below
            with self._wg() as ws_:
                vsp = vec_branch(ws_)
        ops.nchw_to_nhwc_pad(spec, self.x4)
        prev = self.x4
        for l in range(1, self.L + 1):
            ops.conv2d_fwd(self.geo[f"enc{l}.down"], prev, self.wf(f"enc{l}.down.kernel"), p[f"enc{l}.down.bias"], self.down[l],
                           w_packed=self.ppk.get(f"enc{l}.down.kernel"))
            self._conv_bn_relu_fwd(f"enc{l}.cb1", self.down[l], self.y[l], self.a[l])
            prev = self.a[l]
        if self.wg_stream is None:
            vsp = vec_branch(self.ws)
        else:
            self._join_wg()
        # Add (dl_models/u_net.py:229)
        if self.dtype == "f32":
            ops.conv2d_fwd(self.geo["vec.conv"], vsp, p["vec.conv.kernel"], p["vec.conv.bias"], self.z, addend=self.a[self.L])
        else:
            ops.add_f32_to_bf16(self.a[self.L], self.v1x1, self.z)
        cur = self.z
        for l in range(D, 0, -1):
            c = self.ch[l - 1]
            ops.conv2d_transpose_fwd(self.geo[f"dec{l}.up"], cur, self.wb(f"dec{l}.up.kernel"), p[f"dec{l}.up.bias"],
                                     self.cat[l].slice(c, c))
            self._conv_bn_relu_fwd(f"dec{l}.cb1a", self.cat[l], self.ya[l], self.aa[l])
            self._conv_bn_relu_fwd(f"dec{l}.cb1b", self.aa[l], self.yb[l], self.ab[l])
            cur = self.ab[l]
        if self.head_direct:
            ops.head6x6_fwd(cur, p["head.kernel"], p["head.bias"], self.logits)
        else:
            ops.conv2d_fwd(self.geo["head"], cur, p["head.kernel"], p["head.bias"], self.logits)
        if target is not None:
            gb = B if global_batch is None else global_batch
            inv_norm = 1.0 / (2.0 * self.H * self.W * gb)
            ops.sigmoid_loss(self.logits, target, alpha, inv_norm, self.pred, self.g_logits, self.loss_out, self.ws, **self._loss_extras(spec))
        else:
            self._last_spec = spec
            ops.sigmoid_nchw(self.logits, self.pred)
        return self.pred

    def reg_loss(self, into=None, accumulate=False):
        """sum(model.losses) / replicas (main_training.py:232-233), evaluated on device into reg_out[0] (or added to `into`[0])."""
        out = self.reg_out if into is None else into
        first = not accumulate
        for n in self.l2_names:
            s_ = self.specs[n]
            ops.sumsq(self.theta[s_.offset:s_.offset + s_.numel], L2_COEF / self.n_replicas, out, not first, self.ws)
            first = False
        return out

    def loss_from_logits(self, target, global_batch=None, alpha=0.9):
        """compute_loss for the logits of the last forward pass: rewrites the prediction (same values), the data loss and
        dL/dlogits, which seeds backward()."""
        gb = self.B if global_batch is None else global_batch
        if tuple(target.shape) != (self.B, 2, self.H, self.W) or target.dtype != torch.float32 or not target.is_contiguous():
            raise ValueError(f"target must be a contiguous float32 [{self.B},2,{self.H},{self.W}] tensor")
        ops.sigmoid_loss(self.logits, target, alpha, 1.0 / (2.0 * self.H * self.W * gb), self.pred, self.g_logits, self.loss_out, self.ws,
                         **self._loss_extras())

    def loss_total(self):
        """Data loss + l2 terms as one device scalar (a 4-byte copy and the l2 reductions accumulating onto it)."""
        self.loss_tot.copy_(self.loss_out[0:1])
        self.reg_loss(into=self.loss_tot, accumulate=True)
        return self.loss_tot

    # ------------------------------------------------------------------ backward
    def backward(self, dpred=None, on_ready=None, include_reg=True):
        """Gradients of every trainable variable into the flat gradient buffer.  Seeds from the loss kernel's
        dL/dlogits (forward(target=...)) or from an upstream dL/dpred (NCHW).  `on_ready(offset_end)` is called
        after the launches that finalise the gradient range [0, offset_end) of the flat buffer have been enqueued."""
        D, p, pt, g, ws = self.depth, self.p, self.pt, self.g, self.ws
        # d/dw of (l2(0.001) * sum w^2) / replicas, folded into the split-K reduction of the weight gradient
        reg = 2.0 * L2_COEF / self.n_replicas if include_reg else 0.0
        if dpred is not None:
            ops.sigmoid_bwd(self.pred, dpred, self.g_logits)

        def ready(name):
            if on_ready is None:
                return
            s_ = self.specs[name]
            off = s_.offset + (-(-s_.numel // ALIGN) * ALIGN)
            if self.wg_stream is None:
                self._hand_over(on_ready, off)
                return
            # A bucket's gradients come from both streams (weight gradients: side stream; BatchNorm / fused bias
            # gradients: main stream).  The SIDE stream hands the bucket over once it has waited for the main stream's
            # progress: the all-reduce orders after both and the main stream never blocks on the side stream.  That wait is
            # the one the next `with self._wg()` performs anyway, so the hand-over is parked until then instead of putting an
            # event record of its own into the main stream (each costs ~8 us of dispatch gap; later is always safe).
            if self._defer_ready:
                self._pending_ready.append((on_ready, off))
                return
            self.rt.wait(self.wg_stream, self.rt.record())
            with self.rt.on(self.wg_stream):
                self._hand_over(on_ready, off)

        gl = self.g_logits
        if self.head_direct:
            with self._wg() as ws_:
                ops.head6x6_wgrad(self.ab[1] if D >= 1 else self.a[1], gl, g["head.kernel"], ws_)   # rows 2,3 of the padded kernel stay 0
        else:
            with self._wg() as ws_:
                ops.conv2d_wgrad(self.geo["head"], self.ab[1] if D >= 1 else self.a[1], gl, g["head.kernel"], ws_)
        with self._wg() as ws_:
            ops.colsum(gl, g["head.bias"], ws_)
        top = self.ab[1] if D >= 1 else self.a[1]
        g_cur = self.g_ab[1] if D >= 1 else self.g_z
        if self.head_direct and self.dtype == "bf16" and ops.head6x6_dgrad_supported(self.W, self.ch[0]):
            ops.head6x6_dgrad(gl, self.p["head.kernel"], g_cur)         # reads the fp32 master kernel: before ready()
        else:
            ops.conv2d_dgrad(self.geo["head"], gl, self.wb("head.kernel"), g_cur)
        ready("head.bias")      # a group is handed over (all-reduce / optimizer) only after the last reader of its parameters
        for l in range(1, D + 1):
            c = self.ch[l - 1]
            # cb1b
            self._bn_relu_bwd(f"dec{l}.cb1b", self.g_ab[l], self.yb[l], self.g_yb[l])
            with self._wg() as ws_:
                ops.conv2d_wgrad(self.geo[f"dec{l}.cb1b"], self.aa[l], self.g_yb[l], g[f"dec{l}.cb1b.kernel"], ws_, defer=self._rb if self.park_reduces else None)
            if not self.batchnorm:      # a bias in front of BatchNorm has an identically zero gradient (dy sums to 0 per channel)
                with self._wg() as ws_:
                    ops.colsum(self.g_yb[l], g[f"dec{l}.cb1b.bias"], ws_)
            ops.conv2d_dgrad(self.geo[f"dec{l}.cb1b"], self.g_yb[l], self.wb(f"dec{l}.cb1b.kernel"), self.g_aa[l])
            # cb1a
            self._bn_relu_bwd(f"dec{l}.cb1a", self.g_aa[l], self.ya[l], self.g_ya[l])
            with self._wg() as ws_:
                ops.conv2d_wgrad(self.geo[f"dec{l}.cb1a"], self.cat[l], self.g_ya[l], g[f"dec{l}.cb1a.kernel"], ws_, defer=self._rb if self.park_reduces else None)
            if not self.batchnorm:      # a bias in front of BatchNorm has an identically zero gradient (dy sums to 0 per channel)
                with self._wg() as ws_:
                    ops.colsum(self.g_ya[l], g[f"dec{l}.cb1a.bias"], ws_)
            up_bias_done = self._dgrad_colsum(f"dec{l}.cb1a", self.g_ya[l], self.g_cat[l], g[f"dec{l}.up.bias"], c, c)
            # Conv2DTranspose
            g_up = self.g_cat[l].slice(c, c)
            x_in = self.ab[l + 1] if l < D else self.z
            g_in = self.g_ab[l + 1] if l < D else self.g_z
            with self._wg() as ws_:
                ops.conv2d_transpose_wgrad(self.geo[f"dec{l}.up"], x_in, g_up, g[f"dec{l}.up.kernel"], ws_, reg=reg,
                                           w=p[f"dec{l}.up.kernel"], defer=self._rb if self.park_reduces else None)
            if not up_bias_done:
                with self._wg() as ws_:
                    ops.colsum(g_up, g[f"dec{l}.up.bias"], ws_)
            ops.conv2d_transpose_dgrad(self.geo[f"dec{l}.up"], g_up, self.wf(f"dec{l}.up.kernel"), g_in,
                                       w_packed=self.ppk.get(f"dec{l}.up.kernel"))
            ready(f"dec{l}.up.bias")            # after the last reader of this group's parameters (fp32: wf() is the master kernel)
        # bottleneck: z = a_L + conv1x1(dropout(dense(embedding))).  Nothing downstream of the information-vector branch feeds the
        # encoder's backward chain (that needs only g_z), so the whole branch runs on the weight-gradient stream when there is one.
        B = self.B
        has_do = self.dropout_mask is not None
        vsp = self.vd_sp if has_do else Act(self.v.base.view(B, self.h5, self.w5, VEC_CH))
        with self._wg() as ws_:
            gz = self.g_z
            if self.dtype == "bf16":      # the information-vector branch is fp32: give it an fp32 copy of dL/dz
                ops.cast_bf16_to_f32(self.g_z, self.g_z32)
                gz = self.g_z32
            ops.conv2d_wgrad(self.geo["vec.conv"], vsp, gz, g["vec.conv.kernel"], ws_)
            ops.colsum(gz, g["vec.conv.bias"], ws_)
            ops.conv2d_dgrad(self.geo["vec.conv"], gz, pt["vec.conv.kernel"], self.g_vd_sp)
            if has_do:
                ops.mul(self.g_vd.base, self.dropout_mask, self.g_v.base)
                gv = self.g_v
            else:
                gv = self.g_vd
            ops.conv2d_wgrad(self.geo["vec.dense"], self.flat, gv, g["vec.dense.kernel"], ws_)
            ops.colsum(gv, g["vec.dense.bias"], ws_)
            if self.dense_direct:
                ops.dense_dgrad(gv, p["vec.dense.kernel"], self.g_flat, ws_)           # dL/dflat = dv . W from the kernel as stored
            else:
                ops.dense_fwd(gv, pt["vec.dense.kernel"], None, self.g_flat, ws_)      # the same through the [in][out] copy
            ops.embedding_bwd(self.emb_idx, self.g_emb_out, g["vec.embedding"])
        ready("vec.embedding")
        # encoder, deepest level first; the gradient of a_l is (skip half of g_cat_l) + dgrad of the next strided conv
        g_a = self.g_z
        for l in range(self.L, 0, -1):
            self._bn_relu_bwd(f"enc{l}.cb1", g_a, self.y[l], self.g_y[l])
            with self._wg() as ws_:
                ops.conv2d_wgrad(self.geo[f"enc{l}.cb1"], self.down[l], self.g_y[l], g[f"enc{l}.cb1.kernel"], ws_, defer=self._rb if self.park_reduces else None)
            if not self.batchnorm:
                with self._wg() as ws_:
                    ops.colsum(self.g_y[l], g[f"enc{l}.cb1.bias"], ws_)
            down_bias_done = self._dgrad_colsum(f"enc{l}.cb1", self.g_y[l], self.g_down[l], g[f"enc{l}.down.bias"], 0, self.g_down[l].C)
            x_in = self.a[l - 1] if l > 1 else self.x4
            with self._wg() as ws_:
                ops.conv2d_wgrad(self.geo[f"enc{l}.down"], x_in, self.g_down[l], g[f"enc{l}.down.kernel"], ws_, reg=reg,
                                 w=p[f"enc{l}.down.kernel"], defer=self._rb if self.park_reduces else None)
            if not down_bias_done:
                with self._wg() as ws_:
                    ops.colsum(self.g_down[l], g[f"enc{l}.down.bias"], ws_)
            ready(f"enc{l}.down.bias")
            if l > 1:
                skip = self.g_cat[l - 1].slice(0, self.ch[l - 2])
                ops.conv2d_dgrad(self.geo[f"enc{l}.down"], self.g_down[l], self.wb(f"enc{l}.down.kernel"), skip, addend=skip)
                g_a = skip
        if self._pending_ready or (self._rb is not None and len(self._rb)):
            with self._wg():    # hands over what is still parked (the bucketer runs the parked reductions first) ...
                self.flush_reduces()        # ... and without a bucketer, or behind its last boundary: the reductions still parked
        self._join_wg()     # the optimizer (and the next forward, which overwrites activations) must see every weight gradient

    def _wg(self):
        return _SideStream(self)

    def _hand_over(self, fn, off):
        """on_ready(off): the gradient range [0, off) is final.  A consumer that does not run the parked split-K reductions itself
        (the trainer's bucketer does, lazily, at its bucket boundaries: GradBucketer.before_bucket) gets them run first."""
        if getattr(getattr(fn, "__self__", None), "before_bucket", None) is None:
            self.flush_reduces()
        fn(off)

    def flush_reduces(self):
        """Run the parked split-K reductions (on the current stream: the one the weight gradients ran on).  Called by the trainer's
        bucketer before a bucket's gradients are first read, and at the end of backward()."""
        if self._rb is not None:
            self._rb.flush()

    def _flush_ready(self):
        """Called on the side stream right after it has waited for the main stream: hand over the parked buckets."""
        pend, self._pending_ready = self._pending_ready, []
        for fn, off in pend:
            self._hand_over(fn, off)

    def _join_wg(self):
        if self.wg_stream is not None:
            self.rt.wait(self.rt.current_stream(), self.rt.record(self.wg_stream))

    # ------------------------------------------------------------------ optimizer: DeviceCounters.adam_step / adam_begin / adam_range
    def make_dropout_mask(self, generator=None):
        """Keep mask of Dropout(.3) scaled by 1/(1-p), [B, vec_dim].  Default: the HIP generator kernel, draw number
        `dropout_step` of stream `dropout_seed` (reproducible; the trainer offsets the seed by the replica rank), written
        into one reused buffer.  With a torch generator: torch's own stream of random numbers (tests)."""
        if generator is not None:
            keep = (torch.rand((self.B, self.vec_dim), device=self.device, generator=generator) >= DROPOUT_P)
            return keep.to(torch.float32) / (1.0 - DROPOUT_P)
        if self._mask_buf is None:
            self._mask_buf = torch.empty((self.B, self.vec_dim), dtype=torch.float32, device=self.device)
        return self._draw_mask(self._mask_buf)

    def n_params(self):
        """Trainable parameter count in the reference's sense (padding excluded)."""
        return sum(int(math.prod(s_.keras_shape)) for s_ in self.specs.values())
