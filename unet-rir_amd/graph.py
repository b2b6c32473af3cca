"""A small static-graph executor over the HIP kernels: the graph is built once as a list of (forward, backward)
closures over preallocated NHWC buffers; forward runs the list, backward runs it in reverse.

Two storage modes, as engine.UNetEngine: "f32" (everything fp32) and "bf16" (trunk activations, their gradients and the
trunk kernels' work copies bf16; accumulation, BatchNorm statistics, biases, weight gradients, master weights, Adam and the
Dense / Embedding branches fp32).  A convolution runs in the storage type of its INPUT node; `_cast` nodes join the two.

Used for the operator graphs that are not hand-scheduled in engine.py: ResAE (dl_models/res_ae.py) and the U-Net feature
block modes 1-3 (dl_models/u_net.py:324-386).  A tensor with several consumers gets its gradient from several writers:
the first one writes, the others accumulate in place (conv data gradients through the kernels' `addend` epilogue).
Parameters, gradients and Adam moments live in flat buffers ordered by backward completion, as in engine.py.
"""
import math
from collections import OrderedDict

import torch

from . import ops
from .device import HipRuntime
from .ops import Act
from .engine import ALIGN, BN_EPS, BN_MOMENTUM, L2_COEF, DROPOUT_P, DeviceCounters, ParamSpec, _SideStream

RELU, LEAKY = 1, 2       # activation codes of the C ABI (LeakyReLU: keras default alpha 0.3)


class Node:
    """An activation buffer (or a channel slice of one) and the buffer of its gradient."""
    __slots__ = ("a", "g", "g_set", "needs_grad", "children", "cst")

    def __init__(self, a: Act, needs_grad=True, g: Act = None):
        self.a = a
        self.g = g if g is not None else (Act(torch.empty_like(a.base)) if needs_grad else None)
        self.g_set = False
        self.needs_grad = needs_grad
        self.children = []           # channel-slice views: written whenever this node's gradient is written
        self.cst = None              # (rows, buffer): fused column statistics the producing convolution wrote (BatchNorm statistics)


class GraphEngine(DeviceCounters):
    mask_on_side_stream = False      # the dropout masks are consumed by nodes of the main stream (trainer.Trainer._make_mask)

    def __init__(self, B, device="cuda:0", n_replicas=1, runtime=None, share=None, dtype="f32", overlap_wgrad=False):
        """dtype: storage type of the trunk ("f32" or "bf16").
        overlap_wgrad: the side-stream schedule of engine.UNetEngine for the backward pass - weight and bias gradients of every
        convolution / Dense / head / Embedding node on a second stream (they are leaves: nothing in the backward pass reads them),
        gradient buckets handed over from that stream, the optimizer bucket by bucket on a third (trainer.Trainer).  These
        graphs are chains of small launches (ResAE cfg 5: ~770 per step, 12 us on average), so the two chains run side by side.
        share: another engine of the same class and configuration whose parameters, gradients, Adam moments, work copies
        and BatchNorm moving statistics this one aliases (only the activation buffers depend on the batch size)."""
        self.rt = runtime if runtime is not None else HipRuntime(device)
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        self.dtype = dtype
        self.adt = torch.float32 if dtype == "f32" else torch.bfloat16
        self.fused_junction = True                     # BatchNorm -> Add -> activation backward in one reduce / finalize / apply sequence
        self.fused_stats = True                        # BatchNorm statistics from the convolution epilogues where the kernel has them (bf16)
        self.PAD = 4 if dtype == "f32" else 8          # channel granule = 16 bytes (zero-padded 2-channel ends)
        self._h_kernels = []                           # kernels of the convolutions that run in bf16 (work copies needed)
        self._s2_kernels = []                          # 3x3 stride-2 kernels among them
        self._ppk = {}
        self._share = share
        self._shared = share._shared if share is not None else {"adam_t": 0, "t_dirty": True, "dropout_step": 0}
        self.include_reg = True       # backward adds d/dw of the l2 terms (False: the caller differentiates them itself)
        self.B = B
        self.device = torch.device(device)
        self.n_replicas = n_replicas
        self.training = True
        self.nodes, self.specs_fwd, self.ops = [], [], []
        self._op_params, self._n_specs_seen = [], 0
        self.dropout_seed, self._mask_bufs = (share.dropout_seed if share is not None else torch.initial_seed() & 0xFFFFFFFF), {}
        self.bn_names, self.l2_names = [], []
        self.moving = {}
        self.masks = {}              # dropout keep masks by name (None = no dropout)
        self.ws = ops.Workspace(self.device, 1 << 20)
        if share is not None and overlap_wgrad and share.wg_stream is not None:
            self.wg_stream, self.opt_stream = share.wg_stream, share.opt_stream
        else:
            self.wg_stream, self.opt_stream = self.rt.concurrent_streams(2) if overlap_wgrad else (None, None)
        self.ws_w = ops.Workspace(self.device, 1 << 20) if self.wg_stream is not None else self.ws     # the side stream's own
        self._pending_ready = []
        # the split-K reductions of the weight gradients are parked and run together (ops.ReduceBatch): one launch per bucket
        # hand-over / full arena instead of one per convolution (configs[4]: 57 per step).  Every slab set is parked here: measured on
        # configs[4] (scripts/sweep_park.py, engines alternating in one process) 6.78 ms per step with none parked, 6.69-6.79 with only
        # the sets <= 4 / 16 / 64 MB parked, 6.47 with all of them (single stream: 6.82 / 6.61-6.76 / 6.49)
        self._rb = ops.ReduceBatch(self.device, 512 << 20) if ops.wgrad_defer_supported(dtype) else None
        self.park_reduces = self._rb is not None
        self._p, self._g, self._pt = {}, {}, {}

    @property
    def adam_t(self):
        return self._shared["adam_t"]

    @adam_t.setter
    def adam_t(self, v):
        self._shared["adam_t"] = v

    @property
    def t_dirty(self):
        """The transposed kernel copies are older than the parameters."""
        return self._shared["t_dirty"]

    @t_dirty.setter
    def t_dirty(self, v):
        self._shared["t_dirty"] = v

    # ------------------------------------------------------------------ construction helpers
    def _param(self, name, shape, kind, keras_shape, l2=False):
        self.specs_fwd.append(ParamSpec(name, shape, kind, keras_shape))
        if l2:
            self.l2_names.append(name)

    def _reg(self, node):
        self.nodes.append(node)
        return node

    def _push(self, fwd, bwd):
        """Append one op; the parameters declared since the previous op are the ones whose gradients its `bwd` finalises."""
        self.ops.append((fwd, bwd))
        self._op_params.append([s_.name for s_ in self.specs_fwd[self._n_specs_seen:]])
        self._n_specs_seen = len(self.specs_fwd)

    # ---- side stream (as engine.UNetEngine: _SideStream waits for the main stream, switches, hands parked buckets over)
    def _wg(self):
        return _SideStream(self)

    def _flush_ready(self):
        pend, self._pending_ready = self._pending_ready, []
        for fn, off in pend:
            self._hand_over(fn, off)

    def _hand_over(self, fn, off):
        """on_ready(off) for a consumer that may not run the parked reductions itself (engine.UNetEngine._hand_over)."""
        if getattr(getattr(fn, "__self__", None), "before_bucket", None) is None:
            self.flush_reduces()
        fn(off)

    def flush_reduces(self):
        """Run the parked split-K reductions on the current stream (the one the weight gradients ran on): the trainer's bucketer calls
        this before a bucket's gradients are first read; backward() at its end."""
        if self._rb is not None:
            self._rb.flush()

    def _join_wg(self):
        if self.wg_stream is not None:
            self.rt.wait(self.rt.current_stream(), self.rt.record(self.wg_stream))

    def _new(self, h, w, c, needs_grad=True, f32=False):
        """A trunk activation (storage type of the engine) or, f32=True, a node of the fp32 branches."""
        return self._reg(Node(ops.new_act(self.B, h, w, c, self.device, dtype=torch.float32 if f32 else self.adt), needs_grad))

    def _cast(self, x: Node):
        """The same tensor in the other storage type (fp32 <-> bf16 seam between a Dense branch and the trunk)."""
        to16 = x.a.sfx == "f32"
        y = self._reg(Node(ops.new_act(self.B, x.a.H, x.a.W, x.a.C, self.device, dtype=torch.bfloat16 if to16 else torch.float32)))

        def fwd():
            (ops.cast_f32_to_bf16 if to16 else ops.cast_bf16_to_f32)(x.a, y.a)

        def bwd():
            (ops.cast_bf16_to_f32 if to16 else ops.cast_f32_to_bf16)(y.g, x.g)
            x.g_set = True
        self._push(fwd, bwd)
        return y

    def _view(self, parent: Node, c0, c):
        """Channel slice [c0, c0+c) of a buffer (one half of a skip concat)."""
        v = self._reg(Node(parent.a.slice(c0, c), True, parent.g.slice(c0, c)))
        parent.children.append(v)
        return v

    def _emit(self, node, writer):
        """writer(dst, addend): dst = value (+ addend).  First writer of a gradient writes, later ones accumulate."""
        if not node.needs_grad:
            return
        writer(node.g, node.g if node.g_set else None)
        node.g_set = True
        for ch in node.children:
            ch.g_set = True

    def _conv(self, x: Node, name, cout, k, stride, transpose=False, followed_by_bn=True, pad_in=0, pad_out=0, l2=True,
              dense=False, out: Node = None):
        """Conv2D / Conv2DTranspose (TF 'same'; 1x1 'valid' is the same geometry) + bias."""
        B, cin = self.B, x.a.C
        if transpose:
            H, W = x.a.H * stride, x.a.W * stride
        else:
            H, W = -(-x.a.H // stride), -(-x.a.W // stride)
        co = pad_out if pad_out else cout
        h16 = x.a.sfx == "bf16"           # the convolution runs in the storage type of its input
        if dense and h16:
            raise ValueError("Dense layers run in fp32: cast the input first")
        y = out if out is not None else self._new(H, W, co, f32=not h16)
        kname, bname = name + ".kernel", name + ".bias"
        if h16:
            self._h_kernels.append(kname)
        real_in = 2 if pad_in else cin
        if transpose:      # primary layout [Cin][k][k][Cout]; keras (k,k,Cout,Cin)
            self._param(kname, (cin, k, k, co), "convT_padout" if pad_out else "convT", (k, k, cout, real_in), l2)
        else:              # [Cout][k][k][Cin]; keras (k,k,Cin,Cout) or Dense [in,out]
            kind = "conv_padin" if pad_in else ("conv_padout" if pad_out else "conv")
            self._param(kname, (co, k, k, cin), kind, (cin, cout) if dense else (k, k, real_in, cout), l2)
        self._param(bname, (co,), "bias_pad" if pad_out else "bias", (cout,))
        g = ops.geom(B, x.a.H, x.a.W, cin, co, k, stride)
        reg = lambda: (2.0 * L2_COEF / self.n_replicas) if (l2 and self.include_reg) else 0.0

        if h16 and k == 3 and stride == 2:
            self._s2_kernels.append(kname)             # gets a packed copy for the stride-2 forward kernel (csrc/conv3x3d.hip)
        wpk = lambda: self._ppk.get(kname)
        wf = (lambda: self._ph[kname]) if h16 else (lambda: self._p[kname])       # kernel as stored ([N][T][C])
        wb = (lambda: self._pth[kname]) if h16 else (lambda: self._pt[kname])     # channel roles swapped ([C][T][N])
        # BatchNormalization statistics from the convolution's own epilogue (bf16 trunk): the tensor is not read again for them
        want_cst = bool(h16 and followed_by_bn and self.fused_stats and not dense and out is None)
        cs = {"gen": None, "rows": 0, "buf": None}

        def colstat():
            """(rows, buffer) of this layer under the kernel-selection switches in effect (the row count depends on which kernel
            serves the layer: sized again when ops.set_config has changed a switch since)."""
            if cs["gen"] != ops.config_generation():
                rows = 0
                if want_cst:
                    rows = ops.conv2d_transpose_colstat_rows(g, x.a) if transpose else ops.conv2d_colstat_rows(g, 0, x.a)
                if rows != cs["rows"] or (rows and cs["buf"] is None):
                    cs["buf"] = torch.empty((rows, co, 2), dtype=torch.float32, device=self.device) if rows else None
                cs["rows"], cs["gen"] = rows, ops.config_generation()
            return cs["rows"], cs["buf"]

        colstat()

        def dense_dgrad(dst, add):
            if add is None:      # split-K path: the weight matrix streams from every CU
                ops.dense_fwd(y.g, self._pt[kname], None, dst, self.ws)
            else:
                ops.conv2d_dgrad(g, y.g, self._pt[kname], dst, addend=add)

        def fwd():
            y.cst = None
            rows, cst = colstat()
            if dense:
                ops.dense_fwd(x.a, self._p[kname], self._p[bname], y.a, self.ws)
            elif cst is not None and self.training:
                if transpose:
                    ops.conv2d_transpose_fwd_colstat(g, x.a, wb(), self._p[bname], y.a, cst)
                else:
                    ops.conv2d_fwd_colstat(g, x.a, wf(), self._p[bname], y.a, cst)
                y.cst = (rows, cst)
            elif transpose:
                ops.conv2d_transpose_fwd(g, x.a, wb(), self._p[bname], y.a)
            else:
                ops.conv2d_fwd(g, x.a, wf(), self._p[bname], y.a, w_packed=wpk())

        def bwd():
            with self._wg() as ws_:       # leaves of the backward pass: side stream when there is one
                if transpose:
                    ops.conv2d_transpose_wgrad(g, x.a, y.g, self._g[kname], ws_, reg=reg(), w=self._p[kname], defer=self._rb if self.park_reduces else None)
                else:
                    ops.conv2d_wgrad(g, x.a, y.g, self._g[kname], ws_, reg=reg(), w=self._p[kname], defer=self._rb if self.park_reduces else None)
                if not followed_by_bn:        # a bias in front of BatchNorm has an identically zero gradient
                    ops.colsum(y.g, self._g[bname], ws_)
            if dense:
                self._emit(x, dense_dgrad)
            elif transpose:
                self._emit(x, lambda dst, add: ops.conv2d_transpose_dgrad(g, y.g, wf(), dst, addend=add, w_packed=wpk()))
            else:
                self._emit(x, lambda dst, add: ops.conv2d_dgrad(g, y.g, wb(), dst, addend=add))
        self._push(fwd, bwd)
        return y

    def _head6x6(self, x: Node, name):
        """Conv2D(2, (6,6), 'same') in front of the sigmoid (dl_models/u_net.py:248).  fp32 trunk: an ordinary conv node with the
        2 output channels padded to 4.  bf16 trunk: the head kernels of engine.UNetEngine - bf16 activations in, fp32 logits
        [.., 4] out, bf16 dL/dlogits [.., 8] back."""
        if x.a.sfx == "f32":
            return self._conv(x, name, 2, 6, 1, followed_by_bn=False, pad_out=4, l2=False)
        B, H, W, c, PAD = self.B, x.a.H, x.a.W, x.a.C, self.PAD
        if not ops.head6x6_supported(c):
            raise ValueError("the bf16 head needs number_filters_0 % 8 == 0")
        kname, bname = name + ".kernel", name + ".bias"
        self._param(kname, (PAD, 6, 6, c), "conv_padout", (6, 6, c, 2))
        self._param(bname, (PAD,), "bias_pad", (2,))
        self._h_kernels.append(kname)
        y = self._reg(Node(ops.new_act(B, H, W, 4, self.device), True, ops.new_act(B, H, W, PAD, self.device, dtype=self.adt)))
        g = ops.geom(B, H, W, c, PAD, 6, 1)
        self.ws.reserve(512 * 2 * 36 * c * 4)
        self.ws_w.reserve(512 * 2 * 36 * c * 4)

        def fwd():
            ops.head6x6_fwd(x.a, self._p[kname], self._p[bname], y.a)

        def dgrad(dst, add):
            if add is None and ops.head6x6_dgrad_supported(W, c):
                ops.head6x6_dgrad(y.g, self._p[kname], dst)
            else:
                ops.conv2d_dgrad(g, y.g, self._pth[kname], dst, addend=add)

        def bwd():
            with self._wg() as ws_:
                ops.head6x6_wgrad(x.a, y.g, self._g[kname], ws_)          # rows 2.. of the padded kernel gradient stay 0
                ops.colsum(y.g, self._g[bname], ws_)
            self._emit(x, dgrad)
        self._push(fwd, bwd)
        return y

    def _bn_act(self, x: Node, name, act, addend: Node = None, out: Node = None, batchnorm=True):
        """BatchNormalization (+ Add) (+ activation).  x must have this op as its only consumer."""
        c = x.a.C
        y = out if out is not None else self._new(x.a.H, x.a.W, c, f32=x.a.sfx == "f32")
        if batchnorm:
            self._param(name + ".gamma", (c,), "gamma", (c,))
            self._param(name + ".beta", (c,), "beta", (c,))
            self.bn_names.append(name)
            aff = torch.empty(2 * c, dtype=torch.float32, device=self.device)
            saved = torch.empty(2 * c, dtype=torch.float32, device=self.device)
            if self._share is not None:
                mm, mv = self._share.moving[name + ".moving_mean"], self._share.moving[name + ".moving_variance"]
            else:
                mm = torch.zeros(c, dtype=torch.float32, device=self.device)
                mv = torch.ones(c, dtype=torch.float32, device=self.device)
            self.moving[name + ".moving_mean"], self.moving[name + ".moving_variance"] = mm, mv
        gj = Act(torch.empty_like(x.a.base)) if (addend is not None and not (batchnorm and self.fused_junction)) else None

        def fwd():
            if batchnorm and self.training and x.cst is not None:      # statistics rows -> affine -> apply (+ Add, activation): one call
                ops.bn_colstat_act_add(x.cst[1], x.cst[0], x.a, self._p[name + ".gamma"], self._p[name + ".beta"], aff, saved, y.a, act,
                                       addend.a if addend is not None else None, mm, mv, BN_EPS, BN_MOMENTUM)
                return
            elif batchnorm and self.training:
                ops.bn_stats(x.a, self._p[name + ".gamma"], self._p[name + ".beta"], aff, saved, self.ws, mm, mv, BN_EPS, BN_MOMENTUM)
            elif batchnorm:          # training=False: normalise with the moving statistics
                ops.bn_inference_affine(self._p[name + ".gamma"], self._p[name + ".beta"], mm, mv, BN_EPS, aff)
            ops.bn_act_add(x.a, aff if batchnorm else None, y.a, act, addend.a if addend is not None else None)

        def bwd():
            if addend is None and batchnorm:
                ops.bn_bwd(y.g, x.a, None, aff, saved, x.g, self._g[name + ".gamma"], self._g[name + ".beta"], self.ws, relu=act)
            elif addend is None:
                ops.act_bwd(y.g, y.a, x.g, act)
            elif batchnorm and self.fused_junction:
                # junction y = act(bn(x) + addend): g = dy * act'(y) feeds both branches; one reduce / finalize / apply sequence
                # writes dx, dgamma, dbeta AND the addend's gradient (written, or accumulated in place behind an earlier writer)
                def junction(dst, add):
                    ops.bn_bwd_junction(y.g, x.a, y.a, aff, saved, x.g, self._g[name + ".gamma"], self._g[name + ".beta"], self.ws, act,
                                        gskip=dst, gskip_add=add)
                if addend.needs_grad:
                    self._emit(addend, junction)
                else:
                    junction(None, None)
            else:      # the same in separate passes
                if act:
                    ops.act_bwd(y.g, y.a, gj, act)
                    gsrc = gj
                else:
                    gsrc = y.g
                if batchnorm:
                    ops.bn_bwd(gsrc, x.a, None, aff, saved, x.g, self._g[name + ".gamma"], self._g[name + ".beta"], self.ws, relu=0)
                else:
                    ops.bn_act_add(gsrc, None, x.g, 0, None)
                self._emit(addend, lambda dst, add: ops.bn_act_add(gsrc, None, dst, 0, add))
            x.g_set = True
        self._push(fwd, bwd)
        return y

    def _add(self, x: Node, y: Node, out: Node = None):
        """Add()([x, y]) without activation (dl_models/u_net.py:229, :337, :359)."""
        z = out if out is not None else self._new(x.a.H, x.a.W, x.a.C, f32=x.a.sfx == "f32")
        mixed = x.a.sfx != y.a.sfx                # bf16 trunk + fp32 information-vector branch (dl_models/u_net.py:229)
        if mixed and not (x.a.sfx == "bf16" and y.a.sfx == "f32"):
            raise ValueError("mixed Add: the first operand is the bf16 trunk, the second the fp32 branch")

        def fwd():
            if mixed:
                ops.add_f32_to_bf16(x.a, y.a, z.a)
            else:
                ops.bn_act_add(x.a, None, z.a, 0, y.a)

        def y_grad(dst, add):
            if not mixed:
                ops.bn_act_add(z.g, None, dst, 0, add)
            elif add is None:
                ops.cast_bf16_to_f32(z.g, dst)
            else:
                raise NotImplementedError("the fp32 operand of a mixed Add has one consumer")

        def bwd():
            self._emit(x, lambda dst, add: ops.bn_act_add(z.g, None, dst, 0, add))
            self._emit(y, y_grad)
        self._push(fwd, bwd)
        return z

    def _dense(self, x: Node, name, n_out):
        return self._conv(x, name, n_out, 1, 1, False, followed_by_bn=False, l2=False, dense=True)

    def _dropout(self, x: Node, which):
        """Dropout(.3) with an externally supplied keep mask self.masks[which] (already scaled by 1/(1-p))."""
        y = self._new(x.a.H, x.a.W, x.a.C, f32=x.a.sfx == "f32")
        if x.a.sfx != "f32":
            raise ValueError("Dropout sits on the fp32 Dense branches")
        self.masks.setdefault(which, None)

        def fwd():
            m = self.masks[which]
            if m is None:
                y.a.base.copy_(x.a.base)
            else:
                ops.mul(x.a.base, m, y.a.base)

        def bwd():
            m = self.masks[which]
            if m is None:
                x.g.base.copy_(y.g.base)
            else:
                ops.mul(y.g.base, m, x.g.base)
            x.g_set = True
        self._push(fwd, bwd)
        return y

    def _embedding(self, n_idx, name="embedding", vocab=2000, dim=256):
        """Embedding(2000, 256) -> Flatten; returns a [B,1,1,n_idx*dim] node."""
        B, dev = self.B, self.device
        self._param(name, (vocab, dim), "embedding", (vocab, dim))
        self.emb_idx = torch.zeros(B * n_idx, dtype=torch.int32, device=dev)
        self.n_idx = n_idx
        emb_out = torch.empty((B * n_idx, dim), dtype=torch.float32, device=dev)
        g_emb_out = torch.empty_like(emb_out)
        node = self._reg(Node(Act(emb_out.view(B, 1, 1, n_idx * dim)), True, Act(g_emb_out.view(B, 1, 1, n_idx * dim))))

        def fwd():
            ops.embedding_fwd(self.emb_idx, self._p[name], emb_out)

        def bwd():
            with self._wg():
                ops.embedding_bwd(self.emb_idx, g_emb_out, self._g[name])
        self._push(fwd, bwd)
        return node

    def set_indices(self, emb):
        if emb.dtype not in (torch.int32, torch.int64):
            emb = emb.to(torch.int64)
        if emb.device != self.device:      # host arrays from a DataGenerator: a small copy, never a host pointer to the kernel
            emb = emb.to(self.device)
        ops.index_to_i32(emb.contiguous(), self.emb_idx)

    def _reshape(self, x: Node, h, w, c):
        """Reshape of a [B,1,1,h*w*c] node to NHWC [B,h,w,c] (Keras Reshape is NHWC): a view, gradients alias."""
        v = self._reg(Node(Act(x.a.base.view(self.B, h, w, c)), True, Act(x.g.base.view(self.B, h, w, c))))

        def bwd():
            x.g_set = True
        self._push(lambda: None, bwd)
        return v

    # ------------------------------------------------------------------ parameters
    def _finalize_params(self):
        specs = list(reversed(self.specs_fwd))          # backward completion order
        off = 0
        for s_ in specs:
            s_.offset = off
            off += -(-s_.numel // ALIGN) * ALIGN
        self.specs = OrderedDict((s_.name, s_) for s_ in specs)
        dev = self.device
        sh = self._share
        if sh is not None:
            if [(n, s_.shape) for n, s_ in sh.specs.items()] != [(n, s_.shape) for n, s_ in self.specs.items()]:
                raise ValueError("share= needs an engine of the same configuration (only the batch size may differ)")
            self.theta, self.grad, self.adam_m, self.adam_v = sh.theta, sh.grad, sh.adam_m, sh.adam_v
        else:
            self.theta = torch.zeros(off, dtype=torch.float32, device=dev)
            self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
            self.adam_m = torch.zeros(off, dtype=torch.float32, device=dev)
            self.adam_v = torch.zeros(off, dtype=torch.float32, device=dev)
        for n, s_ in self.specs.items():
            self._p[n] = self.theta[s_.offset:s_.offset + s_.numel].view(s_.shape)
            self._g[n] = self.grad[s_.offset:s_.offset + s_.numel].view(s_.shape)
        self.p, self.g = self._p, self._g
        self._tnames = [n for n, s_ in self.specs.items() if s_.kind.startswith("conv")]
        toff = sum(-(-self.specs[n].numel // ALIGN) * ALIGN for n in self._tnames)
        self.theta_t = sh.theta_t if sh is not None else torch.zeros(max(toff, 4), dtype=torch.float32, device=dev)
        o = 0
        for n in self._tnames:
            k_ = self.specs[n].numel
            self._pt[n] = self.theta_t[o:o + k_]
            o += -(-k_ // ALIGN) * ALIGN
        # bf16 work copies (both orientations) of the kernels whose convolution runs in bf16
        self._ph, self._pth, self._cast_table = {}, {}, None
        if self._h_kernels:
            hoff = sum(-(-self.specs[n].numel // ALIGN) * ALIGN for n in self._h_kernels)
            self.theta_h = sh.theta_h if sh is not None else torch.zeros(max(hoff, 8), dtype=torch.bfloat16, device=dev)
            self.theta_th = sh.theta_th if sh is not None else torch.zeros(max(hoff, 8), dtype=torch.bfloat16, device=dev)
            o = 0
            for n in self._h_kernels:
                k_ = self.specs[n].numel
                self._ph[n], self._pth[n] = self.theta_h[o:o + k_], self.theta_th[o:o + k_]
                o += -(-k_ // ALIGN) * ALIGN
            if sh is not None:
                self._ppk = sh._ppk
            else:
                for n in self._s2_kernels:
                    ne = ops.conv3x3s2_packed_elems(self.specs[n].shape[0], self.specs[n].shape[3])
                    if ne:
                        self._ppk[n] = torch.zeros(ne, dtype=torch.bfloat16, device=dev)

    def refresh_transposed(self):
        for n in self._tnames:
            if n in self._ph:
                continue                                    # bf16 path: both work copies come from the cast below
            s_ = self.specs[n]
            ops.transpose_weight(self._p[n], self._pt[n], s_.shape[0], s_.shape[1] * s_.shape[2], s_.shape[3])
        if self._ph:
            if self._cast_table is None:
                ent = []
                for n in self._ph:
                    s_ = self.specs[n]
                    N, T, C_ = s_.shape[0], s_.shape[1] * s_.shape[2], s_.shape[3]
                    ent.append((self._p[n], self._ph[n], self._pth[n], N, T, C_, C_, N, self._ppk.get(n)))
                self._cast_table = ops.make_cast_table(ent, self.device)
            ops.cast_weights_batched(self._cast_table)

    def load_keras_params(self, params):
        """params: name -> array in Keras layout (HWIO Conv2D, HWOI Conv2DTranspose, [in,out] Dense)."""
        with torch.no_grad():
            for n, s_ in self.specs.items():
                a = torch.as_tensor(params[n]).to(torch.float32)
                if tuple(a.shape) != s_.keras_shape:
                    raise ValueError(f"{n}: expected Keras shape {s_.keras_shape}, got {tuple(a.shape)}")
                t = self._p[n]
                if s_.kind in ("conv", "convT"):
                    if a.dim() == 4:
                        t.copy_(a.permute(3, 0, 1, 2).to(self.device))
                    else:
                        t.copy_(a.t().reshape(t.shape).to(self.device))
                elif s_.kind in ("conv_padin", "convT_padout"):
                    t.zero_(); t[..., :2].copy_(a.permute(3, 0, 1, 2).to(self.device))
                elif s_.kind == "conv_padout":
                    t.zero_(); t[:2].copy_(a.permute(3, 0, 1, 2).to(self.device))
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
                a = t.permute(1, 2, 3, 0) if len(s_.keras_shape) == 4 else t.reshape(t.shape[0], -1).t()
            elif s_.kind in ("conv_padin", "convT_padout"):
                a = t[..., :2].permute(1, 2, 3, 0)
            elif s_.kind == "conv_padout":
                a = t[:2].permute(1, 2, 3, 0)
            elif s_.kind == "bias_pad":
                a = t[:2]
            else:
                a = t
            out[n] = a.contiguous().cpu()
        return out

    def export_keras_grads(self):
        return self._to_keras(self._g)

    def export_keras_params(self):
        return self._to_keras(self._p)

    def reset_parameters(self, generator=None):
        """Keras defaults: glorot_uniform kernels, zero biases, gamma 1, beta 0, Embedding U(-0.05, 0.05)."""
        with torch.no_grad():
            for n, s_ in self.specs.items():
                t, ks = self._p[n], s_.keras_shape
                if s_.kind == "embedding":
                    t.copy_((torch.rand(s_.shape, generator=generator) * 0.1 - 0.05).to(self.device))
                elif s_.kind.startswith("conv"):
                    rf = ks[0] * ks[1] if len(ks) == 4 else 1
                    fan_in, fan_out = (ks[2] * rf, ks[3] * rf) if len(ks) == 4 else ks
                    lim = math.sqrt(6.0 / (fan_in + fan_out))
                    w = ((torch.rand(s_.shape, generator=generator) * 2 - 1) * lim).to(self.device)
                    if s_.kind in ("conv_padin", "convT_padout"):
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

    # ------------------------------------------------------------------ step pieces
    def run_forward(self, lo=0, hi=None):
        if self.t_dirty or self.training:
            self.refresh_transposed()
            self.t_dirty = False
        for fwd, _ in self.ops[lo:hi]:
            fwd()

    def backward(self, on_ready=None, dpred=None, include_reg=True):
        """Gradients of every trainable variable into the flat gradient buffer, seeded by the loss kernel's dL/dlogits (or by an
        upstream dL/dpred, NCHW).  The flat buffer is laid out in backward-completion order, so after an op's backward closure
        everything up to the end of that op's parameters is final: `on_ready(offset_end)` hands that prefix to the trainer
        (gradient bucket all-reduce, trainer.GradBucketer)."""
        self.include_reg = include_reg
        if dpred is not None:
            ops.sigmoid_bwd(self.pred, dpred, self.logits.g)
            self.logits.g_set = True
        for (_, bwd), names in zip(reversed(self.ops), reversed(self._op_params)):
            bwd()
            if on_ready is not None and names:
                off = max(self.specs[n].offset + (-(-self.specs[n].numel // ALIGN) * ALIGN) for n in names)
                if self.wg_stream is None:
                    self._hand_over(on_ready, off)
                else:
                    # side-stream schedule: the prefix is final once the side stream has run this op's leaves AND the main
                    # stream its last reader of these parameters (the data gradient just queued).  The hand-over is parked
                    # until the side stream next waits for the main stream (the next op's `with self._wg()`), as in
                    # engine.UNetEngine.backward: no event record of its own in the main stream.
                    self._pending_ready.append((on_ready, off))
        if self.wg_stream is not None:
            if self._pending_ready or (self._rb is not None and len(self._rb)):
                with self._wg():
                    self.flush_reduces()        # behind the hand-over of what was still parked: the reductions nobody asked for yet
            self._join_wg()         # the optimizer and the next forward must see every weight gradient
        else:
            self.flush_reduces()
        for node in self.nodes:          # next step: the first writer of every gradient writes again
            node.g_set = False

    def loss_or_sigmoid(self, logits: Node, target, global_batch, alpha):
        la = logits.a
        if la.sfx == "bf16":          # a bf16 output layer (ResAE / Autoencoder): the sigmoid + loss kernel reads fp32 logits
            if self._logits32 is None:
                self._logits32 = ops.new_act(self.B, la.H, la.W, la.ld, self.device)
            ops.cast_bf16_to_f32(la, self._logits32)
            la = self._logits32
        if target is not None:
            gb = self.B if global_batch is None else global_batch
            ops.sigmoid_loss(la, target, alpha, 1.0 / (2.0 * self.H * self.W * gb), self.pred, logits.g, self.loss_out, self.ws,
                             **self._loss_extras())
            logits.g_set = True
        else:
            ops.sigmoid_nchw(la, self.pred)
        return self.pred

    def loss_from_logits(self, target, global_batch=None, alpha=0.9):
        """compute_loss for the logits of the last forward pass (seeds backward())."""
        if tuple(target.shape) != (self.B, 2, self.H, self.W) or target.dtype != torch.float32 or not target.is_contiguous():
            raise ValueError(f"target must be a contiguous float32 [{self.B},2,{self.H},{self.W}] tensor")
        self.loss_or_sigmoid(self.logits, target, global_batch, alpha)

    def loss_total(self):
        self.loss_tot.copy_(self.loss_out[0:1])
        self.reg_loss(into=self.loss_tot, accumulate=True)
        return self.loss_tot

    def _alloc_outputs(self):
        dev = self.device
        self.loss_tot = torch.zeros(1, dtype=torch.float32, device=dev)
        self._logits32 = None
        self.pred = torch.empty((self.B, 2, self.H, self.W), dtype=torch.float32, device=dev)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=dev)
        self.reg_out = torch.zeros(1, dtype=torch.float32, device=dev)

    def reg_loss(self, into=None, accumulate=False):
        """sum(model.losses) / replicas evaluated on device into reg_out[0] (or added to `into`[0])."""
        out = self.reg_out if into is None else into
        first = not accumulate
        for n in self.l2_names:
            s_ = self.specs[n]
            ops.sumsq(self.theta[s_.offset:s_.offset + s_.numel], L2_COEF / self.n_replicas, out, not first, self.ws)
            first = False
        if first:
            out.zero_()
        return out

    def dropout_mask(self, n, generator=None, slot=0):
        """Keep mask [B, n] of Dropout(.3) scaled by 1/(1-p): HIP generator kernel into a reused buffer per `slot`, or torch's
        generator when one is passed (tests)."""
        if generator is not None:
            return (torch.rand((self.B, n), device=self.device, generator=generator) >= DROPOUT_P).to(torch.float32) / (1.0 - DROPOUT_P)
        buf = self._mask_bufs.get(slot)
        if buf is None or buf.shape[1] != n:
            buf = self._mask_bufs[slot] = torch.empty((self.B, n), dtype=torch.float32, device=self.device)
        return self._draw_mask(buf)

    def n_params(self):
        return sum(int(math.prod(s_.keras_shape)) for s_ in self.specs.values())
