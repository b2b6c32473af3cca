"""ResAE (dl_models/res_ae.py) on the same HIP kernels: the second operator graph of BASELINE.json configs[4].

Residual bottleneck blocks (1x1 -> kxk -> 1x1 Conv2D / Conv2DTranspose, BatchNormalization, LeakyReLU(0.3), Add), a Dense
latent that concatenates the information vector, and a mirrored Conv2DTranspose decoder.  The graph is built once as a
list of ops over preallocated NHWC fp32 buffers; forward runs the list, backward runs it in reverse.  A tensor with two
consumers (a block input feeds the residual path and the skip) gets its gradient from two writers: the first writes,
the second accumulates in place (conv data gradients through the kernels' `addend` epilogue).
"""
import math
from collections import OrderedDict

import torch

from . import ops
from .ops import Act
from .engine import ALIGN, BN_EPS, BN_MOMENTUM, L2_COEF, VOCAB, EMB_DIM, DROPOUT_P, ParamSpec

LEAKY = 2       # activation code of the C ABI: LeakyReLU(0.3)


class Node:
    """An activation buffer and the buffer of its gradient."""
    __slots__ = ("a", "g", "g_set", "needs_grad")

    def __init__(self, a: Act, needs_grad=True):
        self.a = a
        self.g = Act(torch.empty_like(a.base)) if needs_grad else None
        self.g_set = False
        self.needs_grad = needs_grad


class ResAEEngine:
    """One replica of ResAE for a fixed per-replica batch size (constructor mirrors dl_models/res_ae.py:41-50)."""

    def __init__(self, H, W, B, conv_filters=(32, 64, 128, 256), conv_kernels=(3, 3, 3, 3), conv_strides=(2, 2, 2, 2),
                 latent_space_dim=32, n_neurons=1024, inf_vector_shape=(2, 16), device="cuda:0", n_replicas=1):
        self.H, self.W, self.B = H, W, B
        self.filters, self.kernels, self.strides = tuple(conv_filters), tuple(conv_kernels), tuple(conv_strides)
        if any(f % 4 for f in self.filters) or any(s not in (1, 2) for s in self.strides):
            raise ValueError("conv_filters must be multiples of 4 and conv_strides 1 or 2")
        self.latent, self.n_neurons = latent_space_dim, n_neurons
        if latent_space_dim % 4 or n_neurons % 4:
            raise ValueError("latent_space_dim and n_neurons must be multiples of 4")
        self.inf_vector_shape = tuple(inf_vector_shape)
        self.device = torch.device(device)
        self.n_replicas = n_replicas
        self.n_idx = int(math.prod(self.inf_vector_shape))
        self.nodes = []              # every activation/gradient buffer pair (accumulate-or-write flags are reset per step)
        self.specs_fwd = []          # ParamSpec in creation order
        self.ops = []                # (fwd, bwd) closures
        self.bn_names = []
        self.l2_names = []
        self.ws = ops.Workspace(self.device, 1 << 20)
        self._p, self._g, self._pt = {}, {}, {}            # filled by _finalize_params
        self._build()
        self._finalize_params()
        self.adam_t = 0

    # ------------------------------------------------------------------ graph construction helpers
    def _param(self, name, shape, kind, keras_shape, l2=False):
        self.specs_fwd.append(ParamSpec(name, shape, kind, keras_shape))
        if l2:
            self.l2_names.append(name)

    def _new(self, h, w, c, needs_grad=True):
        return self._reg(Node(ops.new_act(self.B, h, w, c, self.device), needs_grad))

    def _reg(self, node):
        self.nodes.append(node)
        return node

    def _emit(self, node, writer):
        """writer(dst, addend): dst = value (+ addend).  First writer of a gradient writes, later ones accumulate."""
        if not node.needs_grad:
            return
        writer(node.g, node.g if node.g_set else None)
        node.g_set = True

    def _conv(self, x: Node, name, cout, k, stride, transpose, followed_by_bn=True, pad_in=0, pad_out=0, l2=True, dense=False):
        """Conv2D / Conv2DTranspose(padding 'same' or 1x1 'valid') + bias."""
        B = self.B
        cin = x.a.C
        if transpose:
            H, W = x.a.H * stride, x.a.W * stride
        else:
            H, W = -(-x.a.H // stride), -(-x.a.W // stride)
        co = cout if not pad_out else pad_out
        y = self._new(H, W, co)
        kname, bname = name + ".kernel", name + ".bias"
        real_in = cin if not pad_in else 2
        if transpose:      # primary layout [Cin][k][k][Cout]; keras (k,k,Cout,Cin)
            self._param(kname, (cin, k, k, co), "convT_padout" if pad_out else "convT", (k, k, cout, real_in), l2)
        else:              # [Cout][k][k][Cin]; keras (k,k,Cin,Cout)
            self._param(kname, (co, k, k, cin), "conv_padin" if pad_in else "conv", (cin, cout) if dense else (k, k, real_in, cout), l2)
        self._param(bname, (co,), "bias_pad" if pad_out else "bias", (cout,))
        g = ops.geom(B, x.a.H, x.a.W, cin, co, k, stride)
        reg = lambda: (2.0 * L2_COEF / self.n_replicas) if l2 else 0.0

        def fwd():
            if transpose:
                ops.conv2d_transpose_fwd(g, x.a, self._pt[kname], self._p[bname], y.a)
            else:
                ops.conv2d_fwd(g, x.a, self._p[kname], self._p[bname], y.a)

        def bwd():
            if transpose:
                ops.conv2d_transpose_wgrad(g, x.a, y.g, self._g[kname], self.ws, reg=reg(), w=self._p[kname])
            else:
                ops.conv2d_wgrad(g, x.a, y.g, self._g[kname], self.ws, reg=reg(), w=self._p[kname])
            if not followed_by_bn:        # a bias in front of BatchNorm has an identically zero gradient
                ops.colsum(y.g, self._g[bname], self.ws)
            if transpose:
                self._emit(x, lambda dst, add: ops.conv2d_transpose_dgrad(g, y.g, self._p[kname], dst, addend=add))
            else:
                self._emit(x, lambda dst, add: ops.conv2d_dgrad(g, y.g, self._pt[kname], dst, addend=add))
        self.ops.append((fwd, bwd))
        return y

    def _bn_act(self, x: Node, name, act, addend: Node = None):
        """BatchNormalization (+ Add) (+ LeakyReLU)."""
        c = x.a.C
        self._param(name + ".gamma", (c,), "gamma", (c,))
        self._param(name + ".beta", (c,), "beta", (c,))
        self.bn_names.append(name)
        y = self._new(x.a.H, x.a.W, c)
        aff = torch.empty(2 * c, dtype=torch.float32, device=self.device)
        saved = torch.empty(2 * c, dtype=torch.float32, device=self.device)
        mm = torch.zeros(c, dtype=torch.float32, device=self.device)
        mv = torch.ones(c, dtype=torch.float32, device=self.device)
        self.moving[name + ".moving_mean"], self.moving[name + ".moving_variance"] = mm, mv
        gj = Act(torch.empty_like(x.a.base)) if addend is not None else None

        def fwd():
            ops.bn_stats(x.a, self._p[name + ".gamma"], self._p[name + ".beta"], aff, saved, self.ws, mm, mv, BN_EPS, BN_MOMENTUM)
            ops.bn_act_add(x.a, aff, y.a, act, addend.a if addend is not None else None)

        def bwd():
            if addend is None:
                ops.bn_bwd(y.g, x.a, None, aff, saved, x.g, self._g[name + ".gamma"], self._g[name + ".beta"], self.ws, relu=act)
            else:      # junction y = act(bn(x) + addend): g = dy * act'(y) feeds both branches
                ops.act_bwd(y.g, y.a, gj, act)
                ops.bn_bwd(gj, x.a, None, aff, saved, x.g, self._g[name + ".gamma"], self._g[name + ".beta"], self.ws, relu=0)
                self._emit(addend, lambda dst, add: ops.bn_act_add(gj, None, dst, 0, add))
            x.g_set = True
        self.ops.append((fwd, bwd))
        return y

    def _res_block(self, x: Node, name, f, k, stride, transpose, with_skip, pad_in=0):
        """res_conv / res_identity and the Conv2DTranspose twins (dl_models/res_ae.py:310-371, :453-514)."""
        tag = "conv" if with_skip else "id"
        c1 = self._conv(x, f"{name}_{tag}.1", f, 1, stride if with_skip else 1, transpose, pad_in=pad_in)
        a1 = self._bn_act(c1, f"{name}_{tag}.1", LEAKY)
        c2 = self._conv(a1, f"{name}_{tag}.2", f, k, 1, transpose)
        a2 = self._bn_act(c2, f"{name}_{tag}.2", LEAKY)
        c3 = self._conv(a2, f"{name}_{tag}.3", f, 1, 1, transpose)
        if with_skip:
            cs = self._conv(x, f"{name}_conv.s", f, 1, stride, transpose, pad_in=pad_in)
            skip = self._bn_act(cs, f"{name}_conv.s", 0)
        else:
            skip = x
        return self._bn_act(c3, f"{name}_{tag}.3", LEAKY, addend=skip)

    def _dense(self, x: Node, name, n_out, followed_by_bn=False):
        return self._conv(x, name, n_out, 1, 1, False, followed_by_bn=followed_by_bn, l2=False, dense=True)

    # ------------------------------------------------------------------ the graph (dl_models/res_ae.py:210-530)
    def _build(self):
        B, dev = self.B, self.device
        self.moving = {}
        n = len(self.filters)
        self.x4 = self._reg(Node(ops.new_act(B, self.H, self.W, 4, dev), needs_grad=False))
        x = self.x4
        for i in range(n):        # encoder: _add_conv_layers (:424-451)
            x = self._res_block(x, f"e_res_{i + 1}", self.filters[i], self.kernels[i], self.strides[i], False, True, pad_in=4 if i == 0 else 0)
            x = self._res_block(x, f"e_res_{i + 1}", self.filters[i], self.kernels[i], 1, False, False)
        h, w, c = x.a.H, x.a.W, x.a.C
        self.shape_before_bottleneck = (h, w, c)
        n_feat = h * w * c
        # information vector: Embedding -> Flatten -> Dense(n_neurons) (:411-422)
        self._param("embedding", (VOCAB, EMB_DIM), "embedding", (VOCAB, EMB_DIM))
        self.emb_idx = torch.zeros(B * self.n_idx, dtype=torch.int32, device=dev)
        emb_out = torch.empty((B * self.n_idx, EMB_DIM), dtype=torch.float32, device=dev)
        g_emb_out = torch.empty_like(emb_out)
        flat_vec = self._reg(Node(Act(emb_out.view(B, 1, 1, self.n_idx * EMB_DIM)), needs_grad=False))
        flat_vec.g, flat_vec.needs_grad = Act(g_emb_out.view(B, 1, 1, self.n_idx * EMB_DIM)), True

        def emb_fwd():
            ops.embedding_fwd(self.emb_idx, self._p["embedding"], emb_out)

        def emb_bwd():
            ops.embedding_bwd(self.emb_idx, g_emb_out, self._g["embedding"])
        self.ops.append((emb_fwd, emb_bwd))
        vec = self._dense(flat_vec, "e_dense_vector", self.n_neurons)
        # concatenate([Flatten(x), vec]) -> Dense(latent) -> Dropout (:516-530); the concat is a copy of two row blocks
        cat = self._new(1, 1, n_feat + self.n_neurons)
        x_last = x

        def cat_fwd():
            cat.a.base.view(B, -1)[:, :n_feat].copy_(x_last.a.base.view(B, -1))
            cat.a.base.view(B, -1)[:, n_feat:].copy_(vec.a.base.view(B, -1))

        def cat_bwd():
            x_last.g.base.view(B, -1).copy_(cat.g.base.view(B, -1)[:, :n_feat]); x_last.g_set = True
            vec.g.base.view(B, -1).copy_(cat.g.base.view(B, -1)[:, n_feat:]); vec.g_set = True
        self.ops.append((cat_fwd, cat_bwd))
        z = self._dense(cat, "e_out", self.latent)
        self.mask_latent = self.mask_dec = None
        zd = self._dropout(z, "latent")
        d = self._dense(zd, "decoder_dense", n_feat)             # decoder: Dense -> Dropout -> Reshape (:247-268)
        dd = self._dropout(d, "dec")
        xr = self._reg(Node(Act(dd.a.base.view(B, h, w, c)), needs_grad=False))
        xr.g, xr.needs_grad = Act(dd.g.base.view(B, h, w, c)), True
        link = xr

        def reshape_bwd():
            dd.g_set = True
        self.ops.append((lambda: None, reshape_bwd))
        x = self._res_block(link, "d_res_0", self.filters[-1], self.kernels[-1], 1, True, True)
        x = self._res_block(x, "d_res_0", self.filters[-1], self.kernels[-1], 1, True, False)
        for layer_index in reversed(range(1, n)):                # _add_conv_transpose_layers (:272-308)
            name = f"d_res_{n - layer_index}"
            f, k = self.filters[layer_index - 1], self.kernels[layer_index]
            x = self._res_block(x, name, f, k, self.strides[layer_index - 1], True, True)
            x = self._res_block(x, name, f, k, 1, True, False)
        # _add_decoder_output (:373-389): Conv2DTranspose(2, k0, s0, 'same') + sigmoid; Cout padded 2 -> 4
        self.logits = self._conv(x, "d_out", 2, self.kernels[0], self.strides[0], True, followed_by_bn=False, pad_out=4, l2=False)
        if (self.logits.a.H, self.logits.a.W) != (self.H, self.W):
            raise ValueError("decoder output size does not match the input size")
        self.pred = torch.empty((B, 2, self.H, self.W), dtype=torch.float32, device=dev)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=dev)
        self.reg_out = torch.zeros(1, dtype=torch.float32, device=dev)

    def _dropout(self, x: Node, which):
        y = self._new(1, 1, x.a.C)

        def fwd():
            m = self.mask_latent if which == "latent" else self.mask_dec
            if m is None:
                y.a.base.copy_(x.a.base)
            else:
                ops.mul(x.a.base, m, y.a.base)

        def bwd():
            m = self.mask_latent if which == "latent" else self.mask_dec
            if m is None:
                x.g.base.copy_(y.g.base)
            else:
                ops.mul(y.g.base, m, x.g.base)
            x.g_set = True
        self.ops.append((fwd, bwd))
        return y

    # ------------------------------------------------------------------ parameters
    def _finalize_params(self):
        specs = list(reversed(self.specs_fwd))          # backward completion order
        off = 0
        for s_ in specs:
            s_.offset = off
            off += -(-s_.numel // ALIGN) * ALIGN
        self.specs = OrderedDict((s_.name, s_) for s_ in specs)
        dev = self.device
        self.theta = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(off, dtype=torch.float32, device=dev)
        for n, s_ in self.specs.items():
            self._p[n] = self.theta[s_.offset:s_.offset + s_.numel].view(s_.shape)
            self._g[n] = self.grad[s_.offset:s_.offset + s_.numel].view(s_.shape)
        self.p, self.g = self._p, self._g
        toff = 0
        self._tnames = [n for n, s_ in self.specs.items() if s_.kind.startswith("conv")]
        for n in self._tnames:
            toff += -(-self.specs[n].numel // ALIGN) * ALIGN
        self.theta_t = torch.zeros(max(toff, 4), dtype=torch.float32, device=dev)
        o = 0
        for n in self._tnames:
            k_ = self.specs[n].numel
            self._pt[n] = self.theta_t[o:o + k_]
            o += -(-k_ // ALIGN) * ALIGN
        need = 1 << 20
        for n in self._tnames:
            s_ = self.specs[n]
            need = max(need, 4 * s_.numel * 1024 if s_.numel < (1 << 16) else 4 * s_.numel * 64)
        self.ws.reserve(min(need, 1 << 30))

    def refresh_transposed(self):
        for n in self._tnames:
            s_ = self.specs[n]
            N, T, C_ = s_.shape[0], s_.shape[1] * s_.shape[2], s_.shape[3]
            ops.transpose_weight(self._p[n], self._pt[n], N, T, C_)

    def load_keras_params(self, params):
        with torch.no_grad():
            for n, s_ in self.specs.items():
                a = torch.as_tensor(params[n]).to(torch.float32)
                if tuple(a.shape) != s_.keras_shape:
                    raise ValueError(f"{n}: expected Keras shape {s_.keras_shape}, got {tuple(a.shape)}")
                t = self._p[n]
                if s_.kind in ("conv", "convT"):
                    t.copy_(a.permute(3, 0, 1, 2).to(self.device)) if len(a.shape) == 4 else t.copy_(a.t().reshape(t.shape).to(self.device))
                elif s_.kind == "conv_padin":
                    t.zero_(); t[..., :2].copy_(a.permute(3, 0, 1, 2).to(self.device))
                elif s_.kind == "convT_padout":
                    t.zero_(); t[..., :2].copy_(a.permute(3, 0, 1, 2).to(self.device))
                elif s_.kind == "bias_pad":
                    t.zero_(); t[:2].copy_(a.to(self.device))
                else:
                    t.copy_(a.to(self.device))

    def export_keras_grads(self):
        return self._to_keras(self._g)

    def export_keras_params(self):
        return self._to_keras(self._p)

    def _to_keras(self, views):
        out = {}
        for n, s_ in self.specs.items():
            t = views[n].detach()
            if s_.kind in ("conv", "convT"):
                a = t.permute(1, 2, 3, 0) if len(s_.keras_shape) == 4 else t.reshape(t.shape[0], -1).t()
            elif s_.kind == "conv_padin":
                a = t[..., :2].permute(1, 2, 3, 0)
            elif s_.kind == "convT_padout":
                a = t[..., :2].permute(1, 2, 3, 0)
            elif s_.kind == "bias_pad":
                a = t[:2]
            else:
                a = t
            out[n] = a.contiguous().cpu()
        return out

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
                    t.copy_(w)
                elif s_.kind == "gamma":
                    t.fill_(1.0)
                else:
                    t.zero_()

    # ------------------------------------------------------------------ step
    def forward(self, spec, emb, mask_latent=None, mask_dec=None, target=None, global_batch=None, alpha=0.9):
        B = self.B
        if tuple(spec.shape) != (B, 2, self.H, self.W) or spec.dtype != torch.float32 or not spec.is_contiguous():
            raise ValueError(f"spec must be a contiguous float32 [{B},2,{self.H},{self.W}] tensor")
        self.refresh_transposed()
        self.emb_idx.copy_(emb.reshape(-1).to(torch.int32))
        self.mask_latent, self.mask_dec = mask_latent, mask_dec
        ops.nchw_to_nhwc_pad(spec, self.x4.a)
        for fwd, _ in self.ops:
            fwd()
        if target is not None:
            gb = B if global_batch is None else global_batch
            ops.sigmoid_loss(self.logits.a, target, alpha, 1.0 / (2.0 * self.H * self.W * gb), self.pred, self.logits.g,
                             self.loss_out, self.ws)
            self.logits.g_set = True
        else:
            ops.sigmoid_nchw(self.logits.a, self.pred)
        return self.pred

    def backward(self):
        for _, bwd in reversed(self.ops):
            bwd()
        for node in self.nodes:          # next step: first writer of every gradient writes again
            node.g_set = False

    def reg_loss(self):
        first = True
        for n in self.l2_names:
            s_ = self.specs[n]
            ops.sumsq(self.theta[s_.offset:s_.offset + s_.numel], L2_COEF / self.n_replicas, self.reg_out, not first, self.ws)
            first = False
        return self.reg_out

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-7):
        self.adam_t += 1
        t = self.adam_t
        ops.adam(self.theta, self.grad, self.adam_m, self.adam_v, lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t),
                 beta1, beta2, eps, 1.0)

    def make_dropout_masks(self, generator=None):
        mk = lambda n: (torch.rand((self.B, n), device=self.device, generator=generator) >= DROPOUT_P).to(torch.float32) / (1.0 - DROPOUT_P)
        h, w, c = self.shape_before_bottleneck
        return mk(self.latent), mk(h * w * c)

    def n_params(self):
        return sum(int(math.prod(s_.keras_shape)) for s_ in self.specs.values())
