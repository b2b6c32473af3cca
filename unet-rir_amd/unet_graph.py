"""The U-Net of dl_models/u_net.py with every feature-block mode (0-3) on graph.GraphEngine (fp32 or bf16 storage).

engine.UNetEngine is the hand-scheduled fast path for mode 0 (the only mode the live driver and every BASELINE config
use); this engine covers mode 1 (convolutional_block_2), mode 2 (residual_block_1) and mode 3 (residual_block_2)
(dl_models/u_net.py:324-386) with the same kernels, and mode 0 as a cross-check of the hand schedule.
"""
import math

import torch

from . import ops
from .graph import GraphEngine, Node, RELU
from .engine import VEC_CH, EMB_DIM


class UNetGraphEngine(GraphEngine):
    def __init__(self, H, W, B, F0=32, k=3, depth=4, mode=0, batchnorm=True, inf_vector_shape=(2, 16), device="cuda:0",
                 n_replicas=1, runtime=None, share=None, dtype="f32", overlap_wgrad=False):
        super().__init__(B, device, n_replicas, runtime, share, dtype, overlap_wgrad)
        if mode not in (0, 1, 2, 3):
            raise ValueError("mode must be 0..3")
        if F0 % self.PAD:
            raise ValueError(f"number_filters_0 must be a multiple of {self.PAD} for dtype {dtype}")
        self.H, self.W, self.F0, self.k, self.depth, self.mode, self.batchnorm = H, W, F0, k, depth, mode, batchnorm
        self.inf_vector_shape = tuple(inf_vector_shape)
        self.n_idx = int(math.prod(self.inf_vector_shape))
        self.ch = [F0 * 2 ** l for l in range(depth + 1)]
        self._build()
        self._finalize_params()
        self._alloc_outputs()

    def _unit(self, x, name, out=None):
        """convolutional_block_1: Conv2D(3x3) -> BatchNormalization -> ReLU (dl_models/u_net.py:363-371)."""
        c = self._conv(x, name, x.a.C if out is None else out.a.C, 3, 1, followed_by_bn=self.batchnorm, l2=False)
        return self._bn_act(c, name, RELU, out=out, batchnorm=self.batchnorm)

    def _conv_bn_relu(self, x, name, cout, k):
        c = self._conv(x, name, cout, k, 1, followed_by_bn=self.batchnorm, l2=False)
        return self._bn_act(c, name, RELU, batchnorm=self.batchnorm)

    def _feature_block(self, x, base, out=None):
        """dl_models/u_net.py:280-287 / :312-319."""
        if self.mode == 0:
            return self._unit(x, base + (".cb1" if base.startswith("enc") else ".cb1b"), out)
        a = self._unit(x, base + ".fb.c1")
        if self.mode == 1:
            return self._unit(a, base + ".fb.c2", out)
        b = self._unit(a, base + ".fb.c2")
        if self.mode == 2:
            return self._add(b, x, out)                    # residual_block_1: Add()([x, input_layer])
        c = self._unit(x, base + ".fb.c3")
        return self._add(b, c, out)                        # residual_block_2

    def _build(self):
        """UNet._build (dl_models/u_net.py:201-251)."""
        B, dev, D, ch, k = self.B, self.device, self.depth, self.ch, self.k
        self.x4 = self._reg(Node(ops.new_act(B, self.H, self.W, self.PAD, dev, dtype=self.adt), needs_grad=False))
        x, h, w = self.x4, self.H, self.W
        skips, cats = [], []
        for l in range(1, D + 2):
            c = ch[l - 1]
            stride = 1 if l == 1 else 2
            d = self._conv(x, f"enc{l}.down", c, k, stride, followed_by_bn=False, pad_in=self.PAD if l == 1 else 0, l2=True)
            h, w = d.a.H, d.a.W
            if l <= D:         # the block output is the skip: written straight into the lower half of the concat buffer
                cat = self._new(h, w, 2 * c)
                skip, up = self._view(cat, 0, c), self._view(cat, c, c)
                x = self._feature_block(d, f"enc{l}", out=skip)
                skips.append((cat, skip, up))
            else:
                x = self._feature_block(d, f"enc{l}")
        h5, w5, cL = x.a.H, x.a.W, x.a.C
        # vector_block (dl_models/u_net.py:253-263) + Add (:229)
        flat = self._embedding(self.n_idx, "vec.embedding")
        v = self._dense(flat, "vec.dense", h5 * w5 * VEC_CH)
        vd = self._dropout(v, "vec")
        vsp = self._reshape(vd, h5, w5, VEC_CH)
        vc = self._conv(vsp, "vec.conv", cL, 1, 1, followed_by_bn=False, l2=False)
        x = self._add(x, vc)
        for l in range(D, 0, -1):
            c = ch[l - 1]
            cat, skip, up = skips[l - 1]
            self._conv(x, f"dec{l}.up", c, k, 2, transpose=True, followed_by_bn=False, out=up, l2=True)
            a = self._conv_bn_relu(cat, f"dec{l}.cb1a", c, k)
            x = self._feature_block(a, f"dec{l}")
        self.logits = self._head6x6(x, "head")
        self.vec_dim = h5 * w5 * VEC_CH      # l2(0.001) sits only on the strided and the transposed convs (:274, :302)

    def make_dropout_mask(self, generator=None):
        return self.dropout_mask(self.vec_dim, generator)

    def forward(self, spec, emb, dropout_mask=None, target=None, global_batch=None, alpha=0.9):
        B = self.B
        if tuple(spec.shape) != (B, 2, self.H, self.W) or spec.dtype != torch.float32 or not spec.is_contiguous():
            raise ValueError(f"spec must be a contiguous float32 [{B},2,{self.H},{self.W}] tensor")
        self.set_indices(emb)
        self.masks["vec"] = dropout_mask
        self._last_spec = spec
        ops.nchw_to_nhwc_pad(spec, self.x4.a)
        self.run_forward()
        return self.loss_or_sigmoid(self.logits, target, global_batch, alpha)
