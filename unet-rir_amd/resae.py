"""ResAE (dl_models/res_ae.py) on the same HIP kernels: the second operator graph of BASELINE.json configs[4].

Residual bottleneck blocks (1x1 -> kxk -> 1x1 Conv2D / Conv2DTranspose, BatchNormalization, LeakyReLU(0.3), Add), a Dense
latent that concatenates the information vector, and a mirrored Conv2DTranspose decoder, built on graph.GraphEngine.
"""
import math

import torch

from . import ops
from .graph import GraphEngine, Node, LEAKY


class ResAEEngine(GraphEngine):
    """One replica of ResAE for a fixed per-replica batch size (constructor mirrors dl_models/res_ae.py:41-50)."""
    n_dropout_draws = 2          # two Dropout layers: two masks per step

    def __init__(self, H, W, B, conv_filters=(32, 64, 128, 256), conv_kernels=(3, 3, 3, 3), conv_strides=(2, 2, 2, 2),
                 latent_space_dim=32, n_neurons=1024, inf_vector_shape=(2, 16), device="cuda:0", n_replicas=1, runtime=None, share=None, dtype="f32",
                 overlap_wgrad=False):
        super().__init__(B, device, n_replicas, runtime, share, dtype, overlap_wgrad)
        self.H, self.W = H, W
        self.filters, self.kernels, self.strides = tuple(conv_filters), tuple(conv_kernels), tuple(conv_strides)
        if any(f % 4 for f in self.filters) or any(s not in (1, 2) for s in self.strides):
            raise ValueError("conv_filters must be multiples of 4 and conv_strides 1 or 2")
        self.latent, self.n_neurons = latent_space_dim, n_neurons
        if latent_space_dim % 4 or n_neurons % 4:
            raise ValueError("latent_space_dim and n_neurons must be multiples of 4")
        self.inf_vector_shape = tuple(inf_vector_shape)
        self.n_idx = int(math.prod(self.inf_vector_shape))
        self._build()
        self._finalize_params()
        self._alloc_outputs()

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

    def _build(self):
        """dl_models/res_ae.py:210-530."""
        B, dev = self.B, self.device
        n = len(self.filters)
        self.x4 = self._reg(Node(ops.new_act(B, self.H, self.W, self.PAD, dev, dtype=self.adt), needs_grad=False))
        x = self.x4
        for i in range(n):        # encoder: _add_conv_layers (:424-451)
            x = self._res_block(x, f"e_res_{i + 1}", self.filters[i], self.kernels[i], self.strides[i], False, True, pad_in=self.PAD if i == 0 else 0)
            x = self._res_block(x, f"e_res_{i + 1}", self.filters[i], self.kernels[i], 1, False, False)
        h, w, c = x.a.H, x.a.W, x.a.C
        self.shape_before_bottleneck = (h, w, c)
        n_feat = h * w * c
        flat_vec = self._embedding(self.n_idx)                  # Embedding -> Flatten (:411-420)
        vec = self._dense(flat_vec, "e_dense_vector", self.n_neurons)
        # concatenate([Flatten(x), vec]) -> Dense(latent) -> Dropout (:516-530); the concat is a copy of two row blocks
        cat = self._new(1, 1, n_feat + self.n_neurons, f32=True)     # fp32 (the copies below convert the trunk half)
        x_last = x

        def cat_fwd():
            cat.a.base.view(B, -1)[:, :n_feat].copy_(x_last.a.base.view(B, -1))
            cat.a.base.view(B, -1)[:, n_feat:].copy_(vec.a.base.view(B, -1))

        def cat_bwd():
            x_last.g.base.view(B, -1).copy_(cat.g.base.view(B, -1)[:, :n_feat]); x_last.g_set = True
            vec.g.base.view(B, -1).copy_(cat.g.base.view(B, -1)[:, n_feat:]); vec.g_set = True
        self._push(cat_fwd, cat_bwd)
        z = self._dense(cat, "e_out", self.latent)
        zd = self._dropout(z, "latent")
        self._latent, self._n_enc_ops = zd, len(self.ops)          # model.encoder ends here (dl_models/res_ae.py:516-530)
        d = self._dense(zd, "decoder_dense", n_feat)             # decoder: Dense -> Dropout -> Reshape (:247-268)
        dd = self._dropout(d, "dec")
        x = self._reshape(dd, h, w, c)
        if self.dtype == "bf16":
            x = self._cast(x)                                  # the Dense branch is fp32, the transposed-conv trunk bf16
        x = self._res_block(x, "d_res_0", self.filters[-1], self.kernels[-1], 1, True, True)
        x = self._res_block(x, "d_res_0", self.filters[-1], self.kernels[-1], 1, True, False)
        for layer_index in reversed(range(1, n)):                # _add_conv_transpose_layers (:272-308)
            name = f"d_res_{n - layer_index}"
            f, k = self.filters[layer_index - 1], self.kernels[layer_index]
            x = self._res_block(x, name, f, k, self.strides[layer_index - 1], True, True)
            x = self._res_block(x, name, f, k, 1, True, False)
        # _add_decoder_output (:373-389): Conv2DTranspose(2, k0, s0, 'same') + sigmoid; Cout padded 2 -> 4
        self.logits = self._conv(x, "d_out", 2, self.kernels[0], self.strides[0], True, followed_by_bn=False, pad_out=self.PAD, l2=False)
        if (self.logits.a.H, self.logits.a.W) != (self.H, self.W):
            raise ValueError("decoder output size does not match the input size")

    def forward(self, spec, emb, mask_latent=None, mask_dec=None, target=None, global_batch=None, alpha=0.9, dropout_mask=None):
        """dropout_mask: the (latent, decoder) pair make_dropout_mask() returns (the trainer's calling convention)."""
        B = self.B
        if dropout_mask is not None:
            mask_latent, mask_dec = dropout_mask
        if tuple(spec.shape) != (B, 2, self.H, self.W) or spec.dtype != torch.float32 or not spec.is_contiguous():
            raise ValueError(f"spec must be a contiguous float32 [{B},2,{self.H},{self.W}] tensor")
        self.set_indices(emb)
        self.masks["latent"], self.masks["dec"] = mask_latent, mask_dec
        self._last_spec = spec
        ops.nchw_to_nhwc_pad(spec, self.x4.a)
        self.run_forward()
        return self.loss_or_sigmoid(self.logits, target, global_batch, alpha)

    def encode(self, spec, emb, dropout_mask=None):
        """model.encoder([spec, emb]) (dl_models/res_ae.py:62, :391-403): the latent vector [B, latent_space_dim] (a copy)."""
        B = self.B
        if tuple(spec.shape) != (B, 2, self.H, self.W) or spec.dtype != torch.float32 or not spec.is_contiguous():
            raise ValueError(f"spec must be a contiguous float32 [{B},2,{self.H},{self.W}] tensor")
        self.set_indices(emb)
        self.masks["latent"] = dropout_mask[0] if dropout_mask is not None else None
        self._last_spec = spec
        ops.nchw_to_nhwc_pad(spec, self.x4.a)
        self.run_forward(0, self._n_enc_ops)
        return self._latent.a.base.view(B, self.latent).clone()

    def decode(self, z, dropout_mask=None):
        """model.decoder(z) (dl_models/res_ae.py:63, :233-245): z [B, latent_space_dim] -> prediction [B,2,H,W] (NCHW buffer)."""
        if tuple(z.shape) != (self.B, self.latent) or z.dtype != torch.float32:
            raise ValueError(f"z must be float32 [{self.B},{self.latent}]")
        self._latent.a.base.view(self.B, self.latent).copy_(z)
        self.masks["dec"] = dropout_mask[1] if dropout_mask is not None else None
        self.run_forward(self._n_enc_ops, None)
        return self.loss_or_sigmoid(self.logits, None, None, 0.9)

    def make_dropout_masks(self, generator=None):
        h, w, c = self.shape_before_bottleneck
        return self.dropout_mask(self.latent, generator, 0), self.dropout_mask(h * w * c, generator, 1)

    make_dropout_mask = make_dropout_masks
