"""Autoencoder (dl_models/autoencoder.py) on the same HIP kernels: the conv / conv-transpose BatchNorm-ReLU stack with a Dense
latent that main_training.py:118-129 builds for name == "ae" - the remaining graph of the reference's autoencoder family
(SURVEY.md 8(f) rank 4).  Built on graph.GraphEngine; no kernel of its own.
"""
import math

import torch

from . import ops
from .graph import GraphEngine, Node, RELU


class AutoencoderEngine(GraphEngine):
    """One replica of Autoencoder for a fixed per-replica batch size (constructor mirrors dl_models/autoencoder.py:41-46)."""
    n_dropout_draws = 2          # two Dropout layers: two masks per step

    def __init__(self, H, W, B, conv_filters=(64, 128, 256, 512), conv_kernels=(3, 3, 3, 3), conv_strides=(2, 2, 2, 2),
                 latent_space_dim=64, n_neurons=2048, inf_vector_shape=(2, 16), device="cuda:0", n_replicas=1, runtime=None,
                 share=None, dtype="f32", overlap_wgrad=False):
        super().__init__(B, device, n_replicas, runtime, share, dtype, overlap_wgrad)
        self.H, self.W = H, W
        self.filters, self.kernels, self.strides = tuple(conv_filters), tuple(conv_kernels), tuple(conv_strides)
        if any(f % 4 for f in self.filters) or any(s not in (1, 2) for s in self.strides):
            raise ValueError("conv_filters must be multiples of 4 and conv_strides 1 or 2")
        if latent_space_dim % 4 or n_neurons % 4:
            raise ValueError("latent_space_dim and n_neurons must be multiples of 4")
        self.latent, self.n_neurons = latent_space_dim, n_neurons
        self.inf_vector_shape = tuple(inf_vector_shape)
        self.n_idx = int(math.prod(self.inf_vector_shape))
        self._build()
        self._finalize_params()
        self._alloc_outputs()

    def _build(self):
        """dl_models/autoencoder.py:210-417."""
        B, dev = self.B, self.device
        n = len(self.filters)
        self.x4 = self._reg(Node(ops.new_act(B, self.H, self.W, self.PAD, dev, dtype=self.adt), needs_grad=False))
        x = self.x4
        for i in range(n):        # encoder: Conv2D(l2) -> BatchNormalization -> ReLU (:384-402)
            c = self._conv(x, f"encoder_conv_layer_{i + 1}", self.filters[i], self.kernels[i], self.strides[i], False,
                           pad_in=self.PAD if i == 0 else 0)
            x = self._bn_act(c, f"encoder_bn_{i + 1}", RELU)
        h, w, c = x.a.H, x.a.W, x.a.C
        self.shape_before_bottleneck = (h, w, c)
        n_feat = h * w * c
        flat_vec = self._embedding(self.n_idx)                  # Embedding -> Flatten -> Dense -> Dropout (:357-369)
        vec = self._dense(flat_vec, "encoder_inf_dense", self.n_neurons)
        vecd = self._dropout(vec, "inf")
        # concatenate([Flatten(x), y]) -> Dense(latent) (:404-417); the concat is a copy of two row blocks
        cat = self._new(1, 1, n_feat + self.n_neurons, f32=True)     # fp32 (the copies below convert the trunk half)
        x_last = x

        def cat_fwd():
            cat.a.base.view(B, -1)[:, :n_feat].copy_(x_last.a.base.view(B, -1))
            cat.a.base.view(B, -1)[:, n_feat:].copy_(vecd.a.base.view(B, -1))

        def cat_bwd():
            x_last.g.base.view(B, -1).copy_(cat.g.base.view(B, -1)[:, :n_feat]); x_last.g_set = True
            vecd.g.base.view(B, -1).copy_(cat.g.base.view(B, -1)[:, n_feat:]); vecd.g_set = True
        self._push(cat_fwd, cat_bwd)
        z = self._dense(cat, "encoder_output", self.latent)
        self._latent, self._n_enc_ops = z, len(self.ops)          # model.encoder ends here
        d = self._dense(z, "decoder_dense", n_feat)               # decoder: Dense -> Dropout -> Reshape (:245-265)
        dd = self._dropout(d, "dec")
        x = self._reshape(dd, h, w, c)
        if self.dtype == "bf16":
            x = self._cast(x)                                  # the Dense branch is fp32, the transposed-conv trunk bf16
        ct = self._conv(x, "decoder_conv_transpose_layer_0", self.filters[-1], self.kernels[-1], 1, True)      # stride 1 (:267-285)
        x = self._bn_act(ct, "decoder_bn_0", RELU)
        for layer_index in reversed(range(1, n)):                 # _add_conv_transpose_layers (:287-320)
            num = n - layer_index
            ct = self._conv(x, f"decoder_conv_transpose_layer_{num}", self.filters[layer_index - 1], self.kernels[layer_index - 1],
                            self.strides[layer_index - 1], True)
            x = self._bn_act(ct, f"decoder_bn_{num}", RELU)
        # _add_decoder_output (:322-335): Conv2DTranspose(2, k0, s0, 'same') + sigmoid; Cout padded 2 -> 4, no l2
        self.logits = self._conv(x, f"decoder_out_{n}", 2, self.kernels[0], self.strides[0], True, followed_by_bn=False, pad_out=self.PAD,
                                 l2=False)
        if (self.logits.a.H, self.logits.a.W) != (self.H, self.W):
            raise ValueError("decoder output size does not match the input size")

    def forward(self, spec, emb, mask_inf=None, mask_dec=None, target=None, global_batch=None, alpha=0.9, dropout_mask=None):
        """dropout_mask: the (information-vector, decoder) pair make_dropout_mask() returns (the trainer's calling convention)."""
        B = self.B
        if dropout_mask is not None:
            mask_inf, mask_dec = dropout_mask
        if tuple(spec.shape) != (B, 2, self.H, self.W) or spec.dtype != torch.float32 or not spec.is_contiguous():
            raise ValueError(f"spec must be a contiguous float32 [{B},2,{self.H},{self.W}] tensor")
        self.set_indices(emb)
        self.masks["inf"], self.masks["dec"] = mask_inf, mask_dec
        self._last_spec = spec
        ops.nchw_to_nhwc_pad(spec, self.x4.a)
        self.run_forward()
        return self.loss_or_sigmoid(self.logits, target, global_batch, alpha)

    def encode(self, spec, emb, dropout_mask=None):
        """model.encoder([spec, emb]) (dl_models/autoencoder.py:337-346): the latent vector [B, latent_space_dim] (a copy)."""
        B = self.B
        if tuple(spec.shape) != (B, 2, self.H, self.W) or spec.dtype != torch.float32 or not spec.is_contiguous():
            raise ValueError(f"spec must be a contiguous float32 [{B},2,{self.H},{self.W}] tensor")
        self.set_indices(emb)
        self.masks["inf"] = dropout_mask[0] if dropout_mask is not None else None
        self._last_spec = spec
        ops.nchw_to_nhwc_pad(spec, self.x4.a)
        self.run_forward(0, self._n_enc_ops)
        return self._latent.a.base.view(B, self.latent).clone()

    def decode(self, z, dropout_mask=None):
        """model.decoder(z) (dl_models/autoencoder.py:222-233): z [B, latent_space_dim] -> prediction [B,2,H,W] (NCHW buffer)."""
        if tuple(z.shape) != (self.B, self.latent) or z.dtype != torch.float32:
            raise ValueError(f"z must be float32 [{self.B},{self.latent}]")
        self._latent.a.base.view(self.B, self.latent).copy_(z)
        self.masks["dec"] = dropout_mask[1] if dropout_mask is not None else None
        self.run_forward(self._n_enc_ops, None)
        return self.loss_or_sigmoid(self.logits, None, None, 0.9)

    def make_dropout_mask(self, generator=None):
        h, w, c = self.shape_before_bottleneck
        return self.dropout_mask(self.n_neurons, generator, 0), self.dropout_mask(h * w * c, generator, 1)
