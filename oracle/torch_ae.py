"""torch.nn.functional CPU restatement of the reference Autoencoder graph (dl_models/autoencoder.py), the conv / conv-transpose
BN-ReLU stack with a Dense latent that main_training.py:118-129 builds for name == "ae".

Oracle / test infrastructure only (see oracle/__init__.py; PARITY UNPINNED: the TensorFlow/Keras reference cannot run
here and has no tests or fixtures).  Parameters are held in Keras layouts: Conv2D [kh,kw,Cin,Cout], Conv2DTranspose
[kh,kw,Cout,Cin], Dense [in,out].  Activations NCHW inside this file; Flatten/Reshape follow the Keras NHWC order.
"""
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .torch_ref import conv2d_same, conv2d_transpose_same, data_loss, BN_EPS, BN_MOMENTUM, L2_COEF, VOCAB, EMB_DIM
from .torch_resae import init_from_shapes

DROPOUT_P = 0.3      # Dropout(.3) (dl_models/autoencoder.py:255, :368)


@dataclass
class AEConfig:
    """Autoencoder.__init__ arguments (dl_models/autoencoder.py:41-46); main_training.py:120-129 uses filters
    (64,128,256,512), kernels 3, strides 2, latent 64, n_neurons 2048."""
    H: int
    W: int
    conv_filters: tuple = (64, 128, 256, 512)
    conv_kernels: tuple = (3, 3, 3, 3)
    conv_strides: tuple = (2, 2, 2, 2)
    latent_space_dim: int = 64
    n_neurons: int = 2048
    inf_vector_shape: tuple = (2, 16)

    def bottleneck_shape(self):
        h, w = self.H, self.W
        for s in self.conv_strides:
            h, w = -(-h // s), -(-w // s)
        return h, w, self.conv_filters[-1]


def param_shapes(cfg: AEConfig) -> Dict[str, tuple]:
    """Trainable variables in creation order: encoder (dl_models/autoencoder.py:337-417), decoder (:222-335)."""
    shapes = {}
    n = len(cfg.conv_filters)
    cin = 2
    for i in range(n):                                                         # _add_conv_layer (:384-402)
        f, k = cfg.conv_filters[i], cfg.conv_kernels[i]
        base = f"encoder_conv_layer_{i + 1}"
        shapes[base + ".kernel"] = (k, k, cin, f)
        shapes[base + ".bias"] = (f,)
        shapes[f"encoder_bn_{i + 1}.gamma"] = (f,)
        shapes[f"encoder_bn_{i + 1}.beta"] = (f,)
        cin = f
    h, w, c = cfg.bottleneck_shape()
    n_in = int(np.prod(cfg.inf_vector_shape)) * EMB_DIM
    shapes["embedding"] = (VOCAB, EMB_DIM)                                     # _add_dense_to_inf (:357-369)
    shapes["encoder_inf_dense.kernel"] = (n_in, cfg.n_neurons)
    shapes["encoder_inf_dense.bias"] = (cfg.n_neurons,)
    shapes["encoder_output.kernel"] = (h * w * c + cfg.n_neurons, cfg.latent_space_dim)      # _add_bottleneck (:404-417)
    shapes["encoder_output.bias"] = (cfg.latent_space_dim,)
    shapes["decoder_dense.kernel"] = (cfg.latent_space_dim, h * w * c)         # _add_dense_layer (:245-256)
    shapes["decoder_dense.bias"] = (h * w * c,)
    f, k = cfg.conv_filters[-1], cfg.conv_kernels[-1]                          # _add_first_conv (:267-285): stride 1
    shapes["decoder_conv_transpose_layer_0.kernel"] = (k, k, f, c)
    shapes["decoder_conv_transpose_layer_0.bias"] = (f,)
    shapes["decoder_bn_0.gamma"] = (f,)
    shapes["decoder_bn_0.beta"] = (f,)
    cin = f
    for layer_index in reversed(range(1, n)):                                  # _add_conv_transpose_layer (:300-320)
        num = n - layer_index
        f, k = cfg.conv_filters[layer_index - 1], cfg.conv_kernels[layer_index - 1]
        shapes[f"decoder_conv_transpose_layer_{num}.kernel"] = (k, k, f, cin)
        shapes[f"decoder_conv_transpose_layer_{num}.bias"] = (f,)
        shapes[f"decoder_bn_{num}.gamma"] = (f,)
        shapes[f"decoder_bn_{num}.beta"] = (f,)
        cin = f
    k0 = cfg.conv_kernels[0]
    shapes[f"decoder_out_{n}.kernel"] = (k0, k0, 2, cin)                       # _add_decoder_output (:322-335)
    shapes[f"decoder_out_{n}.bias"] = (2,)
    return shapes


def l2_regularized(cfg: AEConfig):
    """kernel_regularizer=l2(0.001) on every encoder Conv2D and every decoder Conv2DTranspose except the output layer."""
    return [n for n in param_shapes(cfg) if n.endswith(".kernel") and ("conv_layer" in n or "conv_transpose_layer" in n)]


def init_params(cfg: AEConfig, seed_name="ap", randomize_all=False, dtype=np.float32):
    """Keras default initialisers; values from detrand (platform independent)."""
    return init_from_shapes(param_shapes(cfg), seed_name, randomize_all, dtype)


def _bn_relu(x, P, base):
    y = F.batch_norm(x, None, None, P[base + ".gamma"], P[base + ".beta"], training=True, momentum=1 - BN_MOMENTUM, eps=BN_EPS)
    return F.relu(y)


def forward(P, spec, emb, cfg: AEConfig, mask_inf: Optional[torch.Tensor] = None, mask_dec: Optional[torch.Tensor] = None,
            inter=None):
    """Autoencoder.model([spec, emb]) in training mode.  spec [B,2,H,W] NCHW, emb int [B,2,16].
    mask_inf [B, n_neurons] / mask_dec [B, h*w*c]: dropout keep masks already scaled by 1/(1-p) (None = no dropout)."""
    n = len(cfg.conv_filters)
    B = spec.shape[0]
    x = spec
    for i in range(n):
        x = conv2d_same(x, P[f"encoder_conv_layer_{i + 1}.kernel"], P[f"encoder_conv_layer_{i + 1}.bias"], cfg.conv_strides[i])
        x = _bn_relu(x, P, f"encoder_bn_{i + 1}")
    h, w, c = cfg.bottleneck_shape()
    flat = x.permute(0, 2, 3, 1).reshape(B, -1)                               # Flatten of the NHWC tensor (:413)
    vec = P["embedding"][emb.long()].reshape(B, -1) @ P["encoder_inf_dense.kernel"] + P["encoder_inf_dense.bias"]
    if mask_inf is not None:
        vec = vec * mask_inf                                                  # Dropout(.3) (:368)
    z = torch.cat([flat, vec], dim=1) @ P["encoder_output.kernel"] + P["encoder_output.bias"]
    if inter is not None:
        inter["latent"] = z
    d = z @ P["decoder_dense.kernel"] + P["decoder_dense.bias"]
    if mask_dec is not None:
        d = d * mask_dec                                                      # Dropout(.3) (:255)
    x = d.view(B, h, w, c).permute(0, 3, 1, 2)
    x = _bn_relu(conv2d_transpose_same(x, P["decoder_conv_transpose_layer_0.kernel"], P["decoder_conv_transpose_layer_0.bias"], 1),
                 P, "decoder_bn_0")
    for layer_index in reversed(range(1, n)):
        num = n - layer_index
        x = conv2d_transpose_same(x, P[f"decoder_conv_transpose_layer_{num}.kernel"], P[f"decoder_conv_transpose_layer_{num}.bias"],
                                  cfg.conv_strides[layer_index - 1])
        x = _bn_relu(x, P, f"decoder_bn_{num}")
    x = conv2d_transpose_same(x, P[f"decoder_out_{n}.kernel"], P[f"decoder_out_{n}.bias"], cfg.conv_strides[0])
    if inter is not None:
        inter["logits"] = x
    return torch.sigmoid(x)


def reg_loss(P, cfg: AEConfig, n_replicas=1):
    tot = 0.0
    for n in l2_regularized(cfg):
        tot = tot + L2_COEF * (P[n] ** 2).sum()
    return tot / n_replicas


def loss_and_grads(params, spec_in, emb, spec_out, cfg: AEConfig, alpha=0.9, global_batch=None, n_replicas=1,
                   mask_inf=None, mask_dec=None, dtype=torch.float64, inter=None):
    P = {k: torch.tensor(np.asarray(v), dtype=dtype).requires_grad_(True) for k, v in params.items()}
    t = lambda a: None if a is None else torch.as_tensor(np.asarray(a)).to(dtype)
    pred = forward(P, t(spec_in), torch.as_tensor(np.asarray(emb)), cfg, t(mask_inf), t(mask_dec), inter)
    dl = data_loss(t(spec_out), pred, alpha, global_batch)
    loss = dl + reg_loss(P, cfg, n_replicas)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach() for k, v in P.items()}
    return float(loss.detach()), float(dl.detach()), pred.detach(), grads
