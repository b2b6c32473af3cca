"""torch.nn.functional CPU restatement of the reference ResAE graph (dl_models/res_ae.py), BASELINE.json configs[4].

Oracle / test infrastructure only (see oracle/__init__.py; PARITY UNPINNED: the TensorFlow/Keras reference cannot run
here and has no tests or fixtures).  Parameters are held in Keras layouts: Conv2D [kh,kw,Cin,Cout], Conv2DTranspose
[kh,kw,Cout,Cin], Dense [in,out].  Activations NCHW inside this file; Flatten/Reshape follow the Keras NHWC order.
"""
import math
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import detrand
from .torch_ref import conv2d_same, conv2d_transpose_same, data_loss, BN_EPS, BN_MOMENTUM, L2_COEF, VOCAB, EMB_DIM

LEAKY = 0.3          # keras LeakyReLU() default alpha (dl_models/res_ae.py:470)
DROPOUT_P = 0.3      # Dropout(.3) (dl_models/res_ae.py:256, :528)


@dataclass
class ResAEConfig:
    """ResAE.__init__ arguments (dl_models/res_ae.py:41-50); main_training.py:132-141 uses filters (32,64,128,256),
    kernels 3, strides 2, latent 32, n_neurons 1024."""
    H: int
    W: int
    conv_filters: tuple = (32, 64, 128, 256)
    conv_kernels: tuple = (3, 3, 3, 3)
    conv_strides: tuple = (2, 2, 2, 2)
    latent_space_dim: int = 32
    n_neurons: int = 1024
    inf_vector_shape: tuple = (2, 16)

    def bottleneck_shape(self):
        h, w = self.H, self.W
        for s in self.conv_strides:
            h, w = -(-h // s), -(-w // s)
        return h, w, self.conv_filters[-1]


def _block_params(shapes, name, cin, f, k, transpose, with_skip):
    def kern(kk, ci, co):
        return (kk, kk, co, ci) if transpose else (kk, kk, ci, co)
    tag = "conv" if with_skip else "id"
    for idx, (kk, ci) in enumerate(((1, cin), (k, f), (1, f)), start=1):
        base = f"{name}_{tag}.{idx}"
        shapes[base + ".kernel"] = kern(kk, ci, f)
        shapes[base + ".bias"] = (f,)
        shapes[base + ".gamma"] = (f,)
        shapes[base + ".beta"] = (f,)
    if with_skip:
        base = f"{name}_conv.s"
        shapes[base + ".kernel"] = kern(1, cin, f)
        shapes[base + ".bias"] = (f,)
        shapes[base + ".gamma"] = (f,)
        shapes[base + ".beta"] = (f,)


def param_shapes(cfg: ResAEConfig) -> Dict[str, tuple]:
    """Trainable variables in creation order (encoder dl_models/res_ae.py:391-451, decoder :230-308)."""
    shapes = {}
    n = len(cfg.conv_filters)
    cin = 2
    for i in range(n):
        f, k = cfg.conv_filters[i], cfg.conv_kernels[i]
        _block_params(shapes, f"e_res_{i + 1}", cin, f, k, False, True)       # res_conv  (:482-514)
        _block_params(shapes, f"e_res_{i + 1}", f, f, k, False, False)        # res_identity (:453-480)
        cin = f
    h, w, c = cfg.bottleneck_shape()
    n_in = int(np.prod(cfg.inf_vector_shape)) * EMB_DIM
    shapes["embedding"] = (VOCAB, EMB_DIM)
    shapes["e_dense_vector.kernel"] = (n_in, cfg.n_neurons)
    shapes["e_dense_vector.bias"] = (cfg.n_neurons,)
    shapes["e_out.kernel"] = (h * w * c + cfg.n_neurons, cfg.latent_space_dim)
    shapes["e_out.bias"] = (cfg.latent_space_dim,)
    shapes["decoder_dense.kernel"] = (cfg.latent_space_dim, h * w * c)
    shapes["decoder_dense.bias"] = (h * w * c,)
    f, k = cfg.conv_filters[-1], cfg.conv_kernels[-1]
    _block_params(shapes, "d_res_0", f, f, k, True, True)                      # _add_first_conv (:259-270)
    _block_params(shapes, "d_res_0", f, f, k, True, False)
    cin = f
    for layer_index in reversed(range(1, n)):                                  # _add_conv_transpose_layers (:272-308)
        name = f"d_res_{n - layer_index}"
        f, k = cfg.conv_filters[layer_index - 1], cfg.conv_kernels[layer_index]
        _block_params(shapes, name, cin, f, k, True, True)
        _block_params(shapes, name, f, f, k, True, False)
        cin = f
    k0 = cfg.conv_kernels[0]
    shapes["d_out.kernel"] = (k0, k0, 2, cin)                                  # _add_decoder_output (:373-389)
    shapes["d_out.bias"] = (2,)
    return shapes


def l2_regularized(cfg: ResAEConfig):
    """kernel_regularizer=l2(0.001) sits on every conv of the residual blocks (not on Dense, not on the output layer)."""
    return [n for n in param_shapes(cfg) if n.endswith(".kernel") and ("_conv." in n or "_id." in n)]


def init_params(cfg: ResAEConfig, seed_name="rp", randomize_all=False, dtype=np.float32):
    return init_from_shapes(param_shapes(cfg), seed_name, randomize_all, dtype)


def init_from_shapes(shapes, seed_name, randomize_all=False, dtype=np.float32):
    """Keras default initialisers for a name -> Keras-shape table (glorot_uniform kernels, zero biases, gamma 1, beta 0,
    Embedding U(-0.05, 0.05)); `randomize_all` perturbs biases / gamma / beta so parity tests exercise them."""
    out = {}
    for name, shp in shapes.items():
        key = f"{seed_name}/{name}"
        if name == "embedding":
            a = detrand.uniform(key, shp, -0.05, 0.05)
        elif name.endswith(".kernel"):
            if len(shp) == 4:
                rf = shp[0] * shp[1]
                fan_in, fan_out = shp[2] * rf, shp[3] * rf
            else:
                fan_in, fan_out = shp
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            a = detrand.uniform(key, shp, -lim, lim)
        elif name.endswith(".gamma"):
            a = detrand.uniform(key, shp, 0.5, 1.5) if randomize_all else np.ones(shp, np.float32)
        else:
            a = detrand.uniform(key, shp, -0.2, 0.2) if randomize_all else np.zeros(shp, np.float32)
        out[name] = a.astype(dtype)
    return out


def _bn(x, P, base):
    return F.batch_norm(x, None, None, P[base + ".gamma"], P[base + ".beta"], training=True, momentum=1 - BN_MOMENTUM, eps=BN_EPS)


def _conv(x, P, base, stride, transpose):
    if transpose:
        return conv2d_transpose_same(x, P[base + ".kernel"], P[base + ".bias"], stride)
    return conv2d_same(x, P[base + ".kernel"], P[base + ".bias"], stride)


def res_block(x, P, name, stride, transpose, with_skip, inter=None):
    """res_conv / res_identity (dl_models/res_ae.py:482-514, :453-480) and their Conv2DTranspose twins (:339-371, :310-337)."""
    tag = "conv" if with_skip else "id"
    y = F.leaky_relu(_bn(_conv(x, P, f"{name}_{tag}.1", stride if with_skip else 1, transpose), P, f"{name}_{tag}.1"), LEAKY)
    y = F.leaky_relu(_bn(_conv(y, P, f"{name}_{tag}.2", 1, transpose), P, f"{name}_{tag}.2"), LEAKY)
    y = _bn(_conv(y, P, f"{name}_{tag}.3", 1, transpose), P, f"{name}_{tag}.3")
    if with_skip:
        skip = _bn(_conv(x, P, f"{name}_conv.s", stride, transpose), P, f"{name}_conv.s")
    else:
        skip = x
    out = F.leaky_relu(y + skip, LEAKY)
    if inter is not None:
        inter[f"{name}_{tag}.out"] = out
    return out


def forward(P, spec, emb, cfg: ResAEConfig, mask_latent: Optional[torch.Tensor] = None,
            mask_dec: Optional[torch.Tensor] = None, inter=None):
    """ResAE.model([spec, emb]) in training mode.  spec [B,2,H,W] NCHW, emb int [B,2,16].
    mask_latent [B, latent] / mask_dec [B, h*w*c]: dropout keep masks already scaled by 1/(1-p) (None = no dropout)."""
    n = len(cfg.conv_filters)
    B = spec.shape[0]
    x = spec
    for i in range(n):
        x = res_block(x, P, f"e_res_{i + 1}", cfg.conv_strides[i], False, True, inter)
        x = res_block(x, P, f"e_res_{i + 1}", 1, False, False, inter)
    h, w, c = cfg.bottleneck_shape()
    flat = x.permute(0, 2, 3, 1).reshape(B, -1)                       # Flatten of the NHWC tensor (:527)
    vec = P["embedding"][emb.long()].reshape(B, -1) @ P["e_dense_vector.kernel"] + P["e_dense_vector.bias"]   # :411-422
    z = torch.cat([flat, vec], dim=1) @ P["e_out.kernel"] + P["e_out.bias"]                                    # :529-530
    if mask_latent is not None:
        z = z * mask_latent
    if inter is not None:
        inter["latent"] = z
    d = z @ P["decoder_dense.kernel"] + P["decoder_dense.bias"]       # :247-257
    if mask_dec is not None:
        d = d * mask_dec
    x = d.view(B, h, w, c).permute(0, 3, 1, 2)                        # Reshape(shape_before_bottleneck) is NHWC
    x = res_block(x, P, "d_res_0", 1, True, True, inter)
    x = res_block(x, P, "d_res_0", 1, True, False, inter)
    for layer_index in reversed(range(1, n)):
        name = f"d_res_{n - layer_index}"
        x = res_block(x, P, name, cfg.conv_strides[layer_index - 1], True, True, inter)
        x = res_block(x, P, name, 1, True, False, inter)
    x = conv2d_transpose_same(x, P["d_out.kernel"], P["d_out.bias"], cfg.conv_strides[0])
    if inter is not None:
        inter["logits"] = x
    return torch.sigmoid(x)


def reg_loss(P, cfg: ResAEConfig, n_replicas=1):
    tot = 0.0
    for n in l2_regularized(cfg):
        tot = tot + L2_COEF * (P[n] ** 2).sum()
    return tot / n_replicas


def loss_and_grads(params, spec_in, emb, spec_out, cfg: ResAEConfig, alpha=0.9, global_batch=None, n_replicas=1,
                   mask_latent=None, mask_dec=None, dtype=torch.float64, inter=None):
    P = {k: torch.tensor(np.asarray(v), dtype=dtype).requires_grad_(True) for k, v in params.items()}
    t = lambda a: None if a is None else torch.as_tensor(np.asarray(a)).to(dtype)
    pred = forward(P, t(spec_in), torch.as_tensor(np.asarray(emb)), cfg, t(mask_latent), t(mask_dec), inter)
    dl = data_loss(t(spec_out), pred, alpha, global_batch)
    loss = dl + reg_loss(P, cfg, n_replicas)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach() for k, v in P.items()}
    return float(loss.detach()), float(dl.detach()), pred.detach(), grads
