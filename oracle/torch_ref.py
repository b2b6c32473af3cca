"""torch.nn.functional CPU restatement of the reference U-Net graph, loss and Adam step.

Oracle / test infrastructure only (see oracle/__init__.py; PARITY UNPINNED: the
TensorFlow/Keras reference cannot run here and has no tests or fixtures).

Parameters are held in the *reference's own layouts* (what Keras would save):
  Conv2D kernel          [kh, kw, Cin, Cout]   (HWIO)
  Conv2DTranspose kernel [kh, kw, Cout, Cin]   (HWOI)
  Dense kernel           [in, out]
  Embedding table        [2000, 256]
Activations are NCHW inside this file (the north_star's boundary layout); the
mathematics is the Keras NHWC graph of dl_models/u_net.py:201-251.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import detrand

BN_EPS = 1e-3        # keras BatchNormalization default epsilon (dl_models/u_net.py:368)
BN_MOMENTUM = 0.99   # keras default momentum
L2_COEF = 1e-3       # kernel_regularizer=l2(0.001) (dl_models/u_net.py:274, :302)
VOCAB, EMB_DIM = 2000, 256   # Embedding(2000, 256) (dl_models/u_net.py:257)
VEC_CH = 16          # shape[2] = 16 (dl_models/u_net.py:255)
DROPOUT_P = 0.3      # Dropout(.3) (dl_models/u_net.py:260)


@dataclass
class Config:
    """Hyper-parameters of UNet.__init__ (dl_models/u_net.py:40-49) that shape the graph."""
    H: int
    W: int
    F0: int = 32                 # number_filters_0
    k: int = 3                   # kernels
    depth: int = 4               # number of stride-2 levels (hard-coded 4 in the reference, :213-243)
    batchnorm: bool = True
    inf_vector_shape: tuple = (2, 16)
    s0: int = 1                  # resize_factor_0[0]
    s: int = 2                   # res_factor[0]
    mode: int = 0                # feature block: 0 convolutional_block_1, 1 convolutional_block_2, 2 residual_block_1,
                                 # 3 residual_block_2 (dl_models/u_net.py:280-287, :312-319)

    def enc_channels(self) -> List[int]:
        return [self.F0 * (2 ** l) for l in range(self.depth + 1)]

    def bottleneck_hw(self):
        h, w = self.H, self.W
        h, w = -(-h // self.s0), -(-w // self.s0)
        for _ in range(self.depth):
            h, w = -(-h // self.s), -(-w // self.s)
        return h, w


def same_pads(n_in, k, s):
    n_out = -(-n_in // s)
    total = max((n_out - 1) * s + k - n_in, 0)
    return n_out, total // 2, total - total // 2


# --------------------------------------------------------------------------- parameters

def _feature_block_names(cfg, base):
    """Conv-BN-ReLU units of the per-level feature block (dl_models/u_net.py:324-386): mode 0 keeps the historical names."""
    if cfg.mode == 0:
        return [base + (".cb1" if base.startswith("enc") else ".cb1b")]
    return [base + ".fb.c1", base + ".fb.c2"] + ([base + ".fb.c3"] if cfg.mode == 3 else [])


def param_shapes(cfg: Config) -> Dict[str, tuple]:
    """Trainable variables in the order Keras creates them in UNet._build
    (dl_models/u_net.py:201-251).  Names are this repo's; shapes are Keras'."""
    shapes = {}
    ch = cfg.enc_channels()
    cin = 2

    def unit(name, c):
        shapes[name + ".kernel"] = (3, 3, c, c)
        shapes[name + ".bias"] = (c,)
        if cfg.batchnorm:
            shapes[name + ".gamma"] = (c,)
            shapes[name + ".beta"] = (c,)
    for l, c in enumerate(ch, start=1):
        shapes[f"enc{l}.down.kernel"] = (cfg.k, cfg.k, cin, c)
        shapes[f"enc{l}.down.bias"] = (c,)
        for u in _feature_block_names(cfg, f"enc{l}"):
            unit(u, c)
        cin = c
    h5, w5 = cfg.bottleneck_hw()
    n_in = int(np.prod(cfg.inf_vector_shape)) * EMB_DIM
    shapes["vec.embedding"] = (VOCAB, EMB_DIM)
    shapes["vec.dense.kernel"] = (n_in, h5 * w5 * VEC_CH)
    shapes["vec.dense.bias"] = (h5 * w5 * VEC_CH,)
    shapes["vec.conv.kernel"] = (1, 1, VEC_CH, ch[-1])
    shapes["vec.conv.bias"] = (ch[-1],)
    for l in range(cfg.depth, 0, -1):
        c = ch[l - 1]
        shapes[f"dec{l}.up.kernel"] = (cfg.k, cfg.k, c, ch[l])      # HWOI
        shapes[f"dec{l}.up.bias"] = (c,)
        shapes[f"dec{l}.cb1a.kernel"] = (cfg.k, cfg.k, 2 * c, c)
        shapes[f"dec{l}.cb1a.bias"] = (c,)
        if cfg.batchnorm:
            shapes[f"dec{l}.cb1a.gamma"] = (c,)
            shapes[f"dec{l}.cb1a.beta"] = (c,)
        for u in _feature_block_names(cfg, f"dec{l}"):
            unit(u, c)
    shapes["head.kernel"] = (6, 6, ch[0], 2)
    shapes["head.bias"] = (2,)
    return shapes


def l2_regularized(cfg: Config) -> List[str]:
    """Only the strided Conv2D and the Conv2DTranspose kernels carry l2(0.001)
    (dl_models/u_net.py:274, :302)."""
    names = [f"enc{l}.down.kernel" for l in range(1, cfg.depth + 2)]
    names += [f"dec{l}.up.kernel" for l in range(cfg.depth, 0, -1)]
    return names


def init_params(cfg: Config, seed_name="p", randomize_all=False, dtype=np.float32) -> Dict[str, np.ndarray]:
    """Keras default initialisers (no initialiser argument appears anywhere in
    dl_models/u_net.py): glorot_uniform kernels, zero biases, gamma=1, beta=0,
    Embedding U(-0.05, 0.05).  ``randomize_all`` perturbs biases / gamma / beta
    so parity tests exercise them.  Values come from detrand (platform independent)."""
    out = {}
    for name, shp in param_shapes(cfg).items():
        key = f"{seed_name}/{name}"
        if name == "vec.embedding":
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
        elif name.endswith(".beta") or name.endswith(".bias"):
            a = detrand.uniform(key, shp, -0.2, 0.2) if randomize_all else np.zeros(shp, np.float32)
        else:
            raise KeyError(name)
        out[name] = a.astype(dtype)
    return out


def synthetic_batch(cfg: Config, B: int, seed_name="d"):
    """Inputs with the contract of DataGenerator.__getitem__ (datageneratorv2.py:64-102):
    spec_in/spec_out f32 [B,2,H,W] (NCHW here), amp and phase in [0,1)
    (Normalizer.normalize, preprocess.py:26-32) with the zero-padded border the
    TensorPadder adds (129/144 rows, 151/160 cols: dataset.py:70), emb int [B,2,16]
    in [26, 1282) (range of rooms.py:96-99 over the whole UTS set)."""
    spec_in = detrand.uniform(f"{seed_name}/spec_in", (B, 2, cfg.H, cfg.W))
    spec_out = detrand.uniform(f"{seed_name}/spec_out", (B, 2, cfg.H, cfg.W))
    r0, c0 = math.ceil(0.896 * cfg.H), math.ceil(0.944 * cfg.W)
    for a in (spec_in, spec_out):
        a[:, :, r0:, :] = 0.0
        a[:, :, :, c0:] = 0.0
    emb = detrand.randint(f"{seed_name}/emb", (B,) + tuple(cfg.inf_vector_shape), 26, 1282)
    return spec_in, emb, spec_out


# --------------------------------------------------------------------------- ops (NCHW)

def conv2d_same(x, w_hwio, b, stride=1):
    """Conv2D padding='same' (dl_models/u_net.py:269-276, :366, :248, :262); TF pads
    asymmetrically (extra on bottom/right)."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    _, pt, pb = same_pads(x.shape[2], kh, stride)
    _, pl, pr = same_pads(x.shape[3], kw, stride)
    x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w_hwio.permute(3, 2, 0, 1), b, stride=stride)


def conv2d_transpose_same(x, w_hwoi, b, stride=2):
    """Conv2DTranspose strides=2 padding='same' (dl_models/u_net.py:297-304): full
    scatter then crop [pad_before : pad_before + 2n], pad_before = (k - s)//2."""
    kh, kw = w_hwoi.shape[0], w_hwoi.shape[1]
    full = F.conv_transpose2d(x, w_hwoi.permute(3, 2, 0, 1), None, stride=stride)
    pt, pl = max(kh - stride, 0) // 2, max(kw - stride, 0) // 2
    Ho, Wo = x.shape[2] * stride, x.shape[3] * stride
    # k < s (the 1x1 stride-2 'valid' layers of ResAE, dl_models/res_ae.py:357): the scatter is shorter than n*s, the
    # remaining positions receive the bias only
    eh, ew = max(pt + Ho - full.shape[2], 0), max(pl + Wo - full.shape[3], 0)
    if eh or ew:
        full = F.pad(full, (0, ew, 0, eh))
    y = full[:, :, pt:pt + Ho, pl:pl + Wo]
    return y + b.view(1, -1, 1, 1)


def bn_relu(x, gamma, beta, state, name, training, relu=True):
    """BatchNormalization (training: biased batch stats, eps 1e-3) + ReLU
    (dl_models/u_net.py:367-369).  Moving statistics: moving = 0.99*moving + 0.01*batch,
    batch variance Bessel-corrected as TF's fused kernel does (documented choice,
    SURVEY.md appendix A.1; does not influence the train step)."""
    if state is None:
        rm = rv = None
    else:
        rm, rv = state[name + ".moving_mean"], state[name + ".moving_variance"]
    y = F.batch_norm(x, rm, rv, gamma, beta, training=training or rm is None,
                     momentum=1.0 - BN_MOMENTUM, eps=BN_EPS)
    return F.relu(y) if relu else y


class _QuantSTE(torch.autograd.Function):
    """Models a tensor STORED in bfloat16: the value is rounded to bf16 on the way forward and its gradient is
    rounded to bf16 on the way back (the product keeps both the activation and its gradient in bf16 buffers)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _QuantFwd(torch.autograd.Function):
    """bf16 work copy of an fp32 master weight: rounded forward, gradient passed through (weight gradients are fp32)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _QuantBwd(torch.autograd.Function):
    """fp32 tensor whose gradient is stored in bf16 (the logits)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _ident(x):
    return x


def conv_block_1(x, P, name, cfg, state, training, inter, q=_ident, qw=_ident):
    """UNet.convolutional_block_1 (dl_models/u_net.py:363-371)."""
    y = q(conv2d_same(x, qw(P[name + ".kernel"]), P[name + ".bias"], 1))
    if inter is not None:
        inter[name + ".conv"] = y
    if cfg.batchnorm:
        y = bn_relu(y, P[name + ".gamma"], P[name + ".beta"], state, name, training, relu=False)
    if inter is not None:
        inter[name + ".pre"] = y                         # the ReLU input (tests: distance of the nearest one from zero)
    y = q(F.relu(y))
    if inter is not None:
        inter[name + ".out"] = y
    return y


def feature_block(x, P, base, cfg, state, training, inter, q, qw):
    """mode 0: convolutional_block_1; 1: convolutional_block_2 (dl_models/u_net.py:373-386); 2: residual_block_1
    (:324-339, two units + identity Add, no activation after the Add); 3: residual_block_2 (:341-361, third unit on the skip)."""
    names = _feature_block_names(cfg, base)
    if cfg.mode == 0:
        return conv_block_1(x, P, names[0], cfg, state, training, inter, q, qw)
    a = conv_block_1(x, P, names[0], cfg, state, training, inter, q, qw)
    b = conv_block_1(a, P, names[1], cfg, state, training, inter, q, qw)
    if cfg.mode == 1:
        return b
    if cfg.mode == 2:
        return q(b + x)
    c = conv_block_1(x, P, names[2], cfg, state, training, inter, q, qw)
    return q(b + c)


def forward(P: Dict[str, torch.Tensor], spec, emb, cfg: Config, training=True,
            dropout_mask: Optional[torch.Tensor] = None, bn_state=None, inter=None, storage=None):
    """UNet._build graph (dl_models/u_net.py:201-251), mode 0.
    spec [B,2,H,W] NCHW, emb int [B,2,16] -> [B,2,H,W] in (0,1).
    dropout_mask: [B, h5*w5*16] keep mask already scaled by 1/(1-p) (None = no dropout).
    storage="bf16" restates the product's bf16 mode: every trunk activation and its gradient are rounded to bf16
    where the product stores them, trunk kernels are used as bf16 work copies; arithmetic stays in the tensor dtype."""
    q = _QuantSTE.apply if storage == "bf16" else _ident
    qw = _QuantFwd.apply if storage == "bf16" else _ident
    x = q(spec)
    skips = []
    n_levels = cfg.depth + 1
    for l in range(1, n_levels + 1):
        stride = cfg.s0 if l == 1 else cfg.s
        # encoding_block (dl_models/u_net.py:265-289): strided conv (bias, no BN/act) + block
        x = q(conv2d_same(x, qw(P[f"enc{l}.down.kernel"]), P[f"enc{l}.down.bias"], stride))
        if inter is not None:
            inter[f"enc{l}.down"] = x
        x = feature_block(x, P, f"enc{l}", cfg, bn_state, training, inter, q, qw)
        skips.append(x)
    # vector_block (dl_models/u_net.py:253-263)
    B = spec.shape[0]
    h5, w5 = x.shape[2], x.shape[3]
    f = P["vec.embedding"][emb.long()]                   # [B,2,16,256]
    flat = f.reshape(B, -1)                              # Flatten: row-major (2,16,256)
    v = flat @ P["vec.dense.kernel"] + P["vec.dense.bias"]
    if dropout_mask is not None:
        v = v * dropout_mask
    if inter is not None:
        inter["vec.dense"] = v
    v = v.view(B, h5, w5, VEC_CH).permute(0, 3, 1, 2)    # Reshape((h5,w5,16)) is NHWC
    v = conv2d_same(v, P["vec.conv.kernel"], P["vec.conv.bias"], 1)
    x = q(x + v)                                         # Add() (dl_models/u_net.py:229); the vector branch stays fp32
    if inter is not None:
        inter["bottleneck"] = x
    # decoding_block (dl_models/u_net.py:291-321)
    for l in range(cfg.depth, 0, -1):
        x = q(conv2d_transpose_same(x, qw(P[f"dec{l}.up.kernel"]), P[f"dec{l}.up.bias"], cfg.s))
        if inter is not None:
            inter[f"dec{l}.up"] = x
        x = q(torch.cat([skips[l - 1], x], dim=1))       # concatenate([skip, x]) (:308)
        x = conv_block_1(x, P, f"dec{l}.cb1a", cfg, bn_state, training, inter, q, qw)
        x = feature_block(x, P, f"dec{l}", cfg, bn_state, training, inter, q, qw)
    # UpSampling2D((1,1)) is the identity; Conv2D(2,(6,6),'same') + sigmoid (:247-249)
    x = conv2d_same(x, qw(P["head.kernel"]), P["head.bias"], 1)  # logits stay fp32 in every mode (bf16 mode: bf16 kernel copy)
    if storage == "bf16":
        x = _QuantBwd.apply(x)                                   # ... but dL/dlogits is stored in bf16
    if inter is not None:
        inter["head.logits"] = x
    return torch.sigmoid(x)


def sigmoid_weight(beta, W, dtype=torch.float64):
    """preprocess.sigmoid (preprocess.py:116-121): the column weights of the `sigmoid_loss` switch - z = 1/(1+exp(-(x+5) beta)) on
    x = linspace(-10, 10, W), flipped (every row of the [H, W] mask is this vector)."""
    x = torch.linspace(-10.0, 10.0, W, dtype=torch.float64)
    return torch.flip(1.0 / (1.0 + torch.exp(-(x + 5.0) * beta)), dims=(0,)).to(dtype)


def data_loss(y_true, y_pred, alpha=0.9, global_batch=None, x_in=None, phase_weight=None):
    """compute_loss without the regulariser (main_training.py:203-231), NCHW.
    x_in (the `diff_loss` switch, :214-217): the network input; the phase target becomes phase_true - phase_x.
    phase_weight (the `sigmoid_loss` switch, :221-222): [W] column weights multiplied into the phase term (sigmoid_weight)."""
    B, _, H, W = y_true.shape
    gb = B if global_batch is None else global_batch
    e_amp = (y_true[:, 0] - y_pred[:, 0]) ** 2
    pt = y_true[:, 1] if x_in is None else y_true[:, 1] - x_in[:, 1]
    yt = pt * 2 * math.pi - math.pi
    yp = y_pred[:, 1] * 2 * math.pi - math.pi
    ph = torch.remainder((yt - yp) + math.pi, 2 * math.pi) - math.pi   # phase_loss :184-190
    e_ph = 1.0 - torch.cos(ph)
    if phase_weight is not None:
        e_ph = e_ph * phase_weight.to(e_ph.dtype).view(1, 1, W)
    per = alpha * e_amp + (1.0 - alpha) * e_ph
    return per.sum() / (H * W * 2) / gb


def reg_loss(P, cfg: Config, n_replicas=1):
    """sum(model.losses) scaled by 1/replicas (main_training.py:232-233)."""
    tot = 0.0
    for n in l2_regularized(cfg):
        tot = tot + L2_COEF * (P[n] ** 2).sum()
    return tot / n_replicas


def to_torch(params: Dict[str, np.ndarray], dtype=torch.float32, requires_grad=False):
    return {k: torch.tensor(np.asarray(v), dtype=dtype).requires_grad_(requires_grad)
            for k, v in params.items()}


def loss_and_grads(params, spec_in, emb, spec_out, cfg: Config, alpha=0.9, global_batch=None,
                   n_replicas=1, dropout_mask=None, dtype=torch.float32, inter=None, bn_state=None, storage=None):
    """One forward + backward of train_step (main_training.py:253-268) up to the gradients."""
    P = to_torch(params, dtype, True)
    spec_in = torch.as_tensor(np.asarray(spec_in)).to(dtype)
    spec_out = torch.as_tensor(np.asarray(spec_out)).to(dtype)
    emb = torch.as_tensor(np.asarray(emb))
    if dropout_mask is not None:
        dropout_mask = torch.as_tensor(np.asarray(dropout_mask)).to(dtype)
    pred = forward(P, spec_in, emb, cfg, True, dropout_mask, bn_state, inter, storage)
    dl = data_loss(spec_out, pred, alpha, global_batch)
    loss = dl + reg_loss(P, cfg, n_replicas)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach() for k, v in P.items()}
    return float(loss.detach()), float(dl.detach()), pred.detach(), grads


def adam_update(theta, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-7):
    """Keras Adam (main_training.py:168-169): epsilon outside the sqrt, bias
    correction folded into lr_t."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return theta - lr_t * m / (v.sqrt() + eps), m, v


def sgd_update(theta, g, lr):
    """tf.keras.optimizers.SGD(learning_rate=lr) without momentum (main_training.py:166-167)."""
    return theta - lr * g


def nadam_update(theta, g, m, v, t, lr, m_schedule, b1=0.9, b2=0.999, eps=1e-7):
    """tf.keras.optimizers.Nadam (main_training.py:164-165) as TF 2.x's optimizer_v2 implements it (Dozat's Nesterov Adam with the
    momentum schedule mu_t = beta_1 (1 - 0.5 * 0.96^(0.004 t)); m_schedule is the running product of the mu_t, 1.0 before the first
    step).  Restated from the published algorithm: PARITY UNPINNED like the rest of the oracle."""
    mu_t = b1 * (1.0 - 0.5 * 0.96 ** (0.004 * t))
    mu_t1 = b1 * (1.0 - 0.5 * 0.96 ** (0.004 * (t + 1)))
    ms_new = m_schedule * mu_t
    ms_next = ms_new * mu_t1
    g_prime = g / (1.0 - ms_new)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    m_prime = m / (1.0 - ms_next)
    v_prime = v / (1.0 - b2 ** t)
    m_bar = (1.0 - mu_t) * g_prime + mu_t1 * m_prime
    return theta - lr * m_bar / (v_prime.sqrt() + eps), m, v, ms_new


class TrainState:
    """Holds parameters + Adam slots for repeated oracle train steps (CPU baseline)."""

    def __init__(self, cfg: Config, params, lr=5e-7, dtype=torch.float32):
        self.cfg, self.lr, self.dtype = cfg, lr, dtype
        self.P = to_torch(params, dtype, True)
        self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.bn_state = {}
        for k, v in self.P.items():
            if k.endswith(".gamma"):
                base = k[:-len(".gamma")]
                self.bn_state[base + ".moving_mean"] = torch.zeros_like(v.detach())
                self.bn_state[base + ".moving_variance"] = torch.ones_like(v.detach())
        self.t = 0

    def step(self, spec_in, emb, spec_out, alpha=0.9, global_batch=None, n_replicas=1, dropout_mask=None):
        """train_step (main_training.py:253-268): fwd, loss, grads, Adam apply."""
        for p in self.P.values():
            p.grad = None
        pred = forward(self.P, spec_in, emb, self.cfg, True, dropout_mask, self.bn_state)
        loss = data_loss(spec_out, pred, alpha, global_batch) + reg_loss(self.P, self.cfg, n_replicas)
        loss.backward()
        self.t += 1
        with torch.no_grad():
            for k, p in self.P.items():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                new, self.m[k], self.v[k] = adam_update(p, g, self.m[k], self.v[k], self.t, self.lr)
                p.copy_(new)
        return float(loss.detach())
