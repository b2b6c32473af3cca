"""Direct-loop NumPy (fp64) definitions of the ops on the hot path.

Oracle / test infrastructure only (see oracle/__init__.py; parity unpinned).
Layout here is the reference's own: NHWC activations, HWIO ``Conv2D`` kernels,
HWOI ``Conv2DTranspose`` kernels, [in, out] ``Dense`` kernels.  Loops run over
output pixels and taps and are vectorised over channels only, so keep shapes
tiny.
"""
import math
import numpy as np


def same_pads(n_in, k, s):
    """TF ``padding='same'`` geometry (tf.nn.convolution semantics used by
    Conv2D at dl_models/u_net.py:269-276, :366, :248): out = ceil(in/s),
    pad_total = max((out-1)*s + k - in, 0), pad_before = pad_total // 2."""
    n_out = -(-n_in // s)
    total = max((n_out - 1) * s + k - n_in, 0)
    return n_out, total // 2, total - total // 2


def conv2d_same(x, w, b=None, stride=1):
    """Conv2D(..., padding='same') - dl_models/u_net.py:269-276 (strided, bias,
    no activation), :366 (3x3 s1), :248 (6x6 head), :262 (1x1).
    x [B,H,W,Ci], w [kh,kw,Ci,Co] (HWIO), b [Co]."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    B, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    Ho, pt, _ = same_pads(H, kh, stride)
    Wo, pl, _ = same_pads(W, kw, stride)
    y = np.zeros((B, Ho, Wo, Co))
    for oy in range(Ho):
        for ox in range(Wo):
            acc = np.zeros((B, Co))
            for a in range(kh):
                iy = oy * stride + a - pt
                if iy < 0 or iy >= H:
                    continue
                for c in range(kw):
                    ix = ox * stride + c - pl
                    if ix < 0 or ix >= W:
                        continue
                    acc += x[:, iy, ix, :] @ w[a, c]
            y[:, oy, ox, :] = acc
    if b is not None:
        y += np.asarray(b, np.float64)
    return y


def conv2d_transpose_same(x, w, b=None, stride=2):
    """Conv2DTranspose(..., strides=2, padding='same') - dl_models/u_net.py:297-304.
    Defined (as TF does) as the adjoint of the SAME strided conv that maps the
    2n-sized output grid back to n: out[2j + a - pad_before] += x[j] * w[a],
    pad_before = max(k - s, 0) // 2, output cropped to [0, n*s).
    x [B,H,W,Ci], w [kh,kw,Co,Ci] (Keras HWOI), b [Co]."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    B, H, W, Ci = x.shape
    kh, kw, Co, _ = w.shape
    Ho, Wo = H * stride, W * stride
    pt = max(kh - stride, 0) // 2
    pl = max(kw - stride, 0) // 2
    y = np.zeros((B, Ho, Wo, Co))
    for j in range(H):
        for i in range(W):
            for a in range(kh):
                oy = j * stride + a - pt
                if oy < 0 or oy >= Ho:
                    continue
                for c in range(kw):
                    ox = i * stride + c - pl
                    if ox < 0 or ox >= Wo:
                        continue
                    y[:, oy, ox, :] += x[:, j, i, :] @ w[a, c].T
    if b is not None:
        y += np.asarray(b, np.float64)
    return y


def batchnorm_train(x, gamma, beta, eps=1e-3):
    """BatchNormalization() in training mode - dl_models/u_net.py:368 (Keras
    defaults axis=-1, epsilon=1e-3): biased batch variance over (B,H,W)."""
    x = np.asarray(x, np.float64)
    mean = x.mean(axis=(0, 1, 2))
    var = x.var(axis=(0, 1, 2))
    return (x - mean) / np.sqrt(var + eps) * gamma + beta, mean, var


def relu(x):
    """Activation('relu') - dl_models/u_net.py:369."""
    return np.maximum(x, 0.0)


def sigmoid(x):
    """Activation('sigmoid') - dl_models/u_net.py:249."""
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))


def embedding_dense(v, table, wd, bd):
    """Embedding(2000,256) -> Flatten -> Dense - dl_models/u_net.py:257-259.
    v int [B,2,16], table [2000,256], wd [8192, dim] ([in,out]), bd [dim]."""
    f = np.asarray(table, np.float64)[np.asarray(v)]          # [B,2,16,256]
    flat = f.reshape(f.shape[0], -1)                          # row-major over (2,16,256)
    return flat @ np.asarray(wd, np.float64) + np.asarray(bd, np.float64)


def amp_phase_loss(y_true, y_pred, alpha=0.9, global_batch=None):
    """compute_loss / phase_loss - main_training.py:184-190, :203-235 (without
    the regulariser term).  y_* [B,H,W,2] NHWC."""
    y_true = np.asarray(y_true, np.float64)
    y_pred = np.asarray(y_pred, np.float64)
    B, H, W, _ = y_true.shape
    gb = B if global_batch is None else global_batch
    e_amp = (y_true[..., 0] - y_pred[..., 0]) ** 2
    yt = y_true[..., 1] * 2 * math.pi - math.pi
    yp = y_pred[..., 1] * 2 * math.pi - math.pi
    ph = np.mod((yt - yp) + math.pi, 2 * math.pi) - math.pi
    e_ph = 1.0 - np.cos(ph)
    per = alpha * e_amp + (1 - alpha) * e_ph
    return per.sum() / (H * W * 2) / gb


def adam_step(theta, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-7):
    """tf.keras.optimizers.Adam defaults (main_training.py:168-169):
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return theta - lr_t * m / (np.sqrt(v) + eps), m, v
