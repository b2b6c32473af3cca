"""Test infrastructure: CPU stand-ins for the HIP kernels behind ``unet_rir_amd.ops``, built on the oracle.

``install(monkeypatch, runtime)`` replaces the functions of ``unet_rir_amd.ops`` that launch kernels with restatements in
torch CPU fp64 (oracle/torch_ref.py's own ops + autograd for the two gradient kernels of every layer), each reporting the
tensors it reads and writes to the simulated runtime (tests/sim_runtime.py).  With that the PRODUCT's UNetEngine /
GraphEngine / Trainer code - buffer plan, launch order, stream hand-overs, bucket schedule - runs unmodified on CPU tensors,
single process or under gloo.  Nothing here is reachable from the product: without the monkeypatch the same calls go to
libunetrir.so and need an MI355X.
"""
import math

import torch
import torch.nn.functional as F

from oracle import torch_ref as R

D = torch.float64


def _nchw(a):
    """Act -> fp64 NCHW tensor of its channels."""
    return a.base[..., a.c0:a.c0 + a.C].to(D).permute(0, 3, 1, 2)


def _put(a, y_nchw):
    a.base[..., a.c0:a.c0 + a.C] = y_nchw.permute(0, 2, 3, 1).to(a.base.dtype)


def _flat2(a):
    """Act -> [P, C] fp64."""
    return a.base[..., a.c0:a.c0 + a.C].reshape(-1, a.C).to(D)


def _put2(a, y):
    a.base[..., a.c0:a.c0 + a.C] = y.reshape(a.B, a.H, a.W, a.C).to(a.base.dtype)


def _act(y, code):
    if code == 1:
        return F.relu(y)
    if code == 2:
        return F.leaky_relu(y, 0.3)
    return y


def _dact(out, code):
    if code == 1:
        return (out > 0).to(D)
    if code == 2:
        return torch.where(out > 0, torch.ones_like(out), torch.full_like(out, 0.3))
    return torch.ones_like(out)


class CpuOps:
    def __init__(self, rt):
        self.rt = rt

    # ---- Conv2D: w [Cout][k][k][Cin]; wt [Cin][k][k][Cout]
    def _w_hwio(self, w, g):
        return w.reshape(g.Cout, g.k, g.k, g.Cin).to(D).permute(1, 2, 3, 0)

    def conv3x3s2_packed_elems(self, N, C_):
        return 0                          # the simulated device has no packed kernel copies

    def conv2d_fwd(self, g, x, w, bias, y, addend=None, w_packed=None):
        self.rt.touch([x, w, bias, addend], [y], "conv2d_fwd")
        out = R.conv2d_same(_nchw(x), self._w_hwio(w, g), None if bias is None else bias.to(D)[:g.Cout], g.stride)
        if addend is not None:
            out = out + _nchw(addend)
        _put(y, out)

    def conv2d_dgrad(self, g, dy, wt, dx, addend=None):
        self.rt.touch([dy, wt, addend], [dx], "conv2d_dgrad")
        w = wt.reshape(g.Cin, g.k, g.k, g.Cout).to(D).permute(1, 2, 0, 3)       # -> HWIO
        x0 = torch.zeros((g.B, g.Cin, g.H, g.W), dtype=D, requires_grad=True)
        out = R.conv2d_same(x0, w, None, g.stride)
        (gx,) = torch.autograd.grad(out, x0, _nchw(dy))
        if addend is not None:
            gx = gx + _nchw(addend)
        _put(dx, gx)

    def _finish_wgrad(self, gw, dw, ws, defer, what):
        """Immediately, or - defer: the product's ops.ReduceBatch protocol - when the batch is flushed: until then dw holds NaNs, so a
        reader that was not ordered behind the flush is seen by the results and by the race check (the write happens at the flush,
        on the stream that flushes)."""
        if defer is None:
            self.rt.touch([], [dw, ws], what)
            dw.reshape(gw.shape).copy_(gw.to(dw.dtype))
            return
        dw.fill_(float("nan"))

        def finish(gw=gw, dw=dw):
            self.rt.touch([], [dw], what + " (deferred split-K reduction)")
            dw.reshape(gw.shape).copy_(gw.to(dw.dtype))
        defer.pending.append(finish)

    def conv2d_wgrad(self, g, x, dy, dw, ws, reg=0.0, w=None, defer=None):
        self.rt.touch([x, dy, w if reg else None], [], "conv2d_wgrad")
        w0 = torch.zeros((g.k, g.k, g.Cin, g.Cout), dtype=D, requires_grad=True)
        out = R.conv2d_same(_nchw(x), w0, None, g.stride)
        (gw,) = torch.autograd.grad(out, w0, _nchw(dy))
        gw = gw.permute(3, 0, 1, 2)
        if reg:
            gw = gw + reg * w.reshape(gw.shape).to(D)
        self._finish_wgrad(gw, dw, ws, defer, "conv2d_wgrad")

    def wgrad_defer_supported(self, storage):
        return True                # the simulated operators park reductions in every storage type (the schedule is what is tested)

    def conv2d_colstat_rows(self, g, dgrad, x):
        return 0

    def conv2d_transpose_colstat_rows(self, g, x):
        return 0

    def conv2d_wgrad_ws_bytes(self, g):
        return 1 << 12

    conv2d_transpose_wgrad_ws_bytes = conv2d_wgrad_ws_bytes

    # ---- Conv2DTranspose: primary [Cin][k][k][Cout]; forward takes [Cout][k][k][Cin]
    def conv2d_transpose_fwd(self, g, x, wt, bias, y):
        self.rt.touch([x, wt, bias], [y], "conv2d_transpose_fwd")
        k_hwoi = wt.reshape(g.Cout, g.k, g.k, g.Cin).to(D).permute(1, 2, 0, 3)
        _put(y, R.conv2d_transpose_same(_nchw(x), k_hwoi, bias.to(D)[:g.Cout], g.stride))

    def _convT(self, g, x, w_primary):
        k_hwoi = w_primary.reshape(g.Cin, g.k, g.k, g.Cout).permute(1, 2, 3, 0)
        return R.conv2d_transpose_same(x, k_hwoi, torch.zeros(g.Cout, dtype=D), g.stride)

    def conv2d_transpose_dgrad(self, g, dy, w, dx, addend=None, w_packed=None):
        self.rt.touch([dy, w, addend], [dx], "conv2d_transpose_dgrad")
        x0 = torch.zeros((g.B, g.Cin, g.H, g.W), dtype=D, requires_grad=True)
        (gx,) = torch.autograd.grad(self._convT(g, x0, w.to(D)), x0, _nchw(dy))
        if addend is not None:
            gx = gx + _nchw(addend)
        _put(dx, gx)

    def conv2d_transpose_wgrad(self, g, x, dy, dw, ws, reg=0.0, w=None, defer=None):
        self.rt.touch([x, dy, w if reg else None], [], "conv2d_transpose_wgrad")
        w0 = torch.zeros((g.Cin, g.k, g.k, g.Cout), dtype=D, requires_grad=True)
        (gw,) = torch.autograd.grad(self._convT(g, _nchw(x), w0), w0, _nchw(dy))
        if reg:
            gw = gw + reg * w.reshape(gw.shape).to(D)
        self._finish_wgrad(gw, dw, ws, defer, "conv2d_transpose_wgrad")

    # ---- Dense
    def dense_fwd(self, x, w, bias, y, ws):
        self.rt.touch([x, w, bias], [y, ws], "dense_fwd")
        K, N = x.C, y.C
        out = _flat2(x) @ w.reshape(N, K).to(D).t()
        if bias is not None:
            out = out + bias.to(D)
        _put2(y, out)

    def dense_dgrad_supported(self, B, K, N):
        return True

    def dense_dgrad(self, dy, w, dx, ws):
        self.rt.touch([dy, w], [dx, ws], "dense_dgrad")
        _put2(dx, _flat2(dy) @ w.reshape(dy.C, dx.C).to(D))

    def transpose_weight(self, w, wt, N, T, C_):
        self.rt.touch([w], [wt], "transpose_weight")
        wt.reshape(C_, T, N).copy_(w.reshape(N, T, C_).permute(2, 1, 0))

    # ---- BatchNormalization
    def bn_stats(self, x, gamma, beta, affine, saved, ws, moving_mean=None, moving_var=None, eps=1e-3, momentum=0.99):
        self.rt.touch([x, gamma, beta, moving_mean, moving_var], [affine, saved, ws, moving_mean, moving_var], "bn_stats")
        v = _flat2(x)
        P, C_ = v.shape
        mean, var = v.mean(0), v.var(0, unbiased=False)
        rstd = 1.0 / torch.sqrt(var + eps)
        scale = gamma.to(D) * rstd
        affine[:C_] = scale.float(); affine[C_:] = (beta.to(D) - mean * scale).float()
        saved[:C_] = mean.float(); saved[C_:] = rstd.float()
        if moving_mean is not None:
            moving_mean.copy_((momentum * moving_mean.to(D) + (1 - momentum) * mean).float())
            moving_var.copy_((momentum * moving_var.to(D) + (1 - momentum) * var * (P / max(P - 1, 1))).float())

    def bn_inference_affine(self, gamma, beta, moving_mean, moving_var, eps, affine):
        self.rt.touch([gamma, beta, moving_mean, moving_var], [affine], "bn_inference_affine")
        C_ = moving_mean.numel()
        scale = gamma.to(D) / torch.sqrt(moving_var.to(D) + eps)
        affine[:C_] = scale.float(); affine[C_:] = (beta.to(D) - moving_mean.to(D) * scale).float()

    def bn_apply(self, x, affine, y, relu=True):
        self.rt.touch([x, affine], [y], "bn_apply")
        v = _flat2(x)
        if affine is not None:
            v = v * affine[:x.C].to(D) + affine[x.C:].to(D)
        _put2(y, _act(v, int(relu)))

    def bn_act_add(self, x, affine, y, act=2, addend=None):
        self.rt.touch([x, affine, addend], [y], "bn_act_add")
        v = _flat2(x)
        if affine is not None:
            v = v * affine[:x.C].to(D) + affine[x.C:].to(D)
        if addend is not None:
            v = v + _flat2(addend)
        _put2(y, _act(v, int(act)))

    def act_bwd(self, da, out, g, act=2):
        self.rt.touch([da, out], [g], "act_bwd")
        _put2(g, _flat2(da) * _dact(_flat2(out), int(act)))

    def bn_bwd(self, da, x, gamma, affine, saved, dx, dgamma, dbeta, ws, relu=True):
        self.rt.touch([da, x, affine, saved], [dx, dgamma, dbeta, ws], "bn_bwd")
        C_ = x.C
        v, P = _flat2(x), x.P
        scale, shift = affine[:C_].to(D), affine[C_:].to(D)
        mean, rstd = saved[:C_].to(D), saved[C_:].to(D)
        xhat = (v - mean) * rstd
        gr = _flat2(da) * _dact(v * scale + shift, int(relu))
        db, dg = gr.sum(0), (gr * xhat).sum(0)
        dgamma.copy_(dg.float()); dbeta.copy_(db.float())
        _put2(dx, scale * (gr - db / P - xhat * dg / P))

    def bn_bwd_junction(self, da, x, out, affine, saved, dx, dgamma, dbeta, ws, act=2, gskip=None, gskip_add=None):
        self.rt.touch([da, x, out, affine, saved, gskip_add], [dx, dgamma, dbeta, ws, gskip], "bn_bwd_junction")
        C_ = x.C
        v, P = _flat2(x), x.P
        scale = affine[:C_].to(D)
        mean, rstd = saved[:C_].to(D), saved[C_:].to(D)
        xhat = (v - mean) * rstd
        gr = _flat2(da) * _dact(_flat2(out), int(act))
        db, dg = gr.sum(0), (gr * xhat).sum(0)
        dgamma.copy_(dg.float()); dbeta.copy_(db.float())
        _put2(dx, scale * (gr - db / P - xhat * dg / P))
        if gskip is not None:
            _put2(gskip, gr + (_flat2(gskip_add) if gskip_add is not None else 0.0))

    def colsum(self, x, out, ws):
        self.rt.touch([x], [out, ws], "colsum")
        out[:x.C] = _flat2(x).sum(0).float()

    def relu_fwd(self, x, y):
        self.rt.touch([x], [y], "relu_fwd")
        _put2(y, F.relu(_flat2(x)))

    def relu_bwd(self, da, x, dx):
        self.rt.touch([da, x], [dx], "relu_bwd")
        _put2(dx, _flat2(da) * (_flat2(x) > 0))

    def add(self, a, b, y):
        self.rt.touch([a, b], [y], "add")
        torch.add(a, b, out=y)

    # ---- boundary, head, loss
    def nchw_to_nhwc_pad(self, x, y):
        self.rt.touch([x], [y], "nchw_to_nhwc_pad")
        y.base.zero_()
        y.base[..., :x.shape[1]] = x.permute(0, 2, 3, 1).to(y.base.dtype)

    def head6x6_supported(self, C_):
        return C_ % 8 == 0

    def head6x6_dgrad_supported(self, W_, C_):
        return False

    def head6x6_fwd(self, x, w, bias, y):
        self.rt.touch([x, w, bias], [y], "head6x6_fwd")
        k = w.reshape(-1, 6, 6, x.C)[:2].to(D).permute(1, 2, 3, 0)
        out = R.conv2d_same(_nchw(x), k, bias.to(D)[:2], 1)
        y.base.zero_()
        y.base[..., :2] = out.permute(0, 2, 3, 1).to(y.base.dtype)

    def head6x6_wgrad(self, x, dy, dw, ws):
        self.rt.touch([x, dy], [dw, ws], "head6x6_wgrad")
        w0 = torch.zeros((6, 6, x.C, 2), dtype=D, requires_grad=True)
        out = R.conv2d_same(_nchw(x), w0, None, 1)
        (gw,) = torch.autograd.grad(out, w0, dy.base[..., :2].to(D).permute(0, 3, 1, 2))
        dw.reshape(-1, 6, 6, x.C)[:2] = gw.permute(3, 0, 1, 2).to(dw.dtype)

    def sigmoid_loss(self, logits, target, alpha, inv_norm, pred, dlogits, loss_out, ws, phase_ref=None, phase_weight=None):
        self.rt.touch([logits, target, phase_ref, phase_weight], [pred, dlogits, loss_out, ws], "sigmoid_loss")
        z = logits.base[..., :2].to(D).permute(0, 3, 1, 2).clone().requires_grad_(True)
        p = torch.sigmoid(z)
        t = target.to(D)
        e_amp = (t[:, 0] - p[:, 0]) ** 2
        t1 = t[:, 1] if phase_ref is None else t[:, 1] - phase_ref.to(D)[:, 1]
        d = (t1 - p[:, 1]) * 2 * math.pi
        ph = torch.remainder(d + math.pi, 2 * math.pi) - math.pi
        e_ph = 1.0 - torch.cos(ph)
        e_w = e_ph if phase_weight is None else e_ph * phase_weight.to(D).view(1, 1, -1)
        loss = (alpha * e_amp + (1 - alpha) * e_w).sum() * inv_norm
        (gz,) = torch.autograd.grad(loss, z)
        pred.copy_(p.detach().float())
        dlogits.base.zero_()
        dlogits.base[..., :2] = gz.permute(0, 2, 3, 1).to(dlogits.base.dtype)
        loss_out[0] = float(loss.detach()); loss_out[1] = float(e_amp.detach().sum()); loss_out[2] = float(e_ph.detach().sum())

    def sigmoid_nchw(self, logits, pred):
        self.rt.touch([logits], [pred], "sigmoid_nchw")
        pred.copy_(torch.sigmoid(logits.base[..., :2].to(D)).permute(0, 3, 1, 2).float())

    def sigmoid_bwd(self, pred, dpred, dlogits):
        self.rt.touch([pred, dpred], [dlogits], "sigmoid_bwd")
        p = pred.to(D)
        dlogits.base.zero_()
        dlogits.base[..., :2] = (dpred.to(D) * p * (1 - p)).permute(0, 2, 3, 1).to(dlogits.base.dtype)

    # ---- bf16 glue
    def make_cast_table(self, entries, device):
        return list(entries), len(entries)

    def cast_weights_batched(self, table):
        ents, _ = table
        for (w, same, tr, N, T, C_, Cp, Np, *_pk) in ents:
            self.rt.touch([w], [same, tr], "cast_weights")
            if same is not None:
                same.reshape(N, T, Cp)[..., :C_] = w.reshape(N, T, C_).to(same.dtype)
            if tr is not None:
                tr.reshape(C_, T, Np)[..., :N] = w.reshape(N, T, C_).permute(2, 1, 0).to(tr.dtype)

    def add_f32_to_bf16(self, a, b, y):
        self.rt.touch([a, b], [y], "add_f32_to_bf16")
        y.base.copy_((a.base.to(D) + b.base.to(D)).to(y.base.dtype))

    def cast_bf16_to_f32(self, a, y):
        self.rt.touch([a], [y], "cast_bf16_to_f32")
        y.base.copy_(a.base.to(y.base.dtype))

    def cast_f32_to_bf16(self, a, y):
        self.rt.touch([a], [y], "cast_f32_to_bf16")
        y.base.copy_(a.base.to(y.base.dtype))

    # ---- information vector
    def index_to_i32(self, idx, out):
        self.rt.touch([idx], [out], "index_to_i32")
        out.copy_(idx.reshape(-1).to(torch.int32))

    def embedding_fwd(self, idx, table, out):
        self.rt.touch([idx, table], [out], "embedding_fwd")
        out.copy_(table[idx.long()])

    def embedding_bwd(self, idx, dout, dtable):
        self.rt.touch([idx, dout], [dtable], "embedding_bwd")
        dtable.zero_()
        dtable.index_add_(0, idx.long(), dout)

    def dropout_mask(self, mask, p, seed, step):
        self.rt.touch([], [mask], "dropout_mask")
        gen = torch.Generator().manual_seed(int(seed) * 1000003 + int(step))
        mask.copy_((torch.rand(mask.shape, generator=gen) >= p).float() / (1.0 - p))

    def mul(self, x, m, y):
        self.rt.touch([x, m], [y], "mul")
        torch.mul(x, m, out=y)

    def sumsq(self, x, coef, out, accumulate, ws):
        self.rt.touch([x, out if accumulate else None], [out, ws], "sumsq")
        r = float(coef * (x.to(D) ** 2).sum())
        out[0] = (float(out[0]) + r) if accumulate else r

    def sgd(self, theta, g, lr, grad_scale=1.0):
        self.rt.touch([theta, g], [theta], "sgd")
        theta.copy_((theta.to(D) - lr * grad_scale * g.to(D)).float())

    def nadam(self, theta, g, m, v, lr, beta1, beta2, eps, c_g, c_m, c_v, grad_scale=1.0):
        self.rt.touch([theta, g, m, v], [theta, m, v], "nadam")
        gg = g.to(D) * grad_scale
        m_ = beta1 * m.to(D) + (1 - beta1) * gg
        v_ = beta2 * v.to(D) + (1 - beta2) * gg * gg
        theta.copy_((theta.to(D) - lr * (c_g * gg + c_m * m_) / ((c_v * v_).sqrt() + eps)).float())
        m.copy_(m_.float()); v.copy_(v_.float())

    def adam(self, theta, g, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
        self.rt.touch([theta, g, m, v], [theta, m, v], "adam")
        gg = g.to(D) * grad_scale
        m_ = beta1 * m.to(D) + (1 - beta1) * gg
        v_ = beta2 * v.to(D) + (1 - beta2) * gg * gg
        theta.copy_((theta.to(D) - lr_t * m_ / (v_.sqrt() + eps)).float())
        m.copy_(m_.float()); v.copy_(v_.float())


def _noop():
    pass


class SimReduceBatch:
    """Stand-in for unet_rir_amd.ops.ReduceBatch: the parked reductions are closures that write the weight gradients when the
    product flushes."""

    def __init__(self, device, arena_bytes, capacity=64, park_max_bytes=None):
        self.pending = []
        self.park_max_bytes = park_max_bytes

    def __len__(self):
        return len(self.pending)

    def flush(self):
        pend, self.pending = self.pending, []
        for fn in pend:
            fn()


def _with_grad(fn):
    """The restatements use torch.autograd for the gradient kernels; they are also called from inside an autograd backward
    (the module path), where grad mode is off."""
    def wrapped(*a, **k):
        with torch.enable_grad():
            return fn(*a, **k)
    wrapped.__name__ = getattr(fn, "__name__", "op")
    return wrapped


def install(monkeypatch, rt):
    """Point every kernel-launching function of unet_rir_amd.ops at the CPU restatements (for the duration of one test)."""
    import unet_rir_amd
    impl = CpuOps(rt)
    for name in dir(impl):
        if not name.startswith("_") and name != "rt":
            if not hasattr(unet_rir_amd.ops, name):
                raise AttributeError(f"unet_rir_amd.ops has no function {name}")
            monkeypatch.setattr(unet_rir_amd.ops, name, _with_grad(getattr(impl, name)))
    monkeypatch.setattr(unet_rir_amd.ops, "bn_ws_bytes", lambda P, C_: 1 << 12)
    monkeypatch.setattr(unet_rir_amd.ops, "ReduceBatch", SimReduceBatch)
    return impl
