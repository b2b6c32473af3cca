"""Oracle self-consistency: the NumPy loop definitions (oracle/np_ops.py) against the
torch.nn.functional restatement (oracle/torch_ref.py), plus TF-SAME geometry KATs
(SURVEY.md appendix A.2).  Parity with the TensorFlow reference itself is UNPINNED
(it cannot run here and ships no fixtures)."""
import math

import numpy as np
import pytest
import torch

from oracle import detrand, np_ops, torch_ref as R


def nhwc(x):
    return np.transpose(x, (0, 2, 3, 1))


def nchw_t(x):
    return torch.tensor(np.transpose(x, (0, 3, 1, 2)))


@pytest.mark.parametrize("H,W,k,s", [(8, 10, 3, 1), (8, 10, 3, 2), (9, 7, 3, 2), (8, 6, 6, 1),
                                     (8, 10, 6, 2), (5, 5, 1, 1), (7, 9, 6, 2)])
def test_conv_same_np_vs_torch(H, W, k, s):
    x = detrand.uniform("x", (2, H, W, 3), -1, 1, np.float64)
    w = detrand.uniform("w", (k, k, 3, 5), -1, 1, np.float64)
    b = detrand.uniform("b", (5,), -1, 1, np.float64)
    y_np = np_ops.conv2d_same(x, w, b, s)
    y_t = R.conv2d_same(nchw_t(x), torch.tensor(w), torch.tensor(b), s)
    assert y_np.shape[1:3] == (math.ceil(H / s), math.ceil(W / s))
    np.testing.assert_allclose(nhwc(y_t.numpy()), y_np, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("H,W,k", [(4, 5, 3), (3, 4, 6), (2, 2, 2)])
def test_conv_transpose_same_np_vs_torch(H, W, k):
    x = detrand.uniform("x", (2, H, W, 3), -1, 1, np.float64)
    w = detrand.uniform("w", (k, k, 4, 3), -1, 1, np.float64)   # HWOI
    b = detrand.uniform("b", (4,), -1, 1, np.float64)
    y_np = np_ops.conv2d_transpose_same(x, w, b, 2)
    y_t = R.conv2d_transpose_same(nchw_t(x), torch.tensor(w), torch.tensor(b), 2)
    assert y_np.shape == (2, 2 * H, 2 * W, 4)
    np.testing.assert_allclose(nhwc(y_t.numpy()), y_np, rtol=1e-12, atol=1e-12)


def test_conv_transpose_is_adjoint_of_strided_conv():
    """<conv_s2(u), x> == <u, convT(x)> with shared kernel: the definition TF uses."""
    k = 3
    u = detrand.uniform("u", (1, 8, 6, 4), -1, 1, np.float64)
    x = detrand.uniform("x", (1, 4, 3, 5), -1, 1, np.float64)
    w_hwio = detrand.uniform("w", (k, k, 4, 5), -1, 1, np.float64)    # conv: 4 -> 5
    w_hwoi = w_hwio                                                    # convT 5 -> 4 holds [kh,kw,out=4,in=5]
    lhs = (np_ops.conv2d_same(u, w_hwio, None, 2) * x).sum()
    rhs = (u * np_ops.conv2d_transpose_same(x, w_hwoi, None, 2)).sum()
    assert abs(lhs - rhs) < 1e-10


def test_same_geometry_kats():
    """Delta-image KATs from SURVEY.md A.2: k=3 s=2 even input pads (0,1); k=6 s=1 pads (2,3);
    convT k=3 keeps [0:2n]."""
    assert np_ops.same_pads(8, 3, 2) == (4, 0, 1)
    assert np_ops.same_pads(9, 3, 2) == (5, 1, 1)
    assert np_ops.same_pads(8, 6, 1) == (8, 2, 3)
    assert np_ops.same_pads(8, 6, 2) == (4, 2, 2)
    assert np_ops.same_pads(8, 3, 1) == (8, 1, 1)
    # one-hot image, ramp kernel
    x = np.zeros((1, 4, 4, 1)); x[0, 2, 2, 0] = 1.0
    w = np.arange(9, dtype=np.float64).reshape(3, 3, 1, 1)
    y = np_ops.conv2d_same(x, w, None, 2)[0, :, :, 0]
    exp = np.zeros((2, 2)); exp[1, 1] = w[0, 0, 0, 0]; exp[0, 0] = w[2, 2, 0, 0]
    exp[0, 1] = w[2, 0, 0, 0]; exp[1, 0] = w[0, 2, 0, 0]
    np.testing.assert_array_equal(y, exp)
    xt = np.zeros((1, 2, 2, 1)); xt[0, 1, 1, 0] = 1.0
    yt = np_ops.conv2d_transpose_same(xt, w.reshape(3, 3, 1, 1), None, 2)[0, :, :, 0]
    exp_t = np.zeros((4, 4)); exp_t[2:4, 2:4] = w[:2, :2, 0, 0]
    np.testing.assert_array_equal(yt, exp_t)


def test_bn_and_loss_np_vs_torch():
    x = detrand.uniform("x", (3, 4, 5, 6), -2, 2, np.float64)
    g = detrand.uniform("g", (6,), 0.5, 1.5, np.float64)
    b = detrand.uniform("b", (6,), -1, 1, np.float64)
    y_np, mean, var = np_ops.batchnorm_train(x, g, b)
    y_t = R.bn_relu(nchw_t(x), torch.tensor(g), torch.tensor(b), None, "bn", True, relu=False)
    np.testing.assert_allclose(nhwc(y_t.numpy()), y_np, rtol=1e-10, atol=1e-10)
    yt = detrand.uniform("yt", (2, 4, 5, 2), 0, 1, np.float64)
    yp = detrand.uniform("yp", (2, 4, 5, 2), 0, 1, np.float64)
    l_np = np_ops.amp_phase_loss(yt, yp, 0.9, 8)
    l_t = R.data_loss(nchw_t(yt), nchw_t(yp), 0.9, 8)
    assert abs(l_np - float(l_t)) < 1e-12


def test_adam_np_vs_torch():
    th = detrand.uniform("th", (50,), -1, 1, np.float64)
    g = detrand.uniform("g", (50,), -1, 1, np.float64)
    m = np.zeros(50); v = np.zeros(50)
    a, ma, va = np_ops.adam_step(th, g, m, v, 1, 1e-3)
    b, mb, vb = R.adam_update(torch.tensor(th), torch.tensor(g), torch.tensor(m), torch.tensor(v), 1, 1e-3)
    np.testing.assert_allclose(b.numpy(), a, rtol=1e-12)
    # first Adam step moves every weight by ~lr * sign(g)
    np.testing.assert_allclose(th - a, 1e-3 * np.sign(g), rtol=5e-3)


def test_graph_shapes_and_param_count():
    """Parameter counts quoted in SURVEY.md appendix B pin the graph wiring."""
    assert sum(int(np.prod(s)) for s in R.param_shapes(R.Config(256, 256, 64, 3)).values()) == 68_613_058
    assert sum(int(np.prod(s)) for s in R.param_shapes(R.Config(144, 160, 32, 3)).values()) == 20_955_010
    assert sum(int(np.prod(s)) for s in R.param_shapes(R.Config(256, 256, 16, 3)).values()) == 36_236_530
    cfg = R.Config(32, 48, 4, 3)
    P = R.to_torch(R.init_params(cfg, randomize_all=True))
    spec_in, emb, _ = R.synthetic_batch(cfg, 2)
    y = R.forward(P, torch.tensor(spec_in), torch.tensor(emb), cfg)
    assert y.shape == (2, 2, 32, 48)
    assert float(y.min()) > 0 and float(y.max()) < 1


def test_vector_block_np_vs_torch():
    cfg = R.Config(16, 16, 4, 3)
    Pn = R.init_params(cfg, randomize_all=True, dtype=np.float64)
    _, emb, _ = R.synthetic_batch(cfg, 2)
    v_np = np_ops.embedding_dense(emb, Pn["vec.embedding"], Pn["vec.dense.kernel"], Pn["vec.dense.bias"])
    inter = {}
    P = R.to_torch(Pn, torch.float64)
    spec = torch.zeros(2, 2, 16, 16, dtype=torch.float64)
    R.forward(P, spec, torch.tensor(emb), cfg, inter=inter)
    np.testing.assert_allclose(inter["vec.dense"].numpy(), v_np, rtol=1e-10, atol=1e-12)


def test_gradcheck_fp64_tiny():
    """Finite-difference check of the oracle's own gradients (fp64, 16x16)."""
    cfg = R.Config(16, 16, 4, 3)
    Pn = R.init_params(cfg, randomize_all=True, dtype=np.float64)
    spec_in, emb, spec_out = R.synthetic_batch(cfg, 2)
    loss, _, _, grads = R.loss_and_grads(Pn, spec_in, emb, spec_out, cfg, dtype=torch.float64)
    for name, idx in [("enc2.down.kernel", (1, 2, 3, 1)), ("dec1.up.kernel", (0, 1, 2, 3)),
                      ("enc3.cb1.gamma", (5,)), ("head.bias", (1,)), ("vec.dense.kernel", (100, 7))]:
        eps = 1e-5
        Pp = {k: v.copy() for k, v in Pn.items()}; Pp[name][idx] += eps
        Pm = {k: v.copy() for k, v in Pn.items()}; Pm[name][idx] -= eps
        lp = R.loss_and_grads(Pp, spec_in, emb, spec_out, cfg, dtype=torch.float64)[0]
        lm = R.loss_and_grads(Pm, spec_in, emb, spec_out, cfg, dtype=torch.float64)[0]
        fd = (lp - lm) / (2 * eps)
        an = float(grads[name][idx])
        assert abs(fd - an) <= 1e-6 + 1e-4 * abs(an), (name, fd, an)
