"""Parity of every HIP kernel (called through the C ABI) against the CPU oracle on seeded inputs.
fp32 tolerances are stated per test; the oracle side runs in fp64."""
import math

import numpy as np
import pytest
import torch

from oracle import detrand, torch_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def U():
    import unet_rir_amd
    unet_rir_amd._lib.lib()
    return unet_rir_amd


DEV = "cuda:0"


def close(actual, expected, tol, what=""):
    a = actual.detach().double().cpu()
    e = expected.detach().double().cpu()
    assert a.shape == e.shape, (what, a.shape, e.shape)
    scale = float(e.abs().max()) + 1e-30
    err = float((a - e).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def to_nhwc_buf(x_nchw, ld, c0, dev):
    """Embed an NCHW tensor into channels [c0, c0+C) of a poisoned [B,H,W,ld] device buffer."""
    B, C, H, W = x_nchw.shape
    buf = torch.full((B, H, W, ld), 777.0, dtype=torch.float32)
    buf[..., c0:c0 + C] = x_nchw.permute(0, 2, 3, 1).float()
    return buf.to(dev)


CONV_CASES = [
    # B, H, W, Cin, Cout, k, s
    (2, 16, 24, 8, 16, 3, 1),
    (2, 16, 24, 16, 32, 3, 2),
    (1, 9, 7, 8, 8, 3, 2),         # odd input: SAME pads (1,1)
    (2, 8, 8, 64, 128, 3, 1),      # BN=128 tile
    (1, 10, 12, 40, 72, 3, 1),     # channels not multiples of 32
    (1, 12, 12, 16, 8, 6, 1),      # head geometry: pads (2,3)
    (1, 12, 12, 8, 8, 6, 2),
    (2, 4, 4, 16, 64, 1, 1),       # 1x1 (information-vector conv)
    (1, 20, 20, 4, 16, 3, 1),      # padded stem (Cin=4)
    (1, 20, 20, 32, 4, 6, 1),      # padded head (Cout=4)
    (3, 1, 1, 256, 64, 1, 1),      # Dense as a 1x1 conv on a 1x1 grid
    (2, 24, 70, 48, 160, 3, 1),    # patch-staged 3x3 kernel: several tiles in x / y / N, ragged edges, 2 channel chunks
    (1, 40, 64, 72, 64, 3, 1),
    (2, 32, 72, 40, 72, 3, 2),     # stride 2, even size: the data gradient takes the fused parity-class kernel
]


def conv_data(case, dtype=torch.float64):
    B, H, W, Ci, Co, k, s = case
    x = torch.tensor(detrand.uniform(f"x{case}", (B, Ci, H, W), -1, 1, np.float64), dtype=dtype)
    w = torch.tensor(detrand.uniform(f"w{case}", (k, k, Ci, Co), -1, 1, np.float64), dtype=dtype)
    b = torch.tensor(detrand.uniform(f"b{case}", (Co,), -1, 1, np.float64), dtype=dtype)
    return x, w, b


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_dgrad_wgrad(U, case):
    ops = U.ops
    B, H, W, Ci, Co, k, s = case
    x, w, b = conv_data(case)
    x.requires_grad_(True); w.requires_grad_(True)
    y = R.conv2d_same(x, w, b, s)
    Ho, Wo = y.shape[2], y.shape[3]
    add = torch.tensor(detrand.uniform(f"a{case}", (B, Co, Ho, Wo), -1, 1, np.float64))
    gy = torch.tensor(detrand.uniform(f"g{case}", (B, Co, Ho, Wo), -1, 1, np.float64))
    (y * gy).sum().backward()
    K = max(Ci * k * k, 1)
    tol = 2e-6 * math.sqrt(K) + 1e-6

    g = ops.geom(B, H, W, Ci, Co, k, s)
    xa = ops.Act(to_nhwc_buf(x.detach(), Ci + 8, 4, DEV), 4, Ci)          # offset view: concat-style input
    w_ohwi = w.detach().permute(3, 0, 1, 2).contiguous().float().to(DEV)
    bias = b.float().to(DEV)
    ya = ops.Act(torch.full((B, Ho, Wo, Co + 4), 555.0, device=DEV), 0, Co)
    adda = ops.Act(to_nhwc_buf(add, Co, 0, DEV))
    ops.conv2d_fwd(g, xa, w_ohwi, bias, ya, adda)
    torch.cuda.synchronize()
    close(ya.dense().permute(0, 3, 1, 2), y.detach() + add, tol, "fwd")
    assert float(ya.base[..., Co:].min()) == 555.0      # pad channels untouched

    # dgrad
    wt = torch.empty((Ci, k * k, Co), device=DEV)
    ops.transpose_weight(w_ohwi, wt, Co, k * k, Ci)
    torch.cuda.synchronize()
    close(wt, w.detach().permute(2, 0, 1, 3).reshape(Ci, k * k, Co), 1e-7, "transpose_weight")
    gya = ops.Act(to_nhwc_buf(gy, Co, 0, DEV))
    dxa = ops.Act(torch.full((B, H, W, Ci), 333.0, device=DEV))
    ops.conv2d_dgrad(g, gya, wt, dxa)
    torch.cuda.synchronize()
    close(dxa.dense().permute(0, 3, 1, 2), x.grad, 2e-6 * math.sqrt(Co * k * k) + 1e-6, "dgrad")

    # wgrad (+ l2 regulariser gradient 2*coef*w folded into the reduction)
    ws = ops.Workspace(DEV)
    dw = torch.full((Co, k, k, Ci), 111.0, device=DEV)
    reg = 0.002
    ops.conv2d_wgrad(g, xa, gya, dw, ws, reg=reg, w=w_ohwi)
    torch.cuda.synchronize()
    exp = (w.grad + reg * w.detach()).permute(3, 0, 1, 2)
    close(dw, exp, 2e-6 * math.sqrt(B * Ho * Wo) + 1e-6, "wgrad")


CONVT_CASES = [(2, 6, 5, 16, 8, 3), (1, 4, 4, 64, 32, 3), (1, 5, 6, 8, 16, 6), (2, 3, 3, 128, 64, 3),
               (2, 12, 40, 48, 72, 3), (1, 16, 32, 64, 64, 3)]      # the last two take the fused parity-class kernel


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv2d_transpose(U, case):
    ops = U.ops
    B, H, W, Ci, Co, k = case
    x = torch.tensor(detrand.uniform(f"tx{case}", (B, Ci, H, W), -1, 1, np.float64), requires_grad=True)
    w = torch.tensor(detrand.uniform(f"tw{case}", (k, k, Co, Ci), -1, 1, np.float64), requires_grad=True)  # HWOI
    b = torch.tensor(detrand.uniform(f"tb{case}", (Co,), -1, 1, np.float64))
    y = R.conv2d_transpose_same(x, w, b, 2)
    gy = torch.tensor(detrand.uniform(f"tg{case}", tuple(y.shape), -1, 1, np.float64))
    (y * gy).sum().backward()

    g = ops.geom(B, H, W, Ci, Co, k, 2)
    w_prim = w.detach().permute(3, 0, 1, 2).contiguous().float().to(DEV)       # [Ci][kh][kw][Co]
    wt = torch.empty((Co, k * k, Ci), device=DEV)
    ops.transpose_weight(w_prim, wt, Ci, k * k, Co)
    xa = ops.Act(to_nhwc_buf(x.detach(), Ci, 0, DEV))
    ya = ops.Act(torch.full((B, 2 * H, 2 * W, 2 * Co), 555.0, device=DEV), Co, Co)   # upper half of a concat buffer
    ops.conv2d_transpose_fwd(g, xa, wt, b.float().to(DEV), ya)
    torch.cuda.synchronize()
    close(ya.dense().permute(0, 3, 1, 2), y.detach(), 2e-6 * math.sqrt(Ci * k * k) + 1e-6, "convT fwd")
    assert float(ya.base[..., :Co].min()) == 555.0

    gya = ops.Act(to_nhwc_buf(gy, 2 * Co, Co, DEV), Co, Co)
    dxa = ops.Act(torch.full((B, H, W, Ci), 333.0, device=DEV))
    ops.conv2d_transpose_dgrad(g, gya, w_prim, dxa)
    torch.cuda.synchronize()
    close(dxa.dense().permute(0, 3, 1, 2), x.grad, 2e-6 * math.sqrt(Co * k * k) + 1e-6, "convT dgrad")

    ws = ops.Workspace(DEV)
    dw = torch.full((Ci, k, k, Co), 111.0, device=DEV)
    ops.conv2d_transpose_wgrad(g, xa, gya, dw, ws, reg=0.002, w=w_prim)
    torch.cuda.synchronize()
    close(dw, (w.grad + 0.002 * w.detach()).permute(3, 0, 1, 2), 2e-6 * math.sqrt(B * H * W) + 1e-6, "convT wgrad")


@pytest.mark.parametrize("shape", [(2, 12, 10, 16), (3, 7, 5, 40), (1, 4, 4, 1024), (2, 33, 31, 8)])
@pytest.mark.parametrize("relu", [True, False])
def test_batchnorm_fwd_bwd(U, shape, relu):
    ops = U.ops
    B, H, W, Cc = shape
    x = torch.tensor(detrand.uniform(f"bx{shape}", (B, Cc, H, W), -2, 2, np.float64), requires_grad=True)
    gamma = torch.tensor(detrand.uniform(f"bg{shape}", (Cc,), 0.5, 1.5, np.float64), requires_grad=True)
    beta = torch.tensor(detrand.uniform(f"bb{shape}", (Cc,), -0.5, 0.5, np.float64), requires_grad=True)
    state = {"bn.moving_mean": torch.zeros(Cc, dtype=torch.float64), "bn.moving_variance": torch.ones(Cc, dtype=torch.float64)}
    y = R.bn_relu(x, gamma, beta, state, "bn", True, relu=relu)
    gy = torch.tensor(detrand.uniform(f"bgy{shape}", (B, Cc, H, W), -1, 1, np.float64))
    (y * gy).sum().backward()

    ws = ops.Workspace(DEV)
    xa = ops.Act(to_nhwc_buf(x.detach(), Cc + 4, 4, DEV), 4, Cc)
    aff = torch.empty(2 * Cc, device=DEV); saved = torch.empty(2 * Cc, device=DEV)
    mm = torch.zeros(Cc, device=DEV); mv = torch.ones(Cc, device=DEV)
    g32, b32 = gamma.detach().float().to(DEV), beta.detach().float().to(DEV)
    ops.bn_stats(xa, g32, b32, aff, saved, ws, mm, mv)
    ya = ops.Act(torch.empty((B, H, W, Cc), device=DEV))
    ops.bn_apply(xa, aff, ya, relu=relu)
    torch.cuda.synchronize()
    close(ya.dense().permute(0, 3, 1, 2), y.detach(), 2e-6, "bn fwd")
    close(mm, state["bn.moving_mean"], 1e-5, "moving mean")
    close(mv, state["bn.moving_variance"], 1e-5, "moving var")

    gya = ops.Act(to_nhwc_buf(gy, Cc, 0, DEV))
    dxa = ops.Act(torch.empty((B, H, W, Cc), device=DEV))
    dg = torch.empty(Cc, device=DEV); db = torch.empty(Cc, device=DEV)
    ops.bn_bwd(gya, xa, g32, aff, saved, dxa, dg, db, ws, relu=relu)
    torch.cuda.synchronize()
    close(dxa.dense().permute(0, 3, 1, 2), x.grad, 1e-5, "bn dx")
    close(dg, gamma.grad, 1e-5, "dgamma")
    close(db, beta.grad, 1e-5, "dbeta")

    cs = torch.empty(Cc, device=DEV)
    ops.colsum(gya, cs, ws)
    torch.cuda.synchronize()
    close(cs, gy.sum(dim=(0, 2, 3)), 1e-5, "colsum")


def test_relu_only(U):
    ops = U.ops
    x = torch.tensor(detrand.uniform("rx", (2, 5, 6, 8), -1, 1))
    g = torch.tensor(detrand.uniform("rg", (2, 5, 6, 8), -1, 1))
    xa, ga = ops.Act(x.to(DEV)), ops.Act(g.to(DEV))
    ya, dxa = ops.Act(torch.empty_like(xa.base)), ops.Act(torch.empty_like(xa.base))
    ops.relu_fwd(xa, ya); ops.relu_bwd(ga, xa, dxa)
    torch.cuda.synchronize()
    assert torch.equal(ya.base.cpu(), x.clamp_min(0))
    assert torch.equal(dxa.base.cpu(), g * (x > 0))


@pytest.mark.parametrize("shape", [(2, 12, 20), (1, 33, 17)])
def test_sigmoid_loss(U, shape):
    ops = U.ops
    B, H, W = shape
    logits = torch.tensor(detrand.uniform(f"sl{shape}", (B, 2, H, W), -3, 3, np.float64), requires_grad=True)
    target = torch.tensor(detrand.uniform(f"st{shape}", (B, 2, H, W), 0, 1, np.float64))
    pred = torch.sigmoid(logits)
    gb = 4 * B
    loss = R.data_loss(target, pred, 0.9, gb)
    loss.backward()
    la = ops.Act(to_nhwc_buf(logits.detach(), 4, 0, DEV), 0, 4)
    pr = torch.empty((B, 2, H, W), device=DEV)
    dl = ops.Act(torch.empty((B, H, W, 4), device=DEV))
    out = torch.zeros(4, device=DEV)
    ws = ops.Workspace(DEV)
    ops.sigmoid_loss(la, target.float().to(DEV), 0.9, 1.0 / (2 * H * W * gb), pr, dl, out, ws)
    torch.cuda.synchronize()
    close(pr, pred.detach(), 1e-6, "pred")
    assert abs(float(out[0]) - float(loss.detach())) <= 2e-6 * abs(float(loss.detach()))
    close(dl.dense()[..., :2].permute(0, 3, 1, 2), logits.grad, 2e-5, "dlogits")
    assert float(dl.dense()[..., 2:].abs().max()) == 0.0
    pr2 = torch.empty_like(pr)
    ops.sigmoid_nchw(la, pr2)
    torch.cuda.synchronize()
    assert torch.equal(pr, pr2)


def test_embedding_adam_misc(U):
    ops = U.ops
    idx = torch.tensor(detrand.randint("ei", (3, 2, 16), 26, 60), dtype=torch.int32)      # many duplicates
    table = torch.tensor(detrand.uniform("et", (2000, 256), -0.05, 0.05))
    out = torch.empty((idx.numel(), 256), device=DEV)
    ops.embedding_fwd(idx.to(DEV), table.to(DEV), out)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), table[idx.long().flatten()])
    dout = torch.tensor(detrand.uniform("ed", (idx.numel(), 256), -1, 1))
    dt = torch.full((2000, 256), 9.0, device=DEV)
    ops.embedding_bwd(idx.to(DEV), dout.to(DEV), dt)
    torch.cuda.synchronize()
    exp = torch.zeros(2000, 256, dtype=torch.float64).index_add_(0, idx.long().flatten(), dout.double())
    close(dt, exp, 1e-6, "embedding bwd")

    n = 10007
    th = torch.tensor(detrand.uniform("ath", (n,), -1, 1)); g = torch.tensor(detrand.uniform("ag", (n,), -1, 1))
    m = torch.tensor(detrand.uniform("am", (n,), -0.1, 0.1)); v = torch.tensor(detrand.uniform("av", (n,), 0, 0.1))
    t, lr = 3, 1e-3
    e_th, e_m, e_v = R.adam_update(th.double(), g.double() * 0.5, m.double(), v.double(), t, lr)
    lr_t = lr * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
    d = [a.clone().to(DEV) for a in (th, g, m, v)]
    ops.adam(d[0], d[1], d[2], d[3], lr_t, grad_scale=0.5)
    torch.cuda.synchronize()
    close(d[0], e_th, 1e-6, "adam theta"); close(d[2], e_m, 1e-6, "adam m"); close(d[3], e_v, 1e-6, "adam v")

    ws = ops.Workspace(DEV)
    acc = torch.tensor([1.5], device=DEV)
    ops.sumsq(d[1], 0.001, acc, True, ws)
    x = torch.tensor(detrand.uniform("mx", (999,), -1, 1)); k = torch.tensor(detrand.uniform("mk", (999,), 0, 2))
    y = torch.empty(999, device=DEV)
    ops.mul(x.to(DEV), k.to(DEV), y)
    nchw = torch.tensor(detrand.uniform("nx", (2, 2, 5, 7), 0, 1))
    pa = ops.Act(torch.full((2, 5, 7, 4), 5.0, device=DEV))
    ops.nchw_to_nhwc_pad(nchw.to(DEV), pa)
    torch.cuda.synchronize()
    assert abs(float(acc) - (1.5 + 0.001 * float((g.double() ** 2).sum()))) < 1e-5
    assert torch.equal(y.cpu(), x * k)
    assert torch.equal(pa.base[..., :2].cpu(), nchw.permute(0, 2, 3, 1)) and float(pa.base[..., 2:].abs().max()) == 0


@pytest.mark.parametrize("B,H,W,Cc", [(2, 20, 37, 16), (1, 32, 32, 64), (3, 16, 16, 32)])
def test_head6x6_direct(U, B, H, W, Cc):
    """The direct (non-MFMA) head kernels against the oracle: Conv2D(2,(6,6),'same') forward and weight gradient."""
    ops = U.ops
    x = torch.tensor(detrand.uniform(f"hx{B,H,W,Cc}", (B, Cc, H, W), -1, 1, np.float64))
    w = torch.tensor(detrand.uniform(f"hw{Cc}", (6, 6, Cc, 2), -1, 1, np.float64), requires_grad=True)
    b = torch.tensor(detrand.uniform("hb", (2,), -1, 1, np.float64))
    y = R.conv2d_same(x, w, b, 1)
    gy = torch.tensor(detrand.uniform(f"hg{B,H,W}", (B, 2, H, W), -1, 1, np.float64))
    (y * gy).sum().backward()
    xa = ops.Act(to_nhwc_buf(x, Cc + 4, 4, DEV), 4, Cc)
    w4 = torch.zeros((4, 6, 6, Cc), device=DEV); w4[:2] = w.detach().permute(3, 0, 1, 2).float().to(DEV)
    b4 = torch.zeros(4, device=DEV); b4[:2] = b.float().to(DEV)
    ya = ops.Act(torch.full((B, H, W, 4), 5.0, device=DEV))
    ops.head6x6_fwd(xa, w4, b4, ya)
    torch.cuda.synchronize()
    close(ya.dense()[..., :2].permute(0, 3, 1, 2), y.detach(), 2e-6 * math.sqrt(36 * Cc) + 1e-6, "head fwd")
    assert float(ya.dense()[..., 2:].abs().max()) == 0.0
    gya = ops.Act(to_nhwc_buf(gy, 4, 0, DEV), 0, 4)
    gya.base[..., 2:] = 0
    dw = torch.zeros((4, 6, 6, Cc), device=DEV)
    ws = ops.Workspace(DEV)
    ops.head6x6_wgrad(xa, gya, dw, ws)
    torch.cuda.synchronize()
    close(dw[:2], w.grad.permute(3, 0, 1, 2), 2e-6 * math.sqrt(B * H * W) + 1e-6, "head wgrad")
    assert float(dw[2:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------------------------
# bf16-storage kernels: inputs are rounded to bf16 first, the oracle then runs in fp64 on the SAME rounded values, so the
# only differences are fp32 accumulation order and the final rounding of bf16 outputs (2^-8 relative).
# ------------------------------------------------------------------------------------------------------------------
def q16(t):
    return t.to(torch.bfloat16).double()


def to_nhwc_bf16(x_nchw, ld, c0, dev):
    B, C, H, W = x_nchw.shape
    buf = torch.full((B, H, W, ld), 768.0, dtype=torch.bfloat16)
    buf[..., c0:c0 + C] = x_nchw.permute(0, 2, 3, 1).to(torch.bfloat16)
    return buf.to(dev)


BF16_CONV_CASES = [
    (2, 16, 24, 8, 16, 3, 1), (2, 16, 24, 16, 32, 3, 2), (1, 9, 7, 8, 8, 3, 2), (2, 8, 8, 64, 128, 3, 1),
    (1, 10, 12, 40, 72, 3, 1), (1, 12, 12, 64, 8, 6, 1), (2, 16, 16, 128, 64, 3, 1), (1, 32, 32, 8, 64, 3, 1),
    (2, 24, 70, 48, 160, 3, 1), (1, 40, 64, 136, 64, 3, 1),          # patch-staged kernel: multi-tile, ragged, 3 chunks
    (2, 32, 72, 40, 72, 3, 2),                                        # stride 2: fused parity-class data gradient
    (1, 32, 64, 24, 64, 3, 2),                                        # stride 2: data gradient on upconv3x3g (C = 64 output channels)
    (2, 32, 64, 64, 128, 3, 2), (2, 16, 64, 136, 72, 3, 2), (3, 24, 32, 32, 160, 3, 2),   # stride 2: weight gradient on wgrad3x3d (LDS-DMA; one tile / ragged channel tails / several n tiles)
    # stride 2 forward on conv3x3d (persistent LDS-DMA): one job / 3 chunks, ragged channel block, 2 x 2 tiles / three channel
    # tiles, the last one 32 wide / 9 jobs (grid not a multiple of 8) / 272 jobs on 256 workgroups (two jobs per workgroup)
    (1, 16, 64, 16, 32, 3, 2), (2, 32, 128, 48, 96, 3, 2), (2, 32, 64, 32, 288, 3, 2), (9, 16, 64, 32, 128, 3, 2), (34, 64, 128, 32, 64, 3, 2),
    (2, 32, 64, 128, 192, 3, 2),                                      # ... with the packed kernel copy: two channel tiles, the second half empty
    (2, 32, 32, 32, 64, 3, 2), (3, 64, 96, 48, 96, 3, 2), (2, 32, 32, 64, 128, 3, 2),   # outputs 16 / 48 wide: the tap-table kernel keeps them (a 16 x 16-tile variant of conv3x3d measured slower: 131 vs 111 us)
    (2, 40, 70, 64, 160, 3, 1), (1, 33, 64, 96, 96, 3, 1),            # LDS-DMA kernel (conv3x3g): ragged tiles / channels, 3 chunks
    (3, 16, 16, 64, 160, 3, 1), (4, 16, 16, 96, 128, 3, 1), (2, 32, 16, 64, 128, 3, 1), (5, 9, 11, 32, 72, 3, 1),   # conv3x3g, two narrow images per tile (odd batch, ragged)
    (2, 32, 32, 128, 256, 3, 1),                                      # LDS-DMA kernel: forward 4 chunks, data gradient 8 chunks
    # strip kernel (conv3x3s, 64 -> 64): ragged rows and columns; many one-tile segments; whole-column strips of 3 tiles on
    # 256 workgroups; two jobs of 2 tiles per workgroup
    (2, 21, 60, 64, 64, 3, 1), (3, 64, 96, 64, 64, 3, 1), (16, 24, 512, 64, 64, 3, 1), (32, 16, 512, 64, 64, 3, 1),
    # ... and its 32 -> 32 form (64-byte pixels, one K chunk): the same shapes, and the reference geometry's 144 x 160
    (2, 21, 60, 32, 32, 3, 1), (3, 64, 96, 32, 32, 3, 1), (16, 24, 512, 32, 32, 3, 1), (32, 16, 512, 32, 32, 3, 1), (4, 144, 160, 32, 32, 3, 1),
]


@pytest.mark.parametrize("case", BF16_CONV_CASES)
def test_conv2d_bf16(U, case, monkeypatch):
    ops = U.ops
    B, H, W, Ci, Co, k, s = case
    if W <= 16:      # narrow images: take the paired-image tile of conv3x3g even at test sizes (it is gated on the workgroup count)
        ops.set_config(conv3x3g_pair=2)
    x, w, b = conv_data(case)
    x, w = q16(x), q16(w)
    x.requires_grad_(True); w.requires_grad_(True)
    y = R.conv2d_same(x, w, b, s)
    Ho, Wo = y.shape[2], y.shape[3]
    add = q16(torch.tensor(detrand.uniform(f"a{case}", (B, Co, Ho, Wo), -1, 1, np.float64)))
    gy = q16(torch.tensor(detrand.uniform(f"g{case}", (B, Co, Ho, Wo), -1, 1, np.float64)))
    (y * gy).sum().backward()

    g = ops.geom(B, H, W, Ci, Co, k, s)
    xa = ops.Act(to_nhwc_bf16(x.detach(), Ci + 8, 8, DEV), 8, Ci)
    w32 = w.detach().permute(3, 0, 1, 2).contiguous().float().to(DEV)            # [Co][k][k][Ci] fp32 master
    wh = torch.empty((Co, k * k, Ci), dtype=torch.bfloat16, device=DEV)
    ops.cast_weight_bf16(w32, wh, Co, k * k, Ci, Ci)
    ya = ops.Act(torch.full((B, Ho, Wo, Co + 8), 512.0, dtype=torch.bfloat16, device=DEV), 0, Co)
    adda = ops.Act(to_nhwc_bf16(add, Co, 0, DEV))
    ops.conv2d_fwd(g, xa, wh, b.float().to(DEV), ya, adda)
    torch.cuda.synchronize()
    close(ya.dense().permute(0, 3, 1, 2), y.detach() + add, 1e-2, "bf16 fwd")
    assert float(ya.base[..., Co:].float().min()) == 512.0
    ne = ops.conv3x3s2_packed_elems(Co, Ci) if (k == 3 and s == 2) else 0
    if ne:      # the packed kernel copy of the stride-2 forward kernel: same values in the same LDS positions -> the same bits out
        pk, pk2 = torch.zeros(ne, dtype=torch.bfloat16, device=DEV), torch.zeros(ne, dtype=torch.bfloat16, device=DEV)
        same2, tr2 = torch.empty_like(wh), torch.empty((Ci, k * k, Co), dtype=torch.bfloat16, device=DEV)
        ops.cast_weights_batched(ops.make_cast_table([(w32, same2, tr2, Co, k * k, Ci, Ci, Co, pk)], DEV))     # fused cast kernel
        ops.cast_weights_batched(ops.make_cast_table([(w32, same2, None, Co, k * k, Ci, Ci, Co, pk2)], DEV))   # element-wise one
        assert torch.equal(same2, wh) and torch.equal(pk, pk2) and float(pk.float().abs().max()) > 0
        ya2 = ops.Act(torch.full((B, Ho, Wo, Co + 8), 512.0, dtype=torch.bfloat16, device=DEV), 0, Co)
        ops.conv2d_fwd(g, xa, wh, b.float().to(DEV), ya2, adda, w_packed=pk)
        torch.cuda.synchronize()
        assert torch.equal(ya2.base, ya.base)

    Np = -(-Co // 8) * 8
    wt = torch.zeros((Ci, k * k, Np), dtype=torch.bfloat16, device=DEV)
    ops.transpose_cast_weight_bf16(w32, wt, Co, k * k, Ci, Np)
    gya = ops.Act(to_nhwc_bf16(gy, Np, 0, DEV), 0, Np)
    if Np != Co:
        gya.base[..., Co:] = 0
    gd = ops.geom(B, H, W, Ci, Np, k, s)
    dxa = ops.Act(torch.full((B, H, W, Ci), 256.0, dtype=torch.bfloat16, device=DEV))
    ops.conv2d_dgrad(gd, gya, wt, dxa)
    torch.cuda.synchronize()
    close(dxa.dense().permute(0, 3, 1, 2), x.grad, 1e-2, "bf16 dgrad")

    if k == 3:
        ws = ops.Workspace(DEV)
        dw = torch.full((Co, k, k, Ci), 111.0, device=DEV)
        ops.conv2d_wgrad(g, xa, ops.Act(to_nhwc_bf16(gy, Co, 0, DEV)), dw, ws, reg=0.002, w=w32)
        torch.cuda.synchronize()
        close(dw, (w.grad + 0.002 * w.detach()).permute(3, 0, 1, 2), 2e-6 * math.sqrt(B * Ho * Wo) + 1e-6, "bf16 wgrad")


@pytest.mark.parametrize("case", [(2, 6, 5, 16, 8, 3), (1, 4, 4, 64, 32, 3), (2, 3, 3, 128, 64, 3), (2, 12, 40, 48, 72, 3),
                                  (1, 16, 32, 136, 64, 3),
                                  (1, 16, 64, 96, 72, 3), (2, 20, 32, 32, 64, 3),    # LDS-DMA kernel (upconv3x3g): 3 chunks / ragged N, ragged rows
                                  (2, 8, 32, 64, 32, 3), (1, 16, 32, 128, 48, 3),    # data gradient on conv3x3d (input in a concat buffer)
                                  (2, 16, 16, 64, 32, 3)])                           # 16-wide adjoint: tap-table kernel
def test_conv2d_transpose_bf16(U, case):
    ops = U.ops
    B, H, W, Ci, Co, k = case
    x = q16(torch.tensor(detrand.uniform(f"tx{case}", (B, Ci, H, W), -1, 1, np.float64))).requires_grad_(True)
    w = q16(torch.tensor(detrand.uniform(f"tw{case}", (k, k, Co, Ci), -1, 1, np.float64))).requires_grad_(True)
    b = torch.tensor(detrand.uniform(f"tb{case}", (Co,), -1, 1, np.float64))
    y = R.conv2d_transpose_same(x, w, b, 2)
    gy = q16(torch.tensor(detrand.uniform(f"tg{case}", tuple(y.shape), -1, 1, np.float64)))
    (y * gy).sum().backward()
    g = ops.geom(B, H, W, Ci, Co, k, 2)
    w32 = w.detach().permute(3, 0, 1, 2).contiguous().float().to(DEV)            # primary [Ci][k][k][Co]
    wprim = torch.empty((Ci, k * k, Co), dtype=torch.bfloat16, device=DEV)
    ops.cast_weight_bf16(w32, wprim, Ci, k * k, Co, Co)
    wt = torch.zeros((Co, k * k, Ci), dtype=torch.bfloat16, device=DEV)
    ops.transpose_cast_weight_bf16(w32, wt, Ci, k * k, Co, Ci)
    xa = ops.Act(to_nhwc_bf16(x.detach(), Ci, 0, DEV))
    ya = ops.Act(torch.full((B, 2 * H, 2 * W, 2 * Co), 512.0, dtype=torch.bfloat16, device=DEV), Co, Co)
    ops.conv2d_transpose_fwd(g, xa, wt, b.float().to(DEV), ya)
    torch.cuda.synchronize()
    close(ya.dense().permute(0, 3, 1, 2), y.detach(), 1e-2, "bf16 convT fwd")
    assert float(ya.base[..., :Co].float().min()) == 512.0
    gya = ops.Act(to_nhwc_bf16(gy, 2 * Co, Co, DEV), Co, Co)
    dxa = ops.Act(torch.full((B, H, W, Ci), 256.0, dtype=torch.bfloat16, device=DEV))
    ops.conv2d_transpose_dgrad(g, gya, wprim, dxa)
    torch.cuda.synchronize()
    close(dxa.dense().permute(0, 3, 1, 2), x.grad, 1e-2, "bf16 convT dgrad")
    ws = ops.Workspace(DEV)
    dw = torch.full((Ci, k, k, Co), 111.0, device=DEV)
    ops.conv2d_transpose_wgrad(g, xa, gya, dw, ws, reg=0.002, w=w32)
    torch.cuda.synchronize()
    close(dw, (w.grad + 0.002 * w.detach()).permute(3, 0, 1, 2), 2e-6 * math.sqrt(B * H * W) + 1e-6, "bf16 convT wgrad")


@pytest.mark.parametrize("shape", [(2, 12, 10, 16), (1, 4, 4, 1024), (2, 33, 31, 8)])
def test_batchnorm_bf16(U, shape):
    ops = U.ops
    B, H, W, Cc = shape
    x = q16(torch.tensor(detrand.uniform(f"bx{shape}", (B, Cc, H, W), -2, 2, np.float64))).requires_grad_(True)
    gamma = torch.tensor(detrand.uniform(f"bg{shape}", (Cc,), 0.5, 1.5, np.float64), requires_grad=True)
    beta = torch.tensor(detrand.uniform(f"bb{shape}", (Cc,), -0.5, 0.5, np.float64), requires_grad=True)
    y = R.bn_relu(x, gamma, beta, None, "bn", True, relu=True)
    gy = q16(torch.tensor(detrand.uniform(f"bgy{shape}", (B, Cc, H, W), -1, 1, np.float64)))
    (y * gy).sum().backward()
    ws = ops.Workspace(DEV)
    xa = ops.Act(to_nhwc_bf16(x.detach(), Cc + 8, 8, DEV), 8, Cc)
    aff = torch.empty(2 * Cc, device=DEV); saved = torch.empty(2 * Cc, device=DEV)
    g32, b32 = gamma.detach().float().to(DEV), beta.detach().float().to(DEV)
    ops.bn_stats(xa, g32, b32, aff, saved, ws)
    ya = ops.Act(torch.empty((B, H, W, Cc), dtype=torch.bfloat16, device=DEV))
    ops.bn_apply(xa, aff, ya, relu=True)
    torch.cuda.synchronize()
    close(ya.dense().permute(0, 3, 1, 2), y.detach(), 1e-2, "bf16 bn fwd")
    gya = ops.Act(to_nhwc_bf16(gy, Cc, 0, DEV))
    dxa = ops.Act(torch.empty((B, H, W, Cc), dtype=torch.bfloat16, device=DEV))
    dg = torch.empty(Cc, device=DEV); db = torch.empty(Cc, device=DEV)
    ops.bn_bwd(gya, xa, g32, aff, saved, dxa, dg, db, ws, relu=True)
    cs = torch.empty(Cc, device=DEV)
    ops.colsum(gya, cs, ws)
    torch.cuda.synchronize()
    close(dxa.dense().permute(0, 3, 1, 2), x.grad, 1e-2, "bf16 bn dx")
    close(dg, gamma.grad, 1e-4, "bf16 dgamma"); close(db, beta.grad, 1e-4, "bf16 dbeta")
    close(cs, gy.sum(dim=(0, 2, 3)), 1e-5, "bf16 colsum")


def test_batchnorm_passes_on_a_tensor_past_the_non_temporal_threshold(U):
    """Tensors of 100 MB and more take the non-temporal-load instances of the three BatchNorm passes (csrc/elementwise.hip: bn_nt) -
    the full-resolution levels of BASELINE.json configs[1].  [8, 256, 256, 128] bf16 = 134 MB: (1) the forward apply equals the apply of
    its two 67 MB halves (plain instances) bit for bit; (2) statistics, backward sums and dx against the same arithmetic in fp64."""
    ops = U.ops
    B, H, W, Cc = 8, 256, 256, 128
    g = torch.Generator(device=DEV); g.manual_seed(5)
    x = (torch.rand((B, H, W, Cc), device=DEV, generator=g) * 4 - 2).to(torch.bfloat16)
    gy = (torch.rand((B, H, W, Cc), device=DEV, generator=g) * 2 - 1).to(torch.bfloat16)
    assert x.numel() * 2 >= 100 << 20 and x[:B // 2].numel() * 2 < 100 << 20
    gamma = torch.rand(Cc, device=DEV, generator=g) + 0.5
    beta = torch.rand(Cc, device=DEV, generator=g) - 0.5
    ws = ops.Workspace(DEV)
    xa, gya = ops.Act(x), ops.Act(gy)
    aff = torch.empty(2 * Cc, device=DEV); saved = torch.empty(2 * Cc, device=DEV)
    ops.bn_stats(xa, gamma, beta, aff, saved, ws)
    ya = ops.Act(torch.empty_like(x))
    ops.bn_apply(xa, aff, ya, relu=True)
    halves = torch.empty_like(x)
    for h in range(2):
        sl = slice(h * B // 2, (h + 1) * B // 2)
        ops.bn_apply(ops.Act(x[sl]), aff, ops.Act(halves[sl]), relu=True)
    dxa = ops.Act(torch.empty_like(x))
    dg = torch.empty(Cc, device=DEV); db = torch.empty(Cc, device=DEV)
    ops.bn_bwd(gya, xa, gamma, aff, saved, dxa, dg, db, ws, relu=True)
    torch.cuda.synchronize()
    assert torch.equal(ya.dense(), halves)
    x64, g64 = x.double(), gy.double()
    mu = x64.mean(dim=(0, 1, 2)); var = x64.var(dim=(0, 1, 2), unbiased=False)
    rs = 1.0 / torch.sqrt(var + 1e-3)
    close(saved[:Cc], mu, 1e-6, "mean"); close(saved[Cc:], rs, 1e-6, "rstd")
    xh = (x64 - mu) * rs
    y64 = torch.relu(xh * gamma.double() + beta.double())
    close(ya.dense(), y64, 1e-2, "apply")
    gm = g64 * (ya.dense() > 0)             # the forward pass's own ReLU decisions (an fp64 mask flips where x * scale + shift rounds to 0)
    close(db, gm.sum(dim=(0, 1, 2)), 1e-5, "dbeta"); close(dg, (gm * xh).sum(dim=(0, 1, 2)), 1e-5, "dgamma")
    dx64 = gamma.double() * rs * (gm - gm.mean(dim=(0, 1, 2)) - xh * (gm * xh).mean(dim=(0, 1, 2)))
    close(dxa.dense(), dx64, 1e-2, "dx")


@pytest.mark.parametrize("B,H,W,Cc", [(2, 20, 37, 32), (1, 70, 200, 64), (1, 33, 256, 128), (1, 12, 300, 64), (2, 9, 20, 16),
                                      (1, 19, 512, 128), (2, 35, 523, 64), (1, 7, 257, 32)])
def test_head_and_loss_bf16(U, B, H, W, Cc):
    """Head forward / sigmoid+loss / head weight gradient on bf16 activations.  The matrix-core head (C in {32,64,128}; images
    wider than 256 pixels run in column blocks: 300, 512 = configs[3]'s width, 523 and 257 exercise the block seams) takes the
    kernel as bf16, so the oracle runs on the bf16-rounded kernel; C = 16 takes the direct kernels (fp32 kernel values: the
    rounded kernel is exactly representable there too)."""
    ops = U.ops
    x = q16(torch.tensor(detrand.uniform("hx16", (B, Cc, H, W), -1, 1, np.float64)))
    w = q16(torch.tensor(detrand.uniform("hw16", (6, 6, Cc, 2), -1, 1, np.float64))).requires_grad_(True)
    b = torch.tensor(detrand.uniform("hb16", (2,), -1, 1, np.float64))
    target = torch.tensor(detrand.uniform("ht16", (B, 2, H, W), 0, 1, np.float64))
    logits = R.conv2d_same(x, w, b, 1)
    logits.retain_grad()
    loss = R.data_loss(target, torch.sigmoid(logits), 0.9, B)
    loss.backward()
    xa = ops.Act(to_nhwc_bf16(x, Cc, 0, DEV))
    w8 = torch.zeros((8, 6, 6, Cc), device=DEV); w8[:2] = w.detach().permute(3, 0, 1, 2).float().to(DEV)
    b8 = torch.zeros(8, device=DEV); b8[:2] = b.float().to(DEV)
    la = ops.Act(torch.full((B, H, W, 4), 5.0, device=DEV))
    ops.head6x6_fwd(xa, w8, b8, la)
    pr = torch.empty((B, 2, H, W), device=DEV)
    dl = ops.Act(torch.full((B, H, W, 8), 9.0, dtype=torch.bfloat16, device=DEV))
    out = torch.zeros(4, device=DEV)
    ws = ops.Workspace(DEV)
    ops.sigmoid_loss(la, target.float().to(DEV), 0.9, 1.0 / (2 * H * W * B), pr, dl, out, ws)
    torch.cuda.synchronize()
    close(la.dense()[..., :2].permute(0, 3, 1, 2), logits.detach(), 2e-5, "head fwd (bf16 in)")
    assert abs(float(out[0]) - float(loss.detach())) <= 1e-5 * float(loss.detach())
    close(dl.dense()[..., :2].permute(0, 3, 1, 2), logits.grad, 1e-2, "dlogits bf16")
    assert float(dl.dense()[..., 2:].float().abs().max()) == 0.0
    # weight gradient from the bf16 dlogits the kernel itself produced
    gl = dl.dense()[..., :2].double().cpu().permute(0, 3, 1, 2)
    w2 = w.detach().clone().requires_grad_(True)
    (R.conv2d_same(x, w2, b, 1) * gl).sum().backward()
    dw = torch.zeros((8, 6, 6, Cc), device=DEV)
    ops.head6x6_wgrad(xa, dl, dw, ws)
    torch.cuda.synchronize()
    close(dw[:2], w2.grad.permute(3, 0, 1, 2), 2e-6 * math.sqrt(B * H * W) + 1e-6, "head wgrad (bf16 in)")
    # data gradient (matrix-core kernel where supported) from the same bf16 dlogits
    if ops.head6x6_dgrad_supported(W, Cc):
        x2 = x.clone().requires_grad_(True)
        (R.conv2d_same(x2, w.detach(), b, 1) * gl).sum().backward()
        dxa = ops.Act(torch.full((B, H, W, Cc + 8), 3.0, dtype=torch.bfloat16, device=DEV), 8, Cc)
        ops.head6x6_dgrad(dl, w8, dxa)
        torch.cuda.synchronize()
        close(dxa.dense().permute(0, 3, 1, 2), x2.grad, 1e-2, "head dgrad (bf16)")
        assert float(dxa.base[..., :8].float().min()) == 3.0 and float(dxa.base[..., :8].float().max()) == 3.0


@pytest.mark.parametrize("B,K,N", [(32, 8192, 4096), (3, 520, 72), (4, 4096, 8192)])
def test_dense_split_k(U, B, K, N):
    """Dense on a small batch through the split-K entry point (dl_models/u_net.py:259) and its data gradient."""
    ops = U.ops
    x = torch.tensor(detrand.uniform(f"dx{B,K}", (B, K), -1, 1))
    w = torch.tensor(detrand.uniform(f"dw{K,N}", (N, K), -1, 1)) * 0.05
    b = torch.tensor(detrand.uniform(f"db{N}", (N,), -1, 1))
    xa = ops.Act(x.view(B, 1, 1, K).to(DEV))
    ya = ops.Act(torch.full((B, 1, 1, N), 7.0, device=DEV))
    ws = ops.Workspace(DEV)
    ops.dense_fwd(xa, w.to(DEV), b.to(DEV), ya, ws)
    torch.cuda.synchronize()
    close(ya.base.view(B, N), x.double() @ w.double().t() + b.double(), 2e-6 * math.sqrt(K) + 1e-6, "dense fwd")
    # data gradient from the kernel as stored (no transposed copy): dx = dy . w, rows beyond B untouched, ld > K honoured
    dy = torch.tensor(detrand.uniform(f"ddy{B,N}", (B, N), -1, 1))
    assert ops.dense_dgrad_supported(B, K, N) == (B <= 32 and K % 4 == 0)
    if ops.dense_dgrad_supported(B, K, N):
        ldx = K + 8
        dxa = ops.Act(torch.full((B, 1, 1, ldx), 7.0, device=DEV), 0, K)
        ops.dense_dgrad(ops.Act(dy.view(B, 1, 1, N).to(DEV)), w.to(DEV), dxa, ws)
        torch.cuda.synchronize()
        close(dxa.base.view(B, ldx)[:, :K], dy.double() @ w.double(), 2e-6 * math.sqrt(N) + 1e-6, "dense dgrad")
        assert bool((dxa.base.view(B, ldx)[:, K:] == 7.0).all())
        again = ops.Act(torch.zeros((B, 1, 1, ldx), device=DEV), 0, K)
        ops.dense_dgrad(ops.Act(dy.view(B, 1, 1, N).to(DEV)), w.to(DEV), again, ws)
        assert torch.equal(again.base[..., :K], dxa.base[..., :K])        # fixed-order reduction: bit-reproducible


@pytest.mark.parametrize("case", [(2, 40, 70, 64, 160, 3, 1), (1, 33, 64, 96, 96, 3, 1), (2, 30, 64, 32, 64, 3, 1), (1, 16, 32, 16, 24, 3, 1),
                                  (5, 16, 16, 96, 128, 3, 1), (4, 24, 12, 96, 160, 3, 1),
                                  (2, 21, 60, 64, 64, 3, 1), (3, 64, 96, 64, 64, 3, 1), (16, 24, 512, 64, 64, 3, 1),
                                  (2, 21, 60, 32, 32, 3, 1), (3, 64, 96, 32, 32, 3, 1), (16, 24, 512, 32, 32, 3, 1)])
def test_conv_fused_column_statistics_bf16(U, case, monkeypatch):
    """Conv epilogue statistics (conv3x3g / conv3x3r<4,1>): the per-tile (sum, sum of squares) rows must add up to the
    statistics of the bf16 tensor the same launch stored, forward and data gradient, and BN statistics / bias gradients
    derived from them must match the stand-alone kernels."""
    ops = U.ops
    B, H, W, Ci, Co, k, s = case
    if W <= 16:
        ops.set_config(conv3x3g_pair=2)
    x, w, b = conv_data(case)
    g = ops.geom(B, H, W, Ci, Co, k, s)
    xa = ops.Act(to_nhwc_bf16(q16(x), Ci, 0, DEV))
    wf = torch.empty((Co, 9, Ci), dtype=torch.bfloat16, device=DEV)
    w32 = w.permute(3, 0, 1, 2).reshape(Co, 9, Ci).float().contiguous().to(DEV)
    ops.cast_weight_bf16(w32, wf, Co, 9, Ci, Ci)
    rows = ops.conv2d_colstat_rows(g, 0, xa)
    paired = W <= 16 and B >= 2 and Co > 64 and Ci % 32 == 0          # narrow images: two per tile
    if (Ci, Co) in ((64, 64), (32, 32)):          # strip kernel: one row per (job = vertical segment of a 32-wide strip, wave), whoever serves the job
        assert ops.conv3x3_kernel(g, 0, xa) == "conv3x3s"
        assert rows % 8 == 0 and 0 < rows <= 8 * B * ((W + 31) // 32) * ((H + 7) // 8)
    else:
        assert rows == ((B + 1) // 2 if paired else B) * ((H + 15) // 16) * ((W + 31) // 32)
    ya = ops.Act(torch.zeros((B, H, W, Co), dtype=torch.bfloat16, device=DEV))
    cst = torch.full((rows, Co, 2), 7.0, device=DEV)
    ops.conv2d_fwd_colstat(g, xa, wf, b.float().to(DEV), ya, cst)
    yref = ops.Act(torch.zeros((B, H, W, Co), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_fwd(g, xa, wf, b.float().to(DEV), yref)
    torch.cuda.synchronize()
    assert torch.equal(ya.base, yref.base)
    yd = ya.base.double()
    tot = cst.double().sum(dim=0)
    close(tot[:, 0], yd.sum(dim=(0, 1, 2)), 2e-6, "colstat sum")
    close(tot[:, 1], (yd * yd).sum(dim=(0, 1, 2)), 2e-6, "colstat sum of squares")
    cst2 = torch.full((rows, Co, 2), 7.0, device=DEV)          # the rows do not depend on which workgroup drew which tile
    ops.conv2d_fwd_colstat(g, xa, wf, b.float().to(DEV), ya, cst2)
    torch.cuda.synchronize()
    assert torch.equal(cst, cst2)
    # BN statistics from the rows == the stand-alone kernel
    gamma = torch.rand(Co, device=DEV) + 0.5
    beta = torch.rand(Co, device=DEV) - 0.5
    aff1, sav1 = torch.zeros(2 * Co, device=DEV), torch.zeros(2 * Co, device=DEV)
    aff2, sav2 = torch.zeros(2 * Co, device=DEV), torch.zeros(2 * Co, device=DEV)
    ws = ops.Workspace(DEV)
    ops.bn_stats(ya, gamma, beta, aff1, sav1, ws)
    ops.bn_stats_colstat(cst, rows, ya.P, Co, gamma, beta, aff2, sav2)
    torch.cuda.synchronize()
    close(aff2, aff1, 1e-5, "bn affine from colstat")
    close(sav2, sav1, 1e-5, "bn saved from colstat")
    # data gradient: statistics of dx, bias-gradient slice
    rows_d = ops.conv2d_colstat_rows(g, 1, ya)
    assert rows_d == rows
    wt = torch.empty((Ci, 9, Co), dtype=torch.bfloat16, device=DEV)
    ops.transpose_cast_weight_bf16(w32, wt, Co, 9, Ci, Co)
    dxa = ops.Act(torch.zeros((B, H, W, Ci), dtype=torch.bfloat16, device=DEV))
    cst_d = torch.full((rows, Ci, 2), 7.0, device=DEV)
    ops.conv2d_dgrad_colstat(g, ya, wt, dxa, cst_d)
    c0, cn = Ci // 2, Ci - Ci // 2
    out = torch.zeros(cn, device=DEV)
    ops.colsum_colstat(cst_d, rows, Ci, c0, cn, out)
    torch.cuda.synchronize()
    close(out, dxa.base.double().sum(dim=(0, 1, 2))[c0:], 2e-6, "bias gradient from colstat")


@pytest.mark.parametrize("case", [(20, 72, 80, 64, 192), (32, 64, 64, 64, 256), (16, 128, 128, 96, 128)])
def test_persistent_conv3x3p_equals_conv3x3g_bit_for_bit(U, case):
    """conv3x3p (persistent: continuous K loop across tiles, direct-store epilogue, per-workgroup statistics) against conv3x3g
    (one workgroup per tile) on layers with >= 512 tiles: the same MFMA sequence per output, so the stored tensors are
    IDENTICAL - forward and data gradient, with bias / addend, ragged rows / columns / channel tiles - and the fused
    column statistics (one row per pixel tile in both; which workgroup serves a tile is decided at run time) add up to the
    same totals and repeat bit for bit from run to run.  (conv3x3g itself is pinned to the oracle by the cases above and by the
    full-size window tests, which run through conv3x3p too.)"""
    ops = U.ops
    B, H, W, Ci, Co = case
    gen = torch.Generator(device=DEV); gen.manual_seed(B * 1000 + H)
    rnd = lambda *sh: ((torch.rand(sh, device=DEV, generator=gen) - 0.5) * 2).to(torch.bfloat16)
    g = ops.geom(B, H, W, Ci, Co, 3, 1)
    x, add = ops.Act(rnd(B, H, W, Ci)), ops.Act(rnd(B, H, W, Co))
    w, wt = rnd(Co, 9, Ci), rnd(Ci, 9, Co)
    bias = (torch.rand(Co, device=DEV, generator=gen) - 0.5)
    res = {}
    for p_on in (1, 0):
        ops.set_config(conv3x3p=p_on)
        rows_f, rows_d = ops.conv2d_colstat_rows(g, 0, x), ops.conv2d_colstat_rows(g, 1, add)
        y = ops.Act(torch.full((B, H, W, Co + 8), 3.0, dtype=torch.bfloat16, device=DEV), 0, Co)
        cs = torch.full((rows_f, Co, 2), 7.0, device=DEV)
        ops.conv2d_fwd_colstat(g, x, w, bias, y, cs, addend=add)
        dx = ops.Act(torch.full((B, H, W, Ci), 5.0, dtype=torch.bfloat16, device=DEV))
        csd = torch.full((rows_d, Ci, 2), 7.0, device=DEV)
        ops.conv2d_dgrad_colstat(g, y, wt, dx, csd)
        y2 = ops.Act(torch.empty((B, H, W, Co), dtype=torch.bfloat16, device=DEV))
        ops.conv2d_fwd(g, x, w, None, y2)
        torch.cuda.synchronize()
        res[p_on] = (rows_f, rows_d, y.base.clone(), cs.double().sum(0), dx.base.clone(), csd.double().sum(0), y2.base.clone())
    assert res[1][0] == res[0][0] == B * -(-H // 16) * -(-W // 32)        # one row of column statistics per pixel tile, either way
    ops.set_config(conv3x3p=1)
    assert ops.conv3x3_kernel(g, 0, x) == "conv3x3p" and (Ci <= 64 or ops.conv3x3_kernel(g, 1, add) == "conv3x3p")
    ops.set_config(conv3x3p=0)
    assert ops.conv3x3_kernel(g, 0, x) == "conv3x3g"
    assert torch.equal(res[1][2], res[0][2]) and float(res[1][2][..., Co:].float().min()) == 3.0
    assert torch.equal(res[1][4], res[0][4]) and torch.equal(res[1][6], res[0][6])
    yd = res[1][2][..., :Co].double()
    close(res[1][3][:, 0], yd.sum(dim=(0, 1, 2)), 2e-6, "colstat sum")
    close(res[1][3][:, 1], (yd * yd).sum(dim=(0, 1, 2)), 2e-6, "colstat sum of squares")
    close(res[1][3], res[0][3], 2e-6, "colstat totals, persistent vs per tile")
    if Ci > 64:
        close(res[1][5], res[0][5], 2e-6, "data-gradient colstat totals")
    # run-to-run: bit-identical
    ops.set_config(conv3x3p=1)
    y3 = ops.Act(torch.empty((B, H, W, Co), dtype=torch.bfloat16, device=DEV))
    cs3 = torch.zeros((res[1][0], Co, 2), device=DEV)
    ops.conv2d_fwd_colstat(g, x, w, None, y3, cs3)
    cs4 = torch.zeros_like(cs3)
    ops.conv2d_fwd_colstat(g, x, w, None, y3, cs4)
    torch.cuda.synchronize()
    assert torch.equal(y3.base, res[1][6]) and torch.equal(cs3, cs4)


@pytest.mark.parametrize("case", [(3, 72, 80, 128, 64), (2, 40, 50, 256, 64), (4, 64, 64, 96, 64)])
def test_conv3x3g_64_channel_tiles_serve_the_n64_layers(U, case):
    """N = 64 output channels with C > 64 (dec1.cb1a: 128 -> 64 behind the first skip concat, dl_models/u_net.py:309) run on
    conv3x3g's 64-channel-tile instantiation: forward (bias, addend, fused column statistics over its eight row groups) and the
    flip = 1 data gradient form, sizes that are not multiples of the 16 x 32 tile (the reference geometry's 72 x 80 among them),
    against conv3x3h for the same layer (set_config(conv3x3g=0)) - different K order, so to bf16 rounding - and the routing itself."""
    ops = U.ops
    B, H, W, Ci, Co = case
    gen = torch.Generator(device=DEV); gen.manual_seed(H * 100 + Ci)
    rnd = lambda *sh: ((torch.rand(sh, device=DEV, generator=gen) - 0.5) * 2).to(torch.bfloat16)
    g = ops.geom(B, H, W, Ci, Co, 3, 1)            # forward: C = Ci > 64, N = 64
    gd = ops.geom(B, H, W, Co, Ci, 3, 1)           # a layer whose DATA GRADIENT has C = Ci > 64, N = 64
    x, add = ops.Act(rnd(B, H, W, Ci + 8), 0, Ci), ops.Act(rnd(B, H, W, Co))
    w, wt = (rnd(Co, 9, Ci).float() * 0.1).to(torch.bfloat16), (rnd(Co, 9, Ci).float() * 0.1).to(torch.bfloat16)
    bias = (torch.rand(Co, device=DEV, generator=gen) - 0.5)
    old = ops.get_config()
    res = {}
    try:
        for on in (1, 0):
            ops.set_config(conv3x3g=on)
            assert ops.conv3x3_kernel(g, 0, x) == ("conv3x3g" if on else "conv3x3h")
            assert ops.conv3x3_kernel(gd, 1, x) == ("conv3x3g" if on else "conv3x3h")
            rows = ops.conv2d_colstat_rows(g, 0, x)
            y = ops.Act(torch.full((B, H, W, Co + 8), 3.0, dtype=torch.bfloat16, device=DEV), 0, Co)
            cs = torch.full((max(rows, 1), Co, 2), 7.0, device=DEV)
            if rows:
                ops.conv2d_fwd_colstat(g, x, w, bias, y, cs, addend=add)
            else:
                ops.conv2d_fwd(g, x, w, bias, y, addend=add)
            dx = ops.Act(torch.full((B, H, W, Co), 5.0, dtype=torch.bfloat16, device=DEV))
            ops.conv2d_dgrad(gd, x, wt, dx)
            torch.cuda.synchronize()
            res[on] = (rows, y.base.clone(), cs.double().sum(0) if rows else None, dx.base.clone())
    finally:
        ops.set_config(**old)
    assert res[1][0] > 0
    assert float(res[1][1][..., Co:].float().min()) == 3.0 and float(res[1][1][..., Co:].float().max()) == 3.0
    for a, b, what in ((res[1][1][..., :Co], res[0][1][..., :Co], "forward"), (res[1][3], res[0][3], "data gradient")):
        a, b = a.float(), b.float()
        assert float((a - b).abs().max()) <= 2.0 ** -6 * float(b.abs().max()), what
    yd = res[1][1][..., :Co].double()
    close(res[1][2][:, 0], yd.sum(dim=(0, 1, 2)), 2e-6, "colstat sum (64-channel tiles)")
    close(res[1][2][:, 1], (yd * yd).sum(dim=(0, 1, 2)), 2e-6, "colstat sum of squares (64-channel tiles)")
    # and against the oracle on the first image
    w_hwio = w.double().cpu().view(Co, 3, 3, Ci).permute(1, 2, 3, 0)
    want = R.conv2d_same(x.base[:1, ..., :Ci].double().cpu().permute(0, 3, 1, 2), w_hwio, bias.double().cpu(), 1) + \
        add.base[:1].double().cpu().permute(0, 3, 1, 2)
    got = res[1][1][:1, ..., :Co].double().cpu().permute(0, 3, 1, 2)
    assert float((got - want).abs().max()) <= 1e-2 * float(want.abs().max())


@pytest.mark.parametrize("case", [(20, 40, 72, 128, 256), (32, 64, 32, 96, 256), (8, 128, 128, 128, 64)])
def test_persistent_upconv3x3q_equals_upconv3x3g_bit_for_bit(U, case):
    """upconv3x3q (persistent) against upconv3x3g (one workgroup per tile) on layers with >= 512 tiles - Conv2DTranspose forward
    (bias, output into a concat buffer) and the strided convolution's data gradient accumulating in place (addend == out): the
    same MFMA sequence per output, so the tensors are identical; ragged coarse rows / columns, two-chunk K loop."""
    ops = U.ops
    B, H, W, Ci, Co = case                 # coarse H x W with Ci channels -> 2H x 2W with Co channels
    gen = torch.Generator(device=DEV); gen.manual_seed(B * 1000 + H)
    rnd = lambda *sh: ((torch.rand(sh, device=DEV, generator=gen) - 0.5) * 2).to(torch.bfloat16)
    x = ops.Act(rnd(B, H, W, Ci))
    wt = rnd(Co, 9, Ci)                    # [out channel][tap][in channel]: the orientation both call sites pass
    bias = (torch.rand(Co, device=DEV, generator=gen) - 0.5)
    skip = rnd(B, 2 * H, 2 * W, Co)
    gt = ops.geom(B, H, W, Ci, Co, 3, 2)                     # Conv2DTranspose: coarse in, fine out
    gs = ops.geom(B, 2 * H, 2 * W, Co, Ci, 3, 2)             # strided Conv2D (fine Co -> coarse Ci): its data gradient is the same kernel
    res = {}
    for q_on in (1, 0):
        ops.set_config(upconv3x3q=q_on)
        y = ops.Act(torch.full((B, 2 * H, 2 * W, 2 * Co), 3.0, dtype=torch.bfloat16, device=DEV), Co, Co)
        ops.conv2d_transpose_fwd(gt, x, wt, bias, y)
        acc = ops.Act(skip.clone())
        ops.conv2d_dgrad(gs, x, wt, acc, addend=acc)
        torch.cuda.synchronize()
        res[q_on] = (y.base.clone(), acc.base.clone())
    assert torch.equal(res[1][0], res[0][0]) and float(res[1][0][..., :Co].float().min()) == 3.0
    assert torch.equal(res[1][1], res[0][1])
    assert float((res[1][1].float() - skip.float()).abs().max()) > 0.5          # the accumulation did happen


def test_cast_weights_batched_bf16(U):
    """The two-launch batched work-copy refresh equals the per-layer calls bit for bit."""
    ops = U.ops
    shapes = [(16, 9, 8), (72, 9, 40), (128, 1, 64), (8, 36, 64), (128, 9, 64), (64, 9, 192), (192, 3, 128)]   # the last three: fused one-read kernel
    ws, same1, tr1, same2, tr2, ent = [], [], [], [], [], []
    for i, (N, T, Cc) in enumerate(shapes):
        w = torch.tensor(detrand.uniform(f"cwb{i}", (N, T, Cc), -1, 1)).to(DEV)
        ws.append(w)
        same1.append(torch.zeros((N, T, Cc), dtype=torch.bfloat16, device=DEV)); tr1.append(torch.zeros((Cc, T, N), dtype=torch.bfloat16, device=DEV))
        same2.append(torch.ones((N, T, Cc), dtype=torch.bfloat16, device=DEV)); tr2.append(torch.ones((Cc, T, N), dtype=torch.bfloat16, device=DEV))
        ops.cast_weight_bf16(w, same1[-1], N, T, Cc, Cc)
        ops.transpose_cast_weight_bf16(w, tr1[-1], N, T, Cc, N)
        ent.append((w, same2[-1], tr2[-1] if i != 2 else None, N, T, Cc, Cc, N))
    table = ops.make_cast_table(ent, DEV)
    ops.cast_weights_batched(table)
    torch.cuda.synchronize()
    for i in range(len(shapes)):
        assert torch.equal(same1[i], same2[i])
        if i != 2:
            assert torch.equal(tr1[i], tr2[i])
        else:
            assert float(tr2[i].float().min()) == 1.0      # NULL destination: untouched


@pytest.mark.parametrize("case", [(2, 40, 70, 8, 64, 3, 1), (1, 32, 64, 8, 64, 3, 1), (3, 7, 33, 8, 64, 3, 1), (2, 20, 40, 8, 128, 3, 1)])
def test_stem_conv_bf16(U, case):
    """First layer (8 stored input channels -> 64, no addend): the direct-operand stem kernel (stem3x3.hip), ragged tiles,
    a strided output buffer, and agreement with the general kernel bit for bit (both accumulate the 72 products in fp32 MFMA
    order per output; equality is not guaranteed in general, so compare against the oracle and bound the difference)."""
    ops = U.ops
    B, H, W, Ci, Co, k, s = case
    x, w, b = conv_data(case)
    x, w = q16(x), q16(w)
    y = R.conv2d_same(x, w, b, s)
    g = ops.geom(B, H, W, Ci, Co, k, s)
    xa = ops.Act(to_nhwc_bf16(x, Ci, 0, DEV))
    w32 = w.permute(3, 0, 1, 2).contiguous().float().to(DEV)
    wh = torch.empty((Co, 9, Ci), dtype=torch.bfloat16, device=DEV)
    ops.cast_weight_bf16(w32, wh, Co, 9, Ci, Ci)
    ya = ops.Act(torch.full((B, H, W, Co + 8), 512.0, dtype=torch.bfloat16, device=DEV), 0, Co)
    ops.conv2d_fwd(g, xa, wh, b.float().to(DEV), ya)
    torch.cuda.synchronize()
    close(ya.dense().permute(0, 3, 1, 2), y, 1e-2, "stem fwd")
    assert float(ya.base[..., Co:].float().min()) == 512.0 and float(ya.base[..., Co:].float().max()) == 512.0
    again = ops.Act(torch.zeros((B, H, W, Co), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_fwd(g, xa, wh, b.float().to(DEV), again)
    assert torch.equal(again.base, ya.base[..., :Co])


@pytest.mark.parametrize("case", [(3, 9, 10, 32, 64), (2, 16, 24, 64, 32), (5, 7, 12, 128, 128), (2, 8, 8, 256, 256), (4, 12, 20, 8, 32),
                                  (2, 33, 17, 16, 96), (2, 10, 12, 48, 64), (2, 6, 10, 96, 32), (2, 5, 8, 64, 96)])
def test_pw1x1_kernel_against_the_tap_table_kernel_and_the_oracle(U, case):
    """The register-streaming 1x1 kernel (pw1x1.hip) in every form the residual graphs launch it (dl_models/res_ae.py:310-371,
    :453-514): Conv2D forward at stride 1 and 2 (bias, fused column statistics, addend), its data gradients (stride 2: scatter to
    the even pixels, zeros - or the untouched in-place addend - elsewhere, odd sizes included), Conv2DTranspose forward at
    stride 1 and at stride 2 'valid' (bias in the three other pixels of every cell) and its data gradient.  Same arithmetic as the
    tap-table kernel (one fp32 MFMA chain over C, one rounding): identical bits; plus the oracle on the forward results.
    Channel counts the kernel has no instantiation for (48, 96 input channels of either direction) must be served by the tap-table
    kernel, not refused."""
    ops = U.ops
    B, H, W, Ci, Co = case
    gen = torch.Generator(device=DEV); gen.manual_seed(B * 100 + Ci)
    rnd = lambda *sh: ((torch.rand(sh, device=DEV, generator=gen) - 0.5) * 2).to(torch.bfloat16)
    old = ops.get_config()
    out = {}
    try:
        for on in (1, 0):
            ops.set_config(pw1x1=on)
            gen.manual_seed(B * 100 + Ci)
            res = []
            for s in (1, 2):
                g = ops.geom(B, H, W, Ci, Co, 1, s)
                Ho, Wo = -(-H // s), -(-W // s)
                x = ops.Act(rnd(B, H, W, Ci + 8), 0, Ci)
                if Ci == 8:
                    x.base[..., 2:8] = 0
                w, wt = (rnd(Co, 1, Ci).float() * 0.2).to(torch.bfloat16), (rnd(Ci, 1, Co).float() * 0.2).to(torch.bfloat16)
                bias = torch.rand(Co, device=DEV, generator=gen) - 0.5
                add = ops.Act(rnd(B, Ho, Wo, Co))
                # Conv2D forward: plain, with addend, with column statistics
                y = ops.Act(torch.full((B, Ho, Wo, Co + 8), 3.0, dtype=torch.bfloat16, device=DEV), 0, Co)
                ops.conv2d_fwd(g, x, w, bias, y, addend=add)
                rows = ops.conv2d_colstat_rows(g, 0, x)
                assert rows > 0
                y2 = ops.Act(torch.empty((B, Ho, Wo, Co), dtype=torch.bfloat16, device=DEV))
                cst = torch.full((rows, Co, 2), 7.0, device=DEV)
                ops.conv2d_fwd_colstat(g, x, w, bias, y2, cst)
                # Conv2D data gradient: written, and accumulated in place behind an earlier writer
                gy = ops.Act(rnd(B, Ho, Wo, Co))
                dx = ops.Act(torch.full((B, H, W, Ci), 5.0, dtype=torch.bfloat16, device=DEV))
                ops.conv2d_dgrad(g, gy, wt, dx)
                acc0 = rnd(B, H, W, Ci)
                dacc = ops.Act(acc0.clone())
                ops.conv2d_dgrad(g, gy, wt, dacc, addend=dacc)
                # Conv2DTranspose(Ci -> Co, 1x1, stride s) on the H x W grid: forward (+ statistics), data gradient
                gt = ops.geom(B, H, W, Ci, Co, 1, s)
                wtr, wprim = (rnd(Co, 1, Ci).float() * 0.2).to(torch.bfloat16), (rnd(Ci, 1, Co).float() * 0.2).to(torch.bfloat16)
                yt = ops.Act(torch.empty((B, H * s, W * s, Co), dtype=torch.bfloat16, device=DEV))
                rows_t = ops.conv2d_transpose_colstat_rows(gt, x)
                assert rows_t > 0
                cst_t = torch.full((rows_t, Co, 2), 7.0, device=DEV)
                ops.conv2d_transpose_fwd_colstat(gt, x, wtr, bias, yt, cst_t)
                gyt = ops.Act(rnd(B, H * s, W * s, Co))
                dxt = ops.Act(torch.empty((B, H, W, Ci), dtype=torch.bfloat16, device=DEV))
                ops.conv2d_transpose_dgrad(gt, gyt, wprim, dxt)
                torch.cuda.synchronize()
                res.append(dict(y=y.base.clone(), y2=y2.base.clone(), cs=cst.double().sum(0), dx=dx.base.clone(), dacc=dacc.base.clone(),
                                acc0=acc0, yt=yt.base.clone(), cst=cst_t.double().sum(0), dxt=dxt.base.clone(), x=x, w=w, wtr=wtr,
                                bias=bias, add=add.base.clone()))
            out[on] = res
    finally:
        ops.set_config(**old)
    for s_i, (a, b) in enumerate(zip(out[1], out[0])):
        for k in ("y", "y2", "dx", "dacc", "yt", "dxt"):
            assert torch.equal(a[k], b[k]), (k, s_i + 1, float((a[k].float() - b[k].float()).abs().max()))
        assert float(a["y"][..., Co:].float().min()) == 3.0                       # the neighbouring channels of the buffer are untouched
        for k, t in (("cs", a["y2"]), ("cst", a["yt"])):
            td = t.double()
            close(a[k][:, 0], td.sum(dim=(0, 1, 2)), 2e-6, f"{k} sum")
            close(a[k][:, 1], (td * td).sum(dim=(0, 1, 2)), 2e-6, f"{k} sum of squares")
        # oracle: forward (+ addend) and the transposed forward
        s = s_i + 1
        xo = a["x"].base[..., :Ci].double().cpu().permute(0, 3, 1, 2)
        w_hwio = a["w"].double().cpu().view(Co, 1, 1, Ci).permute(1, 2, 3, 0)
        want = R.conv2d_same(xo, w_hwio, a["bias"].double().cpu(), s) + a["add"].double().cpu().permute(0, 3, 1, 2)
        got = a["y"][..., :Co].double().cpu().permute(0, 3, 1, 2)
        assert float((got - want).abs().max()) <= 1e-2 * float(want.abs().max())
        k_hwoi = a["wtr"].double().cpu().view(Co, 1, 1, Ci).permute(1, 2, 0, 3)     # [kh,kw,O,I]
        want_t = R.conv2d_transpose_same(xo, k_hwoi, a["bias"].double().cpu(), s)
        got_t = a["yt"].double().cpu().permute(0, 3, 1, 2)
        assert float((got_t - want_t).abs().max()) <= 1e-2 * float(want_t.abs().max())


def test_sgd_and_nadam_steps_against_the_oracle(U):
    """The optimizers main_training.py:164-169 selects besides Adam: three steps of SGD(learning_rate) and of Nadam(learning_rate)
    (momentum schedule, running product of the schedule) through Trainer-level state (engine.adam_begin / adam_range) against the
    oracle's restatement."""
    ops = U.ops
    n = 1000 + 3                                     # not a multiple of 4: the tail path
    gen = torch.Generator(); gen.manual_seed(9)
    theta0 = torch.randn(n, generator=gen)
    gs = [torch.randn(n, generator=gen) * 0.1 for _ in range(3)]
    lr = 1e-2
    # SGD
    th = theta0.clone().to(DEV)
    want = theta0.double()
    for g in gs:
        ops.sgd(th, g.to(DEV), lr)
        want = R.sgd_update(want, g.double(), lr)
    torch.cuda.synchronize()
    close(th, want, 1e-6, "sgd")
    # Nadam through the engine's step bookkeeping
    eng = U.UNetEngine(32, 32, 2, F0=8, device=DEV)
    tr = U.Trainer(eng, lr=lr, optimizer="Nadam")
    assert eng.optimizer == "nadam" and not tr.use_graph
    g0 = torch.Generator(); g0.manual_seed(1)
    eng.reset_parameters(g0)
    t_ref, m, v, ms = eng.theta.double().cpu(), torch.zeros(eng.theta.numel(), dtype=torch.float64), torch.zeros(eng.theta.numel(), dtype=torch.float64), 1.0
    for step in range(1, 4):
        gr = torch.randn(eng.grad.numel(), generator=gen) * 0.1
        eng.grad.copy_(gr.to(DEV))
        eng.adam_step(lr)
        t_ref, m, v, ms = R.nadam_update(t_ref, gr.double(), m, v, step, lr, ms)
    torch.cuda.synchronize()
    assert eng.adam_t == 3 and abs(eng._shared["m_schedule"] - ms) <= 1e-15
    close(eng.theta, t_ref, 2e-6, "nadam theta")
    close(eng.adam_m, m, 2e-6, "nadam m")
    with pytest.raises(ValueError):
        U.Trainer(eng, optimizer="lamb")


@pytest.mark.parametrize("case", [(4, 36, 40, 128, 128, 3, 1), (3, 18, 20, 64, 96, 3, 1), (5, 9, 10, 256, 128, 3, 1), (2, 20, 24, 64, 128, 6, 1),
                                  (3, 18, 20, 64, 64, 6, 2), (2, 9, 11, 32, 64, 3, 2), (2, 12, 16, 24, 40, 5, 1)])
def test_small_problem_tap_table_kernel_equals_the_general_one(U, case):
    """igemm2 (64-pixel tiles, two K chunks in flight: the reference geometry's small levels, main_training.py:27; 6 x 6 kernels,
    dl_models/u_net.py:40-45) against the general tap-table kernel on the same layers: the same MFMA chain over the same K order,
    so identical bits - Conv2D forward (bias, addend, fused column statistics), data gradient, Conv2DTranspose forward and data
    gradient (stride 2: the four parity classes in one grid), odd sizes, channel counts that are not multiples of 64."""
    _tap_table_ab(U, case, "igemm2", {})


def _tap_table_ab(U, case, switch, fixed):
    ops = U.ops
    B, H, W, Ci, Co, k, s = case
    gen = torch.Generator(device=DEV); gen.manual_seed(H * 100 + Ci + k)
    rnd = lambda *sh: ((torch.rand(sh, device=DEV, generator=gen) - 0.5) * 2).to(torch.bfloat16)
    old = ops.get_config()
    out = {}
    try:
        for on in (1, 0):
            ops.set_config(**{switch: on}, conv3x3=0, conv3x3g_pair=0, conv3x3d=0, upconv3x3g=0, upconv3x3q=0, **fixed)     # everything on the tap-table path
            gen.manual_seed(H * 100 + Ci + k)
            g = ops.geom(B, H, W, Ci, Co, k, s)
            Ho, Wo = -(-H // s), -(-W // s)
            x = ops.Act(rnd(B, H, W, Ci + 8), 0, Ci)
            w, wt = (rnd(Co, k * k, Ci).float() * 0.2).to(torch.bfloat16), (rnd(Ci, k * k, Co).float() * 0.2).to(torch.bfloat16)
            bias = torch.rand(Co, device=DEV, generator=gen) - 0.5
            add = ops.Act(rnd(B, Ho, Wo, Co))
            y = ops.Act(torch.full((B, Ho, Wo, Co + 8), 3.0, dtype=torch.bfloat16, device=DEV), 0, Co)
            ops.conv2d_fwd(g, x, w, bias, y, addend=add)
            rows = ops.conv2d_colstat_rows(g, 0, x)
            y2 = ops.Act(torch.empty((B, Ho, Wo, Co), dtype=torch.bfloat16, device=DEV))
            cst = None
            if rows:
                cst = torch.full((rows, Co, 2), 7.0, device=DEV)
                ops.conv2d_fwd_colstat(g, x, w, bias, y2, cst)
            else:
                ops.conv2d_fwd(g, x, w, bias, y2)
            gy = ops.Act(rnd(B, Ho, Wo, Co))
            dx = ops.Act(torch.full((B, H, W, Ci), 5.0, dtype=torch.bfloat16, device=DEV))
            ops.conv2d_dgrad(g, gy, wt, dx)
            # the transposed layer Ci -> Co on the H x W grid
            gt = ops.geom(B, H, W, Ci, Co, k, s)
            wtr, wprim = (rnd(Co, k * k, Ci).float() * 0.2).to(torch.bfloat16), (rnd(Ci, k * k, Co).float() * 0.2).to(torch.bfloat16)
            yt = ops.Act(torch.empty((B, H * s, W * s, Co), dtype=torch.bfloat16, device=DEV))
            ops.conv2d_transpose_fwd(gt, x, wtr, bias, yt)
            gyt = ops.Act(rnd(B, H * s, W * s, Co))
            dxt = ops.Act(torch.empty((B, H, W, Ci), dtype=torch.bfloat16, device=DEV))
            ops.conv2d_transpose_dgrad(gt, gyt, wprim, dxt)
            torch.cuda.synchronize()
            out[on] = dict(y=y.base.clone(), y2=y2.base.clone(), cs=None if cst is None else cst.double().sum(0), dx=dx.base.clone(),
                           yt=yt.base.clone(), dxt=dxt.base.clone(), rows=rows)
    finally:
        ops.set_config(**old)
    for kk in ("y", "y2", "dx", "yt", "dxt"):
        assert torch.equal(out[1][kk], out[0][kk]), (kk, float((out[1][kk].float() - out[0][kk].float()).abs().max()))
    assert float(out[1]["y"][..., Co:].float().min()) == 3.0
    if out[1]["cs"] is not None:
        td = out[1]["y2"].double()
        close(out[1]["cs"][:, 0], td.sum(dim=(0, 1, 2)), 2e-6, "colstat sum")
        close(out[1]["cs"][:, 1], (td * td).sum(dim=(0, 1, 2)), 2e-6, "colstat sum of squares")
        assert out[1]["rows"] >= out[0]["rows"]                   # 64-pixel tiles: at least as many rows as the 128-pixel kernel


def test_parked_split_k_reductions_equal_the_immediate_ones_bit_for_bit(U):
    """ops.ReduceBatch (unetrir_*_wgrad_partials_bf16 + unetrir_splitk_reduce_batched): weight gradients of a mix of layers - 3x3 at
    stride 1 and 2, 1x1, a transposed layer, with and without the l2 term, outputs from 2 K to 150 K floats, few and many slabs -
    parked in one batch and reduced by ONE launch against the same weight gradients reduced layer by layer: identical bits (same code
    over the same slab order per output element), in both the narrow and the wide form of the reduction."""
    ops = U.ops
    gen = torch.Generator(device=DEV); gen.manual_seed(11)
    rnd = lambda *sh: ((torch.rand(sh, device=DEV, generator=gen) - 0.5) * 2).to(torch.bfloat16)
    #        B   H    W   Cin  Cout k  s  transposed  reg
    cases = [(8, 64, 64, 64, 64, 3, 1, False, 0.0), (8, 64, 64, 128, 64, 3, 1, False, 0.0), (4, 64, 64, 64, 128, 3, 2, False, 2e-3),
             (4, 32, 32, 128, 64, 3, 2, True, 2e-3), (8, 32, 32, 32, 32, 1, 1, False, 0.0), (8, 16, 16, 256, 256, 1, 1, False, 0.0),
             (2, 16, 16, 512, 512, 3, 1, False, 0.0), (8, 32, 32, 64, 32, 1, 2, False, 0.0), (3, 24, 40, 64, 96, 3, 1, False, 0.0)]
    layers = []
    for B, H, W, Ci, Co, k, s, tr, reg in cases:
        g = ops.geom(B, H, W, Ci, Co, k, s)
        if tr:      # Conv2DTranspose(Ci -> Co) on H x W: x [B,H,W,Ci], dy [B,sH,sW,Co], kernel [Ci][k][k][Co]
            x, dy = ops.Act(rnd(B, H, W, Ci)), ops.Act(rnd(B, H * s, W * s, Co))
            w = torch.rand((Ci, k, k, Co), device=DEV, generator=gen) - 0.5
        else:
            x, dy = ops.Act(rnd(B, H, W, Ci)), ops.Act(rnd(B, -(-H // s), -(-W // s), Co))
            w = torch.rand((Co, k, k, Ci), device=DEV, generator=gen) - 0.5
        layers.append((g, x, dy, w, reg, tr))
    ws = ops.Workspace(DEV)
    want = []
    for g, x, dy, w, reg, tr in layers:
        dw = torch.full_like(w, 7.0)
        (ops.conv2d_transpose_wgrad if tr else ops.conv2d_wgrad)(g, x, dy, dw, ws, reg=reg, w=w)
        want.append(dw)
    rb = ops.ReduceBatch(DEV, 8 << 20)                    # small on purpose: the arena grows / the batch flushes itself when it is full
    got = []
    for g, x, dy, w, reg, tr in layers:
        dw = torch.full_like(w, 7.0)
        (ops.conv2d_transpose_wgrad if tr else ops.conv2d_wgrad)(g, x, dy, dw, ws, reg=reg, w=w, defer=rb)
        got.append(dw)
    forms = {(int(rb.descs[i].nsplit >= 32), min(int(rb.descs[i].nsplit), 8)) for i in range(len(rb))}
    assert len(rb) >= 3 and len(forms) >= 2, (len(rb), forms)
    rb.flush()
    torch.cuda.synchronize()
    assert len(rb) == 0
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), (cases[i], float((a - b).abs().max()))
    # a descriptor table that fills up before the arena does: the batch flushes itself and goes on
    rb = ops.ReduceBatch(DEV, 1 << 30, capacity=2)
    got = []
    for g, x, dy, w, reg, tr in layers:
        dw = torch.full_like(w, 7.0)
        (ops.conv2d_transpose_wgrad if tr else ops.conv2d_wgrad)(g, x, dy, dw, ws, reg=reg, w=w, defer=rb)
        got.append(dw)
        assert len(rb) <= 2
    rb.flush()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), (cases[i], float((a - b).abs().max()))
    # a threshold on the slab-set size: larger sets reduce at once, smaller ones are parked
    needs = sorted((ops.conv2d_transpose_wgrad_ws_bytes if tr else ops.conv2d_wgrad_ws_bytes)(g) for g, x, dy, w, reg, tr in layers)
    assert needs[0] < needs[-1]
    rb = ops.ReduceBatch(DEV, 1 << 30, park_max_bytes=needs[len(needs) // 2])          # the median slab-set size of these layers
    got = []
    for g, x, dy, w, reg, tr in layers:
        dw = torch.full_like(w, 7.0)
        (ops.conv2d_transpose_wgrad if tr else ops.conv2d_wgrad)(g, x, dy, dw, ws, reg=reg, w=w, defer=rb)
        got.append(dw)
    assert len(rb) < len(cases)                 # the larger half was reduced at once
    rb.flush()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), (cases[i], float((a - b).abs().max()))
    # one big batch too (every layer parked, one flush): an arena that holds them all
    rb = ops.ReduceBatch(DEV, 1 << 30)
    got = []
    for g, x, dy, w, reg, tr in layers:
        dw = torch.full_like(w, 7.0)
        (ops.conv2d_transpose_wgrad if tr else ops.conv2d_wgrad)(g, x, dy, dw, ws, reg=reg, w=w, defer=rb)
        got.append(dw)
    parked = len(rb)
    rb.flush()
    torch.cuda.synchronize()
    assert parked >= len(cases) - 2            # all but the layers whose kernel writes dw directly
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), (cases[i], float((a - b).abs().max()))
