"""Oracle parity AT THE LAUNCHED SHAPES of BASELINE.json configs[1] (batch 32, 256 x 256, number_filters_0 = 64, bf16 storage).

A convolution is local: a window of one image of a full-size layer - with its halo, at tile seams, at the image border and
in the zero-padded region - is seconds on the fp64 CPU oracle.  Every hot kernel of the step runs here ONCE at its real
size through the C ABI (real grid, real tile counts, real pixel strides), and windows of its output are compared with
oracle.torch_ref.conv2d_same / conv2d_transpose_same evaluated in fp64 on the cropped input:

  image 0 and image 31 (first / last of the batch); the top-left corner (border + padding), the bottom-right corner, a
  window straddling the 16 x 32 / 8 x 32 tile seams in the interior, and (synthetic-batch layout) the zero-padded region.

Weight gradients are sums over ALL pixels: a handful of (output channel, tap, input channel) entries are compared with a
direct fp64 sum over the whole batch.  Tolerances are those of the small per-kernel cases: 1e-2 of the tensor's scale for
bf16 outputs (observed ~2e-3: one bf16 rounding), 2e-6 * sqrt(K) for fp32 weight gradients.
Reference call sites: dl_models/u_net.py:269-276 (strided Conv2D), :297-304 (Conv2DTranspose), :366 (3x3 block conv).
"""
import math

import numpy as np
import pytest
import torch

from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B = 32
IMAGES = (0, B - 1)


@pytest.fixture(scope="module")
def U():
    import unet_rir_amd
    unet_rir_amd._lib.lib()
    return unet_rir_amd


def _rand_bf16(shape, seed, scale=1.0, zero_pad=False):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    t = ((torch.rand(shape, device=DEV, generator=g) - 0.5) * 2 * scale).to(torch.bfloat16)
    if zero_pad:          # the synthetic-batch layout: rows >= ceil(0.896 H) and columns >= ceil(0.944 W) exactly zero
        H, W = shape[1], shape[2]
        t[:, math.ceil(0.896 * H):] = 0
        t[:, :, math.ceil(0.944 * W):] = 0
    return t


def _weights(ops, Co, Ci, seed):
    """fp32 master [Co][9][Ci] holding bf16-representable values, its bf16 copy and the transposed bf16 copy [Ci][9][Co]."""
    w32 = _rand_bf16((Co, 9, Ci), seed, 0.1).float().contiguous()
    wh = torch.empty((Co, 9, Ci), dtype=torch.bfloat16, device=DEV)
    wt = torch.empty((Ci, 9, Co), dtype=torch.bfloat16, device=DEV)
    ops.cast_weight_bf16(w32, wh, Co, 9, Ci, Ci)
    ops.transpose_cast_weight_bf16(w32, wt, Co, 9, Ci, Co)
    return w32, wh, wt


def _windows(H, W, step=1):
    """(r0, r1, c0, c1) output windows; even offsets so that they are also valid origins of a stride-2 grid."""
    hh, ww = min(H, 24), min(W, 40)
    out = [(0, hh, 0, ww), (H - hh, H, W - ww, W)]
    if H <= 64 and W <= 64:          # the small levels: one more window across the 16-row / 32-column seams of the interior
        out.append((H // 2 - 8, H // 2 + 8, max(W // 2 - 16, 0), min(W // 2 + 16, W)))
    if H >= 64 and W >= 96:
        out.append((8, 40, 24, 72))                                   # straddles 8- / 16-row and 32-column tile seams
        out.append((H // 2 - 12, H // 2 + 12, W // 2 - 20, W // 2 + 20))
    return out


def _check(got, want, what, tol=1e-2):
    scale = float(want.abs().max()) + 1e-30
    err = float((got.double().cpu() - want).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _nchw64(t):
    return t.double().cpu().permute(0, 3, 1, 2)


def _conv_windows(y_gpu, x_gpu, w_hwio, bias, stride, what, halo=1):
    """y = conv_same(x, w) at full size vs the oracle on cropped input, window by window.  Crops start at rows / columns that
    keep the stride phase; output pixels whose receptive field leaves the crop at an INTERIOR crop edge are not compared."""
    Bn, Ho, Wo, _ = y_gpu.shape
    _, H, W, _ = x_gpu.shape
    for img in IMAGES:
        for (r0, r1, c0, c1) in _windows(Ho, Wo):
            # input crop covering the window's receptive fields plus a margin; clipped at the true image border
            ir0, ir1 = max(r0 * stride - 2 * stride, 0), min(r1 * stride + 2 * stride, H)
            ic0, ic1 = max(c0 * stride - 2 * stride, 0), min(c1 * stride + 2 * stride, W)
            xc = _nchw64(x_gpu[img:img + 1, ir0:ir1, ic0:ic1])
            yc = R.conv2d_same(xc, w_hwio, bias, stride)             # SAME on the crop: exact where the crop edge is the image edge
            # map the window into crop coordinates and drop one output pixel next to interior crop edges
            o_r0, o_c0 = ir0 // stride, ic0 // stride
            a0 = r0 + (1 if ir0 > 0 and r0 == o_r0 else 0)
            b0 = c0 + (1 if ic0 > 0 and c0 == o_c0 else 0)
            a1 = r1 - (1 if ir1 < H and r1 * stride >= ir1 else 0)
            b1 = c1 - (1 if ic1 < W and c1 * stride >= ic1 else 0)
            want = yc[0, :, a0 - o_r0:a1 - o_r0, b0 - o_c0:b1 - o_c0]
            got = y_gpu[img, a0:a1, b0:b1].permute(2, 0, 1)
            _check(got, want, f"{what} image {img} window rows {a0}:{a1} cols {b0}:{b1}")


def _convT_windows(y_gpu, x_gpu, k_hwoi, bias, what):
    """y = conv2d_transpose_same(x, w, stride 2): output window [r0:r1) x [c0:c1) depends on input rows r0/2 - 1 .. r1/2."""
    _, Ho, Wo, _ = y_gpu.shape
    _, H, W, _ = x_gpu.shape
    for img in IMAGES:
        for (r0, r1, c0, c1) in _windows(Ho, Wo):
            ir0, ir1 = max(r0 // 2 - 2, 0), min(r1 // 2 + 2, H)
            ic0, ic1 = max(c0 // 2 - 2, 0), min(c1 // 2 + 2, W)
            xc = _nchw64(x_gpu[img:img + 1, ir0:ir1, ic0:ic1])
            yc = R.conv2d_transpose_same(xc, k_hwoi, bias, 2)         # output rows 2 ir0 .. 2 ir1
            a0 = max(r0, 2 * ir0 + (2 if ir0 > 0 else 0)); a1 = min(r1, 2 * ir1 - (2 if ir1 < H else 0))
            b0 = max(c0, 2 * ic0 + (2 if ic0 > 0 else 0)); b1 = min(c1, 2 * ic1 - (2 if ic1 < W else 0))
            want = yc[0, :, a0 - 2 * ir0:a1 - 2 * ir0, b0 - 2 * ic0:b1 - 2 * ic0]
            got = y_gpu[img, a0:a1, b0:b1].permute(2, 0, 1)
            _check(got, want, f"{what} image {img} window rows {a0}:{a1} cols {b0}:{b1}")


# (Cin, Cout, input size, pixel stride of the input buffer): the stride-1 3x3 layers of configs[1] by serving kernel
S1_LAYERS = [
    (64, 64, 256, 64),        # enc1.cb1 / dec1.cb1b and their data gradients: conv3x3s
    (128, 64, 256, 128),      # dec1.cb1a forward on the concat buffer: conv3x3g with 64-channel tiles;  data gradient 64 -> 128: conv3x3p
    (128, 128, 128, 128),     # enc2.cb1 / dec2.cb1b: conv3x3p
    (256, 128, 128, 256),     # dec2.cb1a on the concat buffer: conv3x3p
    (256, 256, 64, 256),      # enc3.cb1 / dec3.cb1b: conv3x3p, two channel tiles
    (512, 256, 64, 512),      # dec3.cb1a on the concat buffer (data gradient 256 -> 512: four channel tiles)
    (512, 512, 32, 512),      # enc4.cb1 / dec4.cb1b: conv3x3g, four channel tiles
    (1024, 512, 32, 1024),    # dec4.cb1a on the concat buffer (data gradient 512 -> 1024: eight channel tiles)
]


@pytest.mark.parametrize("Ci,Co,HW,ld", S1_LAYERS)
def test_conv3x3_forward_and_data_gradient_windows(U, Ci, Co, HW, ld):
    ops = U.ops
    g = ops.geom(B, HW, HW, Ci, Co, 3, 1)
    x = ops.Act(_rand_bf16((B, HW, HW, ld), 11, zero_pad=True), 0, Ci)
    w32, wh, wt = _weights(ops, Co, Ci, 12)
    bias = (torch.rand(Co, device=DEV) - 0.5)
    y = ops.Act(torch.empty((B, HW, HW, Co), dtype=torch.bfloat16, device=DEV))
    rows = ops.conv2d_colstat_rows(g, 0, x)
    if rows:
        cst = torch.zeros((rows, Co, 2), device=DEV)
        ops.conv2d_fwd_colstat(g, x, wh, bias, y, cst)
    else:
        ops.conv2d_fwd(g, x, wh, bias, y)
    gy = ops.Act(_rand_bf16((B, HW, HW, Co), 13))
    dx = ops.Act(torch.empty((B, HW, HW, Ci), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_dgrad(g, gy, wt, dx)
    torch.cuda.synchronize()
    w_hwio = w32.double().cpu().view(Co, 3, 3, Ci).permute(1, 2, 3, 0)
    _conv_windows(y.base, x.base[..., :Ci], w_hwio, bias.double().cpu(), 1, f"conv {Ci}->{Co}@{HW} fwd")
    # data gradient of a stride-1 'same' conv = conv with the spatially flipped, channel-swapped kernel
    w_flip = w_hwio.flip(0, 1).permute(0, 1, 3, 2)
    _conv_windows(dx.base, gy.base, w_flip, None, 1, f"conv {Ci}->{Co}@{HW} dgrad")
    if rows:          # fused column statistics of the stored bf16 tensor, at the real row count
        tot = cst.double().sum(0).cpu()
        yd = y.base.double()
        _check(tot[:, 0], yd.sum(dim=(0, 1, 2)).cpu(), "colstat sum", 1e-5)
        _check(tot[:, 1], (yd * yd).sum(dim=(0, 1, 2)).cpu(), "colstat sum of squares", 1e-5)


@pytest.mark.parametrize("Ci,Co,HW", [(64, 128, 256), (128, 256, 128), (512, 1024, 32)])     # the last: 16 x 16 output, conv3x3d's narrow tiles
def test_strided_conv_windows(U, Ci, Co, HW):
    """enc2.down / enc3.down (dl_models/u_net.py:269-276): forward (conv3x3d, plain and packed kernel copy), data gradient with the
    in-place skip-gradient addend (upconv3x3q)."""
    ops = U.ops
    g = ops.geom(B, HW, HW, Ci, Co, 3, 2)
    Ho = HW // 2
    x = ops.Act(_rand_bf16((B, HW, HW, Ci), 21, zero_pad=True))
    w32, wh, wt = _weights(ops, Co, Ci, 22)
    bias = (torch.rand(Co, device=DEV) - 0.5)
    y = ops.Act(torch.empty((B, Ho, Ho, Co), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_fwd(g, x, wh, bias, y)
    ne = ops.conv3x3s2_packed_elems(Co, Ci)
    if ne:            # the packed kernel copy the engines pass to the stride-2 forward kernel: the same bits out
        pk = torch.zeros(ne, dtype=torch.bfloat16, device=DEV)
        ops.cast_weights_batched(ops.make_cast_table([(w32, torch.empty_like(wh), torch.empty_like(wt), Co, 9, Ci, Ci, Co, pk)], DEV))
        yp = ops.Act(torch.empty((B, Ho, Ho, Co), dtype=torch.bfloat16, device=DEV))
        ops.conv2d_fwd(g, x, wh, bias, yp, w_packed=pk)
        torch.cuda.synchronize()
        assert torch.equal(yp.base, y.base)
        del yp, pk
    gy = ops.Act(_rand_bf16((B, Ho, Ho, Co), 23))
    skip0 = _rand_bf16((B, HW, HW, 2 * Ci), 24)                        # g_cat: the skip half accumulates in place
    skip = ops.Act(skip0.clone(), 0, Ci)
    ops.conv2d_dgrad(g, gy, wt, skip, addend=skip)
    torch.cuda.synchronize()
    w_hwio = w32.double().cpu().view(Co, 3, 3, Ci).permute(1, 2, 3, 0)
    _conv_windows(y.base, x.base, w_hwio, bias.double().cpu(), 2, f"strided conv {Ci}->{Co}@{HW} fwd")
    assert torch.equal(skip.base[..., Ci:], skip0[..., Ci:])          # the other half of the concat gradient is untouched
    # data gradient: the adjoint of the SAME stride-2 conv = transposed conv with the same HWIO kernel read as HWOI
    k_hwoi = w_hwio                                                    # [kh,kw,O=Ci(out of the adjoint),I=Co]: HWIO of the conv
    dxg = skip.base[..., :Ci].float() - skip0[..., :Ci].float()        # what the kernel added (bf16 rounding of the sum below)
    for img in IMAGES:
        for (r0, r1, c0, c1) in _windows(HW, HW):
            ir0, ir1 = max(r0 // 2 - 2, 0), min(r1 // 2 + 2, Ho)
            ic0, ic1 = max(c0 // 2 - 2, 0), min(c1 // 2 + 2, Ho)
            gc = _nchw64(gy.base[img:img + 1, ir0:ir1, ic0:ic1])
            full = torch.nn.functional.conv_transpose2d(gc, w_hwio.permute(3, 2, 0, 1), None, stride=2)   # rows 2 ir0 .. 2 ir1 + 1
            a0 = max(r0, 2 * ir0 + (2 if ir0 > 0 else 0)); a1 = min(r1, 2 * ir1 - (2 if ir1 < Ho else 0))
            b0 = max(c0, 2 * ic0 + (2 if ic0 > 0 else 0)); b1 = min(c1, 2 * ic1 - (2 if ic1 < Ho else 0))
            want = full[0, :, a0 - 2 * ir0:a1 - 2 * ir0, b0 - 2 * ic0:b1 - 2 * ic0] + \
                skip0[img, a0:a1, b0:b1, :Ci].double().cpu().permute(2, 0, 1)
            got = skip.base[img, a0:a1, b0:b1, :Ci].permute(2, 0, 1)
            _check(got, want, f"strided conv {Ci}->{Co}@{HW} dgrad+addend image {img} rows {a0}:{a1} cols {b0}:{b1}")
    del dxg


@pytest.mark.parametrize("Ci,Co,hw", [(128, 64, 128), (256, 128, 64)])
def test_conv_transpose_windows(U, Ci, Co, hw):
    """dec1.up / dec2.up (dl_models/u_net.py:297-304): forward into the upper half of the concat buffer (upconv3x3q), data
    gradient (conv3x3d on the adjoint geometry)."""
    ops = U.ops
    g = ops.geom(B, hw, hw, Ci, Co, 3, 2)
    HW = 2 * hw
    x = ops.Act(_rand_bf16((B, hw, hw, Ci), 31, zero_pad=True))
    w32 = _rand_bf16((Ci, 9, Co), 32, 0.1).float().contiguous()       # primary [Cin][9][Cout]
    wprim = torch.empty((Ci, 9, Co), dtype=torch.bfloat16, device=DEV)
    wtr = torch.empty((Co, 9, Ci), dtype=torch.bfloat16, device=DEV)
    ops.cast_weight_bf16(w32, wprim, Ci, 9, Co, Co)
    ops.transpose_cast_weight_bf16(w32, wtr, Ci, 9, Co, Ci)
    bias = (torch.rand(Co, device=DEV) - 0.5)
    cat = torch.full((B, HW, HW, 2 * Co), 3.0, dtype=torch.bfloat16, device=DEV)
    y = ops.Act(cat, Co, Co)
    ops.conv2d_transpose_fwd(g, x, wtr, bias, y)
    gy = ops.Act(_rand_bf16((B, HW, HW, 2 * Co), 33), Co, Co)
    dx = ops.Act(torch.empty((B, hw, hw, Ci), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_transpose_dgrad(g, gy, wprim, dx)
    torch.cuda.synchronize()
    assert float(cat[..., :Co].float().min()) == 3.0 and float(cat[..., :Co].float().max()) == 3.0
    k_hwoi = w32.double().cpu().view(Ci, 3, 3, Co).permute(1, 2, 3, 0)          # [kh,kw,O,I]
    _convT_windows(cat[..., Co:], x.base, k_hwoi, bias.double().cpu(), f"convT {Ci}->{Co}@{hw} fwd")
    # data gradient of the transposed conv = the SAME stride-2 conv of the upstream gradient with HWIO = [kh,kw,Co,Ci]
    _conv_windows(dx.base, gy.base[..., Co:], k_hwoi, None, 2, f"convT {Ci}->{Co}@{hw} dgrad")


def _wgrad_entries(Co, Ci):
    rs = np.random.RandomState(7)
    ent = {(0, 0, 0), (Co - 1, 8, Ci - 1), (Co // 2, 4, Ci // 2), (1, 2, Ci - 2), (Co - 2, 6, 1)}
    while len(ent) < 12:
        ent.add((int(rs.randint(Co)), int(rs.randint(9)), int(rs.randint(Ci))))
    return sorted(ent)


@pytest.mark.parametrize("Ci,Co,HW,stride", [(64, 64, 256, 1), (128, 128, 128, 1), (128, 64, 256, 1), (64, 128, 256, 2)])
def test_weight_gradient_entries_against_a_direct_sum(U, Ci, Co, HW, stride):
    """dw[n][kh][kw][c] = sum over every pixel of the batch of dy[p][n] * x[p * s + (kh, kw) - pad][c] (+ reg * w): a few entries in
    fp64 over ALL 32 images (wgrad3x3g for stride 1, the stride-2 patch kernel for the strided conv)."""
    ops = U.ops
    g = ops.geom(B, HW, HW, Ci, Co, 3, stride)
    Ho = HW // stride
    x = ops.Act(_rand_bf16((B, HW, HW, Ci), 41, zero_pad=True))
    gy = ops.Act(_rand_bf16((B, Ho, Ho, Co), 42))
    w32, _, _ = _weights(ops, Co, Ci, 43)
    dw = torch.full((Co, 3, 3, Ci), 9.0, device=DEV)
    ws = ops.Workspace(DEV)
    reg = 0.002
    ops.conv2d_wgrad(g, x, gy, dw, ws, reg=reg, w=w32)
    torch.cuda.synchronize()
    xd, gd = x.base.double(), gy.base.double()
    pad_before = 1 if stride == 1 else 0                               # TF 'same': (1,1) at stride 1, (0,1) at stride 2 / even size
    xp = torch.nn.functional.pad(xd, (0, 0, pad_before, 2, pad_before, 2))      # generous pad after: windows below stay in range
    scale = 0.0
    errs = []
    for (n, tap, c) in _wgrad_entries(Co, Ci):
        kh, kw = divmod(tap, 3)
        xs = xp[:, kh:kh + stride * Ho:stride, kw:kw + stride * Ho:stride, c]
        want = float((gd[..., n] * xs).sum()) + reg * float(w32[n, tap, c])
        got = float(dw[n, kh, kw, c])
        errs.append(abs(got - want))
        scale = max(scale, abs(want))
    K = B * Ho * Ho
    # bf16 products are exact in fp32; the error is fp32 accumulation over K terms in the kernel's (fixed) summation tree
    assert max(errs) <= 2e-6 * math.sqrt(K) * max(scale, 1.0) + 1e-6, (max(errs), scale)
