"""Oracle parity AT THE LAUNCHED SHAPES of BASELINE.json configs[3]'s model on one GPU (depth 5, number_filters_0 = 128,
512 x 512, batch 16 per GPU, bf16 storage): the window method of test_fullsize_windows_gpu.py at the other full-size
configuration bench.py measures (`configs.cfg4_model_1gpu`).  Every kernel class of the step runs once at its real size
through the C ABI - 1 GB activations, 2-8 tiles per workgroup of the persistent kernels, two to four channel tiles - and
windows of the result are compared with the fp64 oracle on the cropped input:

  conv3x3p   128 -> 128 @512 (forward + data gradient, fused column statistics), 256 -> 128 on the concat buffer / 128 -> 256
  conv3x3g   2048 -> 2048 @32 (16 channel tiles: the per-tile kernel keeps the layer)
  conv3x3d   128 -> 256 @512 -> 256 with the PACKED kernel copy the engines pass (and bit-equal without it);
             data gradient of the transposed conv 256 -> 128
  upconv3x3q transposed conv 256 -> 128 @256 -> 512 into the concat buffer; strided data gradient with the in-place addend
  wgrad3x3g / wgrad3x3d: sampled entries against a direct fp64 sum over all 16 images.
Tolerances as in the configs[1] file.  Reference call sites: dl_models/u_net.py:269-276, :297-304, :366."""
import math

import pytest
import torch

import test_fullsize_windows_gpu as W
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B4 = 16


@pytest.fixture(autouse=True)
def _cfg4_batch(monkeypatch):
    monkeypatch.setattr(W, "B", B4)
    monkeypatch.setattr(W, "IMAGES", (0, B4 - 1))


@pytest.fixture(scope="module")
def U():
    import unet_rir_amd
    unet_rir_amd._lib.lib()
    return unet_rir_amd


@pytest.mark.parametrize("Ci,Co,HW,ld", [(128, 128, 512, 128), (256, 128, 512, 256)])
def test_cfg4_conv3x3_forward_and_data_gradient_windows(U, Ci, Co, HW, ld):
    W.test_conv3x3_forward_and_data_gradient_windows(U, Ci, Co, HW, ld)
    ops = U.ops
    g = ops.geom(B4, HW, HW, Ci, Co, 3, 1)
    x = ops.Act(torch.empty((1, 8, 8, ld), dtype=torch.bfloat16, device=DEV), 0, Ci)
    assert ops.conv3x3_kernel(g, 0, x) == "conv3x3p"                              # the persistent kernel serves the layer


def test_cfg4_deep_conv3x3_windows(U, monkeypatch):
    """2048 -> 2048 at 32 x 32: sixteen channel tiles per pixel tile; one image, the top-left window (the oracle is 29 GMAC per window)."""
    monkeypatch.setattr(W, "IMAGES", (B4 - 1,))
    monkeypatch.setattr(W, "_windows", lambda H, Wd, step=1: [(0, 16, 0, 24)])
    W.test_conv3x3_forward_and_data_gradient_windows(U, 2048, 2048, 32, 2048)


def test_cfg4_strided_conv_with_packed_kernel_and_its_data_gradient(U):
    """enc2.down of configs[3] (128 -> 256, 512 -> 256): conv3x3d reads the PACKED copy cast_weights_batched writes (what the engines
    pass); the same launch without it must give the same bits.  Data gradient: upconv3x3q with the in-place addend."""
    ops = U.ops
    Ci, Co, HW = 128, 256, 512
    W.test_strided_conv_windows(U, Ci, Co, HW)                            # plain copy + data gradient against the oracle
    g = ops.geom(B4, HW, HW, Ci, Co, 3, 2)
    x = ops.Act(W._rand_bf16((B4, HW, HW, Ci), 21, zero_pad=True))
    w32, wh, _ = W._weights(ops, Co, Ci, 22)
    ne = ops.conv3x3s2_packed_elems(Co, Ci)
    assert ne == Co * 9 * Ci
    pk = torch.zeros(ne, dtype=torch.bfloat16, device=DEV)
    same2, tr2 = torch.empty_like(wh), torch.empty((Ci, 9, Co), dtype=torch.bfloat16, device=DEV)
    ops.cast_weights_batched(ops.make_cast_table([(w32, same2, tr2, Co, 9, Ci, Ci, Co, pk)], DEV))
    bias = (torch.rand(Co, device=DEV) - 0.5)
    y0 = ops.Act(torch.empty((B4, HW // 2, HW // 2, Co), dtype=torch.bfloat16, device=DEV))
    y1 = ops.Act(torch.empty((B4, HW // 2, HW // 2, Co), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_fwd(g, x, wh, bias, y0)
    ops.conv2d_fwd(g, x, wh, bias, y1, w_packed=pk)
    torch.cuda.synchronize()
    assert torch.equal(same2, wh) and torch.equal(y0.base, y1.base)
    w_hwio = w32.double().cpu().view(Co, 3, 3, Ci).permute(1, 2, 3, 0)
    W._conv_windows(y1.base, x.base, w_hwio, bias.double().cpu(), 2, "strided conv 128->256@512, packed kernel")


def test_cfg4_conv_transpose_windows(U):
    W.test_conv_transpose_windows(U, 256, 128, 256)


@pytest.mark.parametrize("Ci,Co,HW,stride", [(128, 128, 512, 1), (128, 256, 512, 2)])
def test_cfg4_weight_gradient_entries(U, Ci, Co, HW, stride):
    W.test_weight_gradient_entries_against_a_direct_sum(U, Ci, Co, HW, stride)
