"""CPU check of the window-comparison helpers the full-size GPU parity tests use (tests/test_fullsize_windows_gpu.py): fed
with the oracle's own full output they must accept it, and they must reject an output shifted by one pixel."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_fullsize_windows_gpu as T  # noqa: E402
from oracle import torch_ref as R  # noqa: E402


@pytest.mark.parametrize("stride", [1, 2])
def test_conv_window_helper(stride):
    torch.manual_seed(0)
    x = torch.randn(32, 96, 128, 4, dtype=torch.float64)
    w = torch.randn(3, 3, 4, 6, dtype=torch.float64)
    b = torch.randn(6, dtype=torch.float64)
    y = R.conv2d_same(x.permute(0, 3, 1, 2), w, b, stride).permute(0, 2, 3, 1).contiguous()
    T._conv_windows(y, x, w, b, stride, "self")
    with pytest.raises(AssertionError):
        T._conv_windows(torch.roll(y, 1, 2), x, w, b, stride, "shifted")


def test_conv_transpose_window_helper():
    torch.manual_seed(1)
    x = torch.randn(32, 48, 64, 4, dtype=torch.float64)
    k = torch.randn(3, 3, 6, 4, dtype=torch.float64)
    b = torch.randn(6, dtype=torch.float64)
    y = R.conv2d_transpose_same(x.permute(0, 3, 1, 2), k, b, 2).permute(0, 2, 3, 1).contiguous()
    T._convT_windows(y, x, k, b, "self")
    with pytest.raises(AssertionError):
        T._convT_windows(torch.roll(y, 1, 1), x, k, b, "shifted")
