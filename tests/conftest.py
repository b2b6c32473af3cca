import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


_DEFAULT_OFF = ()      # include/unetrir.h: every switch defaults to 1


@pytest.fixture(autouse=True)
def _restore_kernel_switches():
    """Tests may flip kernel-selection switches (ops.set_config); every test starts from and returns to the defaults."""
    yield
    try:
        import unet_rir_amd
        if unet_rir_amd._lib._LIB is not None:
            unet_rir_amd.ops.set_config(**{n: (0 if n in _DEFAULT_OFF else 1) for n, _ in unet_rir_amd._lib.Config._fields_})
    except Exception:
        pass
