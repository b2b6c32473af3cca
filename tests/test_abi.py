"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/unetrir.h declares
(no compute calls here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "unetrir.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(unetrir_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import unet_rir_amd
    L = unet_rir_amd._lib.lib()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"libunetrir.so does not export {n}"
    # and the ctypes table binds exactly the declared set
    assert sorted(unet_rir_amd._lib.EXPORTS) == names
    assert L.unetrir_abi_version() == 1


def test_argument_validation_without_gpu():
    """Entry points reject bad geometry before touching the device (UNETRIR_EINVAL = 10001)."""
    import ctypes as C
    import unet_rir_amd
    from unet_rir_amd._lib import ConvGeom
    L = unet_rir_amd._lib.lib()
    g = ConvGeom(1, 8, 8, 6, 8, 3, 1)            # Cin not a multiple of 4
    assert L.unetrir_conv2d_fwd_f32(C.byref(g), 16, 8, 16, None, None, 0, 16, 8, None) == 10001
    g = ConvGeom(1, 8, 8, 8, 8, 3, 3)            # stride 3
    assert L.unetrir_conv2d_fwd_f32(C.byref(g), 16, 8, 16, None, None, 0, 16, 8, None) == 10001
    assert L.unetrir_bn_ws_bytes(1024, 64) > 0
    assert L.unetrir_conv2d_wgrad_ws_bytes(C.byref(ConvGeom(32, 256, 256, 64, 64, 3, 1))) > 0
    assert L.unetrir_head6x6_supported(64) == 1 and L.unetrir_head6x6_supported(8) == 1 and L.unetrir_head6x6_supported(20) == 0
    # deferred split-K reductions: an empty batch is a no-op, a missing descriptor array or a descriptor without buffers is refused
    from unet_rir_amd._lib import ReduceDesc
    assert L.unetrir_splitk_reduce_batched(None, 0, None) == 0
    assert L.unetrir_splitk_reduce_batched(None, 1, None) == 10001
    d = (ReduceDesc * 2)()
    assert L.unetrir_splitk_reduce_batched(d, 2, None) == 0               # nsplit == 0: the weight gradient went straight into dw
    d[1].nsplit = 4
    assert L.unetrir_splitk_reduce_batched(d, 2, None) == 10001           # slabs to reduce but no buffers
    g = ConvGeom(1, 8, 8, 8, 8, 3, 1)
    assert L.unetrir_conv2d_wgrad_partials_bf16(C.byref(g), 16, 8, 16, 8, 16, 0.0, None, 16, 1 << 20, None, None) == 10001      # no descriptor to fill


def test_product_has_no_cpu_fallback_and_does_not_import_oracle():
    """The product package never imports oracle/ and refuses to run without the GPU path."""
    import torch
    import unet_rir_amd
    pkg = os.path.join(ROOT, "unet-rir_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, f)).read(), f
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            unet_rir_amd.UNet((32, 32, 2), (2, 16), kernels=3, batch_size=1)


def test_library_reads_only_the_documented_environment_switches():
    """include/unetrir.h documents the kernel-selection switches; no other UNETRIR_* string (ablation switches, tuning
    knobs) may be compiled into the product library, and the ablation entry point must not be exported."""
    import subprocess
    import unet_rir_amd
    lib = unet_rir_amd.build.LIB
    unet_rir_amd._lib.lib()
    out = subprocess.run(["strings", "-n", "8", lib], capture_output=True, text=True, check=True).stdout
    found = sorted(set(re.findall(r"UNETRIR_[A-Z0-9_]+", out)))
    hdr = open(os.path.join(ROOT, "include", "unetrir.h")).read()
    documented = sorted(set(re.findall(r"(UNETRIR_[A-Z0-9_]+)\s+(?:3x3|bf16|its)", hdr)))
    assert len(documented) == 18, documented
    assert found == documented, (found, documented)
    assert "unetrir_abl_set" not in out
    cfg = unet_rir_amd.ops.get_config()
    assert sorted(cfg) == sorted(n for n, _ in unet_rir_amd._lib.Config._fields_)
    assert all(v == 1 for v in cfg.values())          # measured refusals are not shipped behind switches: they are removed
    old = unet_rir_amd.ops.set_config(conv3x3s=0)
    assert old["conv3x3s"] == 1 and unet_rir_amd.ops.get_config()["conv3x3s"] == 0
