"""unet-rir_amd: MI355X (gfx950) implementation of the igmsalinas/unet-rir U-Net train-step hot path.

Host side in Python over a C ABI (include/unetrir.h) of hand-written HIP kernels; PyTorch provides
device memory, streams and torch.distributed only.
"""
from . import _lib, build, callbacks, ops  # noqa: F401
from ._lib import UnetrirError  # noqa: F401
from .engine import UNetEngine  # noqa: F401
from .device import HipRuntime  # noqa: F401
from .ae import AutoencoderEngine  # noqa: F401
from .model import Autoencoder, ResAE, UNet  # noqa: F401
from .resae import ResAEEngine  # noqa: F401
from .unet_graph import UNetGraphEngine  # noqa: F401
from .data import DeviceBatchPipeline, synthetic_batches  # noqa: F401
from .trainer import CheckpointManager, GradBucketer, Trainer, fit, lr_schedule  # noqa: F401

__all__ = ["ops", "build", "UnetrirError", "UNetEngine", "UNetGraphEngine", "ResAEEngine", "UNet", "ResAE", "Autoencoder", "AutoencoderEngine", "HipRuntime", "Trainer", "GradBucketer", "lr_schedule", "CheckpointManager", "fit", "DeviceBatchPipeline", "synthetic_batches"]
