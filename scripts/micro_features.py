"""Timing of the waveform <-> feature kernels at a full batch: python scripts/micro_features.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
from unet_rir_amd import features as F
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda:0"
x = (torch.randn((B, 9600)) * torch.exp(-torch.arange(9600) / 900.0)[None, :]).to(dev)


def timeit(f, K=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3


for shape in ((144, 160), (256, 256)):
    pre, post = F.PreProcess(desired_shape=shape), F.PostProcess()
    out = torch.empty((B, 2) + shape, device=dev)
    t1 = timeit(lambda: pre(x, out))
    wav = torch.empty((B, 9600), device=dev)
    t2 = timeit(lambda: U.ops.istft_features(out, wav, 129, 151))
    flop = B * 151 * 129 * 128 * 4.0
    print(f"B={B} {shape}: analysis {t1:.1f} us ({B / t1 * 1e6:.0f} waveforms/s, {flop / t1 / 1e6:.2f} TFLOP/s fp64 DFT, "
          f"{(x.numel() * 4 + out.numel() * 4) / t1 / 1e3:.1f} GB/s), synthesis {t2:.1f} us ({B / t2 * 1e6:.0f} waveforms/s)")
