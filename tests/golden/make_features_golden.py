"""Generates tests/golden/features_*.npz from the fp64 oracle of the waveform <-> feature transforms (oracle/features.py).

PARITY UNPINNED: librosa (the third-party module holding this arithmetic in the reference, preprocess.py:15 and
postprocess.py:133) is not available and the reference ships no fixtures, so these vectors pin the oracle - which is itself
checked against torch.stft / torch.istft in tests/test_features.py - and through it the HIP kernels.  Waveforms are
regenerated from oracle/detrand.py (platform independent): a decaying noise burst after a short pre-delay, which is what a
measured room impulse response of the reference's dataset looks like (0.2 s at 48 kHz = 9600 samples, dataset.py:62-70).

    python tests/golden/make_features_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import detrand, features as FO  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (B, T, n_fft, win, hop, (H, W), pad_mode)
    "features_rir_9600": (1, 9600, 256, 128, 64, (144, 160), "reflect"),       # the reference's constants
    "features_short_1000_constant": (3, 1000, 128, 128, 32, (80, 40), "constant"),   # win == n_fft, ragged length, zero padding
}


def waveforms(name):
    B, T = CASES[name][:2]
    t = np.arange(T, dtype=np.float64)
    out = np.zeros((B, T), dtype=np.float32)
    for b in range(B):
        noise = detrand.uniform(f"{name}/noise/{b}", (T,), -1.0, 1.0, np.float64)
        delay = 40 + 25 * b
        env = np.where(t >= delay, np.exp(-(t - delay) / (300.0 + 200.0 * b)), 0.0)
        out[b] = (0.6 * noise * env + 0.01 * (b + 1) + 1e-3 * detrand.uniform(f"{name}/floor/{b}", (T,), -1.0, 1.0, np.float64)).astype(np.float32)
    return out


def compute(name):
    B, T, n_fft, win, hop, shape, pad_mode = CASES[name]
    wav = waveforms(name)
    feat = np.stack([FO.wav_to_feature(wav[b], shape, n_fft, win, hop, True, pad_mode) for b in range(B)])
    nb, nf = n_fft // 2 + 1, 1 + T // hop
    back = np.stack([FO.feature_to_wav(feat[b].astype(np.float32), (nb, nf), n_fft, win, hop) for b in range(B)])
    raw = np.stack([FO.stft(wav[b].astype(np.float64), n_fft, win, hop, pad_mode) for b in range(B)])
    return {"feature": feat.astype(np.float32), "waveform_back": back.astype(np.float32),
            "stft_re": raw.real.astype(np.float32), "stft_im": raw.imag.astype(np.float32)}


if __name__ == "__main__":
    for name in CASES:
        out = compute(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})
