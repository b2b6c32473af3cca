"""CPU checks of the waveform <-> feature oracle (oracle/features.py): against torch.stft / torch.istft (an independent
implementation of the same published algorithm), the analysis -> synthesis round trip, the committed golden vectors, and
argument validation of the C entry points (no GPU needed).  PARITY UNPINNED w.r.t. librosa itself (absent here)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_features_golden as MFG  # noqa: E402
from oracle import features as FO  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", list(MFG.CASES))
def test_oracle_matches_torch_stft_and_istft(name):
    B, T, n_fft, win, hop, shape, pad_mode = MFG.CASES[name]
    wav = MFG.waveforms(name).astype(np.float64)
    w = torch.hann_window(win, periodic=True, dtype=torch.float64)
    for b in range(B):
        S = FO.stft(wav[b], n_fft, win, hop, pad_mode)
        St = torch.stft(torch.tensor(wav[b]), n_fft, hop, win, w, center=True, pad_mode=pad_mode, return_complex=True)
        assert S.shape == (n_fft // 2 + 1, 1 + T // hop)
        assert float(np.abs(S - St.numpy()).max()) <= 1e-12 * max(1.0, float(np.abs(S).max()))
        y = FO.istft(S, n_fft, win, hop)
        yt = torch.istft(St, n_fft, hop, win, w, center=True).numpy()
        assert y.shape == (hop * (T // hop),)
        assert float(np.abs(y - yt).max()) <= 1e-12
        assert float(np.abs(y - wav[b][:len(y)]).max()) <= 1e-12     # Hann at 50 % / 25 % hop reconstructs exactly


def test_reference_constants_give_the_reference_shapes():
    """0.2 s at 48 kHz with n_fft 256 / hop 64 is 129 x 151, padded to (144, 160) (dataset.py:62-70, postprocess.py:51)."""
    wav = MFG.waveforms("features_rir_9600")[0]
    f = FO.wav_to_feature(wav)
    assert f.shape == (2, 144, 160)
    assert float(f.min()) >= 0.0 and float(f.max()) <= 1.0
    assert not f[:, 129:, :].any() and not f[:, :, 151:].any()
    back = FO.feature_to_wav(f)
    assert back.shape == (9600,)
    assert float(np.abs(back - (wav.astype(np.float64) - wav.astype(np.float64).mean())).max()) <= 1e-6


def test_normalizer_round_trip_and_ranges():
    amp = np.array([0.0, 1e-6, 1e-3, 1.0, 128.0 * (1 - FO.EP)])
    ph = np.array([-np.pi, -1.0, 0.0, 1.0, np.pi - 1e-9])
    a, p = FO.normalize(amp, ph)
    assert abs(a[0]) <= 1e-12 and abs(a[-1] - 1.0) <= 1e-12 and abs(p[0]) <= 1e-15 and p[-1] < 1.0
    amp2, ph2 = FO.denormalize(a, p)
    assert np.allclose(amp2, amp, rtol=1e-9, atol=1e-12) and np.allclose(ph2, ph, atol=1e-12)


@pytest.mark.parametrize("name", list(MFG.CASES))
def test_golden_vectors_are_reproduced(name):
    gold = dict(np.load(os.path.join(GOLD, name + ".npz")))
    got = MFG.compute(name)
    for k, v in gold.items():
        assert got[k].shape == v.shape, k
        assert float(np.abs(got[k].astype(np.float64) - v.astype(np.float64)).max()) <= 1e-6 * max(1.0, float(np.abs(v).max())), k


def test_entry_points_reject_bad_arguments_without_gpu():
    import unet_rir_amd
    L = unet_rir_amd._lib.lib()
    assert L.unetrir_stft_frames(9600, 64) == 151 and L.unetrir_stft_frames(1000, 32) == 32
    one = 16                                        # a non-null, never dereferenced pointer: validation happens before launch
    E = 10001
    assert L.unetrir_stft_features_f32(None, 1, 9600, 256, 128, 64, 0, 1, 1, one, 144, 160, None) == E
    assert L.unetrir_stft_features_f32(one, 1, 9600, 250, 128, 64, 0, 1, 1, one, 144, 160, None) == E     # n_fft not 2^k
    assert L.unetrir_stft_features_f32(one, 1, 9600, 256, 300, 64, 0, 1, 1, one, 144, 160, None) == E     # win > n_fft
    assert L.unetrir_stft_features_f32(one, 1, 9600, 256, 128, 64, 0, 1, 1, one, 128, 160, None) == E     # H < 129
    assert L.unetrir_stft_features_f32(one, 1, 9600, 256, 128, 64, 0, 1, 1, one, 144, 150, None) == E     # W < 151
    assert L.unetrir_stft_features_f32(one, 1, 100, 256, 128, 64, 0, 1, 1, one, 144, 160, None) == E      # reflect needs T > n_fft/2
    assert L.unetrir_stft_features_f32(one, 1, 9600, 256, 128, 64, 2, 1, 1, one, 144, 160, None) == E     # pad mode
    assert L.unetrir_istft_features_f32(one, 1, 144, 160, 128, 151, 256, 128, 64, 1, one, None) == E      # bins != n_fft/2+1
    assert L.unetrir_istft_features_f32(one, 1, 144, 160, 129, 161, 256, 128, 64, 1, one, None) == E      # frames > W
    assert L.unetrir_istft_features_f32(one, 1, 144, 160, 129, 1, 256, 128, 64, 1, one, None) == E        # no output samples
