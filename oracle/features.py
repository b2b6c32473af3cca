"""NumPy fp64 restatement of the reference's waveform <-> feature transforms.

Oracle / test infrastructure only (see oracle/__init__.py).  PARITY UNPINNED: the arithmetic lives in librosa, a third-party
dependency that is absent from /root/reference and from this image and is not pinned by the reference (no requirements
file; (c) 2022 headers => librosa 0.9.x).  What is restated here is librosa's published algorithm at the reference's call
sites; tests/test_features.py checks it against torch.stft / torch.istft (an independent implementation) and through the
analysis -> synthesis round trip.

  preprocess.py:13-18   FeatureExtractor.extract : librosa.stft(waveform, n_fft=256, win_length=128, hop_length=64)
                        -> amp = |S|, phase = angle(S); defaults window='hann', center=True, pad_mode='reflect'
                        (librosa < 0.10; 'constant' from 0.10 on - both are implemented, 'reflect' is the default here)
  preprocess.py:26-32   Normalizer.normalize     : amp -> (20 log10(amp/128 + 1e-5) + 100)/100, phase -> (phase + pi)/(2 pi)
  preprocess.py:34-41   Normalizer.denormalize   : the inverse, phase wrapped back into [-pi, pi)
  preprocess.py:65-105  TensorPadder             : zero rows / columns appended up to (144, 160); un_pad deletes them
  preprocess.py:52-57   Loader.load              : signal -= mean(signal)          (file decoding itself is out of scope)
  postprocess.py:127-136 PostProcess.istft ('ph'): librosa.istft(amp (cos p + i sin p), n_fft, win_length, hop_length)
  dataset.py:62-70      the constants: n_fft 256, win 128, hop 64, 0.2 s at 48 kHz = 9600 samples -> 129 x 151 -> (144, 160)
"""
import math

import numpy as np

MD = 100.0                      # Normalizer.md   (preprocess.py:23)
EP = 10.0 ** (-MD / 20.0)       # Normalizer.ep   (preprocess.py:24)
AMP_REF = 128.0                 # the "/ (128)" of preprocess.py:27
TINY32 = float(np.finfo(np.float32).tiny)   # librosa.util.tiny of the float32 window envelope


def hann_padded(n_fft, win_length):
    """scipy.signal.get_window('hann', win_length, fftbins=True) (periodic Hann) centred in n_fft zeros (librosa.util.pad_center)."""
    m = np.arange(win_length, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * m / win_length)
    lpad = (n_fft - win_length) // 2
    out = np.zeros(n_fft, dtype=np.float64)
    out[lpad:lpad + win_length] = w
    return out


def stft(wav, n_fft=256, win_length=128, hop_length=64, pad_mode="reflect"):
    """librosa.stft at preprocess.py:15 for one waveform [T] -> complex [1 + n_fft/2, 1 + T // hop]."""
    y = np.asarray(wav, dtype=np.float64)
    yp = np.pad(y, n_fft // 2, mode=pad_mode)
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    w = hann_padded(n_fft, win_length)
    frames = np.stack([yp[f * hop_length:f * hop_length + n_fft] * w for f in range(n_frames)], axis=1)
    return np.fft.rfft(frames, axis=0)


def istft(S, n_fft=256, win_length=128, hop_length=64):
    """librosa.istft (center=True, window='hann', length=None) at postprocess.py:133: windowed overlap-add of the inverse
    real FFT of every frame, divided by the overlap-added squared window where that exceeds tiny, n_fft/2 trimmed each side."""
    S = np.asarray(S, dtype=np.complex128)
    n_frames = S.shape[1]
    w = hann_padded(n_fft, win_length)
    total = n_fft + hop_length * (n_frames - 1)
    y = np.zeros(total, dtype=np.float64)
    wss = np.zeros(total, dtype=np.float64)
    frames = np.fft.irfft(S, n=n_fft, axis=0)
    for f in range(n_frames):
        y[f * hop_length:f * hop_length + n_fft] += w * frames[:, f]
        wss[f * hop_length:f * hop_length + n_fft] += w * w
    nz = wss > TINY32
    y[nz] /= wss[nz]
    return y[n_fft // 2: total - n_fft // 2]


def normalize(amp, phase):
    """Normalizer.normalize (preprocess.py:26-32)."""
    return (20.0 * np.log10(amp / AMP_REF + EP) + MD) / MD, (phase + math.pi) / (2.0 * math.pi)


def denormalize(amp_norm, phase_norm):
    """Normalizer.denormalize (preprocess.py:34-41)."""
    amp = (10.0 ** ((amp_norm * MD - MD) / 20.0) - EP) * AMP_REF
    phase = phase_norm * 2.0 * math.pi - math.pi
    phase = (phase + math.pi) % (2.0 * math.pi) - math.pi
    return amp, phase


def wav_to_feature(wav, shape=(144, 160), n_fft=256, win_length=128, hop_length=64, remove_mean=True, pad_mode="reflect"):
    """Loader's mean removal (preprocess.py:56) -> extract (:13-18) -> normalize (:26-32) -> TensorPadder.pad_amp_phase
    (:65-70) for one waveform [T]; returns the NCHW planes [2, H, W] (amp, phase) the network boundary takes."""
    y = np.asarray(wav, dtype=np.float64)
    if remove_mean:
        y = y - y.mean()
    S = stft(y, n_fft, win_length, hop_length, pad_mode)
    a, p = normalize(np.abs(S), np.angle(S))
    H, W = shape
    if a.shape[0] > H or a.shape[1] > W:          # TensorPadder.get_needed_transform: larger inputs pass through unpadded
        raise ValueError("feature larger than the padded shape")
    out = np.zeros((2, H, W), dtype=np.float64)
    out[0, :a.shape[0], :a.shape[1]] = a
    out[1, :p.shape[0], :p.shape[1]] = p
    return out


def feature_to_wav(feat, des_shape=(129, 151), n_fft=256, win_length=128, hop_length=64):
    """PostProcess.post_process without the file writes (postprocess.py:68-73): un_pad -> denormalize -> istft ('ph')
    for one feature [2, H, W] -> waveform [hop * (frames - 1)]."""
    feat = np.asarray(feat, dtype=np.float64)
    a = feat[0, :des_shape[0], :des_shape[1]]
    p = feat[1, :des_shape[0], :des_shape[1]]
    amp, phase = denormalize(a, p)
    return istft(amp * (np.cos(phase) + 1j * np.sin(phase)), n_fft, win_length, hop_length)
