"""Waveform <-> feature transforms on the device: the host-side mirror of the reference's pre- and post-processing classes.

The reference builds its network inputs on the host, one file at a time (`Dataset.load_data`, dataset.py:146-176:
`Loader.load` -> `FeatureExtractor.extract` -> `Normalizer.normalize` -> `TensorPadder.pad_amp_phase`, preprocess.py) and
turns predictions back into impulse responses the same way (`PostProcess.post_process`, postprocess.py:51-73).  Here both
directions are one HIP kernel each over a whole batch that is already in HBM (csrc/features.hip), behind classes that keep
the reference's names, constructor arguments and call shapes:

    FeatureExtractor(n_fft, win_length, hop_length).extract(waveform)   -> (amp, phase)          preprocess.py:7-18
    Normalizer().normalize / .denormalize                                                        preprocess.py:21-41
    TensorPadder(desired_shape)                                                                  preprocess.py:60-113
    PreProcess(...)(waveforms)  = mean removal + extract + normalize + pad, fused               dataset.py:146-176
    PostProcess().post_process(feature, ...) -> waveform (un_pad + denormalize + istft, fused)   postprocess.py:51-73

Layouts: waveforms fp32 `[B, T]` (or `[T]`), features fp32 NCHW `[B, 2, H, W]` (plane 0 amplitude, plane 1 phase; row =
frequency bin, column = frame) - the engine's boundary, so a `PreProcess` output feeds `Trainer.step` / `UNet.forward`
directly; `PostProcess.post_process` also accepts the reference's NHWC `[H, W, 2]` / `[B, H, W, 2]` features.  File decoding
and writing (`librosa.load`, `scipy.io.wavfile.write`, `np.save`) and the Griffin-Lim branch stay on the host side of the
boundary and are not reimplemented.  There is no CPU path: tensors must live on the GPU and the calls raise otherwise.
"""
import torch

from . import ops

N_FFT, WIN_LENGTH, HOP_LENGTH = 256, 128, 64        # dataset.py:62-64
SAMPLE_RATE, DURATION = 48000, 0.2                  # dataset.py:66-67
INPUT_SHAPE = (144, 160)                            # dataset.py:70
STFT_SHAPE = (129, 151)                             # postprocess.py:51 des_shape


def _wav2d(waveform):
    if not isinstance(waveform, torch.Tensor) or not waveform.is_cuda:
        raise ValueError("waveforms must be CUDA tensors (there is no CPU path)")
    w = waveform if waveform.dim() == 2 else waveform.unsqueeze(0)
    if w.dim() != 2:
        raise ValueError("waveform must be [T] or [B, T]")
    return w.contiguous().float(), waveform.dim() == 1


class FeatureExtractor:
    """preprocess.py:7-18.  extract(waveform [T] or [B, T]) -> (amp, phase), each [n_fft/2+1, frames] or [B, ...]."""

    def __init__(self, n_fft, win_length, hop_length, pad_mode="reflect"):
        self.n_fft, self.win_length, self.hop_length, self.pad_mode = n_fft, win_length, hop_length, pad_mode

    def extract(self, waveform):
        w, single = _wav2d(waveform)
        B, T = w.shape
        out = torch.empty((B, 2, self.n_fft // 2 + 1, ops.stft_frames(T, self.hop_length)), dtype=torch.float32, device=w.device)
        ops.stft_features(w, out, self.n_fft, self.win_length, self.hop_length, self.pad_mode, remove_mean=False, normalize=False)
        return (out[0, 0], out[0, 1]) if single else (out[:, 0], out[:, 1])


class Normalizer:
    """preprocess.py:21-41 as constants + elementwise forms; the fused kernels apply them in fp64 (`PreProcess`, `PostProcess`).
    These two methods exist for callers that hold raw (amp, phase) tensors; they are plain tensor expressions."""

    def __init__(self):
        self.md = 100
        self.ep = 10 ** (-1 * self.md / 20)

    def normalize(self, amp, phase):
        return (20 * torch.log10(amp / 128 + self.ep) + self.md) / self.md, (phase + torch.pi) / (2 * torch.pi)

    def denormalize(self, amp_norm, phase_norm):
        amp = (10 ** ((amp_norm * self.md - self.md) / 20) - self.ep) * 128
        phase = torch.remainder(phase_norm * 2 * torch.pi, 2 * torch.pi) - torch.pi
        return amp, phase


class TensorPadder:
    """preprocess.py:60-113 on [..., rows, cols] tensors."""

    def __init__(self, desired_shape):
        self.desired_shape = tuple(desired_shape)

    def transform(self, tensor):
        r, c = tensor.shape[-2:]
        if r > self.desired_shape[0] or c > self.desired_shape[1]:        # get_needed_transform: larger inputs pass through
            return tensor
        return torch.nn.functional.pad(tensor, (0, self.desired_shape[1] - c, 0, self.desired_shape[0] - r))

    def pad_amp_phase(self, amp, phase):
        return self.transform(amp), self.transform(phase)

    @staticmethod
    def un_pad(amp, phase, desired_shape):
        return amp[..., :desired_shape[0], :desired_shape[1]], phase[..., :desired_shape[0], :desired_shape[1]]


class PreProcess:
    """The per-file chain of Dataset.load_data (dataset.py:146-176) for a batch of decoded waveforms, as one kernel:
    `signal -= mean` (preprocess.py:56) -> extract -> normalize -> pad -> features fp32 [B, 2, H, W]."""

    def __init__(self, n_fft=N_FFT, win_length=WIN_LENGTH, hop_length=HOP_LENGTH, desired_shape=INPUT_SHAPE, pad_mode="reflect",
                 remove_mean=True):
        self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length
        self.desired_shape, self.pad_mode, self.remove_mean = tuple(desired_shape), pad_mode, remove_mean

    def __call__(self, waveforms, out=None):
        w, _ = _wav2d(waveforms)
        if out is None:
            out = torch.empty((w.shape[0], 2) + self.desired_shape, dtype=torch.float32, device=w.device)
        ops.stft_features(w, out, self.n_fft, self.win_length, self.hop_length, self.pad_mode, self.remove_mean, normalize=True)
        return out


class PostProcess:
    """postprocess.py:21-73 without the file writes: post_process(feature) -> waveform(s).  Only the 'ph' (predicted phase)
    algorithm is provided; 'gl' (librosa.griffinlim) is outside the boundary."""

    def __init__(self, folder=None, algorithm=None):
        if algorithm == "gl":
            raise NotImplementedError("Griffin-Lim reconstruction is not part of the device path")
        self.algorithm = "ph"
        self.waveform = None

    def post_process(self, feature, vector=None, des_shape=STFT_SHAPE, n_fft=N_FFT, win_length=WIN_LENGTH, hop_length=HOP_LENGTH,
                     sr=SAMPLE_RATE, nhwc=None):
        if not isinstance(feature, torch.Tensor) or not feature.is_cuda:
            raise ValueError("features must be CUDA tensors (there is no CPU path)")
        f = feature
        single = f.dim() == 3
        if single:
            f = f.unsqueeze(0)
        if nhwc is None:                               # the reference hands over [H, W, 2]; the engine produces [B, 2, H, W]
            nhwc = f.shape[-1] == 2 and f.shape[1] != 2
        if nhwc:
            f = f.permute(0, 3, 1, 2)
        f = f.contiguous().float()
        wav = torch.empty((f.shape[0], hop_length * (des_shape[1] - 1)), dtype=torch.float32, device=f.device)
        ops.istft_features(f, wav, des_shape[0], des_shape[1], n_fft, win_length, hop_length, denormalize=True)
        self.waveform = wav[0] if single else wav
        return self.waveform
