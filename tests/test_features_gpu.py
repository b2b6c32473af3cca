"""GPU parity of the waveform <-> feature kernels (csrc/features.hip, through the C ABI) against the fp64 oracle
(oracle/features.py), the committed golden vectors, and size-independent properties at a full batch.

Tolerances: the kernels are fp64 direct DFTs with fp32 inputs and outputs, so a feature differs from the fp64 oracle by one
fp32 rounding: 2e-6 absolute on the [0, 1] planes.  The phase of a bin is only defined as well as its amplitude allows
(angle(S) moves by ~eps/|S|), so phases are compared through the complex value amp * exp(i phase)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_features_golden as MFG  # noqa: E402
from oracle import features as FO  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def U():
    import unet_rir_amd
    return unet_rir_amd


def circ(a, b):
    """distance between normalized phases on the unit circle"""
    d = np.abs(a - b) % 1.0
    return np.minimum(d, 1.0 - d)


@pytest.mark.parametrize("name", list(MFG.CASES))
def test_analysis_matches_golden_and_oracle(U, name):
    from unet_rir_amd import features as F
    B, T, n_fft, win, hop, shape, pad_mode = MFG.CASES[name]
    gold = dict(np.load(os.path.join(GOLD, name + ".npz")))
    wav = torch.tensor(MFG.waveforms(name)).to(DEV)
    # raw extract (no mean removal, no normalisation): FeatureExtractor.extract
    amp, ph = F.FeatureExtractor(n_fft, win, hop, pad_mode).extract(wav)
    S = (amp.double() * torch.exp(1j * ph.double())).cpu().numpy()
    S_ref = np.stack([FO.stft(MFG.waveforms(name)[b].astype(np.float64), n_fft, win, hop, pad_mode) for b in range(B)])
    assert S.shape == S_ref.shape
    assert float(np.abs(S - S_ref).max()) <= 2e-6 * float(np.abs(S_ref).max())
    Sg = gold["stft_re"].astype(np.float64) + 1j * gold["stft_im"].astype(np.float64)
    assert float(np.abs(S - Sg).max()) <= 4e-6 * float(np.abs(Sg).max())
    # the fused chain: mean removal + extract + normalize + pad
    feat = F.PreProcess(n_fft, win, hop, shape, pad_mode)(wav).cpu().numpy()
    g = gold["feature"]
    assert feat.shape == g.shape == (B, 2) + shape
    nb, nf = n_fft // 2 + 1, 1 + T // hop
    assert not feat[:, :, nb:, :].any() and not feat[:, :, :, nf:].any()          # appended rows / columns are exactly zero
    assert float(feat.min()) >= 0.0 and float(feat.max()) <= 1.0
    assert float(np.abs(feat[:, 0] - g[:, 0]).max()) <= 2e-6
    loud = g[:, 0, :nb, :nf] > 0.05                                                # above the -100 dB floor the phase is well conditioned
    assert loud.mean() > 0.9
    assert float(circ(feat[:, 1, :nb, :nf], g[:, 1, :nb, :nf])[loud].max()) <= 2e-5
    # everywhere: compare the denormalized complex values
    a1, p1 = FO.denormalize(feat[:, 0, :nb, :nf].astype(np.float64), feat[:, 1, :nb, :nf].astype(np.float64))
    a2, p2 = FO.denormalize(g[:, 0, :nb, :nf].astype(np.float64), g[:, 1, :nb, :nf].astype(np.float64))
    err = np.abs(a1 * np.exp(1j * p1) - a2 * np.exp(1j * p2))
    assert float((err / (a2 + 128 * FO.EP)).max()) <= 2e-5


@pytest.mark.parametrize("name", list(MFG.CASES))
def test_synthesis_matches_golden_and_oracle(U, name):
    from unet_rir_amd import features as F
    B, T, n_fft, win, hop, shape, pad_mode = MFG.CASES[name]
    gold = dict(np.load(os.path.join(GOLD, name + ".npz")))
    nb, nf = n_fft // 2 + 1, 1 + T // hop
    feat = torch.tensor(gold["feature"]).to(DEV)
    post = F.PostProcess()
    wav = post.post_process(feat, des_shape=(nb, nf), n_fft=n_fft, win_length=win, hop_length=hop).cpu().numpy()
    assert wav.shape == gold["waveform_back"].shape == (B, hop * (nf - 1))
    assert float(np.abs(wav - gold["waveform_back"]).max()) <= 2e-6
    ref = np.stack([FO.feature_to_wav(gold["feature"][b], (nb, nf), n_fft, win, hop) for b in range(B)])
    assert float(np.abs(wav - ref).max()) <= 2e-7 + 2e-7 * float(np.abs(ref).max())
    # the reference's calling convention: one NHWC feature [H, W, 2] (rir_generation.py:175, postprocess.py:51)
    one = post.post_process(feat[1 % B].permute(1, 2, 0).contiguous(), None, (nb, nf), n_fft, win, hop)
    assert one.shape == (hop * (nf - 1),) and torch.equal(one.cpu(), torch.tensor(wav[1 % B]))
    # garbage in the padding must not reach the waveform (un_pad)
    dirty = feat.clone()
    dirty[:, :, nb:, :] = 0.7
    dirty[:, :, :, nf:] = 0.3
    again = post.post_process(dirty, des_shape=(nb, nf), n_fft=n_fft, win_length=win, hop_length=hop)
    assert torch.equal(again.cpu(), torch.tensor(wav))


def test_full_batch_round_trip_and_properties(U):
    """B = 32 waveforms of the reference's length: synthesis(analysis(x)) == x - mean(x); analysis is deterministic, batch
    independent and homogeneous of degree one in the raw amplitude."""
    from unet_rir_amd import features as F
    B, T = 32, 9600
    g = torch.Generator(device="cpu").manual_seed(7)
    t = torch.arange(T, dtype=torch.float64)
    x = (torch.randn((B, T), generator=g, dtype=torch.float64) * torch.exp(-t / 900.0)[None, :] * 0.5 + 0.02).float()
    xd = x.to(DEV)
    pre, post = F.PreProcess(), F.PostProcess()
    feat = pre(xd)
    assert feat.shape == (B, 2, 144, 160)
    assert torch.equal(feat, pre(xd))                                              # deterministic
    assert torch.equal(feat[5:9], pre(xd[5:9].contiguous()))                       # a waveform's features do not depend on its batch
    back = post.post_process(feat).cpu().double()
    want = x.double() - x.double().mean(dim=1, keepdim=True)
    assert back.shape == (B, T)
    assert float((back - want).abs().max()) <= 2e-5                                # fp32 rounding of the log-amplitude / phase planes
    # sample 0 of the batch against the oracle
    f0 = FO.wav_to_feature(x[0].numpy())
    assert float(np.abs(feat[0, 0].cpu().numpy() - f0[0]).max()) <= 2e-6
    fe = F.FeatureExtractor(256, 128, 64)
    a1, p1 = fe.extract(xd)
    a2, p2 = fe.extract((2.0 * xd).contiguous())
    assert a1.shape == (B, 129, 151)
    assert float((a2 - 2.0 * a1).abs().max()) <= 1e-6 * float(a1.max()) and torch.allclose(p1, p2, atol=1e-6)
    # Normalizer / TensorPadder mirrors agree with the fused kernel
    an, pn = F.Normalizer().normalize(*fe.extract((xd - xd.double().mean(dim=1, keepdim=True).float()).contiguous()))
    ap, pp = F.TensorPadder((144, 160)).pad_amp_phase(an, pn)
    assert ap.shape == (B, 144, 160) and float((ap - feat[:, 0]).abs().max()) <= 1e-4
    au, pu = F.TensorPadder.un_pad(ap, pp, (129, 151))
    assert au.shape == (B, 129, 151)


def test_waveform_to_network_to_waveform(U):
    """The callers either side of the hot path: waveforms -> PreProcess -> U-Net (inference, moving statistics) ->
    PostProcess -> waveforms, all on the device, at the reference's own (144, 160) input shape (main_training.py:154)."""
    from unet_rir_amd import features as F
    B, T = 2, 9600
    g = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.randn((B, T), generator=g) * torch.exp(-torch.arange(T) / 700.0)[None, :]).to(DEV)
    feat = F.PreProcess()(x)
    m = U.UNet((144, 160, 2), (2, 16), number_filters_0=8, kernels=3, batch_size=B, device=DEV)
    emb = torch.randint(26, 1282, (B, 2, 16), device=DEV, dtype=torch.int32)
    with torch.no_grad():
        pred = m.model([feat.permute(0, 2, 3, 1), emb], training=False)            # NHWC in / out, as the reference calls it
    assert pred.shape == (B, 144, 160, 2)
    wav = F.PostProcess().post_process(pred)
    assert wav.shape == (B, T) and bool(torch.isfinite(wav).all())
    one = F.PostProcess().post_process(pred[0], emb[0, 1, :])                      # rir_generation.py:175
    assert torch.equal(one, wav[0])


def test_errors_are_loud(U):
    from unet_rir_amd import features as F
    with pytest.raises(ValueError):
        F.PreProcess()(torch.zeros(2, 9600))                                       # host tensor: there is no CPU path
    with pytest.raises(U.UnetrirError):
        F.PreProcess(desired_shape=(128, 160))(torch.zeros(2, 9600, device=DEV))   # fewer rows than frequency bins
    with pytest.raises(U.UnetrirError):
        F.PreProcess()(torch.zeros(2, 100, device=DEV))                            # shorter than the reflect padding
    with pytest.raises(NotImplementedError):
        F.PostProcess(algorithm="gl")
