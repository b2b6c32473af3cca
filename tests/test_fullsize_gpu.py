"""Parity at BASELINE.json's FULL sizes (configs[1]: batch 32, 256 x 256, number_filters_0 = 64, bf16 storage), where the
CPU oracle would take minutes per layer: size-independent properties that tie the HIP kernels to each other and to the
definition of the operators.

  * adjointness      <conv(x), y> == <x, dgrad(y)> == <w, wgrad(x, y)>: the forward, data-gradient and weight-gradient
                     kernels (three different programs, different tilings) must be the three faces of one bilinear form;
  * homogeneity      conv(2x) == 2 conv(x) and wgrad(x, 2y) == 2 wgrad(x, y) BIT-exactly (a power-of-two scale commutes with
                     every rounding step);
  * shift / batch    a 'same' convolution commutes with an image shift away from the border and treats every image of the
                     batch alone, bit-exactly: exposes any tile-seam or halo mistake at the real tile counts;
  * cross-kernel     128 output channels in one launch (conv3x3p / conv3x3g) == two 64-channel launches (conv3x3g's 64-channel
                     tiles since round 2; conv3x3h before) to bf16 rounding;
  * whole step       determinism, sigmoid range, BatchNorm output statistics (mean 0, variance 1 per channel), loss
                     invariance under a permutation of the batch, finite decreasing loss.
The layers are the ones the step actually runs: 64->64 @ 256^2 (conv3x3s), 128->128 @ 128^2 (conv3x3p), the stride-2
64->128 @ 256^2 -> 128^2 (conv3x3d forward, upconv3x3q data gradient, wgrad3x3d) and the transposed 128->64; the last test holds the
bf16 engine against the fp32 engine at this size (gradient fidelity of the benchmarked mode).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B = 32


@pytest.fixture(scope="module")
def U():
    import unet_rir_amd
    unet_rir_amd._lib.lib()
    return unet_rir_amd


def _rand_bf16(shape, seed, scale=1.0):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    return ((torch.rand(shape, device=DEV, generator=g) - 0.5) * 2 * scale).to(torch.bfloat16)


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _weights(ops, Co, Ci, seed):
    """fp32 master holding bf16-representable values, its bf16 copy [Co][9][Ci] and the transposed copy [Ci][9][Co]."""
    w32 = _rand_bf16((Co, 9, Ci), seed, 0.1).float().contiguous()
    wh = torch.empty((Co, 9, Ci), dtype=torch.bfloat16, device=DEV)
    wt = torch.empty((Ci, 9, Co), dtype=torch.bfloat16, device=DEV)
    ops.cast_weight_bf16(w32, wh, Co, 9, Ci, Ci)
    ops.transpose_cast_weight_bf16(w32, wt, Co, 9, Ci, Co)
    return w32, wh, wt


@pytest.mark.parametrize("Ci,Co,HW,stride", [(64, 64, 256, 1), (128, 128, 128, 1), (64, 128, 256, 2)])
def test_conv_trilinear_form_at_full_size(U, Ci, Co, HW, stride):
    ops = U.ops
    g = ops.geom(B, HW, HW, Ci, Co, 3, stride)
    Ho = HW // stride
    x = ops.Act(_rand_bf16((B, HW, HW, Ci), 1))
    y = ops.Act(_rand_bf16((B, Ho, Ho, Co), 2))
    w32, wh, wt = _weights(ops, Co, Ci, 3)
    out = ops.Act(torch.empty((B, Ho, Ho, Co), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_fwd(g, x, wh, None, out)
    dx = ops.Act(torch.empty((B, HW, HW, Ci), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_dgrad(g, y, wt, dx)
    dw = torch.empty((Co, 9, Ci), device=DEV)
    ws = ops.Workspace(DEV)
    ops.conv2d_wgrad(g, x, y, dw, ws)
    torch.cuda.synchronize()
    a, b, c = _dot(out.base, y.base), _dot(x.base, dx.base), _dot(w32, dw)
    scale = math.sqrt(_dot(out.base, out.base) * _dot(y.base, y.base))
    # bf16 outputs carry 2^-9 relative rounding per element with random signs: the three sums agree far below 1e-3 of the norm
    assert abs(a - b) <= 1e-3 * scale and abs(a - c) <= 1e-3 * scale, (a, b, c, scale)
    # homogeneity, bit-exact
    x2 = ops.Act((x.base.float() * 2).to(torch.bfloat16))
    out2 = ops.Act(torch.empty_like(out.base))
    ops.conv2d_fwd(g, x2, wh, None, out2)
    y2 = ops.Act((y.base.float() * 2).to(torch.bfloat16))
    dw2 = torch.empty_like(dw)
    ops.conv2d_wgrad(g, x, y2, dw2, ws)
    torch.cuda.synchronize()
    assert torch.equal(out2.base.float(), out.base.float() * 2)
    assert torch.equal(dw2, dw * 2)


def test_conv_transpose_adjoint_pairs_at_full_size(U):
    """Conv2DTranspose(128 -> 64, k 3, s 2) 128^2 -> 256^2: forward (upconv3x3), data gradient (igemm stride-2 forward form)
    and weight gradient against each other."""
    ops = U.ops
    Ci, Co, H = 128, 64, 128
    g = ops.geom(B, H, H, Ci, Co, 3, 2)
    x = ops.Act(_rand_bf16((B, H, H, Ci), 11))
    y = ops.Act(_rand_bf16((B, 2 * H, 2 * H, Co), 12))
    w32 = _rand_bf16((Ci, 9, Co), 13, 0.1).float().contiguous()            # primary Conv2DTranspose layout [Ci][k][k][Co]
    wprim = torch.empty((Ci, 9, Co), dtype=torch.bfloat16, device=DEV)
    wt = torch.empty((Co, 9, Ci), dtype=torch.bfloat16, device=DEV)
    ops.cast_weight_bf16(w32, wprim, Ci, 9, Co, Co)
    ops.transpose_cast_weight_bf16(w32, wt, Ci, 9, Co, Ci)
    out = ops.Act(torch.empty((B, 2 * H, 2 * H, Co), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_transpose_fwd(g, x, wt, None, out)
    dx = ops.Act(torch.empty((B, H, H, Ci), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_transpose_dgrad(g, y, wprim, dx)
    dw = torch.empty((Ci, 9, Co), device=DEV)
    ws = ops.Workspace(DEV)
    ops.conv2d_transpose_wgrad(g, x, y, dw, ws)
    torch.cuda.synchronize()
    a, b, c = _dot(out.base, y.base), _dot(x.base, dx.base), _dot(w32, dw)
    scale = math.sqrt(_dot(out.base, out.base) * _dot(y.base, y.base))
    assert abs(a - b) <= 1e-3 * scale and abs(a - c) <= 1e-3 * scale, (a, b, c, scale)


@pytest.mark.parametrize("Ci,Co,HW", [(64, 64, 256), (128, 128, 128)])
def test_conv_shift_and_batch_independence_at_full_size(U, Ci, Co, HW):
    ops = U.ops
    g = ops.geom(B, HW, HW, Ci, Co, 3, 1)
    x = _rand_bf16((B, HW, HW, Ci), 21)
    _, wh, _ = _weights(ops, Co, Ci, 22)
    bias = (torch.rand(Co, device=DEV) - 0.5)
    out = ops.Act(torch.empty((B, HW, HW, Co), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_fwd(g, ops.Act(x), wh, bias, out)
    # shift the images by (5, 19) pixels: one pixel away from the old and the new borders the output shifts with them
    sy, sx = 5, 19
    xs = torch.zeros_like(x)
    xs[:, sy:, sx:, :] = x[:, :HW - sy, :HW - sx, :]
    outs = ops.Act(torch.empty_like(out.base))
    ops.conv2d_fwd(g, ops.Act(xs), wh, bias, outs)
    # a batch whose images are permuted: every image must come out identical to its original position
    perm = torch.randperm(B, device=DEV)
    outp = ops.Act(torch.empty_like(out.base))
    ops.conv2d_fwd(g, ops.Act(x[perm].contiguous()), wh, bias, outp)
    torch.cuda.synchronize()
    assert torch.equal(outs.base[:, sy + 1:HW - 1, sx + 1:HW - 1, :], out.base[:, 1:HW - sy - 1, 1:HW - sx - 1, :])
    assert torch.equal(outp.base, out.base[perm])


def test_two_kernels_one_operator_at_full_size(U):
    """conv3x3g (128 output channels in one tile) and conv3x3h (two launches of 64) compute the same 3x3 layer; they order
    the K sum differently, so they agree to the bf16 rounding of the stored result."""
    ops = U.ops
    Ci, Co, HW = 64, 128, 256
    x = ops.Act(_rand_bf16((B, HW, HW, Ci), 31))
    w32, wh, _ = _weights(ops, Co, Ci, 32)
    full = ops.Act(torch.empty((B, HW, HW, Co), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_fwd(ops.geom(B, HW, HW, Ci, Co, 3, 1), x, wh, None, full)
    halves = ops.Act(torch.empty((B, HW, HW, Co), dtype=torch.bfloat16, device=DEV))
    g64 = ops.geom(B, HW, HW, Ci, 64, 3, 1)
    for h in range(2):
        ops.conv2d_fwd(g64, x, wh[64 * h:64 * (h + 1)].contiguous(), None, halves.slice(64 * h, 64))
    torch.cuda.synchronize()
    a, b = full.base.float(), halves.base.float()
    err = float((a - b).abs().max())
    assert err <= 2.0 ** -7 * float(a.abs().max()), err
    assert float((a != b).float().mean()) < 0.2                  # most elements round identically


def test_paired_tile_matches_tap_table_kernel_at_full_size(U, monkeypatch):
    """The 16 x 16 level (1024 -> 1024 channels, batch 32): conv3x3g's paired-image tile against the tap-table kernel the layer
    used before, forward and data gradient; different K order, so agreement to the bf16 rounding of the stored result."""
    ops = U.ops
    C, HW = 1024, 16
    x = ops.Act(_rand_bf16((B, HW, HW, C), 41))
    w32, wh, wt = _weights(ops, C, C, 42)
    g = ops.geom(B, HW, HW, C, C, 3, 1)
    res = {}
    monkeypatch.setattr(ops, "_restore_cfg", ops.get_config(), raising=False)
    for mode in ("1", "0"):
        ops.set_config(conv3x3g_pair=int(mode))
        y = ops.Act(torch.empty((B, HW, HW, C), dtype=torch.bfloat16, device=DEV))
        dx = ops.Act(torch.empty((B, HW, HW, C), dtype=torch.bfloat16, device=DEV))
        ops.conv2d_fwd(g, x, wh, None, y)
        ops.conv2d_dgrad(g, x, wt, dx)
        torch.cuda.synchronize()
        res[mode] = (y.base.float(), dx.base.float())
    assert ops.conv2d_colstat_rows(g, 0, x) in (B * HW * HW // 128, B * HW * HW // 64)  # (mode 0 is in force here: the tap-table kernel, one row per pixel tile)
    ops.set_config(conv3x3g_pair=1)
    assert ops.conv2d_colstat_rows(g, 0, x) == B // 2              # one statistics row per image pair
    for a, b in zip(res["1"], res["0"]):
        assert float((a - b).abs().max()) <= 2.0 ** -7 * float(a.abs().max())
        assert float((a != b).float().mean()) < 0.2


def test_train_step_invariants_at_full_size(U):
    H = W = 256
    eng = U.UNetEngine(H, W, B, F0=64, dtype="bf16", device=DEV)
    gen = torch.Generator()
    gen.manual_seed(0)
    eng.reset_parameters(gen)                            # Keras default initialisers
    spec_in, emb, spec_out = next(U.synthetic_batches(1, B, H, W, DEV))
    eng.training = True
    eng.forward(spec_in, emb, target=spec_out, global_batch=B)
    torch.cuda.synchronize()
    # BatchNormalization output of the first block: mean 0 / variance var/(var+eps) per channel (batch statistics)
    name = "enc1.cb1"
    C1 = eng.y[1].C
    sc, sh = eng.bn_affine[name][:C1].double(), eng.bn_affine[name][C1:].double()
    z = eng.y[1].dense().double() * sc + sh
    z = (z - eng.p[name + ".beta"].double()) / eng.p[name + ".gamma"].double()
    m, v = z.mean(dim=(0, 1, 2)), z.var(dim=(0, 1, 2), unbiased=False)
    # Keras normalises by sqrt(var + 1e-3): the output variance is var / (var + eps) = 1 - eps * rstd^2
    want_v = 1.0 - 1e-3 * eng.bn_saved[name][C1:].double() ** 2
    assert float(m.abs().max()) < 1e-3 and float((v - want_v).abs().max()) < 1e-3, (float(m.abs().max()), float((v - want_v).abs().max()))
    del z
    eng.backward()
    torch.cuda.synchronize()
    pred1, grad1, loss1 = eng.pred.clone(), eng.grad.clone(), float(eng.loss_out[0])
    assert float(pred1.min()) > 0.0 and float(pred1.max()) < 1.0 and math.isfinite(loss1)
    # the step is a function of its inputs: same inputs, same bits
    eng.forward(spec_in, emb, target=spec_out, global_batch=B)
    eng.backward()
    torch.cuda.synchronize()
    assert torch.equal(eng.pred, pred1) and torch.equal(eng.grad, grad1)
    # a permutation of the batch permutes the prediction and leaves loss and weight gradients unchanged up to the order of
    # the fp32 split-K sums
    perm = torch.randperm(B, device=DEV)
    eng.forward(spec_in[perm].contiguous(), emb[perm].contiguous(), target=spec_out[perm].contiguous(), global_batch=B)
    eng.backward()
    torch.cuda.synchronize()
    assert abs(float(eng.loss_out[0]) - loss1) <= 1e-5 * abs(loss1)
    assert float((eng.pred - pred1[perm]).abs().max()) <= 2e-2
    gn = float(grad1.double().norm())
    assert float((eng.grad.double() - grad1.double()).norm()) <= 2e-2 * gn
    # and it trains
    tr = U.Trainer(eng, lr=1e-4, dropout=False)
    l0 = tr.step(spec_in, emb, spec_out, return_loss=True)
    for _ in range(5):
        l1 = tr.step(spec_in, emb, spec_out, return_loss=True)
    assert math.isfinite(l1) and l1 < l0


def test_bf16_training_signal_against_the_fp32_engine_at_config_size(U):
    """How far is the benchmarked bf16-storage step from the fp32-storage step (the mode the fp32-tolerance oracle parity is
    asserted in) AT BASELINE.json configs[1] size - batch 32, 256 x 256, number_filters_0 = 64, where BatchNormalization averages
    over 2 M elements?  Same initial variables (Keras initialisers), same batch, same dropout masks.  Observed (rounds 3 and 4,
    DESIGN.md section 5, profiles/r04_bf16_noise_trace.json): loss 1.9e-4 apart; whole gradient relative L2 0.0075, cosine 0.99997;
    per tensor median relative L2 0.02, the deep encoder levels 0.10-0.14, the information-vector branch 0.29 (cosine 0.958).
    Where it comes from (scripts/bf16_noise_trace.py): the stored FORWARD tensors of the two engines are 1.1-1.9 % apart, so at every
    BatchNormalization -> ReLU a per-cent fraction of the ReLU inputs has the other sign, and the ELEMENTWISE gradient tensors move
    apart by 3-10 % per block (dL/dlogits 1.5 %, behind the first BatchNorm backward 9.9 %, dL/dz of the bottleneck 29 %).  Weight
    gradients sum that over ~10^5..10^6 pixels and stay at 1-14 %; the information-vector branch sums dL/dz itself through a 16-channel
    bottleneck and keeps all of it.  Handing that branch dL/dz BEFORE its rounding to bf16 (an fp32 side output of the data-gradient
    kernel: round 4, built and measured) changes nothing: 0.2865 / 0.2906 / 0.2899 either way - the stored rounding of dL/dz is not
    the cause.  50 Adam steps: losses within 7e-4.  The bounds below are those observations x 1.15 (the kernels are deterministic and
    the inputs seeded: a regression of 15 % trips them)."""
    import bench
    H = 256
    spec_in, emb, spec_out = bench.synthetic_batch(B, H, H, torch.device(DEV), 1234)
    engs = {}
    for dt in ("f32", "bf16"):
        eng = U.UNetEngine(H, H, B, F0=64, k=3, device=DEV, dtype=dt)
        if dt == "f32":
            gen = torch.Generator(); gen.manual_seed(0)
            eng.reset_parameters(gen)
        else:
            eng.load_keras_params(engs["f32"].export_keras_params())
        eng.dropout_seed = 4321
        engs[dt] = eng
    grads, loss = {}, {}
    for dt, eng in engs.items():
        eng.training = True
        mask = eng.make_dropout_mask()
        eng.forward(spec_in, emb, dropout_mask=mask, target=spec_out, global_batch=B)
        eng.backward()
        torch.cuda.synchronize()
        loss[dt] = float(eng.loss_out[0])
        grads[dt] = {k: v.double() for k, v in eng.export_keras_grads().items()}
        eng._shared["dropout_step"] = 0
    assert abs(loss["bf16"] - loss["f32"]) <= 1e-3 * loss["f32"]
    rels = []
    for n, g32 in grads["f32"].items():
        g16 = grads["bf16"][n]
        n32 = float(g32.norm())
        if n32 < 1e-12:                       # biases in front of a BatchNorm: analytically zero, both engines write exact zeros
            assert float(g16.abs().max()) == 0.0, n
            continue
        rel = float((g16 - g32).norm()) / n32
        cos = float((g16 * g32).sum()) / (n32 * float(g16.norm()))
        rels.append(rel)
        if n.startswith("vec."):
            assert rel <= 0.335 and cos >= 0.950, (n, rel, cos)       # observed worst: vec.dense.kernel 0.2906 / 0.9578
        else:
            assert rel <= 0.16 and cos >= 0.988, (n, rel, cos)        # observed worst: enc5.cb1.kernel 0.1376 / 0.9905
    assert len(rels) == 77 - 13 and sorted(rels)[len(rels) // 2] <= 0.025         # observed 0.0198
    w32 = torch.cat([g.flatten() for g in grads["f32"].values()])
    w16 = torch.cat([g.flatten() for g in grads["bf16"].values()])
    assert float((w16 - w32).norm() / w32.norm()) <= 0.009                        # observed 0.0075
    assert float((w16 * w32).sum() / (w16.norm() * w32.norm())) >= 0.99995       # observed 0.999972
    # 50 Adam steps from the same variables with the same dropout stream: the two loss trajectories stay together
    traj = {}
    for dt, eng in engs.items():
        tr = U.Trainer(eng, lr=1e-4, dropout=True)
        ls = []
        for _ in range(50):
            tr.step(spec_in, emb, spec_out)
            ls.append(eng.loss_out[0].clone())
        torch.cuda.synchronize()
        traj[dt] = [float(v) for v in ls]
    assert traj["f32"][-1] < 0.8 * traj["f32"][0] and traj["bf16"][-1] < 0.8 * traj["bf16"][0]
    assert max(abs(a - b) / a for a, b in zip(traj["f32"], traj["bf16"])) <= 1e-3         # observed 6.1e-4 ... 6.7e-4
