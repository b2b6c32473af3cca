"""Oracle parity AT THE LAUNCHED SHAPES of BASELINE.json configs[4]: ResAE (dl_models/res_ae.py; main_training.py:130-141:
filters (32, 64, 128, 256), kernels 3, strides 2, latent 32, n_neurons 1024), batch 32 of [2,256,256], bf16 storage.

The layers of this graph are small enough (<= 9.7 GFLOP each) that the fp64 CPU oracle evaluates a WHOLE layer at the real
batch size in a second or two, so - unlike the U-Net windows tests - every distinct launch of the step is compared over its
full output: forward, data gradient and the complete weight gradient, once per distinct (operator, channels, grid) shape:

  Conv2D           1x1 stride 2 (res_conv entry + skip, dl_models/res_ae.py:482-514), 1x1 stride 1, 3x3 stride 1 (:453-480)
  Conv2DTranspose  1x1 stride 1, 3x3 stride 1 (res_t_identity, :310-337), 1x1 stride 2 'valid' (res_t_conv, :339-371),
                   the output layer Conv2DTranspose(2, 3, strides 2) (:373-389)
  glue             BatchNormalization -> Add -> LeakyReLU junction forward / backward at the 128 x 128 level

and the whole step at batch 32 for the invariants an oracle-free run can check (determinism, sigmoid range, BatchNorm moments,
a decreasing loss, the HIP-graph replay equal to the launch-by-launch step).
Tolerances: 1e-2 of the tensor's scale for bf16 outputs (one bf16 rounding is 4e-3), 2e-6 sqrt(K) for fp32 weight gradients.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B = 32
LEVELS = [(32, 128), (64, 64), (128, 32), (256, 16)]          # (filters, grid) of the four encoder / decoder levels


@pytest.fixture(scope="module")
def U():
    import unet_rir_amd
    unet_rir_amd._lib.lib()
    return unet_rir_amd


def _rand_bf16(shape, seed, scale=1.0):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    return ((torch.rand(shape, device=DEV, generator=g) - 0.5) * 2 * scale).to(torch.bfloat16)


def _weights(ops, N, T, C, seed):
    """fp32 master [N][T][C] holding bf16-representable values and its two bf16 work copies."""
    w32 = _rand_bf16((N, T, C), seed, 0.1).float().contiguous()
    same = torch.empty((N, T, C), dtype=torch.bfloat16, device=DEV)
    tr = torch.empty((C, T, N), dtype=torch.bfloat16, device=DEV)
    ops.cast_weight_bf16(w32, same, N, T, C, C)
    ops.transpose_cast_weight_bf16(w32, tr, N, T, C, N)
    return w32, same, tr


def _nchw64(t):
    return t.double().cpu().permute(0, 3, 1, 2).contiguous()


def _close(got, want, what, tol=1e-2, where=None):
    scale = float(want.abs().max()) + 1e-30
    d = (got.double().cpu() - want).abs()
    if where is not None:          # elementwise comparisons away from a decision boundary (the activation's kink)
        d = d * where
    err = float(d.max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def _colstat_close(cst, y, what):
    """Fused column statistics = per-channel (sum, sum of squares) of the STORED bf16 tensor."""
    tot = cst.double().sum(0).cpu()
    yd = y.double()
    for j, want in enumerate((yd.sum(dim=(0, 1, 2)).cpu(), (yd * yd).sum(dim=(0, 1, 2)).cpu())):
        err = float((tot[:, j] - want).abs().max())
        assert err <= 1e-5 * (float(want.abs().max()) + 1.0), f"{what}: colstat column {j} err {err:.3e}"


def _wgrad_close(dw, want, K, what):
    scale = max(float(want.abs().max()), 1.0)
    err = float((dw.double().cpu() - want).abs().max())
    assert err <= 2e-6 * math.sqrt(K) * scale + 1e-6, f"{what}: max err {err:.3e} (K = {K}, scale {scale:.3e})"


# ---------------------------------------------------------------------------------------------------------------- Conv2D
CONV_CASES = [(1, 2, 8, 32, 256)] + [(1, 2, LEVELS[i][0], LEVELS[i + 1][0], LEVELS[i][1]) for i in range(3)] + \
             [(1, 1, f, f, hw) for f, hw in LEVELS] + [(3, 1, f, f, hw) for f, hw in LEVELS]


@pytest.mark.parametrize("k,s,Ci,Co,HW", CONV_CASES)
def test_conv2d_layers_of_the_encoder_over_the_whole_tensor(U, k, s, Ci, Co, HW):
    """res_conv / res_identity convolutions (dl_models/res_ae.py:453-514) at batch 32: forward (+ bias), data gradient, weight
    gradient with the l2 term folded in - every element against the fp64 oracle."""
    ops = U.ops
    T = k * k
    g = ops.geom(B, HW, HW, Ci, Co, k, s)
    Ho = HW // s
    x = ops.Act(_rand_bf16((B, HW, HW, Ci), 11))
    if Ci == 8:
        x.base[..., 2:] = 0                                  # the zero-padded 2-channel network input
    w32, wh, wt = _weights(ops, Co, T, Ci, 12)
    bias = torch.rand(Co, device=DEV) - 0.5
    y = ops.Act(torch.empty((B, Ho, Ho, Co), dtype=torch.bfloat16, device=DEV))
    rows = ops.conv2d_colstat_rows(g, 0, x)                  # every BatchNorm-fed convolution of the graph has fused statistics
    assert rows > 0
    cst = torch.full((rows, Co, 2), 7.0, device=DEV)
    ops.conv2d_fwd_colstat(g, x, wh, bias, y, cst)
    gy = ops.Act(_rand_bf16((B, Ho, Ho, Co), 13))
    dx = ops.Act(torch.empty((B, HW, HW, Ci), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_dgrad(g, gy, wt, dx)
    dw = torch.full((Co, k, k, Ci), 9.0, device=DEV)
    ws = ops.Workspace(DEV)
    reg = 0.002
    ops.conv2d_wgrad(g, x, gy, dw, ws, reg=reg, w=w32)
    torch.cuda.synchronize()
    xo = _nchw64(x.base).requires_grad_(True)
    w_hwio = w32.double().cpu().view(Co, k, k, Ci).permute(1, 2, 3, 0).contiguous().requires_grad_(True)
    yo = R.conv2d_same(xo, w_hwio, bias.double().cpu(), s)
    gx, gw = torch.autograd.grad(yo, (xo, w_hwio), _nchw64(gy.base))
    _close(y.base.permute(0, 3, 1, 2), yo.detach(), f"conv {k}x{k}/{s} {Ci}->{Co}@{HW} fwd")
    _close(dx.base.permute(0, 3, 1, 2), gx, f"conv {k}x{k}/{s} {Ci}->{Co}@{HW} dgrad")
    want_dw = gw.permute(3, 0, 1, 2) + reg * w32.double().cpu().view(Co, k, k, Ci)
    _wgrad_close(dw, want_dw, B * Ho * Ho, f"conv {k}x{k}/{s} {Ci}->{Co}@{HW} wgrad")
    _colstat_close(cst, y.base, f"conv {k}x{k}/{s} {Ci}->{Co}@{HW}")


# ------------------------------------------------------------------------------------------------------- Conv2DTranspose
CONVT_CASES = [(1, 1, f, f, hw) for f, hw in LEVELS] + [(3, 1, f, f, hw) for f, hw in LEVELS] + \
              [(1, 2, LEVELS[i + 1][0], LEVELS[i][0], LEVELS[i + 1][1]) for i in range(3)] + [(3, 2, 32, 8, 128)]


@pytest.mark.parametrize("k,s,Ci,Co,hw", CONVT_CASES)
def test_conv2d_transpose_layers_of_the_decoder_over_the_whole_tensor(U, k, s, Ci, Co, hw):
    """res_t_conv / res_t_identity / the output layer (dl_models/res_ae.py:310-389): Conv2DTranspose forward (+ bias), data
    gradient and weight gradient; primary kernel layout [Cin][k][k][Cout] (Keras HWOI permuted)."""
    ops = U.ops
    T = k * k
    g = ops.geom(B, hw, hw, Ci, Co, k, s)
    HW = hw * s
    x = ops.Act(_rand_bf16((B, hw, hw, Ci), 21))
    w32, wprim, wtr = _weights(ops, Ci, T, Co, 22)            # master / same orientation [Ci][T][Co]; transposed [Co][T][Ci]
    real_out = 2 if Co == 8 else Co
    if Co == 8:                                               # the 2-channel output layer zero-padded to 8
        w32[..., 2:] = 0
        wprim[..., 2:] = 0
        wtr[2:] = 0
    bias = torch.rand(Co, device=DEV) - 0.5
    if Co == 8:
        bias[2:] = 0
    y = ops.Act(torch.empty((B, HW, HW, Co), dtype=torch.bfloat16, device=DEV))
    rows = ops.conv2d_transpose_colstat_rows(g, x)
    assert (rows > 0) == (not (k == 3 and s == 2))            # only the output layer (no BatchNorm behind it) has none
    cst = torch.full((max(rows, 1), Co, 2), 7.0, device=DEV)
    if rows:
        ops.conv2d_transpose_fwd_colstat(g, x, wtr, bias, y, cst)
    else:
        ops.conv2d_transpose_fwd(g, x, wtr, bias, y)
    gy = ops.Act(_rand_bf16((B, HW, HW, Co), 23))
    if Co == 8:
        gy.base[..., 2:] = 0
    dx = ops.Act(torch.empty((B, hw, hw, Ci), dtype=torch.bfloat16, device=DEV))
    ops.conv2d_transpose_dgrad(g, gy, wprim, dx)
    dw = torch.full((Ci, k, k, Co), 9.0, device=DEV)
    ws = ops.Workspace(DEV)
    reg = 0.002
    ops.conv2d_transpose_wgrad(g, x, gy, dw, ws, reg=reg, w=w32)
    torch.cuda.synchronize()
    xo = _nchw64(x.base).requires_grad_(True)
    k_hwoi = w32.double().cpu().view(Ci, k, k, Co)[..., :real_out].permute(1, 2, 3, 0).contiguous().requires_grad_(True)    # [kh,kw,O,I]
    yo = R.conv2d_transpose_same(xo, k_hwoi, bias.double().cpu()[:real_out], s)
    assert tuple(yo.shape) == (B, real_out, HW, HW)
    gx, gw = torch.autograd.grad(yo, (xo, k_hwoi), _nchw64(gy.base)[:, :real_out])
    _close(y.base.permute(0, 3, 1, 2)[:, :real_out], yo.detach(), f"convT {k}x{k}/{s} {Ci}->{Co}@{hw} fwd")
    _close(dx.base.permute(0, 3, 1, 2), gx, f"convT {k}x{k}/{s} {Ci}->{Co}@{hw} dgrad")
    want_dw = gw.permute(3, 0, 1, 2) + reg * w32.double().cpu().view(Ci, k, k, Co)[..., :real_out]
    _wgrad_close(dw[..., :real_out], want_dw, B * hw * hw, f"convT {k}x{k}/{s} {Ci}->{Co}@{hw} wgrad")
    if Co == 8:
        assert float(y.base[..., 2:].float().abs().max()) == 0.0
    if rows:
        _colstat_close(cst, y.base, f"convT {k}x{k}/{s} {Ci}->{Co}@{hw}")


# ------------------------------------------------------------------------------------------------------------ the junction
@pytest.mark.parametrize("f,hw", LEVELS[:1] + LEVELS[3:])
def test_batchnorm_add_leakyrelu_junction_at_full_size(U, f, hw):
    """y = LeakyReLU(BatchNormalization(x) + skip) (dl_models/res_ae.py:331-336, :475-480): statistics, apply, and the backward
    (dgamma, dbeta, dx, d skip) of the kernels the graph engine launches, against autograd in fp64."""
    ops = U.ops
    P = B * hw * hw
    x = ops.Act(_rand_bf16((B, hw, hw, f), 31, 2.0))
    skip = ops.Act(_rand_bf16((B, hw, hw, f), 32))
    gamma = (torch.rand(f, device=DEV) + 0.5)
    beta = (torch.rand(f, device=DEV) - 0.5)
    aff, saved = torch.empty(2 * f, device=DEV), torch.empty(2 * f, device=DEV)
    ws = ops.Workspace(DEV)
    y = ops.Act(torch.empty((B, hw, hw, f), dtype=torch.bfloat16, device=DEV))
    ops.bn_stats(x, gamma, beta, aff, saved, ws, None, None, 1e-3, 0.99)
    ops.bn_act_add(x, aff, y, 2, skip)
    gy = ops.Act(_rand_bf16((B, hw, hw, f), 33))
    gj = ops.Act(torch.empty((B, hw, hw, f), dtype=torch.bfloat16, device=DEV))
    dx = ops.Act(torch.empty((B, hw, hw, f), dtype=torch.bfloat16, device=DEV))
    dgamma, dbeta = torch.empty(f, device=DEV), torch.empty(f, device=DEV)
    ops.act_bwd(gy, y, gj, 2)
    ops.bn_bwd(gj, x, None, aff, saved, dx, dgamma, dbeta, ws, relu=0)
    torch.cuda.synchronize()
    xo = x.base.double().cpu().view(P, f).requires_grad_(True)
    so = skip.base.double().cpu().view(P, f).requires_grad_(True)
    go, bo = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
    mean, var = xo.mean(0), xo.var(0, unbiased=False)
    yo = F.leaky_relu((xo - mean) / torch.sqrt(var + 1e-3) * go + bo + so, 0.3)
    _close(y.base.view(P, f), yo.detach(), "junction forward")
    # backward through the STORED (bf16-rounded) activation sign, as the product decides it; seed = the same upstream gradient
    gyo = gy.base.double().cpu().view(P, f)
    gxo, gso, ggo, gbo = torch.autograd.grad(yo, (xo, so, go, bo), gyo)
    # an output within fp32 rounding of the kink (statistics summed in another order put it on the other side: a handful of the
    # 2 M ... 16 M elements) takes the other slope in the product: those elements are left out of the ELEMENTWISE comparisons
    off_kink = (yo.detach().abs() > 1e-5).double()
    assert float(off_kink.mean()) > 0.9999
    _close(gj.base.view(P, f), gso, "junction d skip", 1.5e-2, off_kink)
    _close(dx.base.view(P, f), gxo, "junction dx", 2e-2, off_kink)
    assert float((dgamma.double().cpu() - ggo).abs().max()) <= 1e-2 * float(ggo.abs().max())
    assert float((dbeta.double().cpu() - gbo).abs().max()) <= 1e-2 * float(gbo.abs().max())
    # the fused form the graph engine launches: one reduce / finalize / apply sequence, the skip gradient accumulated in place
    prev = _rand_bf16((B, hw, hw, f), 34)
    gs = ops.Act(prev.clone())
    dx2 = ops.Act(torch.empty((B, hw, hw, f), dtype=torch.bfloat16, device=DEV))
    dg2, db2 = torch.empty(f, device=DEV), torch.empty(f, device=DEV)
    ops.bn_bwd_junction(gy, x, y, aff, saved, dx2, dg2, db2, ws, act=2, gskip=gs, gskip_add=gs)
    torch.cuda.synchronize()
    _close(dx2.base.view(P, f), gxo, "fused junction dx", 1e-2, off_kink)
    _close(gs.base.view(P, f), gso + prev.double().cpu().view(P, f), "fused junction d skip (accumulated)", 1e-2, off_kink)
    assert float((dg2.double().cpu() - ggo).abs().max()) <= 2e-3 * float(ggo.abs().max())
    assert float((db2.double().cpu() - gbo).abs().max()) <= 2e-3 * float(gbo.abs().max())


# --------------------------------------------------------------------------------------------------------- the whole step
def _cfg5(U, overlap=False):
    eng = U.ResAEEngine(256, 256, B, (32, 64, 128, 256), (3, 3, 3, 3), (2, 2, 2, 2), 32, 1024, device=DEV, dtype="bf16",
                        overlap_wgrad=overlap)
    gen = torch.Generator()
    gen.manual_seed(0)
    eng.reset_parameters(gen)
    eng.dropout_seed = 9
    return eng


def test_resae_train_step_invariants_at_cfg5_size(U):
    eng = _cfg5(U)
    assert eng.n_params() == 17_173_922
    spec_in, emb, spec_out = next(U.synthetic_batches(1, B, 256, 256, DEV))
    eng.training = True
    eng.forward(spec_in, emb, target=spec_out, global_batch=B)
    eng.backward()
    torch.cuda.synchronize()
    pred1, grad1, loss1 = eng.pred.clone(), eng.grad.clone(), float(eng.loss_out[0])
    assert float(pred1.min()) > 0.0 and float(pred1.max()) < 1.0 and math.isfinite(loss1)
    assert bool(torch.isfinite(grad1).all())
    # same inputs, same bits (every reduction has a fixed order)
    eng.forward(spec_in, emb, target=spec_out, global_batch=B)
    eng.backward()
    torch.cuda.synchronize()
    assert torch.equal(eng.pred, pred1) and torch.equal(eng.grad, grad1)
    # moving statistics moved towards batch statistics with momentum 0.99 from (0, 1)
    mm = eng.moving["e_res_1_conv.2.moving_mean"]
    assert float(mm.abs().max()) > 0.0
    # a batch permutation permutes the prediction; loss and gradients change only by summation order / bf16 rounding
    pg = torch.Generator(device=DEV); pg.manual_seed(3)
    perm = torch.randperm(B, device=DEV, generator=pg)
    eng.forward(spec_in[perm].contiguous(), emb[perm].contiguous(), target=spec_out[perm].contiguous(), global_batch=B)
    eng.backward()
    torch.cuda.synchronize()
    assert abs(float(eng.loss_out[0]) - loss1) <= 1e-4 * abs(loss1)
    # bf16 storage: the statistics of a permuted batch are the same sums in another order, a last-bit difference flips bf16 roundings
    # and 56 BatchNorm layers carry the flips on - a few of the 4.2 M outputs move by several per cent (observed maxima 3.3e-2 ...
    # 7.6e-2 over permutations), the bulk by 1e-3
    # How large: test_resae_bf16_storage_against_fp32_storage_at_cfg5_size measures the bf16 engine against the fp32 engine on the
    # same batch - prediction rms 1.0e-2, max 0.115 apart - and the fp32 engine under this very permutation: prediction bit-equal.
    # A permutation of the bf16 batch costs about half of the storage noise itself (rms 5.4e-3, max 6.1e-2 here; gradient 5.3 %
    # in relative L2): the bounds are those observations x 1.5, not a number picked after a failing run.
    d = (eng.pred - pred1[perm]).abs().double()
    assert float(d.pow(2).mean().sqrt()) <= 8e-3 and float(d.max()) <= 0.1, (float(d.pow(2).mean().sqrt()), float(d.max()))
    assert float((eng.grad.double() - grad1.double()).norm()) <= 8e-2 * float(grad1.double().norm())
    # it trains
    tr = U.Trainer(eng, lr=1e-4, dropout=False)
    l0 = tr.step(spec_in, emb, spec_out, return_loss=True)
    for _ in range(8):
        l1 = tr.step(spec_in, emb, spec_out, return_loss=True)
    assert math.isfinite(l1) and l1 < l0


def test_resae_bf16_storage_against_fp32_storage_at_cfg5_size(U):
    """BASELINE.json configs[4] at its own size: how far is the benchmarked bf16-storage step from the fp32-storage step (the mode the
    fp32-tolerance oracle parity is asserted in, tests/test_resae_gpu.py)?  Same variables (Keras initialisers), same batch, same
    dropout stream.  Observed (round 4, scripts/resae_fidelity.py, profiles/r04_resae_fidelity.json): loss 3.4e-5 apart; prediction
    rms 1.0e-2 / max 0.115 apart (56 BatchNormalization layers carry the bf16 rounding of every stored activation forward); whole
    gradient relative L2 0.077, cosine 0.9971; per tensor (177 with a non-zero gradient) median 0.118, worst the BatchNorm scales /
    offsets of the first encoder blocks at 0.60 (cosine 0.855: elementwise the gradient signal differs by tens of per cent wherever
    a LeakyReLU / ReLU input changed sign between the two forward passes, and a per-channel sum over 32 x 128 x 128 of it keeps that
    noise); 50 Adam steps with dropout: the loss trajectories stay within 2.5e-3.  The fp32 engine itself is bit-stable under a batch
    permutation (prediction equal, gradient 5e-8), the bf16 engine moves by about half of its distance to fp32 (see the permutation
    test above).  Bounds = observations with margin."""
    spec_in, emb, spec_out = next(U.synthetic_batches(1, B, 256, 256, DEV))
    engs = {}
    for dt in ("f32", "bf16"):
        eng = U.ResAEEngine(256, 256, B, (32, 64, 128, 256), (3, 3, 3, 3), (2, 2, 2, 2), 32, 1024, device=DEV, dtype=dt)
        if dt == "f32":
            gen = torch.Generator(); gen.manual_seed(0)
            eng.reset_parameters(gen)
        else:
            eng.load_keras_params(engs["f32"].export_keras_params())
        eng.dropout_seed = 9
        engs[dt] = eng
    grads, loss, pred = {}, {}, {}
    for dt, eng in engs.items():
        eng.training = True
        eng.forward(spec_in, emb, target=spec_out, global_batch=B)
        eng.backward()
        torch.cuda.synchronize()
        loss[dt], pred[dt] = float(eng.loss_out[0]), eng.pred.clone()
        grads[dt] = {k: v.double() for k, v in eng.export_keras_grads().items()}
    assert abs(loss["bf16"] - loss["f32"]) <= 3e-4 * loss["f32"]
    d = (pred["bf16"] - pred["f32"]).abs().double()
    assert float(d.pow(2).mean().sqrt()) <= 1.5e-2 and float(d.max()) <= 0.2, (float(d.pow(2).mean().sqrt()), float(d.max()))
    rels, zero = [], 0
    for n, g32 in grads["f32"].items():
        g16 = grads["bf16"][n]
        n32 = float(g32.norm())
        if n32 < 1e-12:                       # biases in front of a BatchNorm: analytically zero, exact zeros in both engines
            assert float(g16.abs().max()) == 0.0, n
            zero += 1
            continue
        rel = float((g16 - g32).norm()) / n32
        cos = float((g16 * g32).sum()) / (n32 * float(g16.norm()))
        assert rel <= 0.75 and cos >= 0.80, (n, rel, cos)
        rels.append(rel)
    assert zero == 56 and len(rels) == len(grads["f32"]) - 56
    assert sorted(rels)[len(rels) // 2] <= 0.15
    w32 = torch.cat([g.flatten() for g in grads["f32"].values()])
    w16 = torch.cat([g.flatten() for g in grads["bf16"].values()])
    assert float((w16 - w32).norm() / w32.norm()) <= 0.10
    assert float((w16 * w32).sum() / (w16.norm() * w32.norm())) >= 0.995
    # the fp32 engine under a batch permutation: the prediction is the permuted prediction bit for bit (BatchNorm sums are fp64 in a
    # fixed order per slab; fp32 storage absorbs the last-bit differences of another slab order)
    pg = torch.Generator(device=DEV); pg.manual_seed(3)
    perm = torch.randperm(B, device=DEV, generator=pg)
    e32 = engs["f32"]
    whole = e32.grad.double().clone()
    e32.forward(spec_in[perm].contiguous(), emb[perm].contiguous(), target=spec_out[perm].contiguous(), global_batch=B)
    e32.backward()
    torch.cuda.synchronize()
    assert float((e32.pred - pred["f32"][perm]).abs().max()) <= 1e-6
    assert float((e32.grad.double() - whole).norm()) <= 1e-5 * float(whole.norm())
    # 50 Adam steps from the same variables with the same dropout stream
    traj = {}
    for dt, eng in engs.items():
        eng._shared["dropout_step"] = 0
        tr = U.Trainer(eng, lr=1e-4, dropout=True)
        ls = []
        for _ in range(50):
            tr.step(spec_in, emb, spec_out)
            ls.append(eng.loss_out[0].clone())
        torch.cuda.synchronize()
        traj[dt] = [float(v) for v in ls]
    assert traj["f32"][-1] < 0.85 * traj["f32"][0] and traj["bf16"][-1] < 0.85 * traj["bf16"][0]
    assert max(abs(a - b) / a for a, b in zip(traj["f32"], traj["bf16"])) <= 6e-3


def test_resae_graph_replay_equals_the_launched_step_at_cfg5_size(U):
    """The benchmarked form of configs[4] (bench.py: side-stream schedule, step replayed as a HIP graph) against the same
    launches issued one by one, three steps with dropout: bit-identical variables."""
    spec_in, emb, spec_out = next(U.synthetic_batches(1, B, 256, 256, DEV))
    out = []
    for graph in (True, False):
        eng = _cfg5(U, overlap=True)
        if not graph:
            eng.use_device_counters(True)
        tr = U.Trainer(eng, lr=1e-4, graph=graph)
        for _ in range(3):
            tr.step(spec_in, emb, spec_out)
        torch.cuda.synchronize()
        out.append((eng.theta.clone(), float(eng.loss_out[0])))
        del tr, eng
    assert torch.equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
