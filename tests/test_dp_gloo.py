"""Data-parallel path on CPU with gloo, world_size 2: the GradBucketer + SUM all-reduce used by Trainer reproduce the
single-process gradients of the concatenated batch (BatchNorm per replica, loss normalised by the GLOBAL batch, l2 term
divided by the replica count) - the MirroredStrategy semantics of main_training.py:56, :230-233, :323-327.
Gradients here come from the oracle (the HIP kernels need a GPU); what is under test is the product's bucketing,
ordering and reduction code."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _flat_layout(cfg):
    """Same flat layout rule as UNetEngine (backward-completion order, 64-float alignment) without a GPU."""
    from oracle import torch_ref as R
    shapes = R.param_shapes(cfg)
    order = ["head.kernel", "head.bias"]
    for l in range(1, cfg.depth + 1):
        for blk in ("cb1b", "cb1a"):
            order += [f"dec{l}.{blk}.kernel", f"dec{l}.{blk}.bias", f"dec{l}.{blk}.gamma", f"dec{l}.{blk}.beta"]
        order += [f"dec{l}.up.kernel", f"dec{l}.up.bias"]
    order += ["vec.conv.kernel", "vec.conv.bias", "vec.dense.kernel", "vec.dense.bias", "vec.embedding"]
    for l in range(cfg.depth + 1, 0, -1):
        order += [f"enc{l}.cb1.kernel", f"enc{l}.cb1.bias", f"enc{l}.cb1.gamma", f"enc{l}.cb1.beta",
                  f"enc{l}.down.kernel", f"enc{l}.down.bias"]
    assert sorted(order) == sorted(shapes)
    off, layout = 0, {}
    for n in order:
        numel = int(np.prod(shapes[n]))
        layout[n] = (off, numel)
        off += -(-numel // 64) * 64
    return order, layout, off


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import torch_ref as R
    from unet_rir_amd.trainer import GradBucketer
    cfg = R.Config(16, 16, 4, 3)
    params = R.init_params(cfg, randomize_all=True, dtype=np.float64)
    B = 2
    spec_in, emb, spec_out = R.synthetic_batch(cfg, B * world)
    sl = slice(rank * B, (rank + 1) * B)
    _, _, _, grads = R.loss_and_grads(params, spec_in[sl], emb[sl], spec_out[sl], cfg, 0.9, B * world, world, None,
                                      torch.float64)
    order, layout, total = _flat_layout(cfg)
    flat = torch.zeros(total, dtype=torch.float64)
    bounds = [layout[n][0] + (-(-layout[n][1] // 64) * 64) for n in order]
    bucketer = GradBucketer(flat, bounds, bucket_bytes=4096)          # tiny buckets: many all-reduces
    assert len(list(bucketer.buckets())) > 3
    # "backward": parameters become final in flat order; announce them as the engine does
    for n in order:
        o, k = layout[n]
        flat[o:o + k] = grads[n].flatten()
        if n.endswith(".bias") and (n.startswith("head") or ".up." in n or ".down." in n) or n == "vec.embedding":
            bucketer.mark_ready(o + (-(-k // 64) * 64))
    bucketer.finish()
    if rank == 0:
        torch.save({n: flat[layout[n][0]:layout[n][0] + layout[n][1]].clone() for n in order}, tmp)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_process(tmp_path):
    from oracle import torch_ref as R
    world = 2
    tmp = str(tmp_path / "reduced.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, tmp), nprocs=world, join=True)
    reduced = torch.load(tmp)
    # single process: the sum over replicas of per-replica losses (BN statistics per replica)
    cfg = R.Config(16, 16, 4, 3)
    params = R.init_params(cfg, randomize_all=True, dtype=np.float64)
    B = 2
    spec_in, emb, spec_out = R.synthetic_batch(cfg, B * world)
    total = None
    for r in range(world):
        sl = slice(r * B, (r + 1) * B)
        _, _, _, g = R.loss_and_grads(params, spec_in[sl], emb[sl], spec_out[sl], cfg, 0.9, B * world, world, None,
                                      torch.float64)
        total = g if total is None else {k: total[k] + v for k, v in g.items()}
    for n, g in total.items():
        np.testing.assert_allclose(reduced[n].numpy(), g.flatten().numpy(), rtol=1e-12, atol=1e-15, err_msg=n)


def test_bucket_boundaries_cover_flat_buffer():
    from unet_rir_amd.trainer import GradBucketer
    flat = torch.zeros(1000)
    b = GradBucketer(flat, [64, 128, 512, 640, 1000], bucket_bytes=1024)
    ranges = list(b.buckets())
    assert ranges[0][0] == 0 and ranges[-1][1] == 1000
    assert all(lo < hi for lo, hi in ranges) and all(ranges[i][1] == ranges[i + 1][0] for i in range(len(ranges) - 1))


def test_bucket_hook_sees_every_bucket_once_in_order_without_a_collective():
    """reduce=False + on_bucket: the schedule the overlapped trainer uses on one replica (Adam per finished bucket).  A bucket
    is handed over only when mark_ready has covered its end, each exactly once, in flat-buffer order, and finish() flushes
    the rest; reset() starts a new step."""
    from unet_rir_amd.trainer import GradBucketer
    flat = torch.zeros(1000)
    seen = []
    b = GradBucketer(flat, [64, 128, 512, 640, 1000], bucket_bytes=1024, reduce=False,
                     on_bucket=lambda lo, hi, work: seen.append((lo, hi, work)))
    want = list(b.buckets())
    for step in range(2):
        seen.clear()
        b.reset()
        b.mark_ready(100)                       # nothing complete yet (first bucket ends at 512)
        assert seen == []
        b.mark_ready(600)
        assert [(lo, hi) for lo, hi, _ in seen] == [w for w in want if w[1] <= 600]
        b.mark_ready(600)                       # idempotent
        n_before = len(seen)
        assert n_before == len([w for w in want if w[1] <= 600])
        b.finish()
        assert [(lo, hi) for lo, hi, _ in seen] == want and all(w is None for _, _, w in seen)


def test_lr_schedule_matches_reference_formula():
    """main_training.py:342-344: lr * 0.9 ** (epoch / 80) from epoch 80 on (note: not epoch - 80)."""
    from unet_rir_amd.trainer import lr_schedule
    assert lr_schedule(5e-7, 79) == 5e-7
    assert abs(lr_schedule(5e-7, 80) - 5e-7 * 0.9) < 1e-20
    assert abs(lr_schedule(5e-7, 160) - 5e-7 * 0.81) < 1e-20
