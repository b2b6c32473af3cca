"""bench.py --gpus N (N > 1) starts its own ranks (the driver calls it exactly like --gpus 1; the reference is
single-process multi-GPU, main_training.py:56, :323-327).  Without GPUs the children must fail loudly and the parent
must hand their status on - no hang, no silent fallback."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    env["CUDA_VISIBLE_DEVICES"] = ""      # also on a GPU box: this test is about the launcher, not the step
    env["HIP_VISIBLE_DEVICES"] = ""
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus2_self_launch_fails_loudly_without_gpus():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "needs 2 visible GPUs" in r.stderr
    assert "rank 0" in r.stderr or "rank 1" in r.stderr       # a child started and reported (the launcher ends the other one
                                                               # as soon as the first has failed, possibly before it gets to print)
    assert r.stdout.strip() == ""                              # no JSON line from a failed run


def test_rank_environment_must_match_gpus():
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_single_gpu_path_refuses_to_run_without_a_gpu():
    r = _run(["--gpus", "1"])
    assert r.returncode != 0 and ("visible GPUs" in r.stderr or "MI355X" in r.stderr)


def test_shared_gpu_rehearsal_also_refuses_to_run_without_a_gpu():
    """--rehearse-shared-gpu (all ranks on cuda:0, collectives over gloo) relaxes the one-GPU-per-rank check, not the
    need for a GPU: no CPU fallback of the product step."""
    r = _run(["--gpus", "2", "--rehearse-shared-gpu", "--lean", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "visible GPUs" in r.stderr or "MI355X" in r.stderr
    assert r.stdout.strip() == ""


def _probe_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, ROOT)
        from _pytest.monkeypatch import MonkeyPatch
        import bench
        import cpu_ops
        import unet_rir_amd as U
        from oracle import torch_ref as R
        from sim_runtime import SimRuntime
        mpatch = MonkeyPatch()
        rt = SimRuntime()
        cpu_ops.install(mpatch, rt)
        H, W, F0, B = 32, 32, 4, 2
        eng = U.UNetEngine(H, W, B, F0=F0, k=3, device="cpu", runtime=rt, overlap_wgrad=True, n_replicas=world)
        g = torch.Generator(); g.manual_seed(0)
        eng.reset_parameters(g)
        tr = U.Trainer(eng, lr=1e-3, dropout=False, world_size=world, bucket_bytes=8192)
        tr.broadcast_parameters(0)
        cfg = R.Config(H, W, F0, 3)
        spec_in, emb, spec_out = (torch.tensor(a) for a in R.synthetic_batch(cfg, B, seed_name=f"r{rank}"))
        with bench.DpProbe(tr) as probe:
            for _ in range(3):
                probe.step(lambda: tr.step(spec_in, emb, spec_out))
        # the probe has put the runtime and the trainer back as they were
        assert rt.all_reduce_sum.__func__ is SimRuntime.all_reduce_sum and tr.apply_gradients.__func__ is U.Trainer.apply_gradients
        torch.save({"summary": probe.summary(), "n_buckets": len(tr.bucketer.bounds), "grad_bytes": eng.grad.numel() * 4,
                    "theta": eng.theta.clone()}, f"{out_path}.{rank}")
        mpatch.undo()
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_data_parallel_probe_of_the_bench_line_on_two_gloo_ranks(tmp_path):
    """What `bench.py --gpus N` adds to its line for N > 1 (`data_parallel`: per-bucket hand-over time and issue -> completion of the
    all-reduce, end of the backward pass, exposed tail) comes from bench.DpProbe around the product Trainer.step.  No 8-GPU node is
    available to the build, so the probe is exercised here: two gloo ranks, the product schedule (side-stream weight gradients,
    bucket-wise Adam) on the simulated runtime - one row per gradient bucket, bytes adding up to the gradient buffer, hand-over times
    in bucket order inside the step, a tail that is the step minus the backward pass; replicas stay in sync with the probe in place."""
    import torch
    import torch.multiprocessing as mp
    world = 2
    out = str(tmp_path / "probe")
    mp.spawn(_probe_worker, args=(world, 32700 + (os.getpid() % 1500), out), nprocs=world, join=True)
    res = [torch.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    assert torch.equal(res[0]["theta"], res[1]["theta"])
    for r in res:
        s = r["summary"]
        assert s["probed_steps"] == 3 and len(s["buckets"]) == r["n_buckets"] > 3
        assert sum(b["bytes"] for b in s["buckets"]) == r["grad_bytes"] == s["allreduce_bytes_per_step"]
        at = [b["issued_at_ms"] for b in s["buckets"]]
        assert all(a2 >= a1 for a1, a2 in zip(at, at[1:])) and at[0] > 0.0
        assert all(b["issue_to_complete_ms"] is not None and b["issue_to_complete_ms"] >= 0.0 for b in s["buckets"])
        assert 0.0 < s["backward_end_at_ms"] <= s["step_ms"] and abs(s["exposed_tail_ms"] - (s["step_ms"] - s["backward_end_at_ms"])) < 1e-9
        assert at[-1] <= s["step_ms"]
