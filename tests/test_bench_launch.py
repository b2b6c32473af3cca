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
