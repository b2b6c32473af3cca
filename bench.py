#!/usr/bin/env python3
"""bench.py - RIR spectrograms/sec for one full U-Net train step (forward + loss + backward + gradient
all-reduce + Adam) on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Either the caller starts the ranks (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE come from the
environment), or - when WORLD_SIZE is not set - this process starts them itself as children through the same launcher BEFORE
touching the GPU, waits, and exits with their status (the reference is single-process multi-GPU: tf.distribute.MirroredStrategy,
main_training.py:56, :323-327, so `bench.py --gpus 8` alone must work).

Workload at every N: BASELINE.json configs[1] per GPU - dl_models/u_net.py UNet, 4 down / 4 up, number_filters_0=64,
kernels=3, batch 32 of [2,256,256] synthetic amp/phase spectrograms + [2,16] information vectors (weak scaling: the
global batch is 32*N, configs[2] at N=8).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# dense MFMA peaks from /opt/skills/guides/MI355X_MICROARCH.md: fp32 v_mfma_f32_32x32x2_f32 = 256 CU x 4 SIMD x 64 FLOP/clk
# x 2.4 GHz; bf16 ~2.5 PFLOP/s dense (the 5 PF headline includes 2:1 sparsity)
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}


def synthetic_batch(B, H, W, device, seed):
    """Device-side synthetic inputs with the contract of DataGenerator.__getitem__ (datageneratorv2.py:64-102):
    amp/phase ~ U[0,1) with the zero-padded border of TensorPadder (129/144 rows, 151/160 cols), emb in [26,1282)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    spec_in = torch.rand((B, 2, H, W), device=device, generator=g)
    spec_out = torch.rand((B, 2, H, W), device=device, generator=g)
    r0, c0 = -(-896 * H // 1000), -(-944 * W // 1000)
    for a in (spec_in, spec_out):
        a[:, :, r0:, :] = 0.0
        a[:, :, :, c0:] = 0.0
    emb = torch.randint(26, 1282, (B, 2, 16), device=device, generator=g, dtype=torch.int64)
    return spec_in.contiguous(), emb, spec_out.contiguous()


def cpu_baseline(F0, H, W, budget_s=45.0):
    """The CPU restatement of the same train step (oracle/torch_ref.py, stock torch CPU ops) on this box's host cores,
    on a bounded sample: the same model at batch 4, a few steps."""
    from oracle import torch_ref as R
    B = 4
    cfg = R.Config(H, W, F0, 3)
    params = R.init_params(cfg)
    st = R.TrainState(cfg, params, lr=5e-7)
    spec_in, emb, spec_out = R.synthetic_batch(cfg, B)
    spec_in, spec_out, emb = torch.tensor(spec_in), torch.tensor(spec_out), torch.tensor(emb)
    t0 = time.time()
    st.step(spec_in, emb, spec_out)               # warm-up
    warm = time.time() - t0
    times = []
    while (not times) or (sum(times) + warm + (sum(times) / len(times)) < budget_s and len(times) < 5):
        t0 = time.time()
        st.step(spec_in, emb, spec_out)
        times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": B / med, "unit": "spectrograms/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"same model (F0={F0}, {H}x{W}) at batch {B}, median of {len(times)} steps after 1 warm-up, "
                      f"oracle/torch_ref.py on torch CPU fp32, os.cpu_count()={os.cpu_count()}"}


def traffic(dtype):
    """HBM bytes per launch of the dominant conv kernel from the committed PMC passes (profiles/pmc_traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 FETCH_SIZE x2 correction); None if the file is absent.  It is a
    recorded profile of this same command, not a live measurement (counters cannot be read from inside the process)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)[dtype]
        return t["traffic_bytes_per_launch"], f'{t["kernel"]}: bytes per launch, profiles/pmc_traffic.json'
    except Exception:
        return None, None


def run_mode(U, args, dtype, world, rank, device, steps, warmup):
    """Build the engine in one storage mode, run `warmup` untimed and `steps` timed train steps; returns a result dict
    (rank 0) or None.  Timing: barrier + synchronize on both sides, MAX over ranks."""
    H = W = args.size
    overlap = not args.no_overlap
    eng = U.UNetEngine(H, W, args.batch, F0=args.f0, k=3, depth=args.depth, device=device, n_replicas=world, dtype=dtype,
                       overlap_wgrad=overlap)
    gen = torch.Generator()
    gen.manual_seed(0)                                   # identical initial variables on every replica
    eng.reset_parameters(gen)
    tr = U.Trainer(eng, lr=5e-7, alpha=0.9, world_size=world, bucket_bytes=args.bucket_mb << 20, dropout=True)
    tr.broadcast_parameters(0)
    spec_in, emb, spec_out = synthetic_batch(args.batch, H, W, device, 1234 + rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(device_ids=[device.index])
            torch.cuda.synchronize()

    for _ in range(warmup):
        tr.step(spec_in, emb, spec_out)
    sync()
    prof = (not args.no_prof) and rank == 0
    if prof:
        # overlapped schedule: only the forward brackets are clean kernel durations (see below), so only those are taken -
        # every bracket is two event records in the stream (~8 us of dispatch gap each)
        U.ops.prof_enable(2 if overlap and os.environ.get("UNETRIR_BENCH_PROF_ALL") != "1" else 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(spec_in, emb, spec_out)
    sync()
    dt = time.perf_counter() - t0
    fam = None
    if prof:
        fam = U.ops.prof_collect()
        U.ops.prof_enable(False)
    loss = tr.last_loss()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    n_params = eng.n_params()
    del tr, eng
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    global_batch = args.batch * world
    res = {"value": global_batch * steps / dt, "ms_per_step": dt * 1e3 / steps, "steps": steps, "warmup": warmup, "dtype": dtype,
           "loss": loss, "params": n_params}
    if fam is not None:
        counts, fms, ffl = fam
        # With the weight gradients on their own stream (default) the backward brackets overlap each other, so a bracket's
        # time is no longer the kernel's own duration: the roofline figure then comes from the FORWARD convolution launches
        # of the same timed region, which run alone on the stream (same kernels as the data gradients).
        fams = (0,) if overlap else (0, 1, 2)
        conv_ms, conv_fl, nl = sum(fms[i] for i in fams), sum(ffl[i] for i in fams), sum(counts[i] for i in fams)
        ach = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[dtype]
        res["roofline"] = {
            "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic(dtype)[0],
            "traffic_note": traffic(dtype)[1],
            "kernel": ("igemm_fwd_kernel / wgrad3x3_kernel (fp32 v_mfma_f32_32x32x2_f32 implicit-GEMM conv: fwd, dgrad, wgrad)"
                       if dtype == "f32" else
                       "conv3x3g / conv3x3h / upconv3x3 / igemm_fwd_bf16 (fwd, dgrad) + wgrad3x3g / wgrad3x3_bf16 (wgrad): bf16 "
                       "v_mfma_f32_16x16x32_bf16 / 32x32x16 convolutions, fp32 accumulate") + "; layers with Cin,Cout > 8",
            "measured_on": ("forward conv launches of the timed region (backward brackets overlap: wgrad runs on a side stream)"
                            if overlap else "all conv launches (fwd, dgrad, wgrad) of the timed region"),
            "calls_per_step": nl / steps, "avg_call_ms": conv_ms / max(nl, 1),
            "algorithmic_gflop_per_step": conv_fl / steps / 1e9, "conv_ms_per_step": conv_ms / steps,
            "by_family": {name: {"calls": counts[i], "ms": fms[i], "tflops": (ffl[i] / (fms[i] * 1e-3) / 1e12 if fms[i] > 0 else 0.0)}
                          for i, name in ((0, "fwd"), (1, "dgrad"), (2, "wgrad"), (5, "stem_head_dgrad"))},
        }
    return res


def launch_ranks(n):
    """Start n ranks of this script (one per GPU) through torch.distributed.run as a CHILD process and return its exit
    status.  Nothing in this parent has touched the GPU (device_count() does not), and nothing is exec'ed."""
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = str(s_.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--no-overlap", action="store_true", help="weight gradients on the main stream (clean per-kernel brackets)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--f0", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-launch HIP-event bracketing of conv kernels")
    ap.add_argument("--bucket-mb", type=int, default=32)
    ap.add_argument("--dtype", choices=["bf16", "f32", "both"], default="both",
                    help="storage type of activations/gradients (accumulation, statistics, master weights are always fp32). "
                         "'both' (default): bf16 is the headline line (BASELINE.json configs[1] names bf16) and the fp32 "
                         "parity mode is measured as well and reported under 'fp32_mode'")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU")
    n_dev = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if n_dev < world or local_rank >= n_dev:
        raise SystemExit(f"bench.py rank {rank}: --gpus {world} needs {world} visible GPUs, this node shows {n_dev}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("UNETRIR_FORCE_DP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # No device_id: RCCL's communicator and its internal stream are then created at the first collective, AFTER the engine
        # has probed and chosen its own concurrent streams.  HIP multiplexes streams onto a few hardware queues; created first,
        # RCCL's stream can share a queue with the main or the weight-gradient stream, and its event waits (a gradient bucket
        # waiting for the backward pass) then stall every kernel queued behind them: measured 15.0-15.3 instead of 13.9 ms per
        # step in about half of the runs (1-rank rehearsal, UNETRIR_FORCE_DP=1); created afterwards it lands on a free queue.
        if os.environ.get("UNETRIR_BENCH_EAGER_PG") == "1":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world)

    import unet_rir_amd as U
    H = W = args.size
    head_dtype = "bf16" if args.dtype in ("bf16", "both") else "f32"
    main_res = run_mode(U, args, head_dtype, world, rank, device, args.steps, args.warmup)
    f32_res = None
    if args.dtype == "both" and world == 1:          # the fp32 parity mode is a single-GPU side figure
        f32_res = run_mode(U, args, "f32", world, rank, device, min(args.steps, 5), min(args.warmup, 2))
    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    global_batch = args.batch * world
    if (args.f0, args.size, args.depth, args.batch) == (64, 256, 4, 32):
        cfg_name = "BASELINE.json configs[1]"
    elif (args.f0, args.size, args.depth, args.batch) == (128, 512, 5, 16):
        cfg_name = "BASELINE.json configs[3]'s model"
    else:
        cfg_name = "non-BASELINE shape"
    out = {
        "metric": "RIR spectrograms/sec (train step) on [B,2,256,256] U-Net",
        "value": main_res["value"],
        "unit": "spectrograms/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": head_dtype, "data": "synthetic",
        "config": {"workload": f"UNet {args.depth} down/{args.depth} up, number_filters_0={args.f0}, kernels=3, per-GPU batch "
                               f"{args.batch} of [2,{H},{W}] + [2,16] ({cfg_name} per GPU; global batch "
                               f"{global_batch}), full train step: fwd + loss + bwd + grad all-reduce + Adam; "
                               + ("bf16 activations/gradients, fp32 accumulate/statistics/master weights" if head_dtype == "bf16"
                                  else "fp32 throughout"),
                   "global_batch": global_batch, "parallelism": f"dp{world}", "params": main_res["params"]},
        "loss": main_res["loss"],
    }
    if "roofline" in main_res:
        out["roofline"] = main_res["roofline"]
    if f32_res is not None:
        out["fp32_mode"] = {k: f32_res[k] for k in ("value", "ms_per_step", "steps", "warmup", "loss") if k in f32_res}
        out["fp32_mode"]["unit"] = "spectrograms/s"
        out["fp32_mode"]["note"] = "same workload with fp32 storage: the mode the fp32-tolerance parity tests run in"
        if "roofline" in f32_res:
            out["fp32_mode"]["roofline"] = f32_res["roofline"]
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.f0, H, W)
    print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
