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

What the line holds besides the contract fields (N = 1):
  roofline      the MFMA convolution kernels, HIP-event brackets on the launch stream inside the timed region: `frac` /
                `frac_fwd` over the forward launches of the timed (side-stream) run, `frac_all` + `by_family` over forward,
                data-gradient and weight-gradient launches of a short single-stream sub-run of the same process (with the
                weight gradients on their own stream the backward brackets overlap and are not kernel durations)
  module_path   the same workload driven through the drop-in boundary: UNet(nn.Module) -> model.model([spec, emb],
                training=True) -> Trainer.compute_loss -> loss.backward() -> Trainer.apply_gradients()
  fp32_mode     the same workload with fp32 storage (the reference's own arithmetic; the fp32-tolerance parity mode)
  configs       one-GPU figures of the other BASELINE.json configurations (configs[3]'s model, configs[4] ResAE)
  cpu_baseline  the CPU restatement (oracle/torch_ref.py) on this box's host cores, bounded samples
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


def _cpu_steps(F0, H, W, B, steps=5, warmups=2, budget_s=150.0):
    """median of >= 5 timed steps after 2 warm-ups (a step of the benchmarked model at batch 4 takes ~10 s on
    the GPU box's host cores).  `budget_s` only guards against a pathologically slow host: the count of steps actually timed is
    reported."""
    from oracle import torch_ref as R
    cfg = R.Config(H, W, F0, 3)
    params = R.init_params(cfg)
    st = R.TrainState(cfg, params, lr=5e-7)
    spec_in, emb, spec_out = R.synthetic_batch(cfg, B)
    spec_in, spec_out, emb = torch.tensor(spec_in), torch.tensor(spec_out), torch.tensor(emb)
    t_begin = time.time()
    for _ in range(warmups):
        st.step(spec_in, emb, spec_out)
    times = []
    while len(times) < steps and (len(times) < 1 or time.time() - t_begin < budget_s):
        t0 = time.time()
        st.step(spec_in, emb, spec_out)
        times.append(time.time() - t0)
    times.sort()
    return B / times[len(times) // 2], len(times), warmups


def cpu_baseline(F0, H, W):
    """The CPU restatement of the same train step (oracle/torch_ref.py, stock torch CPU ops) on this box's host cores, on
    bounded samples: the benchmarked model at batch 4, and BASELINE.json configs[0] exactly (B=4, F0=16, 256x256, fp32)."""
    v, n, wu = _cpu_steps(F0, H, W, 4)
    out = {"value": v, "unit": "spectrograms/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"same model (F0={F0}, {H}x{W}) at batch 4, median of {n} steps after {wu} warm-ups, "
                     f"oracle/torch_ref.py on torch CPU fp32, os.cpu_count()={os.cpu_count()}"}
    v1, n1, wu1 = _cpu_steps(16, 256, 256, 4)
    out["cfg1"] = {"value": v1, "unit": "spectrograms/s",
                   "sample": f"BASELINE.json configs[0] as stated: UNet base=16ch, batch=4 of [2,256,256], fp32, torch CPU; median of "
                             f"{n1} steps after {wu1} warm-ups"}
    return out


def traffic(dtype):
    """HBM bytes per launch of the dominant conv kernel from the committed PMC passes: profiles/r04_roofline_table.json (made by
    scripts/roofline_table.py from the kernel-trace statistics and the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh,
    gfx950 FETCH_SIZE x2 correction), else the older profiles/pmc_traffic.json; None if neither is there.  It is a recorded profile
    of this same command, not a live measurement (counters cannot be read from inside the process)."""
    if dtype == "bf16":
        try:
            with open(os.path.join(ROOT, "profiles", "r04_roofline_table.json")) as f:
                t = json.load(f)
            k = t["kernels"][t["dominant_kernel"]]
            if k.get("traffic_bytes_per_launch"):
                return k["traffic_bytes_per_launch"], f'{t["dominant_kernel"]}: bytes per launch, profiles/r04_roofline_table.json'
        except Exception:
            pass
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)[dtype]
        return t["traffic_bytes_per_launch"], f'{t["kernel"]}: bytes per launch, profiles/pmc_traffic.json'
    except Exception:
        return None, None


def _fam_stats(fam, fams, dtype):
    counts, fms, ffl = fam
    ms, fl, nl = sum(fms[i] for i in fams), sum(ffl[i] for i in fams), sum(counts[i] for i in fams)
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return ach, ach / MFMA_PEAK_TFLOPS[dtype], ms, fl, nl


def _by_family(fam):
    counts, fms, ffl = fam
    return {name: {"calls": counts[i], "ms": fms[i], "tflops": (ffl[i] / (fms[i] * 1e-3) / 1e12 if fms[i] > 0 else 0.0)}
            for i, name in ((0, "fwd"), (1, "dgrad"), (2, "wgrad"), (5, "stem_head_dgrad"))}


def _dominant(fam, steps, dtype, where):
    """The dominant kernel's OWN figure (bf16: conv3x3p_bf16_kernel; its launches are counted under family 6 in addition to
    their forward / data-gradient family): algorithmic FLOPs of its launches / their HIP-event time."""
    counts, fms, ffl = fam
    if counts[6] == 0 or fms[6] <= 0:
        return None
    ach = ffl[6] / (fms[6] * 1e-3) / 1e12
    return {"kernel": "conv3x3p_bf16_kernel", "calls_per_step": counts[6] / steps, "avg_call_ms": fms[6] / counts[6],
            "algorithmic_gflop_per_step": ffl[6] / steps / 1e9, "achieved": ach, "frac": ach / MFMA_PEAK_TFLOPS[dtype], "measured_on": where}


class DpProbe:
    """What the data-parallel schedule does with its gradient buckets, measured on INSTRUMENTED steps outside the timed region
    (every bracket is an event record in a stream; the timed steps carry none).  Per bucket: bytes, the time from the start of the
    step to the point where the bucket was handed to the collective (its gradients final, on the stream that hands it over), the time
    from there until the stream that waits for the all-reduce saw it complete, and the effective rate; per step: the length of the
    backward pass and the EXPOSED TAIL - what is left of the step after the last backward kernel (all-reduce of the last buckets +
    their optimizer launches), which is what bounds weak-scaling efficiency from above.  Works on any runtime object of
    unet_rir_amd/device.py's interface (tests/test_bench_launch.py drives it on two gloo ranks over the simulated runtime)."""

    def __init__(self, trainer):
        self.tr, self.rt = trainer, trainer.rt
        self.steps = []

    def __enter__(self):
        rt, tr = self.rt, self.tr
        self._saved = (rt.all_reduce_sum, rt.wait_work, tr.apply_gradients)
        probe = self

        def all_reduce_sum(tensor, group=None):
            ev = rt.record(timing=True)
            work = probe._saved[0](tensor, group)
            probe._cur["buckets"].append({"bytes": tensor.numel() * tensor.element_size(), "issue": ev, "done": None, "work": work})
            return work

        def wait_work(work):
            probe._saved[1](work)
            for b in probe._cur["buckets"]:
                if b["work"] is work and b["done"] is None:
                    b["done"] = rt.record(timing=True)

        def apply_gradients():
            probe._cur["bwd_end"] = rt.record(timing=True)       # every backward launch (and the join with the side stream) is queued
            probe._saved[2]()
            probe._cur["end"] = rt.record(timing=True)

        rt.all_reduce_sum, rt.wait_work, tr.apply_gradients = all_reduce_sum, wait_work, apply_gradients
        return self

    def __exit__(self, *exc):
        self.rt.all_reduce_sum, self.rt.wait_work, self.tr.apply_gradients = self._saved
        return False

    def step(self, fn):
        self._cur = {"buckets": []}
        self._cur["start"] = self.rt.record(timing=True)
        fn()
        self.steps.append(self._cur)

    def summary(self):
        """Means over the probed steps."""
        rt, n = self.rt, len(self.steps)
        if n == 0:
            return None
        nb = len(self.steps[0]["buckets"])
        out = {"probed_steps": n, "buckets": [], "note": "instrumented steps outside the timed region"}
        for i in range(nb):
            issue = sum(rt.elapsed_ms(st["start"], st["buckets"][i]["issue"]) for st in self.steps) / n
            have = [st for st in self.steps if st["buckets"][i]["done"] is not None]
            dur = sum(rt.elapsed_ms(st["buckets"][i]["issue"], st["buckets"][i]["done"]) for st in have) / len(have) if have else None
            by = self.steps[0]["buckets"][i]["bytes"]
            out["buckets"].append({"bytes": by, "issued_at_ms": issue, "issue_to_complete_ms": dur,
                                   "effective_GBps": (by / (dur * 1e-3) / 1e9) if dur else None})
        out["backward_end_at_ms"] = sum(rt.elapsed_ms(st["start"], st["bwd_end"]) for st in self.steps) / n
        out["step_ms"] = sum(rt.elapsed_ms(st["start"], st["end"]) for st in self.steps) / n
        out["exposed_tail_ms"] = out["step_ms"] - out["backward_end_at_ms"]
        out["allreduce_bytes_per_step"] = sum(b["bytes"] for b in self.steps[0]["buckets"])
        return out


class Runner:
    def __init__(self, U, args, world, rank, device):
        self.U, self.args, self.world, self.rank, self.device = U, args, world, rank, device

    def sync(self):
        torch.cuda.synchronize()
        if self.world > 1:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[self.device.index])
            else:
                dist.barrier()
            torch.cuda.synchronize()

    def replicas_in_sync(self, theta):
        """MirroredStrategy's invariant after the timed steps (outside the timed region): every replica holds bit-identical
        variables - the SUM all-reduce and the optimizer ran on the same values everywhere.  Compared through two fp64
        checksums (sum and sum of squares) gathered from all ranks."""
        t = theta.double()
        mine = torch.stack([t.sum(), (t * t).sum()])
        allv = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(allv, mine)
        return bool(all(torch.equal(v, allv[0]) for v in allv))

    def timed(self, step, steps, warmup, prof_level=0):
        """`warmup` untimed and EXACTLY `steps` timed calls of step(); barrier + synchronize on both sides, MAX over ranks."""
        for _ in range(warmup):
            step()
        self.sync()
        if prof_level:
            self.U.ops.prof_enable(prof_level)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.sync()
        dt = time.perf_counter() - t0
        fam = None
        if prof_level:
            fam = self.U.ops.prof_collect()
            self.U.ops.prof_enable(False)
        if self.world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        return dt, fam

    def engine_run(self, dtype, steps, warmup, overlap, prof=True, f0=None, size=None, depth=None, batch=None, hw=None):
        """Trainer.step over a UNetEngine.  Returns a result dict.  hw: (H, W) of a non-square input."""
        U, a = self.U, self.args
        f0, size, depth, batch = f0 or a.f0, size or a.size, depth or a.depth, batch or a.batch
        H_, W_ = hw if hw is not None else (size, size)
        eng = U.UNetEngine(H_, W_, batch, F0=f0, k=3, depth=depth, device=self.device, n_replicas=self.world, dtype=dtype,
                           overlap_wgrad=overlap)
        gen = torch.Generator()
        gen.manual_seed(0)                                   # identical initial variables on every replica
        eng.reset_parameters(gen)
        eng.dropout_seed = 1234 + self.rank
        tr = U.Trainer(eng, lr=5e-7, alpha=0.9, world_size=self.world, bucket_bytes=a.bucket_mb << 20, dropout=True,
                       force_dp=a.force_dp)
        tr.broadcast_parameters(0)
        spec_in, emb, spec_out = synthetic_batch(batch, H_, W_, self.device, 1234 + self.rank)
        # overlapped schedule: only the forward brackets are clean kernel durations, so only those are taken (level 2) - every
        # bracket is two event records in the stream (~8 us of dispatch gap each)
        level = 0 if (not prof or a.no_prof or self.rank != 0) else (2 if overlap and not a.prof_all else 1)
        dt, fam = self.timed(lambda: tr.step(spec_in, emb, spec_out), steps, warmup, level)
        res = {"value": batch * self.world * steps / dt, "ms_per_step": dt * 1e3 / steps, "steps": steps, "warmup": warmup,
               "dtype": dtype, "loss": tr.last_loss(), "params": eng.n_params(), "fam": fam, "overlap": overlap}
        if (self.world > 1 or a.force_dp) and prof:
            with DpProbe(tr) as probe:           # after the timed region
                for _ in range(5):
                    probe.step(lambda: tr.step(spec_in, emb, spec_out))
                self.sync()
                res["dp_probe"] = probe.summary()
        if self.world > 1:
            res["replicas_in_sync"] = self.replicas_in_sync(eng.theta)
        del tr, eng
        torch.cuda.empty_cache()
        return res

    def module_run(self, dtype, steps, warmup, overlap):
        """The same workload through the drop-in boundary (the reference's loop shape, main_training.py:253-290)."""
        U, a = self.U, self.args
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(0)
        model = U.UNet((a.size, a.size, 2), (2, 16), mode=0, number_filters_0=a.f0, kernels=3, name="U-Net", depth=a.depth,
                       batch_size=a.batch, device=self.device, n_replicas=self.world, dtype=dtype, overlap=overlap)
        torch.random.set_rng_state(gen_state)
        model.engine.dropout_seed = 1234 + self.rank
        tr = U.Trainer(model, lr=5e-7, alpha=0.9, world_size=self.world, bucket_bytes=a.bucket_mb << 20, force_dp=a.force_dp)
        tr.broadcast_parameters(0)
        spec_in, emb, spec_out = synthetic_batch(a.batch, a.size, a.size, self.device, 1234 + self.rank)
        # DataGenerator.__getitem__ yields NHWC (datageneratorv2.py:101-102)
        spec_in, spec_out = spec_in.permute(0, 2, 3, 1).contiguous(), spec_out.permute(0, 2, 3, 1).contiguous()

        def step():
            pred = model.model([spec_in, emb], training=True)
            loss = tr.compute_loss(spec_out, pred)
            loss.backward()
            tr.apply_gradients()

        dt, _ = self.timed(step, steps, warmup, 0)
        res = {"value": a.batch * self.world * steps / dt, "ms_per_step": dt * 1e3 / steps, "steps": steps, "warmup": warmup,
               "loss": tr.last_loss()}
        del tr, model
        torch.cuda.empty_cache()
        return res

    def resae_run(self, steps, warmup, dtype="bf16"):
        """BASELINE.json configs[4]: ResAE (main_training.py:130-140: filters 32..256, k 3, s 2, latent 32, n_neurons 1024),
        batch 32 of [2,256,256], through its boundary class and the Trainer; and the same step captured once into a HIP graph and
        replayed (single stream: one host call per step)."""
        U = self.U
        spec_in, emb, spec_out = synthetic_batch(32, 256, 256, self.device, 1234)
        out = {}
        for key, overlap, graph in (("launched", True, False), ("graph", False, True)):
            model = U.ResAE((256, 256, 2), (2, 16), (32, 64, 128, 256), (3, 3, 3, 3), (2, 2, 2, 2), 32, 16 * 64, name="resae",
                            batch_size=32, device=self.device, dtype=dtype, overlap=overlap)
            tr = U.Trainer(model, lr=5e-7, alpha=0.9, graph=graph)
            for _ in range(warmup):
                tr.step(spec_in, emb, spec_out)
            self.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                tr.step(spec_in, emb, spec_out)
            host = time.perf_counter() - t0
            self.sync()
            dt = time.perf_counter() - t0
            out[key] = {"value": 32 * steps / dt, "ms_per_step": dt * 1e3 / steps, "host_ms_per_step": host * 1e3 / steps}
            params = model.engine.n_params()
            del tr, model
            torch.cuda.empty_cache()
        # The launched form keeps the host busy for 4.7-5.4 ms of a 6.3 ms step (~590 launches): on a box with a slow host it becomes
        # host-bound (9.3 ms seen once), the graph replay does not.  `value` is ALWAYS the default form (launched, side-stream
        # schedule: what `ResAE(...)` + `Trainer(model)` give); the graph replay (`Trainer(graph=True)`) stands beside it.
        best = "launched"
        res = {"workload": "BASELINE.json configs[4]: ResAE filters (32,64,128,256), k 3, s 2, latent 32, n_neurons 1024, batch 32 of "
                           "[2,256,256], full train step, " + ("bf16 storage of the convolutional trunk" if dtype == "bf16" else "fp32 storage"),
               "schedule": "launches, side-stream weight gradients" if best == "launched" else "HIP-graph replay, single stream (Trainer(graph=True))",
               "value": out[best]["value"], "unit": "spectrograms/s", "ms_per_step": out[best]["ms_per_step"],
               "host_ms_per_step": out[best]["host_ms_per_step"], "steps": steps, "warmup": warmup, "params": params, "dtype": dtype,
               "launched_side_stream": out["launched"], "hip_graph_replay_single_stream": out["graph"]}
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


def roofline_block(main, sub, dtype):
    """main: the timed headline run (forward brackets when overlapped, all families otherwise); sub: optional single-stream
    sub-run with all three families bracketed."""
    overlap = main["overlap"]
    fams_main = (0,) if overlap else (0, 1, 2)
    ach, frac, ms, fl, nl = _fam_stats(main["fam"], fams_main, dtype)
    steps = main["steps"]
    r = {
        "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK_TFLOPS[dtype], "unit": "TFLOP/s", "frac": frac,
        "traffic": traffic(dtype)[0], "traffic_note": traffic(dtype)[1],
        "kernel": ("igemm_fwd_kernel / conv3x3_kernel / wgrad3x3_kernel (fp32 v_mfma_f32_32x32x2_f32 implicit-GEMM conv: fwd, dgrad, wgrad)"
                   if dtype == "f32" else
                   "conv3x3p / conv3x3g / conv3x3s / conv3x3h / upconv3x3q / conv3x3d (fwd, dgrad) + wgrad3x3g / wgrad3x3d (wgrad): bf16 "
                   "v_mfma_f32_16x16x32_bf16 / 32x32x16 convolutions, fp32 accumulate") + "; layers with Cin,Cout > 8",
        "measured_on": ("forward conv launches of the timed region (backward brackets overlap: wgrad runs on a side stream)"
                        if overlap else "all conv launches (fwd, dgrad, wgrad) of the timed region"),
        "calls_per_step": nl / steps, "avg_call_ms": ms / max(nl, 1),
        "algorithmic_gflop_per_step": fl / steps / 1e9, "conv_ms_per_step": ms / steps,
    }
    if overlap:
        r["frac_fwd"] = frac
    dom = _dominant(main["fam"], steps, dtype, "its forward launches in the timed (side-stream) run" if overlap else
                    "all its launches (forward + data gradient) in the timed run")
    if dom is not None:
        r["dominant_kernel"] = dom
    if sub is not None and sub.get("fam") is not None:
        dsub = _dominant(sub["fam"], sub["steps"], dtype, "all its launches (forward + data gradient) in the single-stream sub-run")
        if dsub is not None:
            r["dominant_kernel_single_stream"] = dsub
        a_all, f_all, ms_a, fl_a, nl_a = _fam_stats(sub["fam"], (0, 1, 2), dtype)
        a_f, f_f, _, _, _ = _fam_stats(sub["fam"], (0,), dtype)
        r["frac_all"] = f_all
        r["single_stream_run"] = {"steps": sub["steps"], "ms_per_step": sub["ms_per_step"], "achieved_all": a_all, "frac_all": f_all,
                                  "achieved_fwd": a_f, "frac_fwd": f_f, "calls_per_step": nl_a / sub["steps"],
                                  "algorithmic_gflop_per_step": fl_a / sub["steps"] / 1e9, "conv_ms_per_step": ms_a / sub["steps"],
                                  "by_family": _by_family(sub["fam"]),
                                  "note": "weight gradients on the main stream: every bracket is one kernel's own duration"}
    else:
        r["by_family"] = _by_family(main["fam"])
        if not overlap:
            r["frac_all"] = frac
    return r


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
    ap.add_argument("--prof-all", action="store_true", help="bracket all three conv families in the overlapped run as well")
    ap.add_argument("--bucket-mb", type=int, default=32)
    ap.add_argument("--force-dp", action="store_true", help="1-rank rehearsal of the bucketed RCCL all-reduce path")
    ap.add_argument("--eager-pg", action="store_true", help="create the RCCL communicator before the engine's streams (A/B)")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="REHEARSAL of the N > 1 rank path on a one-GPU box: every rank uses cuda:0 and the collectives go through "
                         "gloo (RCCL refuses two ranks on one device).  Exercises the product step at world_size N with real kernels; "
                         "the value it prints is not a scaling measurement and says so")
    ap.add_argument("--through-module", action="store_true",
                    help="headline value from the drop-in boundary (UNet module + Trainer.compute_loss) instead of Trainer.step")
    ap.add_argument("--lean", action="store_true", help="headline run only (no sub-runs, other configs or CPU baseline)")
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
    if args.rehearse_shared_gpu:
        local_rank = 0
    if n_dev < (1 if args.rehearse_shared_gpu else world) or local_rank >= n_dev:
        raise SystemExit(f"bench.py rank {rank}: --gpus {world} needs {world} visible GPUs, this node shows {n_dev}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # No device_id: RCCL's communicator and its internal stream are then created at the first collective, AFTER the engine
        # has probed and chosen its own concurrent streams.  HIP multiplexes streams onto a few hardware queues; created first,
        # RCCL's stream can share a queue with the main or the weight-gradient stream, and its event waits (a gradient bucket
        # waiting for the backward pass) then stall every kernel queued behind them: measured 15.0-15.3 instead of 13.9 ms per
        # step in about half of the runs (1-rank rehearsal, --force-dp); created afterwards it lands on a free queue.
        if args.rehearse_shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        elif args.eager_pg:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world)

    import unet_rir_amd as U
    run = Runner(U, args, world, rank, device)
    H = W = args.size
    overlap = not args.no_overlap
    head_dtype = "bf16" if args.dtype in ("bf16", "both") else "f32"
    full = world == 1 and not args.lean              # side figures are single-GPU material

    eng_res = run.engine_run(head_dtype, args.steps, args.warmup, overlap)
    single_ref = None
    if world > 1 and not args.lean and not args.rehearse_shared_gpu:
        # thr(N) / (N thr(1)) against a 1-GPU figure of the SAME invocation: rank 0 runs the same per-GPU workload alone (no
        # collective, the other GPUs idle at the barrier behind it) - same box, same process, same kernels
        if rank == 0:
            one = Runner(U, args, 1, 0, device).engine_run(head_dtype, args.steps, args.warmup, overlap, prof=False)
            single_ref = {"value": one["value"], "ms_per_step": one["ms_per_step"], "steps": one["steps"], "warmup": one["warmup"]}
        run.sync()
    mod_res = None
    if args.through_module or full:
        mod_res = run.module_run(head_dtype, args.steps if args.through_module else min(args.steps, 10), min(args.warmup, 3), overlap)
    sub = f32_res = f32_sub = None
    configs = {}
    if full:
        if overlap and not args.no_prof:
            sub = run.engine_run(head_dtype, min(args.steps, 5), 2, False)
        if args.dtype == "both":
            f32_res = run.engine_run("f32", max(args.steps, 20), max(min(args.warmup, 3), 2), overlap)
            f32_sub = run.engine_run("f32", 5, 2, False) if overlap and not args.no_prof else None
        if (args.f0, args.size, args.depth, args.batch) == (64, 256, 4, 32):
            c4 = run.engine_run("bf16", 10, 2, overlap, prof=False, f0=128, size=512, depth=5, batch=16)
            configs["cfg4_model_1gpu"] = {
                "workload": "BASELINE.json configs[3]'s model on ONE GPU: UNet 5 down/5 up, number_filters_0=128, kernels=3, batch 16 of "
                            "[2,512,512], full train step, bf16 storage", "value": c4["value"], "unit": "spectrograms/s",
                "ms_per_step": c4["ms_per_step"], "steps": c4["steps"], "warmup": c4["warmup"], "params": c4["params"], "dtype": "bf16"}
            rg = run.engine_run("bf16", 10, 3, overlap, prof=False, f0=32, hw=(144, 160), depth=4, batch=32)
            configs["reference_geometry"] = {
                "workload": "the reference's own training geometry (main_training.py:154-161: target_size (144,160,2), number_filters_0=32, "
                            "kernels=3; 9 x 10 bottleneck), batch 32, full train step, bf16 storage - not a BASELINE.json config",
                "value": rg["value"], "unit": "spectrograms/s", "ms_per_step": rg["ms_per_step"], "steps": rg["steps"],
                "warmup": rg["warmup"], "params": rg["params"], "dtype": "bf16"}
            configs["cfg5_resae"] = run.resae_run(20, 3, "bf16")
            configs["cfg5_resae_fp32"] = run.resae_run(10, 3, "f32")
    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    head = mod_res if args.through_module else eng_res
    global_batch = args.batch * world
    if (args.f0, args.size, args.depth, args.batch) == (64, 256, 4, 32):
        cfg_name = "BASELINE.json configs[1]"
    elif (args.f0, args.size, args.depth, args.batch) == (128, 512, 5, 16):
        cfg_name = "BASELINE.json configs[3]'s model"
    else:
        cfg_name = "non-BASELINE shape"
    out = {
        "metric": "RIR spectrograms/sec (train step) on [B,2,256,256] U-Net",
        "value": head["value"],
        "unit": "spectrograms/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": head_dtype, "data": "synthetic",
        "config": {"workload": f"UNet {args.depth} down/{args.depth} up, number_filters_0={args.f0}, kernels=3, per-GPU batch "
                               f"{args.batch} of [2,{H},{W}] + [2,16] ({cfg_name} per GPU; global batch "
                               f"{global_batch}), full train step: fwd + loss + bwd + grad all-reduce + Adam; "
                               + ("bf16 activations/gradients, fp32 accumulate/statistics/master weights" if head_dtype == "bf16"
                                  else "fp32 throughout"),
                   "global_batch": global_batch, "parallelism": f"dp{world}", "rccl_world_size": world if dist.is_initialized() else 1,
                   **({"replicas_in_sync": eng_res["replicas_in_sync"]} if "replicas_in_sync" in eng_res else {}),
                   **({"REHEARSAL": "all ranks share cuda:0, collectives over gloo: exercises the N-rank code path, NOT a scaling "
                                    "measurement"} if args.rehearse_shared_gpu else {}),
                   "params": eng_res["params"],
                   "driven_by": ("UNet(nn.Module).model(...) + Trainer.compute_loss/apply_gradients" if args.through_module
                                 else "Trainer.step(UNetEngine)"),
                   "schedule": "side-stream weight gradients + bucket-wise Adam" if overlap else "single stream"},
        "loss": head["loss"],
    }
    if eng_res.get("dp_probe") is not None or single_ref is not None:
        out["data_parallel"] = {"world_size": world, "bucket_mb": args.bucket_mb}
        if eng_res.get("dp_probe") is not None:
            out["data_parallel"].update(eng_res["dp_probe"])
        if single_ref is not None:
            out["data_parallel"]["single_gpu_same_invocation"] = single_ref
            out["data_parallel"]["scaling_efficiency"] = head["value"] / (world * single_ref["value"])
            out["data_parallel"]["scaling_efficiency_note"] = "thr(N) / (N thr(1)), thr(1) = rank 0 alone in this same process; the driver computes its own from the per-N lines"
    if eng_res.get("fam") is not None:
        out["roofline"] = roofline_block(eng_res, sub, head_dtype)
    if mod_res is not None:
        out["module_path"] = {"value": mod_res["value"], "unit": "spectrograms/s", "ms_per_step": mod_res["ms_per_step"],
                              "steps": mod_res["steps"], "warmup": mod_res["warmup"], "loss": mod_res["loss"],
                              "vs_engine_path": mod_res["ms_per_step"] / eng_res["ms_per_step"],
                              "note": "UNet(nn.Module, dtype, overlap) -> model.model([spec NHWC, emb], training=True) -> "
                                      "Trainer.compute_loss -> loss.backward() -> Trainer.apply_gradients()"}
        if args.through_module:
            out["engine_path"] = {"value": eng_res["value"], "ms_per_step": eng_res["ms_per_step"]}
    if f32_res is not None:
        out["fp32_mode"] = {k: f32_res[k] for k in ("value", "ms_per_step", "steps", "warmup", "loss") if k in f32_res}
        out["fp32_mode"]["unit"] = "spectrograms/s"
        out["fp32_mode"]["note"] = "same workload with fp32 storage: the mode the fp32-tolerance parity tests run in"
        if f32_res.get("fam") is not None:
            out["fp32_mode"]["roofline"] = roofline_block(f32_res, f32_sub, "f32")
    if configs:
        out["configs"] = configs
    if full and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.f0, H, W)
    print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
