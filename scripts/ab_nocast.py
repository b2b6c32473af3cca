"""Upper bound of what moving the per-step bf16 work-copy cast off the critical path could buy: the same step with the cast launch
skipped (stale but non-zero work copies; timing only).  python scripts/ab_nocast.py"""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import unet_rir_amd as U
import bench
dev = torch.device("cuda:0")
trs = []
for v in (0, 1):
    eng = U.UNetEngine(256, 256, 32, F0=64, k=3, device=dev, dtype="bf16", overlap_wgrad=True)
    g = torch.Generator(); g.manual_seed(0); eng.reset_parameters(g)
    trs.append(U.Trainer(eng, lr=5e-7))
batch = bench.synthetic_batch(32, 256, 256, dev, 1234)
real = U.ops.cast_weights_batched
res = {0: [], 1: []}
for r in range(6):
    for v in (0, 1):
        tr = trs[v]
        U.ops.cast_weights_batched = real
        tr.step(*batch)                      # the work copies hold real (non-zero) kernels: zero operands would draw less power and clock higher
        U.ops.cast_weights_batched = (lambda t: None) if v else real
        for _ in range(3): tr.step(*batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(12): tr.step(*batch)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 12 * 1e3)
for v in (0, 1):
    t = sorted(res[v]); print("cast skipped" if v else "cast as is", "median", round(t[len(t)//2], 3), "min", round(t[0], 3))
