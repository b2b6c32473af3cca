"""Pinned host -> device copy rate of this box: python scripts/h2d_rate.py"""
import os, time, torch
buf = torch.empty((64 << 20,), dtype=torch.uint8).pin_memory()
dev = torch.empty_like(buf, device="cuda:0")
s = torch.cuda.Stream()
for _ in range(2):
    with torch.cuda.stream(s): dev.copy_(buf, non_blocking=True)
    s.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    with torch.cuda.stream(s): dev.copy_(buf, non_blocking=True)
    s.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"HSA_ENABLE_SDMA={os.environ.get('HSA_ENABLE_SDMA','unset')}: pinned H2D 64 MiB in {dt*1e3:.2f} ms = {64 * 1.048576 / dt / 1e3:.1f} GB/s")
