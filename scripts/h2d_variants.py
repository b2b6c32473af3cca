"""How a 16.8 MB batch tensor gets to the device: .to() vs copy_ into a preallocated buffer: python scripts/h2d_variants.py"""
import time, torch, numpy as np
a = np.random.default_rng(0).random((32, 256, 256, 2), dtype=np.float32)
buf = torch.empty(a.shape, dtype=torch.float32).pin_memory(); buf.copy_(torch.from_numpy(a))
s = torch.cuda.Stream()
def t(f, n=10):
    for _ in range(2): f(); s.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f(); s.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
dev = torch.empty(a.shape, device="cuda:0")
def v1():
    with torch.cuda.stream(s): return buf.to("cuda:0", non_blocking=True)
def v2():
    with torch.cuda.stream(s): dev.copy_(buf, non_blocking=True)
def v3():
    with torch.cuda.stream(s):
        d = torch.empty(a.shape, device="cuda:0"); d.copy_(buf, non_blocking=True); return d
print("to():", t(v1), "ms; copy_ prealloc:", t(v2), "ms; empty+copy_:", t(v3), "ms; is_pinned", buf.is_pinned(), flush=True)
t0 = time.perf_counter()
for _ in range(10): buf.copy_(torch.from_numpy(a))
print("pageable->pinned:", (time.perf_counter() - t0) / 10 * 1e3, "ms", flush=True)
