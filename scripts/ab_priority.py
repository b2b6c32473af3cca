"""HIP stream priorities on the full configs[1] step (bf16, side-stream schedule): one process per setting, order rotated.
python scripts/ab_priority.py [rounds]      settings: (priority of the stream the step is issued on | default stream, priority of the side streams | default)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, contextlib
sys.path.insert(0, %r)
mainp, sidep = sys.argv[1], sys.argv[2]
import torch
import unet_rir_amd as U
import bench
dev = torch.device("cuda:0")
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (None, None)
ctx = torch.cuda.stream(torch.cuda.Stream(device=dev, priority=int(mainp))) if mainp != "none" else contextlib.nullcontext()
if sidep != "none":      # every stream the engine creates from here on (its side streams: engine.pick_concurrent_streams) gets this priority
    _Stream = torch.cuda.Stream
    class _PrioStream(_Stream):
        def __new__(cls, device=None, priority=0, **kw):
            return _Stream.__new__(cls, device=device, priority=int(sidep), **kw)
    torch.cuda.Stream = _PrioStream
with ctx:
    eng = U.UNetEngine(256, 256, 32, F0=64, k=3, device=dev, dtype="bf16", overlap_wgrad=True)
    g = torch.Generator(); g.manual_seed(0); eng.reset_parameters(g)
    tr = U.Trainer(eng, lr=5e-7)
    batch = bench.synthetic_batch(32, 256, 256, dev, 1234)
    out = []
    for r in range(4):
        for _ in range(3): tr.step(*batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(12): tr.step(*batch)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 12 * 1e3)
print(min(out), sorted(out)[len(out) // 2], lo, hi)
''' % ROOT
settings = [("none", "none"), ("none", "1"), ("-1", "none"), ("-1", "1"), ("none", "-1")]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = {s: [] for s in settings}
for r in range(rounds):
    for s in settings[r % len(settings):] + settings[:r % len(settings)]:
        p = subprocess.run([sys.executable, "-c", CHILD, s[0], s[1]], capture_output=True, text=True)
        o = p.stdout.strip().splitlines()
        if not o:
            print("FAILED", s, p.stderr[-400:], flush=True); continue
        res[s].append(float(o[-1].split()[1])); rng = o[-1].split()[2:]
for s in settings:
    print(f"main {s[0]:>4s} side {s[1]:>4s}: medians {['%.3f' % x for x in sorted(res[s])]}", flush=True)
print("priority range reported by torch:", rng)
