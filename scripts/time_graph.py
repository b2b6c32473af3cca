"""The train step launched kernel by kernel against the same step replayed as a HIP graph (trainer.Trainer(graph=True)), per config:
python scripts/time_graph.py [resae] [refgeom] [cfg2] [--no-overlap] [--steps N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
import bench

dev = torch.device("cuda:0")
which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["resae", "refgeom", "cfg2"]
overlap = "--no-overlap" not in sys.argv
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 20


def build(kind):
    if kind == "resae":
        eng = U.ResAEEngine(256, 256, 32, (32, 64, 128, 256), (3, 3, 3, 3), (2, 2, 2, 2), 32, 1024, device=dev, dtype="bf16", overlap_wgrad=overlap)
        H = W = 256
    elif kind == "refgeom":
        H, W = 144, 160
        eng = U.UNetEngine(H, W, 32, F0=32, k=3, device=dev, dtype="bf16", overlap_wgrad=overlap)
    else:
        H = W = 256
        eng = U.UNetEngine(H, W, 32, F0=64, k=3, device=dev, dtype="bf16", overlap_wgrad=overlap)
    g = torch.Generator(); g.manual_seed(0)
    eng.reset_parameters(g)
    eng.dropout_seed = 1234
    return eng, bench.synthetic_batch(32, H, W, dev, 1234)


for kind in which:
    for graph in (False, True):
        eng, batch = build(kind)
        tr = U.Trainer(eng, lr=5e-7, graph=graph)
        for _ in range(3):
            tr.step(*batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(*batch)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        print(f"{kind} overlap={overlap} graph={graph}: {dt * 1e3:.3f} ms/step, {32 / dt:.0f} spectrograms/s (host {t_host / steps * 1e3:.3f} ms/step), loss {float(eng.loss_out[0]):.6f}",
              flush=True)
        del tr, eng
        torch.cuda.empty_cache()
