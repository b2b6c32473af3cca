"""Train step at the reference's own geometry (main_training.py: target_size (144, 160, 2), 9 x 10 bottleneck) beside the BASELINE
256 x 256 one: python scripts/time_refgeom.py [F0] [--only-ref] [--no-overlap]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
if "--lib" in sys.argv:
    U._lib.use_library(os.path.abspath(sys.argv[sys.argv.index("--lib") + 1]))
import bench
F0 = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
for (H, W) in (((144, 160),) if "--only-ref" in sys.argv else ((144, 160), (256, 256))):
    eng = U.UNetEngine(H, W, 32, F0=F0, k=3, device=dev, dtype="bf16", overlap_wgrad="--no-overlap" not in sys.argv)
    g = torch.Generator(); g.manual_seed(0)
    eng.reset_parameters(g)
    tr = U.Trainer(eng, lr=5e-7)
    gen = torch.Generator(); gen.manual_seed(1)
    spec_in = torch.rand((32, 2, H, W), generator=gen).to(dev); spec_out = torch.rand((32, 2, H, W), generator=gen).to(dev)
    emb = torch.randint(26, 1282, (32, 2, 16), generator=gen).to(dev)
    for _ in range(3): tr.step(spec_in, emb, spec_out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tr.step(spec_in, emb, spec_out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    fl = 306.9e9 * 32 * (H * W) / 65536.0 * (F0 / 64.0) ** 2
    print(f"F0={F0} {H}x{W} B=32 bf16: {dt*1e3:.2f} ms/step, {32/dt:.0f} spectrograms/s, ~{fl/dt/1e12:.0f} TFLOP/s", flush=True)
    del eng, tr; torch.cuda.empty_cache()
