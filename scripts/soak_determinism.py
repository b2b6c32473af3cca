"""Two identical 40-step trainings at BASELINE configs[1] (bf16, side-stream weight gradients, dropout with a fixed generator)
must end in bit-identical parameters: a cross-stream race or an unordered accumulation would show up here."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
dev = "cuda:0"
B, H = 32, 256
batches = list(U.synthetic_batches(4, B, H, H, dev))


def run():
    eng = U.UNetEngine(H, H, B, F0=64, dtype="bf16", device=dev, overlap_wgrad=True)
    g = torch.Generator(); g.manual_seed(0); eng.reset_parameters(g)
    tr = U.Trainer(eng, lr=1e-4, dropout=True)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    losses = []
    for i in range(40):
        si, e, so = batches[i % 4]
        losses.append(tr.step(si, e, so, dropout_mask=eng.make_dropout_mask(gen), return_loss=(i % 10 == 9)))
    torch.cuda.synchronize()
    th = eng.theta.clone()
    del eng, tr
    torch.cuda.empty_cache()
    return th, [l for l in losses if l is not None]


t1, l1 = run()
t2, l2 = run()
print("losses", l1)
ok = torch.equal(t1, t2) and l1 == l2 and all(math.isfinite(x) for x in l1) and l1[-1] < l1[0]
print("soak determinism:", "OK" if ok else "MISMATCH", float((t1 - t2).abs().max()))
sys.exit(0 if ok else 1)
