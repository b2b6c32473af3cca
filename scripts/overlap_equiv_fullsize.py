"""At BASELINE configs[1] (bf16): the side-stream schedule (weight gradients, vector branch, bucket-wise Adam) and the plain
single-stream schedule end in bit-identical parameters after 6 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
dev = "cuda:0"
B, H = 32, 256
batches = list(U.synthetic_batches(2, B, H, H, dev))
res = []
for ov in (False, True):
    eng = U.UNetEngine(H, H, B, F0=64, dtype="bf16", device=dev, overlap_wgrad=ov)
    g = torch.Generator(); g.manual_seed(0); eng.reset_parameters(g)
    tr = U.Trainer(eng, lr=1e-4, dropout=True)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    for i in range(6):
        si, e, so = batches[i % 2]
        tr.step(si, e, so, dropout_mask=eng.make_dropout_mask(gen))
    torch.cuda.synchronize()
    res.append((eng.theta.clone(), eng.adam_v.clone()))
    del eng, tr; torch.cuda.empty_cache()
ok = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
print("overlap == plain at full size:", "OK" if ok else "MISMATCH", float((res[0][0] - res[1][0]).abs().max()))
sys.exit(0 if ok else 1)
