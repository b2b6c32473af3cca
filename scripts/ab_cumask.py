"""A/B of a CU mask on the weight-gradient stream: configs[1], side-stream schedule, engines alternating in one
process.  The side stream is created with hipExtStreamCreateWithCUMask (first n bits set: the driver spreads them over the 8 XCDs),
so that the HBM-bound BatchNorm passes of the main stream keep CUs the persistent weight-gradient workgroups cannot take.
python scripts/ab_cumask.py [rounds]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
import bench

dev = torch.device("cuda:0")
torch.cuda.init()
path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)      # the HIP runtime torch has loaded
hip = ctypes.CDLL(path)


def masked_stream(n_cus):
    words = (ctypes.c_uint32 * 8)()
    for i in range(n_cus):
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    err = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert err == 0, err
    return torch.cuda.ExternalStream(s.value, device=dev)


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
variants = [None, 224, 192, 128]
trainers = []
for v in variants:
    eng = U.UNetEngine(256, 256, 32, F0=64, k=3, device=dev, dtype="bf16", overlap_wgrad=True)
    if v is not None:
        eng.wg_stream = masked_stream(v)
    g = torch.Generator(); g.manual_seed(0)
    eng.reset_parameters(g)
    trainers.append(U.Trainer(eng, lr=5e-7))
batch = bench.synthetic_batch(32, 256, 256, dev, 1234)
res = [[] for _ in variants]
for r in range(rounds):
    for i, tr in enumerate(trainers):
        for _ in range(3):
            tr.step(*batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(12):
            tr.step(*batch)
        torch.cuda.synchronize()
        res[i].append((time.perf_counter() - t0) / 12 * 1e3)
for v, t in zip(variants, res):
    t = sorted(t)
    print(f"weight-gradient stream on {'all' if v is None else v} CUs: median {t[len(t)//2]:.3f} min {t[0]:.3f} ms/step", flush=True)
