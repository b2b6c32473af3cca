"""Micro-benchmark of 3x3 stride-1 conv layers (bf16 or f32) through the C ABI, forward and data gradient:
python scripts/micro_conv.py C N HW [dtype] [B]   (several shapes: C,N,HW;C,N,HW...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
if "--lib" in sys.argv:      # another build of the library (A/B of two builds, one process each)
    U._lib.use_library(os.path.abspath(sys.argv[sys.argv.index("--lib") + 1]))
    del sys.argv[sys.argv.index("--lib"):sys.argv.index("--lib") + 2]
ops = U.ops
dev = "cuda:0"


def run(C, N, HW, dt, B=32, K=20):
    x = ops.Act((torch.rand((B, HW, HW, C), device=dev) - 0.5).to(dt))
    w = ((torch.rand((N, 9, C), device=dev) - 0.5) * 0.1).to(dt)
    wt = ((torch.rand((C, 9, N), device=dev) - 0.5) * 0.1).to(dt)
    bias = torch.zeros(N, device=dev)
    y = ops.Act(torch.empty((B, HW, HW, N), device=dev, dtype=dt))
    dx = ops.Act(torch.empty((B, HW, HW, C), device=dev, dtype=dt))
    g = ops.geom(B, HW, HW, C, N, 3, 1)
    out = []
    for name, fn in (("fwd", lambda: ops.conv2d_fwd(g, x, w, bias, y)), ("dgrad", lambda: ops.conv2d_dgrad(g, y, wt, dx))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        fl = 2.0 * B * HW * HW * C * N * 9
        by = B * HW * HW * (C + N) * x.base.element_size()
        out.append(f"{name} {ms*1e3:7.1f} us {fl/ms/1e9:6.0f} TF/s {by/ms/1e6:5.0f} GB/s")
    print(f"C={C:4d} N={N:4d} {HW}x{HW} B={B} {'bf16' if dt == torch.bfloat16 else 'f32'}: " + " | ".join(out), flush=True)


if __name__ == "__main__":
    for a_ in [v for v in sys.argv[1:] if v.startswith("--")]:       # --conv3x3p=0: kernel-selection switches
        k_, v_ = a_[2:].split("=")
        ops.set_config(**{k_: int(v_)})
    sys.argv = [v for v in sys.argv if not v.startswith("--")]
    if len(sys.argv) < 2:      # the > 64-channel 3x3 layers of BASELINE configs[1] (forward shape; the data gradient swaps C and N)
        sys.argv.append("128,128,128;256,128,128;256,256,64;512,256,64;512,512,32;128,64,256;64,64,256")
    if ";" in sys.argv[1] or "," in sys.argv[1]:
        dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
        for sh in sys.argv[1].split(";"):
            C, N, HW = (int(v) for v in sh.split(","))
            run(C, N, HW, dt)
    else:
        C, N, HW = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
        dt = torch.bfloat16 if (len(sys.argv) < 5 or sys.argv[4] == "bf16") else torch.float32
        run(C, N, HW, dt, int(sys.argv[5]) if len(sys.argv) > 5 else 32)
