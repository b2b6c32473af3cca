"""Race screen for the LDS-DMA kernels (counted-vmcnt pipelines): the same launch repeated many times, under memory load from a
second stream, must reproduce its first result bit for bit (a DMA read one barrier early shows up as rare differing tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_rir_amd as U
ops = U.ops
dev = "cuda:0"
torch.manual_seed(0)
side = torch.cuda.Stream()
junk_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
junk_b = torch.empty_like(junk_a)
N_IT = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def rnd(shape, scale=1.0):
    return ((torch.rand(shape, device=dev) - 0.5) * 2 * scale).to(torch.bfloat16)


def screen(name, launch, out):
    launch(); torch.cuda.synchronize()
    ref = out.clone()
    bad = 0
    for it in range(N_IT):
        with torch.cuda.stream(side):
            junk_b.copy_(junk_a, non_blocking=True)          # HBM / L2 pressure while the kernel runs
        launch()
        if it % 10 == 9:
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
    torch.cuda.synchronize()
    print(f"{name:40s} {'OK' if bad == 0 else f'{bad} MISMATCHING CHECKS'}")
    return bad


total = 0
for (B, HW, Ci, Co) in [(8, 128, 128, 128), (8, 64, 256, 256), (4, 256, 64, 64), (4, 256, 128, 64), (16, 32, 512, 512),
                       (32, 16, 1024, 1024)]:      # the last: paired-image tile of conv3x3g, two-quartet wgrad3x3g
    g = ops.geom(B, HW, HW, Ci, Co, 3, 1)
    x, y = ops.Act(rnd((B, HW, HW, Ci))), ops.Act(rnd((B, HW, HW, Co)))
    wh, wt = rnd((Co, 9, Ci), 0.1), rnd((Ci, 9, Co), 0.1)
    out = ops.Act(torch.empty((B, HW, HW, Co), dtype=torch.bfloat16, device=dev))
    dx = ops.Act(torch.empty((B, HW, HW, Ci), dtype=torch.bfloat16, device=dev))
    dw = torch.empty((Co, 9, Ci), device=dev)
    ws = ops.Workspace(dev)
    total += screen(f"conv fwd  {Ci}->{Co} @{HW}", lambda: ops.conv2d_fwd(g, x, wh, None, out), out.base)
    total += screen(f"conv dgrad {Ci}->{Co} @{HW}", lambda: ops.conv2d_dgrad(g, y, wt, dx), dx.base)
    total += screen(f"conv wgrad {Ci}->{Co} @{HW}", lambda: ops.conv2d_wgrad(g, x, y, dw, ws), dw)
for (B, H, Ci, Co) in [(8, 64, 128, 64), (8, 32, 256, 128)]:
    g = ops.geom(B, H, H, Ci, Co, 3, 2)
    x = ops.Act(rnd((B, H, H, Ci)))
    wt = rnd((Co, 9, Ci), 0.1)
    out = ops.Act(torch.empty((B, 2 * H, 2 * H, Co), dtype=torch.bfloat16, device=dev))
    total += screen(f"convT fwd {Ci}->{Co} @{H}->{2*H}", lambda: ops.conv2d_transpose_fwd(g, x, wt, None, out), out.base)
for (B, H, Ci, Co) in [(8, 128, 64, 128), (8, 64, 128, 256)]:          # stride-2 weight gradient: wgrad3x3d
    g = ops.geom(B, H, H, Ci, Co, 3, 2)
    x, y = ops.Act(rnd((B, H, H, Ci))), ops.Act(rnd((B, H // 2, H // 2, Co)))
    dw = torch.empty((Co, 9, Ci), device=dev)
    ws = ops.Workspace(dev)
    total += screen(f"conv s2 wgrad {Ci}->{Co} @{H}", lambda: ops.conv2d_wgrad(g, x, y, dw, ws), dw)
# the persistent kernels at their launched sizes (batch 32: >= 512 / 1024 tiles; tiles drawn at run time, counted waits across tile
# boundaries): conv3x3p forward / data gradient, conv3x3d (strided forward, plain kernel copy), upconv3x3q (transposed forward)
for (B, HW, Ci, Co) in [(32, 128, 128, 128), (32, 64, 256, 256)]:
    g = ops.geom(B, HW, HW, Ci, Co, 3, 1)
    x, y = ops.Act(rnd((B, HW, HW, Ci))), ops.Act(rnd((B, HW, HW, Co)))
    wh, wt = rnd((Co, 9, Ci), 0.1), rnd((Ci, 9, Co), 0.1)
    out = ops.Act(torch.empty((B, HW, HW, Co), dtype=torch.bfloat16, device=dev))
    dx = ops.Act(torch.empty((B, HW, HW, Ci), dtype=torch.bfloat16, device=dev))
    assert ops.conv3x3_kernel(g, 0, x) == "conv3x3p"
    total += screen(f"conv3x3p fwd  {Ci}->{Co} @{HW} B={B}", lambda: ops.conv2d_fwd(g, x, wh, None, out), out.base)
    total += screen(f"conv3x3p dgrad {Ci}->{Co} @{HW} B={B}", lambda: ops.conv2d_dgrad(g, y, wt, dx), dx.base)
# conv3x3s at its launched size: 1 024 strip segments drawn at run time, the column statistics leave the registers per segment
for (B, HW, CH) in [(32, 256, 64), (32, 128, 64), (32, 128, 32), (32, 256, 32)]:
    g = ops.geom(B, HW, HW, CH, CH, 3, 1)
    x, y = ops.Act(rnd((B, HW, HW, CH))), ops.Act(rnd((B, HW, HW, CH)))
    wh, wt = rnd((CH, 9, CH), 0.1), rnd((CH, 9, CH), 0.1)
    out = ops.Act(torch.empty((B, HW, HW, CH), dtype=torch.bfloat16, device=dev))
    dx = ops.Act(torch.empty((B, HW, HW, CH), dtype=torch.bfloat16, device=dev))
    assert ops.conv3x3_kernel(g, 0, x) == "conv3x3s"
    rows = ops.conv2d_colstat_rows(g, 0, x)
    cst = torch.zeros((rows, CH, 2), device=dev)
    both = torch.zeros(out.base.numel() + cst.numel(), device=dev)

    def fwd_stats():
        ops.conv2d_fwd_colstat(g, x, wh, None, out, cst)
        both[:out.base.numel()].copy_(out.base.flatten()); both[out.base.numel():].copy_(cst.flatten())
    total += screen(f"conv3x3s fwd + statistics {CH}->{CH} @{HW} B={B} ({rows} rows)", fwd_stats, both)
    total += screen(f"conv3x3s dgrad {CH}->{CH} @{HW} B={B}", lambda: ops.conv2d_dgrad(g, y, wt, dx), dx.base)
for (B, H, Ci, Co) in [(32, 256, 64, 128), (32, 128, 128, 256)]:
    g = ops.geom(B, H, H, Ci, Co, 3, 2)
    x = ops.Act(rnd((B, H, H, Ci)))
    wh, wt = rnd((Co, 9, Ci), 0.1), rnd((Ci, 9, Co), 0.1)
    out = ops.Act(torch.empty((B, H // 2, H // 2, Co), dtype=torch.bfloat16, device=dev))
    total += screen(f"conv3x3d fwd {Ci}->{Co} @{H}->{H//2} B={B}", lambda: ops.conv2d_fwd(g, x, wh, None, out), out.base)
    dx = ops.Act(torch.empty((B, H, H, Ci), dtype=torch.bfloat16, device=dev))
    total += screen(f"upconv3x3q dgrad {Co}->{Ci} @{H//2}->{H} B={B}", lambda: ops.conv2d_dgrad(g, out, wt, dx), dx.base)
print("race screen:", "clean" if total == 0 else f"{total} failures")
sys.exit(1 if total else 0)
