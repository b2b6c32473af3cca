"""Per-kernel roofline table of the configs[1] train step from the committed profile summaries - a reader can recompute every number:

  python scripts/roofline_table.py profiles/r04_bf16_kernel_stats_ss.csv profiles/r04_bf16_pmc_fetch_size.txt \
         profiles/r04_bf16_pmc_write_size.txt profiles/r04_bf16_pmc_mfma_busy.txt > profiles/r04_roofline_table.json

Inputs (scripts/profile_round.sh): rocprofv3 --kernel-trace --stats of the single-stream step (the step count is the number of
adam_kernel launches), and the FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes (separate runs).  FETCH_SIZE is doubled
(MI355X_MICROARCH.md: gfx950 reports half of a wide coalesced read); both byte counters are in KB.

Algorithmic work comes from the layer table of BASELINE.json configs[1] (UNet 4 down / 4 up, F0 = 64, batch 32 of
[2,256,256]) and the dispatch rules of csrc/api.hip, restated in `serving_kernel` below; which kernel serves a 3x3 stride-1 layer is
asked of the library itself (unetrir_conv3x3_kernel_id_bf16 - no GPU needed).  Per kernel: launches per step, mean duration,
algorithmic GFLOP and bytes per launch (mean over the launches it serves), fraction of the dense bf16 MFMA peak (2.5 PFLOP/s), its
algorithmic bytes as a fraction of the HBM rate at 6.29 TB/s (measured copy rate) and 8.0 TB/s (spec), measured traffic / algorithmic
bytes, clock and MFMA-pipe-busy from the PMC pass."""
import csv, ctypes, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B, H0, F0, DEPTH = 32, 256, 64, 4
PEAK_TF, HBM_MEAS, HBM_SPEC = 2500.0, 6.29e12, 8.0e12


def layers():
    """(name, Cin, Cout, H of the layer INPUT, k, stride, transposed) of every convolution of the graph (dl_models/u_net.py:201-251)."""
    ch = [F0 * 2 ** l for l in range(DEPTH + 1)]
    hw = [H0 >> l for l in range(DEPTH + 1)]
    out = [("enc1.down", 8, ch[0], hw[0], 3, 1, False)]             # 2 input channels stored as 8
    for l in range(1, DEPTH + 2):
        if l > 1:
            out.append((f"enc{l}.down", ch[l - 2], ch[l - 1], hw[l - 2], 3, 2, False))
        out.append((f"enc{l}.cb1", ch[l - 1], ch[l - 1], hw[l - 1], 3, 1, False))
    for l in range(DEPTH, 0, -1):
        out.append((f"dec{l}.up", ch[l], ch[l - 1], hw[l], 3, 2, True))            # H = the low-resolution input
        out.append((f"dec{l}.cb1a", 2 * ch[l - 1], ch[l - 1], hw[l - 1], 3, 1, False))
        out.append((f"dec{l}.cb1b", ch[l - 1], ch[l - 1], hw[l - 1], 3, 1, False))
    return out


def lib():
    import unet_rir_amd
    return unet_rir_amd._lib.lib(), unet_rir_amd._lib.ConvGeom


K3 = ("tap-table", "conv3x3r", "conv3x3g", "conv3x3g pair", "conv3x3h", "conv3x3s", "conv3x3p", "stem")


def serving_kernel(L, ConvGeom, name, cin, cout, h, k, s, tr, what):
    """Kernel family (prefix of the rocprof name) that serves launch `what` in {fwd, dgrad, wgrad} of this layer: csrc/api.hip."""
    if what == "wgrad":
        return "wgrad3x3g_bf16_kernel" if s == 1 else "wgrad3x3d_bf16_kernel"
    if s == 1:
        g = ConvGeom(B, h, h, cin, cout, k, 1)
        ld_in = (2 * cout if False else (cout if what == "dgrad" else cin))
        kid = K3[L.unetrir_conv3x3_kernel_id_bf16(ctypes.byref(g), 1 if what == "dgrad" else 0, ld_in)]
        if kid == "conv3x3p":
            return "conv3x3p_bf16_kernel"
        if kid == "conv3x3s":
            return "conv3x3s_bf16_kernel"
        if kid == "stem":
            return "stem3x3_bf16_kernel"
        if kid == "conv3x3g pair":
            return "conv3x3g_bf16_kernel<2, true"
        if kid == "conv3x3g":
            n_out = cin if what == "dgrad" else cout
            return "conv3x3g_bf16_kernel<2, false, 64>" if n_out <= 64 else "conv3x3g_bf16_kernel<2, false, 128>"
        if kid == "conv3x3h":
            return "conv3x3h_bf16_kernel"
        if kid == "conv3x3r":
            return "conv3x3r_bf16_kernel"
        return "igemm_fwd_bf16_kernel<"
    # stride 2.  "Down" direction (strided Conv2D forward, Conv2DTranspose data gradient): conv3x3d where the OUTPUT grid is >= 32 wide.
    # "Up" direction (Conv2DTranspose forward, strided Conv2D data gradient): upconv3x3q from 1024 coarse tiles, upconv3x3g on grids
    # >= 32 wide, the four parity classes on the tap-table kernel below that.
    coarse = h if tr else h // 2                       # side of the low-resolution grid of the pair
    down = (what == "fwd") != tr
    if down:
        return "conv3x3d_bf16_kernel" if coarse >= 32 else "igemm_fwd_bf16_kernel<128"
    if coarse < 32:
        return "igemm_fwd_bf16_kernel4<128"
    n_out = cin if (what == "dgrad") else cout           # channels the launch writes
    tiles = B * (coarse // 8) * (coarse // 32) * max(1, n_out // 64)
    return "upconv3x3q_bf16_kernel" if tiles >= 1024 else "upconv3x3g_bf16_kernel"


def work(cin, cout, h, k, s, tr, what):
    """(GFLOP, algorithmic bytes) of one launch: 2 B H W Cin Cout k^2 on the grid the arithmetic runs on; bf16 activations in and
    out once, the kernel once (weight gradients: fp32 out)."""
    hi = h * s if tr else h                             # high-resolution side
    lo = hi // s
    pix = B * lo * lo
    flop = 2.0 * pix * cin * cout * k * k
    c_hi, c_lo = (cout, cin) if tr else (cin, cout)     # channels on the high- / low-resolution side
    act = 2.0 * (B * hi * hi * c_hi + B * lo * lo * c_lo)
    wbytes = cin * cout * k * k * (4.0 if what == "wgrad" else 2.0)
    return flop / 1e9, act + wbytes


def parse_pmc(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"(.{0,60}?)\s+" + counter + r"\s+launches\s+(\d+)\s+total\s+([\d.]+)\s+per-launch\s+([\d.]+)", line)
        if m:
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    return out


def parse_mfma(path):
    out = {}
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            out[m.group(1).strip()] = {"clk_ghz": float(m.group(4)), "mfma_busy": float(m.group(5))}
    return out


def find(d, name):
    for k, v in d.items():
        if k and (name.startswith(k[:56]) or k.startswith(name[:56])):
            return v
    return None


def main():
    stats, fetch_p, write_p, mfma_p = sys.argv[1:5]
    rows = list(csv.DictReader(open(stats)))
    steps = next(int(r["Calls"]) for r in rows if r["Name"].startswith("adam_kernel"))
    L, ConvGeom = lib()
    served = {}                                          # kernel prefix -> [(layer, what, gflop, bytes)]
    for name, cin, cout, h, k, s, tr in layers():
        for what in ("fwd", "dgrad", "wgrad"):
            if name == "enc1.down" and what == "dgrad":
                continue                                 # nothing upstream of the first layer
            kn = serving_kernel(L, ConvGeom, name, cin, cout, h, k, s, tr, what)
            gf, by = work(cin, cout, h, k, s, tr, what)
            served.setdefault(kn, []).append((name, what, gf, by))
    # BatchNormalization passes: 13 layers, bf16 tensors [B, h, h, c]: apply = 1 read + 1 write, backward reduce = 2 reads, backward apply = 2 reads + 1 write.
    # Tensors of 100 MB and more run the non-temporal-load instances (csrc/elementwise.hip bn_nt); rocprofv3 prints those instances with a
    # garbled demangling ("<bool _Accum, bool, E>": its demangler does not know the __bf16 mangling DF16b) and the plain ones mangled.
    bn = []
    for name, cin, cout, h, k, s, tr in layers():
        if ".cb1" in name:
            bn.append(2.0 * B * h * h * cout)
    big, small = [t for t in bn if t >= 100 << 20], [t for t in bn if t < 100 << 20]
    for kn_nt, kn, passes in (("bn_apply_kernel<bool _Accum", "_Z15bn_apply_kernelIDF16b", 2), ("chan_partial_kernel<2, bool _Accum", "_Z19chan_partial_kernelILi2EDF16b", 2),
                              ("bn_bwd_apply_kernel<bool _Accum", "_Z19bn_bwd_apply_kernelIDF16b", 3)):
        served[kn_nt] = [("BatchNorm layer >= 100 MB (non-temporal loads)", "pass", 0.0, passes * t) for t in big]
        served[kn] = [("BatchNorm layer < 100 MB", "pass", 0.0, passes * t) for t in small]
    served["adam_kernel"] = [("all parameters", "optimizer", 0.0, 28.0 * 68_613_058)]
    fetch, write, mf = parse_pmc(fetch_p, "FETCH_SIZE"), parse_pmc(write_p, "WRITE_SIZE"), parse_mfma(mfma_p)
    table, total_ms = {}, 0.0
    for r in rows:
        name = r["Name"]
        ms_step = float(r["TotalDurationNs"]) / 1e6 / steps
        total_ms += ms_step
        if ms_step < 0.02:
            continue
        ent = {"launches_per_step": int(r["Calls"]) / steps, "avg_us": float(r["AverageNs"]) / 1e3, "ms_per_step": ms_step}
        key = next((k for k in served if name.startswith(k)), None)
        if key is not None:
            ls = served[key]
            ent["serves"] = [f"{n} {w}" for n, w, _, _ in ls]
            ent["launches_in_table"] = len(ls)
            gf = sum(x[2] for x in ls) / len(ls)
            by = sum(x[3] for x in ls) / len(ls)
            ent["algorithmic_gflop_per_launch"], ent["algorithmic_bytes_per_launch"] = gf, by
            t = ent["avg_us"] * 1e-6
            if gf > 0:
                ent["tflops"] = gf / 1e3 / t
                ent["frac_of_mfma_peak"] = ent["tflops"] / PEAK_TF
            ent["hbm_time_frac_at_6.29TBps"] = by / HBM_MEAS / t
            ent["hbm_time_frac_at_8.0TBps"] = by / HBM_SPEC / t
        f, w = find(fetch, name), find(write, name)
        if f and w:
            ent["fetch_bytes_per_launch_x2"] = 2 * f[1] * 1024.0
            ent["write_bytes_per_launch"] = w[1] * 1024.0
            ent["traffic_bytes_per_launch"] = ent["fetch_bytes_per_launch_x2"] + ent["write_bytes_per_launch"]
            if key is not None:
                ent["traffic_over_algorithmic"] = ent["traffic_bytes_per_launch"] / ent["algorithmic_bytes_per_launch"]
        m = find(mf, name)
        if m:
            ent.update(m)
        table[name] = ent
    dominant = max((k for k in table if "frac_of_mfma_peak" in table[k]), key=lambda k: table[k]["ms_per_step"])
    print(json.dumps({
        "source": {"kernel_stats": os.path.basename(stats), "fetch": os.path.basename(fetch_p), "write": os.path.basename(write_p),
                   "mfma": os.path.basename(mfma_p), "steps": steps,
                   "command": "bench.py --lean --no-prof --no-cpu-baseline --dtype bf16 --no-overlap (scripts/profile_round.sh), 1x MI355X"},
        "peaks": {"mfma_bf16_dense_tflops": PEAK_TF, "hbm_measured_copy_Bps": HBM_MEAS, "hbm_spec_Bps": HBM_SPEC},
        "kernel_ms_per_step_total": total_ms, "dominant_kernel": dominant, "kernels": table}, indent=1))


if __name__ == "__main__":
    main()
