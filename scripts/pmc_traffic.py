"""profiles/pmc_traffic.json from the two PMC summaries of scripts/profile_round.sh (FETCH_SIZE and WRITE_SIZE passes, separate runs):
python scripts/pmc_traffic.py <fetch_size.txt> <write_size.txt> <kernel name prefix> <round tag> > profiles/pmc_traffic.json
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half of a wide coalesced read); both counters are in KB."""
import json, re, sys


def per_launch(path, kernel, counter):
    for line in open(path):
        if line.startswith(kernel) and counter in line:
            m = re.search(r"launches\s+(\d+)\s+total\s+([\d.]+)\s+per-launch\s+([\d.]+)", line)
            return int(m.group(1)), float(m.group(3))
    raise SystemExit(f"{kernel} / {counter} not in {path}")


fetch, write, kernel, tag = sys.argv[1:5]
nl, f_kb = per_launch(fetch, kernel, "FETCH_SIZE")
_, w_kb = per_launch(write, kernel, "WRITE_SIZE")
print(json.dumps({
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/profile_round.sh {tag}) over `bench.py --lean --no-prof "
              f"--no-cpu-baseline --dtype bf16 --steps 2 --warmup 1 --no-overlap`, 1x MI355X: profiles/{tag}_bf16_pmc_fetch_size.txt, "
              f"_write_size.txt; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read)",
    "bf16": {"kernel": kernel, "launches": nl, "fetch_size_kb_per_launch": f_kb, "write_size_kb_per_launch": w_kb,
             "traffic_bytes_per_launch": (2 * f_kb + w_kb) * 1024.0},
}, indent=1))
