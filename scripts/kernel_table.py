"""Per-kernel ms/step table from a rocprofv3 *_kernel_stats.csv: python scripts/kernel_table.py stats.csv n_steps"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = 0.0
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    name = re.sub(r"^_Z\d+", "", name)[:58]
    ms = float(r["TotalDurationNs"]) / 1e6 / n
    tot += ms
    if ms >= 0.004:
        print(f"{name:58s} {int(r['Calls']):5d} {ms:8.3f} ms/step {float(r['AverageNs'])/1e3:9.1f} us avg")
print(f"{'total':58s}       {tot:8.3f} ms/step")
