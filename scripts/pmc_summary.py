"""Sum a rocprofv3 --pmc counter_collection.csv per kernel (short names): launches, total counter value."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60]
    agg[(name, r["Counter_Name"])][0] += 1
    agg[(name, r["Counter_Name"])][1] += float(r["Counter_Value"])
for (name, c), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{name:60s} {c:12s} launches {n:5d} total {v:16.1f} per-launch {v/n:14.1f}")
