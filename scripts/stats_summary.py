"""Shorten a rocprofv3 kernel_stats.csv (kernel names cut at the argument list) for committing under profiles/."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:70]
    w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
