"""Per kernel: launches, total time, effective clock (GRBM_GUI_ACTIVE / 8 / duration), MFMA-busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs)) from a rocprofv3 --pmc counter_collection.csv."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(dict)
for r in rows:
    k = r["Dispatch_Id"]
    d[k]["name"] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:56]
    d[k]["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9
    d[k][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for v in d.values():
    a = agg[v["name"]]
    a[0] += 1; a[1] += v["dur"]; a[2] += v.get("GRBM_GUI_ACTIVE", 0.0); a[3] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
print(f"{'kernel':56s} {'calls':>5s} {'ms':>9s} {'clk GHz':>8s} {'mfma busy':>9s}")
for name, (n, dur, gui, mf) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    cyc = gui / 8
    print(f"{name:56s} {n:5d} {dur*1e3:9.3f} {cyc/dur/1e9 if dur else 0:8.2f} {mf/(cyc*1024) if cyc else 0:9.3f}")
