"""Compact per-launch listing of a rocprofv3 kernel_trace.csv (last N launches)."""
import csv, sys, re
path = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:40]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:10.1f}us {d:9.1f}us grid={r.get('Grid_Size_X','?'):>9}x{r.get('Grid_Size_Y','?'):>5} wg={r.get('Workgroup_Size_X','?')} {name}")
