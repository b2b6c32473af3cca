"""Per-dispatch effective clock (GRBM_GUI_ACTIVE / 8 / duration) and MFMA busy fraction for the long kernels."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(dict)
for r in rows:
    k = r["Dispatch_Id"]
    d[k]["name"] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:40]
    d[k]["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9
    d[k][r["Counter_Name"]] = float(r["Counter_Value"])
    d[k]["grid"] = r["Grid_Size"]
for k, v in d.items():
    if v["dur"] > 0.8e-3 and "igemm" in v["name"]:
        clk = v.get("GRBM_GUI_ACTIVE", 0) / 8 / v["dur"] / 1e9
        mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        # MFMA busy cycles summed over SIMDs(?) - normalise by 256 CU * 4 SIMD * cycles
        cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
        print(f"{v['name']:28s} grid {v['grid']:>9} dur {v['dur']*1e3:7.3f} ms clk {clk:5.2f} GHz  mfma_busy/cyc/1024 {mf/(cyc*1024+1e-9):6.3f} sqbusy {v.get('SQ_BUSY_CYCLES',0)/(cyc+1e-9):7.2f}")
