#!/usr/bin/env python3
"""Copy the rocprofv3 outputs of tools/shadow_profile.sh from gpurun_out/ into profiles/<tag>_shadow_*."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def newest(pat):
    fs = glob.glob(os.path.join(root, pat)); fs.sort(key=os.path.getmtime); return fs[-1]
shutil.copy(newest("gpurun_out/shS/*/*kernel_stats.csv"), os.path.join(root, f"profiles/{tag}_shadow_kernel_stats.csv"))
out = {}
for d in ("shA", "shF", "shW"):
    cc = newest(f"gpurun_out/{d}/*/*counter_collection.csv"); kt = cc.replace("counter_collection", "kernel_trace")
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kt))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); name = None
    for r in csv.DictReader(open(cc)):
        if "fused_a16" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); name = r["Kernel_Name"]
    last = sorted(agg, key=int)[-1]; c = dict(agg[last]); t = trace[last]
    c["duration_ns"] = int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8; c["cycles_per_xcd"] = cyc; c["eff_clock_ghz"] = cyc / c["duration_ns"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c: c["mfma_pipe_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
        if "SQ_LDS_IDX_ACTIVE" in c: c["lds_active_frac_per_cu"] = c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc
    out[d] = c
fetch = out["shF"]["FETCH_SIZE"] * 1024; write = out["shW"]["WRITE_SIZE"] * 1024
out["_notes"] = {"kernel": name, "workload": "tools/kernel_time.py, VDB_SHADOW=1: 1Mx768 cosine batch 256 k=10 (one launch)",
    "fetch_raw_bytes": fetch, "write_bytes": write,
    "correction": "FETCH_SIZE doubled per the gfx950 correction (see profiles/README.md)",
    "hbm_bytes_per_launch": 2 * fetch + write, "algorithmic_bytes": 2 * 1000000 * 768 + 2 * 256 * 768}
json.dump(out, open(os.path.join(root, f"profiles/{tag}_shadow_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
