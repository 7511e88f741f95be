#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output for the fused kernel: per-dispatch sums and derived ratios."""
import collections, csv, glob, json, sys
out = {}
for d in sys.argv[1:]:
    cc = glob.glob(d + "/*/*counter_collection.csv")[0]
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kt))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cc)):
        if "fused" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    last = sorted(agg, key=int)[-1]
    c = dict(agg[last]); t = trace[last]
    c["duration_ns"] = int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    c["cycles_per_xcd"] = cyc
    c["eff_clock_ghz"] = cyc / c["duration_ns"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c: c["mfma_pipe_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
    if "SQ_LDS_IDX_ACTIVE" in c: c["lds_active_frac_per_cu"] = c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc
    if "SQ_LDS_BANK_CONFLICT" in c: c["lds_conflict_frac_per_cu"] = c["SQ_LDS_BANK_CONFLICT"] / 256 / cyc
    out[d] = c
print(json.dumps(out, indent=1))
