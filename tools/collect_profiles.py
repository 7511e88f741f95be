#!/usr/bin/env python3
"""Copy the newest rocprofv3 outputs under gpurun_out/ (tools/round_profile.sh) into profiles/ with a tag."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def newest(pat):
    fs = glob.glob(os.path.join(root, pat)); fs.sort(key=os.path.getmtime); return fs[-1]
shutil.copy(newest("gpurun_out/profS/*/*kernel_stats.csv"), os.path.join(root, f"profiles/{tag}_kernel_stats.csv"))
cm = os.path.join(root, "gpurun_out/cert_margins.json")
if os.path.exists(cm): shutil.copy(cm, os.path.join(root, f"profiles/{tag}_cert_margins.json"))
out = {}
for d in ("pmcA", "pmcB", "pmcF", "pmcW"):
    cc = newest(f"gpurun_out/{d}/*/*counter_collection.csv"); kt = cc.replace("counter_collection", "kernel_trace")
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kt))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); name = None
    for r in csv.DictReader(open(cc)):
        if "fused" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); name = r["Kernel_Name"]
    last = sorted(agg, key=int)[-1]; c = dict(agg[last]); t = trace[last]
    c["duration_ns"] = int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8; c["cycles_per_xcd"] = cyc; c["eff_clock_ghz"] = cyc / c["duration_ns"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c: c["mfma_pipe_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
        if "SQ_LDS_IDX_ACTIVE" in c: c["lds_active_frac_per_cu"] = c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc
    out[d] = c
fetch = out["pmcF"]["FETCH_SIZE"] * 1024; write = out["pmcW"]["WRITE_SIZE"] * 1024
out["_notes"] = {"kernel": name, "workload": "bench.py default: 1Mx768 f32 cosine batch 256 k=10 (one launch)",
    "fetch_raw_bytes": fetch, "write_bytes": write,
    "correction": "gfx950 FETCH_SIZE counts 64 B per 128-B request for 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM section); calibrated in round 1 on the 384 MB device-to-device copy of bench.py's index build (WRITE_SIZE 366.2 MiB exact, FETCH_SIZE 183.1 MiB = 1/2)",
    "hbm_bytes_per_launch": 2 * fetch + write, "algorithmic_bytes": 3072786432}
json.dump(out, open(os.path.join(root, f"profiles/{tag}_pmc_fused.json"), "w"), indent=1)
tpath = os.path.join(root, "profiles/traffic.json")
tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
screened = "bf16" in (name or "")
key = "hbm_bytes_per_launch_bf16_screen" if screened else "hbm_bytes_per_launch"
tj[key] = 2 * fetch + write
tj[key + "_detail"] = {"fetch_size_raw_bytes": fetch, "write_size_bytes": write, "kernel": name,
    "tag": tag,
    "source": f"profiles/{tag}_pmc_fused.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of bench.py --steps 3; FETCH doubled per the gfx950 correction)"}
tj["workload"] = "bench.py default (1Mx768 f32, batch 256, k=10, 1 GPU)"
json.dump(tj, open(tpath, "w"), indent=1)
bench = open(os.path.join(root, "gpurun_out/bench_final.json")).read().strip().splitlines()[-1]
open(os.path.join(root, f"profiles/{tag}_bench_line.json"), "w").write(bench + "\n")
# ---- config 3's shape (tools/round_profile.sh): kernel stats, bench line, FETCH / WRITE of the wide filter kernel per launch
try:
    shutil.copy(newest("gpurun_out/profS3/*/*kernel_stats.csv"), os.path.join(root, f"profiles/{tag}_c3_kernel_stats.csv"))
    for cfg in ("c3", "c1", "c4"):
        f = os.path.join(root, f"gpurun_out/bench_{cfg}.json")
        if os.path.exists(f) and open(f).read().strip():
            open(os.path.join(root, f"profiles/{tag}_{cfg}_bench_line.json"), "w").write(open(f).read().strip().splitlines()[-1] + "\n")
    c3 = {}
    for d in ("pmcF3", "pmcW3"):
        cc = newest(f"gpurun_out/{d}/*/*counter_collection.csv")
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); kname = None
        for r in csv.DictReader(open(cc)):
            if "fused_bf16" in r["Kernel_Name"]:
                agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); kname = r["Kernel_Name"]
        last = sorted(agg, key=int)[-1]
        c3[d] = dict(agg[last]); c3["kernel"] = kname
    f3 = c3["pmcF3"]["FETCH_SIZE"] * 1024; w3 = c3["pmcW3"]["WRITE_SIZE"] * 1024
    rows_bytes = 1250000 * 768 * 4
    c3["_notes"] = {"workload": "bench.py --config c3: one 1.25M x 768 shard, dot, batch 1024 (two launches of 512 queries), k = 100",
                    "fetch_raw_bytes_per_launch": f3, "write_bytes_per_launch": w3, "hbm_bytes_per_launch": 2 * f3 + w3,
                    "rows_bytes": rows_bytes, "launches_per_batch": 2,
                    "hbm_bytes_per_batch_filter_passes": 2 * (2 * f3 + w3), "algorithmic_bytes_per_batch": rows_bytes + 1024 * 768 * 4,
                    "rows_read_per_batch": 2 * (2 * f3 + w3) / rows_bytes,
                    "correction": "FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM section)"}
    json.dump(c3, open(os.path.join(root, f"profiles/{tag}_pmc_c3.json"), "w"), indent=1)
    print("c3:", json.dumps(c3["_notes"]))
except Exception as e:
    print("c3 profile not collected:", e)
print(json.dumps({k: v for k, v in out.items() if k != "_notes"}, indent=1)[:1800]); print(out["_notes"]["hbm_bytes_per_launch"])
