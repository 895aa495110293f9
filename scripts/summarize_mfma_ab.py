#!/usr/bin/env python3
"""gpurun_out/mfma_ab/ (scripts/gpu_mfma_ab.sh) -> gpurun_out/mfma_ab/summary.json (copy to profiles/r02_mfma_ab.json)."""
import csv
import glob
import json
import os

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", "mfma_ab")
res = {"what": "MFMA leaf test (v_mfma_f64_16x16x4_f64, opt-in build -DHDSM_LEAF_MFMA) vs the default build on the bench line "
               "(1024 agents, rounds 165..184, one MI355X, same box, back to back). Counters: mean per timed launch of k_replan."}
for tag in ("default", "mfma"):
    r = {"bench": []}
    for f in sorted(glob.glob(os.path.join(out, f"{tag}_bench?.json"))):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r["bench"].append({k: d[k] for k in ("value", "ms_per_step", "kernel_ms_mean", "p95_solve_latency_ms")})
    for grp in ("mfma", "WRITE_SIZE", "FETCH_SIZE"):
        jf = os.path.join(out, f"{tag}_pmc_{grp}.json")
        if not os.path.exists(jf):
            continue
        line = json.loads(open(jf).read().strip().splitlines()[-1])
        seq = line["k_replan_launch_sequence"]
        lo = seq["setup_flight"] + seq["warmup"]
        hi = lo + seq["timed"]
        for f in glob.glob(os.path.join(out, f"{tag}_pmc_{grp}", "*counter_collection.csv")):
            per = {}
            for row in csv.DictReader(open(f)):
                if "k_replan" in row["Kernel_Name"]:
                    per.setdefault(row["Counter_Name"], []).append((int(row["Start_Timestamp"]), float(row["Counter_Value"])))
                    r["scratch_bytes_per_lane"] = row["Scratch_Size"]
            for name, v in per.items():
                v.sort()
                vals = [x[1] for x in v[lo:hi]]
                r[name] = sum(vals) / max(1, len(vals))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in r and r.get("SQ_BUSY_CYCLES"):
        r["mfma_busy_over_sq_busy"] = r["SQ_VALU_MFMA_BUSY_CYCLES"] / r["SQ_BUSY_CYCLES"]
    res[tag] = r
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:2500])
