#!/usr/bin/env python3
"""Reduce the rocprofv3 CSVs of scripts/gpu_profile.sh to profiles/<tag>_summary.json (+ pmc_latest.json)."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"tag": tag, "kernel": "k_replan"}
stats = os.path.join(root, "gpurun_out", f"{tag}_trace", "bench_kernel_stats.csv")
if os.path.exists(stats):
    for row in csv.DictReader(open(stats)):
        if "k_replan" in row["Name"]:
            out["kernel_stats"] = {k: row[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")}
counters = {}
for f in glob.glob(os.path.join(root, "gpurun_out", f"{tag}_pmc_*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if "k_replan" not in row.get("Kernel_Name", ""):
            continue
        name, val = row["Counter_Name"], float(row["Counter_Value"])
        counters.setdefault(name, []).append(val)
out["pmc_mean_per_launch"] = {k: sum(v) / len(v) for k, v in counters.items()}
out["pmc_launches"] = {k: len(v) for k, v in counters.items()}
pm = out["pmc_mean_per_launch"]
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    # rocprofv3 reports KiB; MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide
    # coalesced streams -> doubled as prescribed there (upper estimate here: this kernel's loads are 8-24 B per lane).
    out["hbm_bytes_per_launch_raw"] = (pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024
    out["hbm_bytes_per_launch"] = (2 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_summary.json"), "w"), indent=1)
json.dump(out, open(os.path.join(root, "profiles", "pmc_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
