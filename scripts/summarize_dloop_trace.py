#!/usr/bin/env python3
"""Per-kernel timeline of the device-resident loop (hdsm_dswarm_round) from a rocprofv3 --kernel-trace of bench.py: the last
ROUNDS rounds of the process are the device-loop pass; a round starts at its k_corridor dispatch. Prints, per kernel, the
mean duration, and the mean idle gap before it (end of the previous dispatch -> start), and the round period.

usage: python scripts/summarize_dloop_trace.py <kernel_trace.csv> [ROUNDS=20] [out.json]"""
import csv
import json
import re
import sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_corridor" in r["Kernel_Name"]]
starts = starts[-rounds:]
per = OrderedDict()
periods = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    seg = rows[a:b]
    if b != len(rows):
        periods.append(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]))
    prev_end = None
    for r in seg:
        name = re.sub(r"^(void )?", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).split("(")[0].split("<")[0][:40]
        d = per.setdefault(name, {"n": 0, "dur": 0, "gap": 0})
        d["n"] += 1
        d["dur"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if prev_end is not None:
            d["gap"] += int(r["Start_Timestamp"]) - prev_end
        prev_end = int(r["End_Timestamp"])
out = {"rounds": len(starts), "round_period_us_mean": sum(periods) / max(1, len(periods)) / 1e3,
       "kernels_per_round_us": {k: {"launches_per_round": v["n"] / len(starts), "dur_us": v["dur"] / len(starts) / 1e3,
                                    "gap_before_us": v["gap"] / len(starts) / 1e3} for k, v in per.items()}}
out["sum_dur_us"] = sum(v["dur_us"] for v in out["kernels_per_round_us"].values())
out["sum_gap_us"] = sum(v["gap_before_us"] for v in out["kernels_per_round_us"].values())
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
