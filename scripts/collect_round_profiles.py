#!/usr/bin/env python3
"""Reduce gpurun_out/<tag>/ (scripts/gpu_round_evidence.sh) to profiles/<tag>_workloads.json (the secondary bench lines) and
profiles/<tag>_phases.json (cycle counters of the -DHDSM_PROFILE build on the timed rounds). usage: collect_round_profiles.py r03"""
import glob
import json
import os
import re
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
KEEP = ("value", "ms_per_step", "ms_per_step_repeats", "config", "solver_stats_timed_rounds", "failed_instances_timed_rounds", "limit_instances_timed_rounds",
        "device_resident_loop", "kernel_ms_mean", "p50_solve_latency_ms", "p95_solve_latency_ms", "host_buffer_path")
lines = {}
for path in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    txt = open(path).read().strip().splitlines()
    if not txt or not txt[-1].startswith("{"):
        continue
    d = json.loads(txt[-1])
    lines[os.path.basename(path)[len("bench_"):-len(".json")]] = {k: d[k] for k in KEEP if k in d}
json.dump({"what": f"secondary bench lines of the round (scripts/gpu_round_evidence.sh {tag}): BASELINE configs 2, 3, 5 and a 4096-agent circle at "
                   "H = 15 on one MI355X, plus the A/B lines of the round's knobs: '_unsplit' = HDSM_SPLIT=0 (one-kernel launches), '_depth1' = "
                   "HDSM_SPLIT_DEPTH=1, '_raw_pick_rule' = HDSM_PICK_RULE=0, '_cold_start' = hdsm_params.warm_start = 0, '_no_dominance' = HDSM_DOMINANCE=0 "
                   "(round 6), '_round5_settings' = HDSM_DOMINANCE=0 with the split budget / item minimum of round 5 (16 / 16). Forest flights do not "
                   "repeat bit for bit from run to run (DESIGN section 4); A/Bs on identical recorded rounds: scripts/gpu_ab_env.sh",
           "lines": lines}, open(os.path.join(root, "profiles", f"{tag}_workloads.json"), "w"), indent=1)


def parse(line):
    return {k: float(v) for k, v in re.findall(r"(\w+)=(-?\d+)", line)}


if not os.path.exists(os.path.join(src, "prof_bench.log")):   # (round 6: no -DHDSM_PROFILE pass in the evidence run)
    print("wrote", len(lines), "bench lines (no phase counters in this run)")
    sys.exit(0)
log = open(os.path.join(src, "prof_bench.log")).read().splitlines()
worst = [ln for ln in log if ln.startswith("HDSM_PROFILE worst")]
mean = [ln for ln in log if ln.startswith("HDSM_PROFILE mean")]
K = 20  # the timed rounds of the last repetition of the bench line
rounds = [{"slowest_instance": parse(w), "mean_instance": parse(m)} for w, m in zip(worst[-K:], mean[-K:])]
json.dump({"what": "cycle counters of the -DHDSM_PROFILE build (scripts/gpu_prof_bench.sh) on the 20 timed rounds of the bench line (last "
                   "repetition), final binary of the round: per round the slowest instance and the mean over the instances; every counter "
                   "read costs a few dozen cycles and orders the memory operations around it. states / select / normal / d / sums / upd / add / "
                   "drop: phases of a regular active-set operation (summed over the instance); setup, sweep, leaf: per instance; w_*: inside "
                   "the warm start; sw_*: inside the sweeps; su_*: inside the set-up",
           "rounds": rounds}, open(os.path.join(root, "profiles", f"{tag}_phases.json"), "w"), indent=1)
print("wrote", len(lines), "bench lines and", len(rounds), "profiled rounds")
