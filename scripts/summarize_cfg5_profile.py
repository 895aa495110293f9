#!/usr/bin/env python3
"""Reduce the rocprofv3 CSVs of scripts/gpu_profile_cfg5.sh to profiles/<tag>_cfg5_roofline.json (+ profiles/pmc_<workload key>.json).

A replan round of cfg 5 is a SPLIT launch: k_plan_prepass, k_replan (pass 1: every instance, with a node budget), k_replan again
(pass 2: one workgroup per queued item of the handed-over searches), k_split_merge — and now and then a rescue pass. The dispatches of
the process are cut into rounds at the pre-pass dispatches; the timed rounds are the LAST K rounds of the process (the profiling
command runs one repetition of warm-up + timed rounds and nothing after it). Per round: durations and counters of pass 1 and pass 2
separately and summed; the roofline block prices the algorithmic bytes of the round (SURVEY 8d formula x 4096 agents) on the summed
solver-kernel time."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
cfg_name = sys.argv[2] if len(sys.argv) > 2 else "cfg5"   # cfg5 (default) or cfg3: which secondary workload of the line was profiled
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "gpurun_out", tag + "_" + cfg_name)
line = json.loads(open(os.path.join(out_dir, "trace_bench.json")).read().strip().splitlines()[-1])
plain = json.loads(open(os.path.join(out_dir, "bench.json")).read().strip().splitlines()[-1])
K = line["steps"]
key = line["config"]["workload_key"]
HBM_PEAK = 8000.0


def rounds_of(path, value_of):
    """{counter or 'ns': [per timed round: {'pass1': v, 'pass2': v, 'other': v}]} from one CSV"""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = sorted({r.get("Counter_Name", "ns") for r in rows}) if "Counter_Name" in rows[0] else ["ns"]
    res = {}
    for name in names:
        rr = [r for r in rows if r.get("Counter_Name", "ns") == name]
        rounds, cur = [], None
        for r in rr:
            kn = r["Kernel_Name"]
            if "k_plan_prepass" in kn:
                cur = {"pass1": 0.0, "pass2": 0.0, "other": 0.0, "n_replan": 0, "grid": []}
                rounds.append(cur)
            elif cur is not None and "k_replan" in kn:
                cur["pass1" if cur["n_replan"] == 0 else ("pass2" if cur["n_replan"] == 1 else "other")] += value_of(r)
                cur["n_replan"] += 1
                cur["grid"].append(int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0))
        res[name] = rounds[-K:]
    return res, rows


summ = {"tag": tag, "workload_key": key, "workload": line["config"]["workload"], "kernel": "k_replan shapes of a split launch (pass 1: all instances; pass 2: the item queue) - kernel_trace.kernel_names",
        "kernel_source_sha16": line["roofline"]["kernel_source_sha16"], "timed_rounds": K,
        "bench_line_plain": {k: plain[k] for k in ("value", "ms_per_step", "limit_instances_timed_rounds", "failed_instances_timed_rounds")},
        "bench_line_of_the_traced_run": {k: line[k] for k in ("value", "ms_per_step")}}
tr = glob.glob(os.path.join(out_dir, "trace", "*kernel_trace.csv"))
t_round_ns = None
if tr:
    res, rows = rounds_of(tr[0], lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rr = res["ns"]
    p1 = [x["pass1"] for x in rr]
    p2 = [x["pass2"] for x in rr]
    ot = [x["other"] for x in rr]
    t_round_ns = (sum(p1) + sum(p2) + sum(ot)) / len(rr)
    first = next(r for r in rows if "k_replan" in r["Kernel_Name"])
    summ["kernel_trace"] = {"pass1_ms_mean": sum(p1) / len(p1) / 1e6, "pass2_ms_mean": sum(p2) / len(p2) / 1e6, "rescue_or_other_ms_mean": sum(ot) / len(ot) / 1e6,
                            "solver_kernels_ms_per_round": t_round_ns / 1e6, "pass1_ms": [x / 1e6 for x in p1], "pass2_ms": [x / 1e6 for x in p2],
                            "k_replan_dispatches_per_round": [x["n_replan"] for x in rr], "grids_last_round": rr[-1]["grid"],
                            "kernel_name": first["Kernel_Name"][:90], "kernel_names": sorted({r["Kernel_Name"].split("(")[0][-60:] for r in rows if "k_replan" in r["Kernel_Name"]}), "VGPR": first["VGPR_Count"], "AGPR": first["Accum_VGPR_Count"], "SGPR": first["SGPR_Count"],
                            "LDS": first["LDS_Block_Size"], "scratch": first["Scratch_Size"], "workgroup": first["Workgroup_Size_X"]}
pm = {}
for f in glob.glob(os.path.join(out_dir, "pmc_*", "*counter_collection.csv")):
    res, _ = rounds_of(f, lambda r: float(r["Counter_Value"]))
    for name, rr in res.items():
        if rr:
            pm[name] = {"per_round_mean": sum(x["pass1"] + x["pass2"] + x["other"] for x in rr) / len(rr),
                        "pass1_mean": sum(x["pass1"] for x in rr) / len(rr), "pass2_mean": sum(x["pass2"] for x in rr) / len(rr)}
summ["pmc"] = pm
v = {k: x["per_round_mean"] for k, x in pm.items()}
B = line["roofline"]["algorithmic_bytes_per_replan"] * line["config"]["agents"]
roof = {"bound": "hbm", "peak": HBM_PEAK, "unit": "GB/s", "algorithmic_bytes_per_round": B}
if t_round_ns:
    roof["achieved"] = B / (t_round_ns * 1e-9) / 1e9
    roof["frac"] = roof["achieved"] / HBM_PEAK
if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
    # rocprofv3 reports KiB; gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM section): an upper estimate
    roof["traffic_raw"] = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    roof["traffic"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    summ["hbm_bytes_per_launch"] = roof["traffic"]
    if t_round_ns:
        roof["traffic_GBps"] = roof["traffic"] / (t_round_ns * 1e-9) / 1e9
if "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"] > 0:
    roof["wait_any_over_wave_cycles"] = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
if "SQ_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0:
    roof["sq_busy_over_gui_active"] = v["SQ_BUSY_CYCLES"] / v["GRBM_GUI_ACTIVE"]
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VALU_MFMA_F64", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
    if k in v:
        roof[k + "_per_round"] = v[k]
if "SQ_INSTS_VALU" in v and t_round_ns:
    # one VALU wave-instruction occupies a 16-lane SIMD for 4 cycles: 1024 SIMDs x 2.4 GHz / 4
    roof["valu_issue_utilisation"] = v["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9 * t_round_ns * 1e-9)
summ["roofline"] = roof
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(summ, open(os.path.join(root, "profiles", f"{tag}_{cfg_name}_roofline.json"), "w"), indent=1)
if "hbm_bytes_per_launch" in summ:
    json.dump(summ, open(os.path.join(root, "profiles", f"pmc_{key}.json"), "w"), indent=1)
print(json.dumps(summ, indent=1)[:4000])
