# per-kernel durations of a secondary workload (rocprofv3 --kernel-trace --stats of bench.py with the given arguments)
# usage: bash scripts/gpu_trace_workload.sh <tag> <bench args...>
TAG=$1; shift
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o w -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-event-pass --repeats 1 "$@" > $OUT/line.json 2> $OUT/err.log)
python - $OUT <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
f = glob.glob(out + "/trace/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
line = json.loads(open(out + "/line.json").read().strip().splitlines()[-1])
steps = line["steps"]
# the last `steps` solver rounds: walk back from the end over the launches of the timed region
names = [r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0] for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
tail = 40 * steps
agg = {}
last = [i for i, n in enumerate(names) if n.startswith("k_split_merge") or n.startswith("k_replan")]
# group per round: a round starts at k_plan_prepass
starts = [i for i, n in enumerate(names) if n.startswith("k_plan_prepass")][-steps:]
for si, s0 in enumerate(starts):
    s1 = starts[si + 1] if si + 1 < len(starts) else len(names)
    rec = {}
    for i in range(s0, s1):
        rec.setdefault(names[i] + " g" + rows[i]["Grid_Size_X"] + " wg" + rows[i]["Workgroup_Size_X"], []).append(round(dur[i], 1))
    print("round", si, {k: v for k, v in rec.items()})
print("ms_per_step", line["ms_per_step"])
PY
rm -rf $OUT/trace
