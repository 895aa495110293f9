# A/B of one environment knob on a bench workload, on IDENTICAL inputs: the rounds are recorded once (default knobs) and replayed
# per setting.   usage: bash scripts/gpu_ab_env.sh VAR "v1 v2 ..." [bench args]      (development aid; runs on the GPU box)
var=$1; vals=$2; shift 2
rec=/tmp/ab_rec_$(echo "$@" | md5sum | cut -c1-10).npz
[ -f $rec ] || timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-event-pass "$@" --save-recording $rec > /dev/null 2>&1
for v in $vals; do
  env $var=$v timeout 600 python bench.py --no-cpu-baseline --no-secondary "$@" --load-recording $rec 2>&1 | tail -1 > /tmp/ab_line.json
  python - "$var" "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_line.json").read())
st = d.get("solver_stats_timed_rounds", {})
print(sys.argv[1], sys.argv[2], "value %.4g" % d["value"], "ms_per_step %.4f" % d["ms_per_step"], ["%.4f" % x for x in d.get("ms_per_step_repeats", [])],
      "failed", d.get("failed_instances_timed_rounds"), "iters mean %.1f max %d nodes max %d" % (st.get("qp_iters_mean", 0), st.get("qp_iters_max", 0), st.get("nodes_max", 0)))
PY
done
