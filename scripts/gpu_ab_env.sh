# A/B of one environment knob on a bench workload: usage  bash scripts/gpu_ab_env.sh VAR "v1 v2 ..." [bench args]
# prints value / ms_per_step / repeats per setting (development aid; runs on the GPU box)
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v timeout 600 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 > /tmp/ab_line.json
  python - "$var" "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_line.json").read())
print(sys.argv[1], sys.argv[2], "value %.4g" % d["value"], "ms_per_step %.4f" % d["ms_per_step"], d.get("ms_per_step_repeats"),
      "failed", d.get("failed_instances_timed_rounds"), "kernel_ms_mean", d.get("kernel_ms_mean"))
PY
done
