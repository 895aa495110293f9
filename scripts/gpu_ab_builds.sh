# bench line (recorded rounds, identical inputs) for several prebuilt libraries multi_agent_pkgs_amd/libhdsm_<name>.so
# usage: bash scripts/gpu_ab_builds.sh "name1 name2 ..." [repeats]
names=$1; reps=${2:-2}
cd $GRAFT_REPO_ROOT
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
rec=/tmp/ab_rec_builds.npz
[ -f $rec ] || timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-event-pass --save-recording $rec > /dev/null 2>&1
for r in $(seq $reps); do
for n in $names; do
  cp multi_agent_pkgs_amd/libhdsm_$n.so multi_agent_pkgs_amd/libhdsm.so
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --load-recording $rec 2>&1 | tail -1 > /tmp/ab_line.json
  python - "$n" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_line.json").read())
print("BUILD", sys.argv[1], "value %.4g" % d["value"], "ms_per_step %.4f" % d["ms_per_step"], ["%.4f" % x for x in d.get("ms_per_step_repeats", [])], "kernel_ms %.4f" % d["kernel_ms_mean"], "failed", d.get("failed_instances_timed_rounds"))
PY
done
done
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
