# round 5 development loop on the GPU box: the -m gpu suite, then an A/B of one environment knob on cfg 5 / cfg 3 / the 4096 x H15 circle,
# then the driver's bench command.   usage: bash scripts/gpu_r5_check.sh [VAR "v1 v2"] [tests: 0/1] [bench: 0/1]
var=${1:-HDSM_CHILD_BOUND}; vals=${2:-"0 1"}; tests=${3:-1}; bench=${4:-1}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
if [ "$tests" = "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r5/gpu_tests.log
fi
run() { timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-event-pass --repeats 2 "${@:3}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], '%.4f ms' % d['ms_per_step'], 'limit', d['limit_instances_timed_rounds'], 'failed', d['failed_instances_timed_rounds'], 'nodes_max', d['solver_stats_timed_rounds']['nodes_max'])" "$1" "$2"; }
for rep in 1 2; do
for v in $vals; do
  export $var=$v
  case "${WL:-cfg5 cfg3 c4096h15}" in *cfg5*) run $var=$v cfg5 --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2;; esac
  case "${WL:-cfg5 cfg3 c4096h15}" in *cfg3*) run $var=$v cfg3 --scenario forest --agents 256 --first-round 60 --steps 8 --warmup 2;; esac
  case "${WL:-cfg5 cfg3 c4096h15}" in *c4096h15*) run $var=$v c4096h15 --agents 4096 --horizon 15 --first-round 20 --steps 8 --warmup 2;; esac
  unset $var
done
done
if [ "$bench" = "1" ]; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/bench.json 2> gpurun_out/r5/bench.err; echo "bench rc=$?"
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench.json').read().strip().splitlines()[-1])
print('value %.3e ms %.4f kernel %.4f frac %.3f' % (d['value'], d['ms_per_step'], d['kernel_ms_mean'], d['roofline']['frac']))
print('parity', d['parity_on_timed_rounds'] and {k: d['parity_on_timed_rounds'][k] for k in ('instances_compared','status_mismatches','max_abs_traj_diff')})
print('second', d.get('second_window') and {k: d['second_window'][k] for k in ('value','ms_per_step','failed_instances')}, 'solved/s %.3e' % d['solved_replans_per_s'])
print('single', d.get('single_instance_call') and {k: v for k, v in d['single_instance_call'].items() if k.endswith(('p50','p95'))})
print('host', d['host_buffer_path'] and (d['host_buffer_path']['ms_per_round'], d['host_buffer_path']['ms_per_round_registered_arrays']), 'dloop', d['device_resident_loop'] and d['device_resident_loop']['ms_per_round'])
for s in d['secondary_workloads'] or []:
    print('secondary', {k: s.get(k) for k in ('workload_key','mip_gap','ms_per_step','limit_instances','failed_instances','nodes_max','error')}, 'parity', s.get('parity_on_timed_rounds') and {k: s['parity_on_timed_rounds'][k] for k in ('instances_compared','status_mismatches','max_abs_traj_diff','limit_instances_checked','limit_incumbents_with_proven_optimum','limit_incumbent_below_optimum','seconds')})
PY
fi
