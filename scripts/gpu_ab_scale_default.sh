# the driver's per-N default workloads (64*N agents, default steps/warmup) replayed on ONE GPU with all instances
cd $GRAFT_REPO_ROOT
VAR=$1; shift
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['agents'], 'agents ms_step', round(d['ms_per_step'],4), 'p50', round(d['p50_solve_latency_ms'],4), 'p95', round(d['p95_solve_latency_ms'],4), 'failed', d['failed_instances_recorded'], d['solver_stats_last_round'])"; }
for V in "$@"; do
  echo "$VAR=$V"
  for n in 64 128 256 512; do env $VAR=$V timeout 900 python bench.py --no-cpu-baseline --agents $n 2>/dev/null | tail -1 | p; done
done
