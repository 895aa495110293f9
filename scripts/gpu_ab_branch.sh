cd $GRAFT_REPO_ROOT
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'][:48], '| ms_step', round(d['ms_per_step'],4), 'p50', round(d['p50_solve_latency_ms'],4), 'p95', round(d['p95_solve_latency_ms'],4), d['solver_stats_last_round'])"; }
for V in 0 1; do
  echo "HDSM_BRANCH_RULE=$V"
  HDSM_BRANCH_RULE=$V timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | p
  HDSM_BRANCH_RULE=$V timeout 900 python bench.py --no-cpu-baseline --agents 128 --radius 22 2>/dev/null | tail -1 | p
  HDSM_BRANCH_RULE=$V timeout 900 python bench.py --no-cpu-baseline --agents 64 --horizon 15 2>/dev/null | tail -1 | p
  HDSM_BRANCH_RULE=$V timeout 900 python bench.py --no-cpu-baseline --agents 128 --horizon 15 --radius 22 2>/dev/null | tail -1 | p
done
HDSM_BRANCH_RULE=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
