# candidate per-N default workloads of the multi-GPU bench (64 N agents, R = 22 N, window shifted with R), on ONE GPU
cd $GRAFT_REPO_ROOT
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'][:95], '| ms_step', round(d['ms_per_step'],4), 'p50', round(d['p50_solve_latency_ms'],4), 'p95', round(d['p95_solve_latency_ms'],4), 'failed', d['failed_instances_recorded'], d['solver_stats_last_round'])"; }
for N in 1 2 4 8; do timeout 1200 python bench.py --no-cpu-baseline --agents $((64*N)) --radius $((22*N)) --first-round $((25*N)) 2>/dev/null | tail -1 | p; done
