cd $GRAFT_REPO_ROOT
for v in $VALUES; do echo -n "$VAR=$v: "; env $VAR=$v python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['kernel_ms_mean'],4), round(d['p95_solve_latency_ms'],4), d['solver_stats_last_round'])"; done
