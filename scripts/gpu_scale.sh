cd $GRAFT_REPO_ROOT
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['agents'], 'H', d['config']['horizon'], 'replans/s', round(d['value']), 'kernel_ms', round(d['kernel_ms_mean'],4), 'p95', round(d['p95_solve_latency_ms'],4), 'failed', d['failed_instances_recorded'], d['solver_stats_last_round'], 'frac', round(d['roofline']['frac'],4))"; }
for n in 128 256; do timeout 600 python bench.py --no-cpu-baseline --agents $n --steps 20 --warmup 5 2>/dev/null | tail -1 | p; done
timeout 1500 python bench.py --no-cpu-baseline --agents 1024 --steps 10 --warmup 2 --first-round 30 2>/dev/null | tail -1 | p
timeout 1500 python bench.py --no-cpu-baseline --agents 1024 --steps 10 --warmup 2 --first-round 150 2>/dev/null | tail -1 | p
timeout 2400 python bench.py --no-cpu-baseline --agents 4096 --horizon 15 --steps 6 --warmup 2 --first-round 20 2>/dev/null | tail -1 | p
