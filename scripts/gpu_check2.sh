cd $GRAFT_REPO_ROOT
for t in 256 64; do echo "== HDSM_THREADS=$t"; HDSM_THREADS=$t timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2; HDSM_THREADS=$t python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms_mean'], d['p95_solve_latency_ms'], d['solver_stats_last_round'])"; done
HDSM_THREADS=256 bash scripts/gpu_scale.sh
