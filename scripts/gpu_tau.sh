cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for tau in 1.5 0.5 0.25 0.1; do echo "tau=$tau"; HDSM_CAND_TAU=$tau python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms_mean'], d['value'], d['solver_stats_last_round'])"; done
