# A/B of an environment toggle on the bench line and two scale points. usage: bash scripts/gpu_ab_env.sh VAR valA valB
cd $GRAFT_REPO_ROOT
VAR=$1; shift
echo "so_md5=$(md5sum multi_agent_pkgs_amd/libhdsm.so | cut -c1-12)"
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['agents'], 'H', d['config']['horizon'], 'ms_step', round(d['ms_per_step'],4), 'p50', round(d['p50_solve_latency_ms'],4), 'failed', d['failed_instances_recorded'], d['solver_stats_last_round'])"; }
for V in "$@"; do
  echo "$VAR=$V"
  env $VAR=$V timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | p
  env $VAR=$V timeout 600 python bench.py --no-cpu-baseline --agents 256 --steps 20 --warmup 5 2>/dev/null | tail -1 | p
  env $VAR=$V timeout 900 python bench.py --no-cpu-baseline --agents 1024 --steps 10 --warmup 2 --first-round 150 2>/dev/null | tail -1 | p
  env $VAR=$V timeout 900 python bench.py --no-cpu-baseline --agents 64 --horizon 15 --steps 20 --warmup 5 2>/dev/null | tail -1 | p
done
