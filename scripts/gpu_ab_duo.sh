cd $GRAFT_REPO_ROOT
echo "so_md5=$(md5sum multi_agent_pkgs_amd/libhdsm.so | cut -c1-12)"
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['config']['agents'], 'H', d['config']['horizon'], 'ms_step', round(d['ms_per_step'],4), 'replans/s', round(d['value']), 'failed', d['failed_instances_recorded'], d['solver_stats_last_round'])"; }
for V in 0 257; do
  echo "HDSM_DUO_MIN=$V"
  HDSM_DUO_MIN=$V timeout 900 python bench.py --no-cpu-baseline --agents 512 2>/dev/null | tail -1 | p
  HDSM_DUO_MIN=$V timeout 900 python bench.py --no-cpu-baseline --agents 1024 --steps 10 --warmup 2 --first-round 30 2>/dev/null | tail -1 | p
  HDSM_DUO_MIN=$V timeout 900 python bench.py --no-cpu-baseline --agents 1024 --steps 10 --warmup 2 --first-round 150 2>/dev/null | tail -1 | p
  HDSM_DUO_MIN=$V timeout 900 python bench.py --no-cpu-baseline --agents 2048 --steps 10 --warmup 2 --first-round 30 2>/dev/null | tail -1 | p
done
HDSM_DUO_MIN=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
