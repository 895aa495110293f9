# cfg 5 with larger node budgets: how many instances still end on the budget, and what the round costs.  usage: bash scripts/gpu_cfg5_nodes.sh "2000 4000 8000"
cd $GRAFT_REPO_ROOT
for n in ${1:-"2000 4000 8000"}; do
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-event-pass --repeats 2 --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2 --max-nodes $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('max_nodes', sys.argv[1], '%.4f ms' % d['ms_per_step'], 'limit', d['limit_instances_timed_rounds'], 'failed', d['failed_instances_timed_rounds'], 'nodes_max', d['solver_stats_timed_rounds']['nodes_max'])" $n
done
