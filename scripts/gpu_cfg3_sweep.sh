cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-secondary --no-event-pass --repeats 2 --scenario forest --agents 256 --first-round 60 --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '%.4f' % d['ms_per_step'], d['limit_instances_timed_rounds'], d['failed_instances_timed_rounds'], d['solver_stats_timed_rounds']['nodes_max'])" "$1"; }
run default
HDSM_SPLIT_DEPTH=2 run depth=2
for b in 4 16 32; do HDSM_SPLIT_BUDGET=$b run "budget=$b"; done
HDSM_SPLIT_DEPTH=2 HDSM_SPLIT_BUDGET=16 run "depth=2,budget=16"
