cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-secondary --no-event-pass --repeats 2 --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['ms_per_step'], d['limit_instances_timed_rounds'], d['failed_instances_timed_rounds'], d['solver_stats_timed_rounds']['nodes_max'])" "$1"; }
for b in 16 32 64 96 160; do HDSM_SPLIT_BUDGET=$b run "budget=$b"; done
HDSM_SPLIT_DEPTH=2 run "depth=2"
HDSM_DUO_MIN=100000 run "one-per-CU pass1"
