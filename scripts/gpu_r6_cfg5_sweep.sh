cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-event-pass --repeats 3 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '%.4f ms' % d['ms_per_step'], ['%.3f' % x for x in d['ms_per_step_repeats']], 'limit', d['limit_instances_timed_rounds'], 'nodes_max', d['solver_stats_timed_rounds']['nodes_max'])" "$1"; }
C5="--scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2"
C5b="--scenario fwf --agents 4096 --horizon 15 --first-round 30 --steps 6 --warmup 2"
for sb in 6 8 10 12; do for im in 8 16; do HDSM_SPLIT_BUDGET=$sb HDSM_ITEM_MIN=$im run "cfg5 SB=$sb IM=$im" $C5; done; done
run "cfg5 rounds 30.. default" $C5b
HDSM_SPLIT_BUDGET=8 run "cfg5 rounds 30.. SB=8" $C5b
HDSM_SPLIT_BUDGET=8 HDSM_ITEM_MIN=8 run "cfg5 rounds 30.. SB=8 IM=8" $C5b
