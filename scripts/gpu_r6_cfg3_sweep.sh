# round 6: the split knobs on cfg 3 (and cfg 5) after the dominance rule made the trees small.  usage: bash scripts/gpu_r6_cfg3_sweep.sh
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-event-pass --repeats 3 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '%.4f ms' % d['ms_per_step'], ['%.3f' % x for x in d['ms_per_step_repeats']], 'limit', d['limit_instances_timed_rounds'], 'nodes_max', d['solver_stats_timed_rounds']['nodes_max'])" "$1"; }
C3="--scenario forest --agents 256 --first-round 60 --steps 8 --warmup 2"
C5="--scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2"
run "cfg3 default" $C3
HDSM_SPLIT=0 run "cfg3 unsplit" $C3
for im in 1 4 8 16; do HDSM_ITEM_MIN=$im run "cfg3 ITEM_MIN=$im" $C3; done
for sb in 1 4 8; do HDSM_SPLIT_BUDGET=$sb run "cfg3 SPLIT_BUDGET=$sb" $C3; done
HDSM_SPLIT_BUDGET=4 HDSM_ITEM_MIN=8 run "cfg3 SB4 IM8" $C3
HDSM_POLL_SLEEP=1 run "cfg3 POLL_SLEEP=1" $C3
run "cfg5 default" $C5
for im in 4 8 32; do HDSM_ITEM_MIN=$im run "cfg5 ITEM_MIN=$im" $C5; done
for sb in 4 8 32; do HDSM_SPLIT_BUDGET=$sb run "cfg5 SPLIT_BUDGET=$sb" $C5; done
HDSM_SPLIT=0 run "cfg5 unsplit" $C5
