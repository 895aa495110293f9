# round 6: launch-order key in 5.12-us units for n > 30 (new) against 0.64-us units (libhdsm_oldkey.so, -DHDSM_OLD_KEY) on cfg 5, alternating
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-event-pass --repeats 3 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '%.4f ms' % d['ms_per_step'], ['%.3f' % x for x in d['ms_per_step_repeats']], 'limit', d['limit_instances_timed_rounds'], 'nodes_max', d['solver_stats_timed_rounds']['nodes_max'])" "$1"; }
C5="--scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2"
C5b="--scenario fwf --agents 4096 --horizon 15 --first-round 30 --steps 6 --warmup 2"
for rep in 1 2; do
  run "cfg5 new" $C5
  HDSM_LIBRARY=$PWD/multi_agent_pkgs_amd/libhdsm_oldkey.so run "cfg5 oldkey" $C5
done
run "cfg5 deep new" $C5b
HDSM_LIBRARY=$PWD/multi_agent_pkgs_amd/libhdsm_oldkey.so run "cfg5 deep oldkey" $C5b
run "c4096h15 new" --agents 4096 --horizon 15 --first-round 20 --steps 8 --warmup 2
HDSM_LIBRARY=$PWD/multi_agent_pkgs_amd/libhdsm_oldkey.so run "c4096h15 oldkey" --agents 4096 --horizon 15 --first-round 20 --steps 8 --warmup 2
