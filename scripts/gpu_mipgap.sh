cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-secondary --no-event-pass --repeats 2 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '%.4f ms' % d['ms_per_step'], 'limit', d['limit_instances_timed_rounds'], 'failed', d['failed_instances_timed_rounds'], 'nodes_max', d['solver_stats_timed_rounds']['nodes_max'])" "$1"; }
run cfg5_exact --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2
run cfg5_gap1e-4 --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2 --mip-gap 1e-4
run cfg5_gap1e-4_tl80ms --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2 --mip-gap 1e-4 --time-limit-s 0.08
run cfg3_exact --scenario forest --agents 256 --first-round 60 --steps 8 --warmup 2
run cfg3_gap1e-4 --scenario forest --agents 256 --first-round 60 --steps 8 --warmup 2 --mip-gap 1e-4
