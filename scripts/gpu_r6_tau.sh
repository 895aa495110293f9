# round 6: the staging radius (HDSM_CAND_TAU, default 0.6 m) on the bench line — does the crossing's three-sweep pattern come from a radius that overflows the 256 slots?
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-event-pass --repeats 3 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '%.4f ms' % d['ms_per_step'], ['%.4f' % x for x in d['ms_per_step_repeats']], 'value %.3f M' % (d['value']/1e6))" "$1"; }
for tau in 0.6 0.3 0.15 0.08 0.6; do HDSM_CAND_TAU=$tau run "headline tau=$tau" --steps 20 --warmup 5; done
