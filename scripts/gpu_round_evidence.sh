# Evidence of the round (GPU box): (1) the driver's bench command with rocprofv3 kernel trace + PMC -> gpurun_out/<tag>/,
# (2) BASELINE's other configurations as secondary bench lines, (3) phase cycle counters of the profile build,
# (4) row f2 / f1 / f4 micro-benchmarks. Summaries are written under gpurun_out/<tag>/ and copied into profiles/ by hand.
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
bash scripts/gpu_profile_bench.sh $TAG > gpurun_out/$TAG/profile.log 2>&1; tail -3 gpurun_out/$TAG/profile.log | cut -c1-200
run() { t=$1; shift; timeout 900 python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/$TAG/bench_$t.json 2> gpurun_out/$TAG/bench_$t.err; echo "$t rc=$?"; cut -c1-240 gpurun_out/$TAG/bench_$t.json; }
run cfg2_circle64 --agents 64 --first-round 35 --steps 50 --warmup 10
run cfg3_forest256 --scenario forest --agents 256 --first-round 60
HDSM_SPLIT=0 run cfg3_forest256_unsplit --scenario forest --agents 256 --first-round 60 --no-event-pass
HDSM_SPLIT=0 run cfg5_fwf4096_h15_unsplit --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2 --no-event-pass
run cfg5_fwf4096_h15 --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2
run cfg5_fwf4096_h15_deep --scenario fwf --agents 4096 --horizon 15 --first-round 30 --steps 6 --warmup 2 --no-event-pass
HDSM_SPLIT_DEPTH=1 run cfg5_fwf4096_h15_deep_depth1 --scenario fwf --agents 4096 --horizon 15 --first-round 30 --steps 6 --warmup 2 --no-event-pass
HDSM_SPLIT_DEPTH=1 run cfg3_forest256_depth1 --scenario forest --agents 256 --first-round 60 --no-event-pass
HDSM_PICK_RULE=0 run circle1024_raw_pick_rule --no-event-pass
HDSM_PICK_RULE=0 run cfg3_forest256_raw_pick_rule --scenario forest --agents 256 --first-round 60 --no-event-pass
run circle1024_cold_start --cold-start --no-event-pass
HDSM_SCANNER=0 run circle1024_no_scanner_wave --no-event-pass
HDSM_QUAD_MIN=1000000 run circle1024_three_per_cu_kernel --no-event-pass
run circle1024_same_box --no-event-pass
run circle4096_h15 --agents 4096 --horizon 15 --first-round 20 --steps 8 --warmup 2
BENCH_ARGS="" bash scripts/gpu_prof_bench.sh > gpurun_out/$TAG/phases.log 2>&1; cp gpurun_out/prof_bench.log gpurun_out/$TAG/prof_bench.log; tail -3 gpurun_out/$TAG/phases.log | cut -c1-400
timeout 300 python scripts/bench_corridor.py > gpurun_out/$TAG/f2_corridor.json 2> gpurun_out/$TAG/f2_corridor.err; cat gpurun_out/$TAG/f2_corridor.json
# the summaries travel back under gpurun_out/ (64 MiB limit): raw traces are dropped once they are reduced
mkdir -p gpurun_out/$TAG/profiles; cp profiles/${TAG}_* profiles/pmc_*.json gpurun_out/$TAG/profiles/ 2>/dev/null
rm -rf gpurun_out/$TAG/trace gpurun_out/$TAG/pmc_*/
