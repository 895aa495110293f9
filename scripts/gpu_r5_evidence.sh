# Evidence of round 5 (GPU box): the driver's bench command with rocprofv3 kernel trace + PMC (-> profiles/r05_summary.json,
# pmc_circle_*.json, incl. the MFMA counters of the pre-pass kernel), the cfg 5 roofline report (scripts/gpu_profile_cfg5.sh), BASELINE's
# other configurations and the round's A/B knobs as secondary bench lines, the launch timeline with its phases, the device-resident loop
# traced in free space and in the forest.    usage: bash scripts/gpu_r5_evidence.sh [tag]
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 1500 bash scripts/gpu_profile_bench.sh $TAG > gpurun_out/$TAG/profile.log 2>&1; tail -3 gpurun_out/$TAG/profile.log | cut -c1-200
timeout 900 bash scripts/gpu_profile_cfg5.sh $TAG > gpurun_out/$TAG/profile_cfg5.log 2>&1; tail -3 gpurun_out/$TAG/profile_cfg5.log | cut -c1-200
run() { t=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/$TAG/bench_$t.json 2> gpurun_out/$TAG/bench_$t.err; echo "$t rc=$?"; cut -c1-200 gpurun_out/$TAG/bench_$t.json; }
run cfg2_circle64 --agents 64 --first-round 35 --steps 50 --warmup 10
run cfg3_forest256 --scenario forest --agents 256 --first-round 60
HDSM_SPLIT=0 run cfg3_forest256_unsplit --scenario forest --agents 256 --first-round 60 --no-event-pass
HDSM_CHILD_BOUND=0 run cfg3_forest256_no_child_bound --scenario forest --agents 256 --first-round 60 --no-event-pass
run cfg5_fwf4096_h15 --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2
HDSM_SPLIT=0 run cfg5_fwf4096_h15_unsplit --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2 --no-event-pass
HDSM_CHILD_BOUND=0 run cfg5_fwf4096_h15_no_child_bound --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2 --no-event-pass
run cfg5_fwf4096_h15_mipgap1e-4 --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2 --no-event-pass --mip-gap 1e-4
run cfg5_fwf4096_h15_deep --scenario fwf --agents 4096 --horizon 15 --first-round 30 --steps 6 --warmup 2 --no-event-pass
run cfg5_fwf4096_h15_4000nodes --scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2 --no-event-pass --max-nodes 4000
HDSM_SETUP_MFMA=0 run circle1024_setup_map_per_instance --no-event-pass
HDSM_OVERLAP_SWEEP=0 run circle1024_sweep_after_warm_start --no-event-pass
run circle1024_cold_start --cold-start --no-event-pass
run circle1024_same_box --no-event-pass
run circle4096_h15 --agents 4096 --horizon 15 --first-round 20 --steps 8 --warmup 2
timeout 600 bash scripts/gpu_timeline_ab.sh HDSM_OVERLAP_SWEEP "1" > gpurun_out/$TAG/timeline.log 2>&1; tail -1 gpurun_out/$TAG/timeline.log | cut -c1-300
timeout 600 bash scripts/gpu_dloop_trace.sh ${TAG}_dloop_circle > gpurun_out/$TAG/dloop_circle.log 2>&1; tail -3 gpurun_out/$TAG/dloop_circle.log | cut -c1-300
timeout 600 bash scripts/gpu_dloop_trace.sh ${TAG}_dloop_forest --scenario forest --agents 256 --first-round 60 > gpurun_out/$TAG/dloop_forest.log 2>&1; tail -3 gpurun_out/$TAG/dloop_forest.log | cut -c1-300
timeout 300 python scripts/bench_corridor.py > gpurun_out/$TAG/f2_corridor.json 2> gpurun_out/$TAG/f2_corridor.err
# the summaries travel back under gpurun_out/ (64 MiB limit): raw traces are dropped once they are reduced
mkdir -p gpurun_out/$TAG/profiles; cp profiles/${TAG}_* profiles/pmc_*.json gpurun_out/$TAG/profiles/ 2>/dev/null
rm -rf gpurun_out/$TAG/trace gpurun_out/$TAG/pmc_*/ gpurun_out/${TAG}_cfg5/trace gpurun_out/${TAG}_cfg5/pmc_*/ gpurun_out/${TAG}_dloop_*/trace
