# Evidence of round 6 (GPU box), everything reduced into profiles/r06_* on the box and copied back under gpurun_out/r06/profiles:
#   1. the driver's bench command plain + rocprofv3 kernel trace + PMC (scripts/gpu_profile_bench.sh -> r06_summary.json, pmc_circle_*.json;
#      summarize_profile.py FAILS when the traced solver kernel uses scratch)
#   2. the cfg 5 roofline report (BASELINE configs[4]) and, new, the same report for cfg 3 (forest, 256 agents)
#   3. BASELINE's other configurations and the round's A/B knobs as bench lines (-> r06_workloads.json)
#   4. launch timeline with phases of the headline kernel; the device-resident loop traced in free space, forest and fwf
# usage: bash scripts/gpu_r6_evidence.sh [tag] [parts: e.g. "1 2 3 4"]
TAG=${1:-r06}; PARTS=${2:-"1 2 3 4"}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
C3="--scenario forest --agents 256 --first-round 60 --steps 8 --warmup 2"
C5="--scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2"
if has 1; then
  timeout 1500 bash scripts/gpu_profile_bench.sh $TAG > gpurun_out/$TAG/profile.log 2>&1; echo "profile_bench rc=$?"; tail -3 gpurun_out/$TAG/profile.log | cut -c1-200
fi
if has 2; then
  timeout 900 bash scripts/gpu_profile_cfg5.sh $TAG > gpurun_out/$TAG/profile_cfg5.log 2>&1; echo "profile_cfg5 rc=$?"; tail -2 gpurun_out/$TAG/profile_cfg5.log | cut -c1-200
  NAME=cfg3 timeout 900 bash scripts/gpu_profile_cfg5.sh $TAG $C3 > gpurun_out/$TAG/profile_cfg3.log 2>&1; echo "profile_cfg3 rc=$?"; tail -2 gpurun_out/$TAG/profile_cfg3.log | cut -c1-200
fi
run() { t=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/$TAG/bench_$t.json 2> gpurun_out/$TAG/bench_$t.err; echo "$t rc=$? $(python -c "import json,sys; d=json.loads(open('gpurun_out/$TAG/bench_$t.json').read().strip().splitlines()[-1]); print('%.4f ms' % d['ms_per_step'], 'limit', d['limit_instances_timed_rounds'], 'failed', d['failed_instances_timed_rounds'], 'nodes_max', d['solver_stats_timed_rounds']['nodes_max'])" 2>/dev/null)"; }
if has 3; then
  run cfg2_circle64 --agents 64 --first-round 35 --steps 50 --warmup 10
  run cfg3_forest256 $C3
  HDSM_DOMINANCE=0 run cfg3_forest256_no_dominance $C3 --no-event-pass
  HDSM_SPLIT=0 run cfg3_forest256_unsplit $C3 --no-event-pass
  run cfg3_forest256_rounds60_79 --scenario forest --agents 256 --first-round 60 --no-event-pass
  run cfg5_fwf4096_h15 $C5
  HDSM_DOMINANCE=0 run cfg5_fwf4096_h15_no_dominance $C5 --no-event-pass
  HDSM_DOMINANCE=0 HDSM_SPLIT_BUDGET=16 HDSM_ITEM_MIN=16 run cfg5_fwf4096_h15_round5_settings $C5 --no-event-pass
  HDSM_SPLIT_BUDGET=16 HDSM_ITEM_MIN=16 run cfg5_fwf4096_h15_budget16 $C5 --no-event-pass
  HDSM_SPLIT=0 run cfg5_fwf4096_h15_unsplit $C5 --no-event-pass
  run cfg5_fwf4096_h15_mipgap1e-4 $C5 --no-event-pass --mip-gap 1e-4
  run cfg5_fwf4096_h15_deep --scenario fwf --agents 4096 --horizon 15 --first-round 30 --steps 6 --warmup 2 --no-event-pass
  HDSM_SETUP_MFMA=0 run circle1024_setup_map_per_instance --no-event-pass
  run circle1024_same_box --no-event-pass
  HDSM_SETUP_MFMA=0 run circle1024_setup_map_per_instance_b --no-event-pass
  run circle1024_same_box_b --no-event-pass
  run circle1024_cold_start --cold-start --no-event-pass
  run circle4096_h15 --agents 4096 --horizon 15 --first-round 20 --steps 8 --warmup 2
  python scripts/collect_round_profiles.py $TAG 2>&1 | tail -1
fi
if has 4; then
  timeout 600 bash scripts/gpu_timeline_ab.sh HDSM_OVERLAP_SWEEP "1" > gpurun_out/$TAG/timeline.log 2>&1; tail -1 gpurun_out/$TAG/timeline.log | cut -c1-300
  tail -1 gpurun_out/$TAG/timeline.log > profiles/${TAG}_timeline_phases.json
  for sc in circle forest fwf; do
    case $sc in circle) A="";; forest) A="--scenario forest --agents 256 --first-round 60";; fwf) A="--scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2";; esac
    timeout 600 bash scripts/gpu_dloop_trace.sh ${TAG}_dloop_$sc $A > gpurun_out/$TAG/dloop_$sc.log 2>&1; tail -3 gpurun_out/$TAG/dloop_$sc.log | cut -c1-200
    cp gpurun_out/${TAG}_dloop_$sc/dloop_trace.json profiles/${TAG}_device_loop_trace_$sc.json 2>/dev/null
  done
fi
# the summaries travel back under gpurun_out/ (64 MiB limit): raw traces are dropped once they are reduced
mkdir -p gpurun_out/$TAG/profiles; cp profiles/${TAG}_* profiles/pmc_*.json gpurun_out/$TAG/profiles/ 2>/dev/null
rm -rf gpurun_out/$TAG/trace gpurun_out/$TAG/pmc_*/ gpurun_out/${TAG}_cfg5/trace gpurun_out/${TAG}_cfg5/pmc_*/ gpurun_out/${TAG}_cfg3/trace gpurun_out/${TAG}_cfg3/pmc_*/ gpurun_out/${TAG}_dloop_*/trace
