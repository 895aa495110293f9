# usage (GPU box): bash scripts/gpu_profile_cfg5.sh <tag> [scenario args]   — the rocprof report BASELINE configs[4] asks for:
# kernel trace + PMC passes (one counter group per pass, --kernel-trace only) of the cfg 5 window (4096 agents, forest + wall +
# forest, H = 15, rounds 8..13), reduced by scripts/summarize_cfg5_profile.py to profiles/<tag>_cfg5_roofline.json and
# profiles/pmc_<workload key>.json. A split launch is several kernels (pre-pass, pass 1, pass 2 = the items, merge): the rounds are
# cut at the pre-pass dispatches, the timed rounds are the last K of the process (--repeats 1, no event pass, no parity pass).
TAG=${1:-r05}; shift
ARGS=${@:-"--scenario fwf --agents 4096 --horizon 15 --first-round 8 --steps 6 --warmup 2"}
cd $GRAFT_REPO_ROOT
NAME=${NAME:-cfg5}   # (round 6: NAME=cfg3 with the forest arguments gives profiles/<tag>_cfg3_roofline.json + pmc_forest_*.json)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_${NAME}
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-secondary --no-event-pass --repeats 1"
python bench.py $ARGS $COMMON > $OUT/bench.json 2> $OUT/bench.err
cut -c1-300 $OUT/bench.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS $COMMON > $OUT/trace_bench.json 2> $OUT/trace.err)
ls $OUT/trace | head -5
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py $ARGS $COMMON > $OUT/pmc_$N.json 2> $OUT/pmc_$N.err)
  ls $OUT/pmc_$N | head -3
done
python scripts/summarize_cfg5_profile.py $TAG $NAME
