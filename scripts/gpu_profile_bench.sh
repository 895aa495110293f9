# usage (GPU box): bash scripts/gpu_profile_bench.sh <tag> [bench args...]   default args = the DRIVER's command line
#   1. the bench line itself                      -> gpurun_out/<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats of the SAME command line (every dispatch of the process is traced; the dispatches of
#      the timed region are picked out of the trace by position, using the launch sequence bench.py prints)
#   3. PMC counters, one pass per counter group (no tracing flags besides --kernel-trace; MI355X_MICROARCH.md "HBM")
#   4. scripts/summarize_profile.py -> profiles/<tag>_* (kernel stats of the timed launches, PMC per launch, workload key)
TAG=${1:-r02}; shift
ARGS=${@:-"--gpus 1 --steps 20 --warmup 5"}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
cut -c1-400 $OUT/bench.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err)
ls $OUT/trace | head -5
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py $ARGS --no-cpu-baseline --no-secondary > $OUT/pmc_$N.json 2> $OUT/pmc_$N.err)
  ls $OUT/pmc_$N | head -3
done
python scripts/summarize_profile.py $TAG
