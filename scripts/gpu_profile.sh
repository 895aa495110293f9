# usage (GPU box): bash scripts/gpu_profile.sh <tag>   -> gpurun_out/<tag>_*  (copy what matters into profiles/)
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CACHE=$GRAFT_REPO_ROOT/gpurun_out/rounds_cache.npz
# 1. the bench line (also writes the recorded rounds)
python bench.py --cache $CACHE > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cat gpurun_out/${TAG}_bench.json | cut -c1-700
export TMPDIR=/tmp
# 2. kernel trace + stats of the SAME workload: rounds come from the cache, warm-up 0, no event pass
#    -> the process launches k_replan exactly `steps` times, on the rounds bench.py times
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -o bench -- \
  python $GRAFT_REPO_ROOT/bench.py --cache $CACHE --no-cpu-baseline --warmup 0 --round-offset 10 --no-event-pass > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace.err
cd $GRAFT_REPO_ROOT
head -3 gpurun_out/${TAG}_trace/bench_kernel_stats.csv
# 3. PMC counters, each in its own pass (no tracing flags besides kernel-trace; see MI355X_MICROARCH.md "HBM")
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  N=$(echo $C | tr ' ' '_')
  (cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$N -o pmc -- \
    python $GRAFT_REPO_ROOT/bench.py --cache $CACHE --no-cpu-baseline --warmup 0 --round-offset 10 --no-event-pass > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_$N.err)
  ls gpurun_out/${TAG}_pmc_$N | head -3
done
python scripts/summarize_pmc.py $TAG
cp gpurun_out/${TAG}_trace/bench_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
# 4. the bench line once more, now that profiles/pmc_latest.json (written by summarize_pmc.py above) holds this
#    kernel's HBM traffic: this is the line to keep
python bench.py --cache $CACHE > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cut -c1-260 gpurun_out/${TAG}_bench.json
