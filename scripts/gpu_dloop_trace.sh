# kernel timeline of the device-resident loop (rocprofv3 --kernel-trace of bench.py; scripts/summarize_dloop_trace.py)
# usage: bash scripts/gpu_dloop_trace.sh [tag] [bench args]
TAG=${1:-dloop}; shift
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export HDSM_BENCH_DLOOP_PLAIN=1   # (the traced process ends with the live rounds: no phase-timing rounds, no downloads)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/trace_bench.json 2> $OUT/trace.err)
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python scripts/summarize_dloop_trace.py $f 20 $OUT/dloop_trace.json | tail -60
