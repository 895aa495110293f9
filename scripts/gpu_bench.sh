# usage (on the GPU box, via gpurun): bash scripts/gpu_bench.sh
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_t64.json 2> gpurun_out/bench_t64.err; tail -3 gpurun_out/bench_t64.err; cat gpurun_out/bench_t64.json
HDSM_THREADS=128 python bench.py --no-cpu-baseline > gpurun_out/bench_t128.json 2>> gpurun_out/bench_t64.err; cat gpurun_out/bench_t128.json
HDSM_THREADS=256 python bench.py --no-cpu-baseline > gpurun_out/bench_t256.json 2>> gpurun_out/bench_t64.err; cat gpurun_out/bench_t256.json
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof | head -20
find gpurun_out/prof -name "*kernel_stats*" | head -1 | xargs -r head -12
