# usage (on the GPU box, via gpurun): bash scripts/gpu_bench.sh <tag>
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -2 gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head -1 | xargs -r head -8
ls gpurun_out/prof_$TAG/* | head
