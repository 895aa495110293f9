set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
python bench.py --no-cpu-baseline > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; tail -3 gpurun_out/bench_quick.err; cat gpurun_out/bench_quick.json
