# round 5, corridor work on the GPU box: the decomposition / corridor / device-loop tests, the decomposition batch timings, the
# device-resident loop traced in the forest.   usage: bash scripts/gpu_r5_corridor.sh [tag]
TAG=${1:-r05c}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "decomposition or corridor or device_resident or config_3" > gpurun_out/$TAG/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/$TAG/tests.log
timeout 300 python scripts/bench_corridor.py > gpurun_out/$TAG/f2_corridor.json 2> gpurun_out/$TAG/f2_corridor.err; cut -c1-600 gpurun_out/$TAG/f2_corridor.json
timeout 600 bash scripts/gpu_dloop_trace.sh ${TAG}_dloop_forest --scenario forest --agents 256 --first-round 60 > gpurun_out/$TAG/dloop_forest.log 2>&1; tail -25 gpurun_out/$TAG/dloop_forest.log | cut -c1-300
rm -rf gpurun_out/${TAG}_dloop_*/trace
