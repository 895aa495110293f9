# launch timeline of k_replan on the bench rounds (-DHDSM_TIMELINE build, on the GPU box only): per launch, the span seen by
# the instances, the slowest instance and when it started, the instance that finished last, late starters
# usage: bash scripts/gpu_timeline.sh [bench args]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
make -C multi_agent_pkgs_amd/csrc -B CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-parameter -DHDSM_TIMELINE $EXTRA_DEFS" 2>&1 | grep -E "error"
rm -f gpurun_out/timeline.bin; HDSM_TIMELINE_DUMP=$GRAFT_REPO_ROOT/gpurun_out/timeline.bin timeout 900 python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/timeline_bench.log 2>&1
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
grep -E "HDSM_TIMELINE" gpurun_out/timeline_bench.log | tail -24 | cut -c1-400
grep -oE '"kernel_ms_mean": [0-9.]+' gpurun_out/timeline_bench.log
