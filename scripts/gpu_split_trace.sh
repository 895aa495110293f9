# -DHDSM_SPLIT_TRACE build (GPU box only): per split launch, how many instances were handed over to pass 2, how many sub-blocks
# worked, and how the nodes spread over them.   usage: bash scripts/gpu_split_trace.sh <bench args>
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
make -C multi_agent_pkgs_amd/csrc -B CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-parameter -DHDSM_SPLIT_TRACE" 2>&1 | grep -E "error"
timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-event-pass --repeats 1 "$@" > gpurun_out/split_trace.log 2>&1
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
grep HDSM_SPLIT_TRACE gpurun_out/split_trace.log | tail -8 | cut -c1-400
