# phase cycle counters (-DHDSM_PROFILE build, on the GPU box only: the box's copy of the repo is scratch) on the bench rounds
# usage: EXTRA_DEFS="-DHDSM_PROF_STAGE" BENCH_ARGS="--agents 1024" bash scripts/gpu_prof_bench.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
make -C multi_agent_pkgs_amd/csrc -B CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-parameter -DHDSM_PROFILE $EXTRA_DEFS" 2>&1 | grep -E "error"
timeout 900 python bench.py --no-cpu-baseline --no-secondary $BENCH_ARGS > gpurun_out/prof_bench.log 2>&1
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
grep -E "HDSM_PROFILE" gpurun_out/prof_bench.log | tail -2 | cut -c1-900
grep -oE '"kernel_ms_mean": [0-9.]+' gpurun_out/prof_bench.log
