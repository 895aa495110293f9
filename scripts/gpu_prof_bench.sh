# phase cycle counters (-DHDSM_PROFILE) on the bench line's rounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SRC="multi_agent_pkgs_amd/csrc/hdsm_api.hip multi_agent_pkgs_amd/csrc/map_kernels.hip multi_agent_pkgs_amd/csrc/hdsm_consts.cpp multi_agent_pkgs_amd/csrc/hdsm_level1.cpp multi_agent_pkgs_amd/csrc/swarm_host.cpp multi_agent_pkgs_amd/csrc/corridor_host.cpp"
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DHDSM_PROFILE $EXTRA_DEFS -std=c++17 -fPIC -shared -o multi_agent_pkgs_amd/libhdsm.so $SRC 2>&1 | grep -E "error"
timeout 900 python bench.py --no-cpu-baseline --steps 20 --warmup 2 $BENCH_ARGS > gpurun_out/prof_bench.log 2>&1
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
grep -E "HDSM_PROFILE" gpurun_out/prof_bench.log | tail -2 | cut -c1-900
grep -oE '"kernel_ms_mean": [0-9.]+' gpurun_out/prof_bench.log
