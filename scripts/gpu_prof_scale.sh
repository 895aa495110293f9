cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SRC="multi_agent_pkgs_amd/csrc/hdsm_api.hip multi_agent_pkgs_amd/csrc/map_kernels.hip multi_agent_pkgs_amd/csrc/hdsm_consts.cpp multi_agent_pkgs_amd/csrc/hdsm_level1.cpp multi_agent_pkgs_amd/csrc/swarm_host.cpp multi_agent_pkgs_amd/csrc/corridor_host.cpp"
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DHDSM_PROFILE -std=c++17 -fPIC -shared -o multi_agent_pkgs_amd/libhdsm.so $SRC 2>&1 | grep -E "error"
run() { tag=$1; shift; timeout 1200 python bench.py --no-cpu-baseline "$@" > gpurun_out/prof_scale_$tag.log 2>&1
  echo "== $tag"; grep -E "HDSM_PROFILE" gpurun_out/prof_scale_$tag.log | tail -2 | cut -c1-700; grep -oE '"kernel_ms_mean": [0-9.]+' gpurun_out/prof_scale_$tag.log; }
run a4096h15 --agents 4096 --horizon 15 --steps 6 --warmup 2 --first-round 20
run a1024late --agents 1024 --steps 10 --warmup 2 --first-round 150
run a1024early --agents 1024 --steps 10 --warmup 2 --first-round 30
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
