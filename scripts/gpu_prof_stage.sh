cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SRC="multi_agent_pkgs_amd/csrc/hdsm_api.hip multi_agent_pkgs_amd/csrc/map_kernels.hip multi_agent_pkgs_amd/csrc/hdsm_consts.cpp multi_agent_pkgs_amd/csrc/hdsm_level1.cpp multi_agent_pkgs_amd/csrc/swarm_host.cpp multi_agent_pkgs_amd/csrc/corridor_host.cpp"
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DHDSM_PROFILE -DHDSM_PROF_STAGE -std=c++17 -fPIC -shared -o multi_agent_pkgs_amd/libhdsm.so $SRC 2>&1 | grep -E "error"
run() { tag=$1; shift; timeout 1200 python bench.py --no-cpu-baseline "$@" > gpurun_out/prof_stage_$tag.log 2>&1
  echo "== $tag (slots sw_cull..sw_tail = issue kt/jeq/bounds | requests | own plan | commit | rest)"; grep -E "HDSM_PROFILE" gpurun_out/prof_stage_$tag.log | tail -2 | cut -c1-900; }
run a1024early --agents 1024 --steps 10 --warmup 2 --first-round 30
run a64 --steps 20 --warmup 2
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
