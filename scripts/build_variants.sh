#!/bin/bash
# builds libhdsm_<name>.so for each "name=DEFS" argument (they travel with gpurun; scripts/gpu_ab_builds.sh benches them)
cd "$(dirname "$0")/../multi_agent_pkgs_amd/csrc"
SRC="hdsm_api.hip map_kernels.hip corridor_kernels.hip swarm_kernels.hip hdsm_consts.cpp hdsm_level1.cpp swarm_host.cpp corridor_host.cpp stats_host.cpp"
for v in "$@"; do
  name=${v%%=*}; defs=${v#*=}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-parameter $defs -shared -o ../libhdsm_$name.so $SRC -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib 2>&1 | grep -E "error" ; echo "built $name ($defs)" ) &
done
wait
