# phase profile of the cooperative voxel decomposition on the GPU box (a -DCD_PROFILE build of the corridor kernels only)
# usage: bash scripts/gpu_corridor_profile.sh [tag] [n_seeds]
TAG=${1:-r05c}; N=${2:-256}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
(cd multi_agent_pkgs_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-parameter -DCD_PROFILE -shared -o /tmp/libcorr_prof.so corridor_kernels.hip corridor_host.cpp 2>&1 | grep -E "error")
timeout 300 python scripts/corridor_profile.py /tmp/libcorr_prof.so $N > gpurun_out/$TAG/corridor_profile.json 2> gpurun_out/$TAG/corridor_profile.err; cat gpurun_out/$TAG/corridor_profile.json; tail -3 gpurun_out/$TAG/corridor_profile.err
