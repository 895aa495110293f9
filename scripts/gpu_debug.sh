cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SRC="multi_agent_pkgs_amd/csrc/hdsm_api.hip multi_agent_pkgs_amd/csrc/hdsm_consts.cpp multi_agent_pkgs_amd/csrc/swarm_host.cpp"
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
for v in "-O3" "-O3 -DHDSM_WSYNC_STRONG" "-O1" "-O3 -DHDSM_DEBUG"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $v -std=c++17 -fPIC -shared -o multi_agent_pkgs_amd/libhdsm.so $SRC 2>&1 | grep -E "error"
  echo "== variant: $v"
  python __graft_entry__.py smoke 2>&1 | grep -v "^it " | tail -2
done > gpurun_out/debug.log 2>&1
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
cat gpurun_out/debug.log
