# builds the library with extra compiler flags on the GPU box and runs the bench line; restores the shipped library
cd $GRAFT_REPO_ROOT
SRC="multi_agent_pkgs_amd/csrc/hdsm_api.hip multi_agent_pkgs_amd/csrc/map_kernels.hip multi_agent_pkgs_amd/csrc/hdsm_consts.cpp multi_agent_pkgs_amd/csrc/hdsm_level1.cpp multi_agent_pkgs_amd/csrc/swarm_host.cpp multi_agent_pkgs_amd/csrc/corridor_host.cpp"
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
p() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ms_step', round(d['ms_per_step'],4), 'p50', round(d['p50_solve_latency_ms'],4))"; }
echo "shipped:"; python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | p
while IFS= read -r FL; do
  [ -z "$FL" ] && continue
  echo "flags: $FL"
  if /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $FL -std=c++17 -fPIC -shared -o multi_agent_pkgs_amd/libhdsm.so $SRC 2>/tmp/err.log; then
    python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | p
  else
    echo "   build failed: $(grep -m1 error /tmp/err.log | cut -c1-160)"
  fi
done
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
