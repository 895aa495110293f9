cd $GRAFT_REPO_ROOT
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
make -C multi_agent_pkgs_amd/csrc -B CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-parameter -DHDSM_CHECK_EXTRAP" 2>&1 | grep -E "error"
timeout 300 python scripts/gpu_mirror_probe.py 3 2>&1 | grep -E "^round|EXTRAP" | head -40
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
