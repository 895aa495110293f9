cd $GRAFT_REPO_ROOT
SRC="multi_agent_pkgs_amd/csrc/hdsm_api.hip multi_agent_pkgs_amd/csrc/map_kernels.hip multi_agent_pkgs_amd/csrc/hdsm_consts.cpp multi_agent_pkgs_amd/csrc/hdsm_level1.cpp multi_agent_pkgs_amd/csrc/swarm_host.cpp multi_agent_pkgs_amd/csrc/corridor_host.cpp"
cp multi_agent_pkgs_amd/libhdsm.so /tmp/libhdsm_orig.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DHDSM_DEBUG -std=c++17 -fPIC -shared -o multi_agent_pkgs_amd/libhdsm.so $SRC 2>&1 | grep -E "error"
python - <<'PY' 2>&1 | tail -60
import sys; sys.path.insert(0,'tests')
import numpy as np, problems
from multi_agent_pkgs_amd import lib
from multi_agent_pkgs_amd.params import agile_params
from oracle import pyoracle as orc
prm = agile_params(10, max_rows_static=18)
sol = lib.Solver(prm, 25, 25)
K=("agent_id","state","ref","n_poly","n_rows","A","b","plans","has_plan")
sn = problems.swarm_snapshot(prm,25,61,spacing=1.2)
args=[sn[k] for k in K]
g=sol.replan(*args)
print("first: status", g['status'][2], 'iters', g['qp_iters'][2])
g=sol.replan(*args); o=orc.replan(prm,*args,n_threads=8)
print("second: status", g['status'][2], o['status'][2], 'iters', g['qp_iters'][2], 'obj', g['obj'][2], o['obj'][2])
PY
cp /tmp/libhdsm_orig.so multi_agent_pkgs_amd/libhdsm.so
