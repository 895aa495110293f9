"""round 6 probe: raw answers of k_replan_duo48<48, 320, 128> on fuzz case 11 (first launch, no rescue) for the library named by HDSM_LIBRARY"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import _case, K
from multi_agent_pkgs_amd import lib as hdsm
from oracle import pyoracle as oracle
import torch
rng = np.random.default_rng(12345)
for case in range(12):
    prm, n_rob, kw, sn = _case(rng, case)
args = [sn[k] for k in K]
big = prm.copy(); big.max_nodes, big.max_qp_iters = 500000, 100000000
o = oracle.replan(big, *args, n_threads=32, search=1)
dev = torch.device("cuda", 0)
dt = dict(agent_id=torch.int32, state=torch.float64, ref=torch.float64, n_poly=torch.int32, n_rows=torch.int32, A=torch.float64, b=torch.float64, plans=torch.float64, has_plan=torch.uint8)
d = {k: torch.from_numpy(np.ascontiguousarray(sn[k])).to(dev).to(dt[k]).contiguous() for k in K}
N, P = prm.n_hor, prm.poly_hor
for env in ({"HDSM_DUO_MIN": "1", "HDSM_DUO48_ROWS": "320"}, {"HDSM_DUO_MIN": "1"}):
    for k, v in env.items(): os.environ[k] = v
    sol = hdsm.Solver(prm, n_rob, n_rob)
    for k in env: del os.environ[k]
    out = dict(traj=torch.zeros((n_rob, N + 1, 9), dtype=torch.float64, device=dev), ctrl=torch.zeros((n_rob, N, 3), dtype=torch.float64, device=dev),
               used=torch.zeros((n_rob, P), dtype=torch.uint8, device=dev), status=torch.full((n_rob,), 7, dtype=torch.int32, device=dev), obj=torch.zeros(n_rob, dtype=torch.float64, device=dev))
    sol.replan_device(*[d[k] for k in K], out["traj"], out["ctrl"], out["used"], out["status"], out["obj"])
    torch.cuda.synchronize()
    st, fl, ss = out["status"].cpu().numpy(), sol.last_sweep_stats(n_rob)["flags"], sol.last_stats(n_rob)
    wrong = (st != o["status"]) & (fl & 8 == 0)
    print("RAW", os.environ.get("HDSM_LIBRARY", "default")[-28:], env.get("HDSM_DUO48_ROWS", "720"), "wrong without flag:", int(wrong.sum()), "flagged", int((fl & 8 != 0).sum()), "iters", ss["qp_iters"].tolist()[:12], flush=True)
    sol.close()
