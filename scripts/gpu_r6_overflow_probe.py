"""round 6 probe: the staging-overflow rescue case (tests/test_gpu_abi.py::test_staging_overflow_in_a_shared_cu_kernel_is_rescued) with details."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import _case, K
from multi_agent_pkgs_amd import lib as hdsm
from oracle import pyoracle as oracle
rng = np.random.default_rng(12345)
for case in range(12):
    prm, n_rob, kw, sn = _case(rng, case)
args = [sn[k] for k in K]
big = prm.copy(); big.max_nodes, big.max_qp_iters = 500000, 100000000
o = oracle.replan(big, *args, n_threads=32, search=1)
for env in ({}, {"HDSM_DUO_MIN": "1"}, {"HDSM_DUO_MIN": "1", "HDSM_DUO48_ROWS": "320"}):
    for k, v in env.items(): os.environ[k] = v
    sol = hdsm.Solver(prm, n_rob, n_rob)
    for k in env: del os.environ[k]
    for rep in range(2):
        g = sol.replan(*args)
        fl = sol.last_sweep_stats(n_rob)["flags"]; st = sol.last_stats(n_rob)
        ok = (o["status"] == 0) & (g["status"] == 0)
        print(os.environ.get("HDSM_LIBRARY", "default")[-24:], env, "rep", rep, "status mism", int((g["status"] != o["status"]).sum()), "dev status", np.bincount(g["status"], minlength=3).tolist(),
              "oracle", np.bincount(o["status"], minlength=3).tolist(), "flags", np.bincount(fl, minlength=9).tolist(), "dtraj", float(np.abs(g["traj"] - o["traj"])[ok].max()) if ok.any() else None,
              "cand max", int(st["cand"].max()), "iters max", int(st["qp_iters"].max()), flush=True)
    sol.close()

# the raw answers of the first launch through the asynchronous entry point (no rescue pass on a fresh handle)
import torch
dev = torch.device("cuda", 0)
dt = dict(agent_id=torch.int32, state=torch.float64, ref=torch.float64, n_poly=torch.int32, n_rows=torch.int32, A=torch.float64, b=torch.float64, plans=torch.float64, has_plan=torch.uint8)
d = {k: torch.from_numpy(np.ascontiguousarray(sn[k])).to(dev).to(dt[k]).contiguous() for k in K}
N, P = prm.n_hor, prm.poly_hor
for env in ({"HDSM_DUO_MIN": "1"}, {"HDSM_DUO_MIN": "1", "HDSM_DUO48_ROWS": "320"}, {"HDSM_DUO_MIN": "1", "HDSM_DUO48_ROWS": "320", "HDSM_SCANNER": "0"}, {"HDSM_DUO_MIN": "1", "HDSM_DUO48_ROWS": "320", "HDSM_THREADS": "64"}):
    for k, v in env.items(): os.environ[k] = v
    sol = hdsm.Solver(prm, n_rob, n_rob)
    for k in env: del os.environ[k]
    out = dict(traj=torch.zeros((n_rob, N + 1, 9), dtype=torch.float64, device=dev), ctrl=torch.zeros((n_rob, N, 3), dtype=torch.float64, device=dev),
               used=torch.zeros((n_rob, P), dtype=torch.uint8, device=dev), status=torch.full((n_rob,), 7, dtype=torch.int32, device=dev), obj=torch.zeros(n_rob, dtype=torch.float64, device=dev))
    sol.replan_device(*[d[k] for k in K], out["traj"], out["ctrl"], out["used"], out["status"], out["obj"])
    torch.cuda.synchronize()
    st, fl, ss = out["status"].cpu().numpy(), sol.last_sweep_stats(n_rob)["flags"], sol.last_stats(n_rob)
    print("RAW", env, "\n status", st.tolist(), "\n oracle", o["status"].tolist(), "\n flags ", fl.tolist(), "\n iters ", ss["qp_iters"].tolist(), "\n nodes ", ss["nodes"].tolist(), "\n sweeps", ss["sweeps"].tolist(), "\n cand  ", ss["cand"].tolist(), flush=True)
    sol.close()
