"""How much work do FEASIBLE instances need, and how much is burnt on infeasible ones? Replays cached bench rounds
(bench.py --cache) and prints, by final status, the distribution of active-set operations and B&B nodes."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_pkgs_amd import lib
from multi_agent_pkgs_amd.params import agile_params
for n_rob, H in [(64, 10), (128, 10), (256, 10), (64, 15), (128, 15)]:
    cache = f"{ROOT}/gpurun_out/limit_cache_{n_rob}_{H}.npz"
    if not os.path.exists(cache):
        subprocess.check_call([sys.executable, "bench.py", "--no-cpu-baseline", "--agents", str(n_rob), "--horizon", str(H),
                               "--cache", cache, "--no-event-pass"], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    z = np.load(cache)
    prm = agile_params(H, max_rows_static=18)
    sol = lib.Solver(prm, n_rob, n_rob)
    keys = ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")
    it, nd, stt = [], [], []
    for r in range(z["state"].shape[0]):
        g = sol.replan(*[z[k][r] for k in keys])
        s = sol.last_stats(n_rob)
        it.append(s["qp_iters"].copy()), nd.append(s["nodes"].copy()), stt.append(g["status"].copy())
    it, nd, stt = np.concatenate(it), np.concatenate(nd), np.concatenate(stt)
    ok, bad = stt == 0, stt == 2
    q = lambda a: [int(np.percentile(a, p)) for p in (50, 90, 99, 100)] if a.size else []
    print(f"{n_rob}xH{H}: optimal {ok.sum()} ops p50/90/99/max {q(it[ok])} nodes {q(nd[ok])} | no-solution {bad.sum()} ops {q(it[bad])} nodes {q(nd[bad])} | limit {(stt == 1).sum()}")
