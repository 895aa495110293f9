import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_pkgs_amd import lib
from multi_agent_pkgs_amd.params import agile_params
z = np.load(f"{ROOT}/gpurun_out/shard_cache_128.npz")
prm = agile_params(10, max_rows_static=18)
keys = ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b")
for sub in (slice(0, 64), slice(64, 128), slice(0, 128)):
    n = sub.stop - sub.start
    sol = lib.Solver(prm, n, 128)
    rows = []
    for r in range(z["state"].shape[0]):
        t0 = time.perf_counter()
        g = sol.replan(*[z[k][r][sub] for k in keys], z["plans"][r], z["has_plan"][r])
        dt = (time.perf_counter() - t0) * 1e3
        st = sol.last_stats(n)
        rows.append((r, round(dt, 3), int(st["qp_iters"].max()), int(st["nodes"].max()), int(st["sweeps"].max()), int((g["status"] == 2).sum()), int((g["status"] == 1).sum())))
    rows.sort(key=lambda x: -x[1])
    print(sub, "slowest rounds (round, host ms, iters max, nodes max, sweeps max, n infeasible, n limit):", rows[:6])
