"""Which instance sets the round time? Replays the cached bench rounds (gpurun_out/rounds_cache.npz) with a warm-started
solver and prints, for the slowest rounds, the operation counts of the slowest instances with their status now and in
the previous round."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_pkgs_amd import lib
from multi_agent_pkgs_amd.params import agile_params
z = np.load(f"{ROOT}/gpurun_out/rounds_cache.npz")
prm = agile_params(10, max_rows_static=18)
n = int(z["n_rob"])
sol = lib.Solver(prm, n, n)
keys = ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")
ops, st = [], []
for r in range(z["state"].shape[0]):
    g = sol.replan(*[z[k][r] for k in keys])
    ops.append(sol.last_stats(n)["qp_iters"].copy()), st.append(g["status"].copy())
ops, st = np.array(ops), np.array(st)
mx = ops[10:].max(axis=1)
print("per-round max ops: mean", mx.mean().round(1), "p50", np.percentile(mx, 50), "p90", np.percentile(mx, 90), "max", mx.max())
who = ops[10:].argmax(axis=1)
cur = st[10:][np.arange(len(who)), who]
prev = st[9:-1][np.arange(len(who)), who]
for name, sel in (("optimal now, optimal before", (cur == 0) & (prev == 0)), ("optimal now, no solution before", (cur == 0) & (prev == 2)),
                  ("no solution now, optimal before", (cur == 2) & (prev == 0)), ("no solution now and before", (cur == 2) & (prev == 2))):
    if sel.any():
        print(f"  slowest instance of the round is '{name}' in {sel.sum()} rounds, its ops: mean {mx[sel].mean():.1f} max {mx[sel].max()}")
