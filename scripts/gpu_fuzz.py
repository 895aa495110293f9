#!/usr/bin/env python3
"""Fuzz: many random swarm snapshots through the HIP path vs the oracle. Reports status mismatches and the worst
trajectory / objective deviation. Usage: python scripts/gpu_fuzz.py [n_cases]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from multi_agent_pkgs_amd import lib  # noqa: E402
from multi_agent_pkgs_amd.params import make_params  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(12345)
K = ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")
tot = mism = 0
kinds = {}
worst_t = worst_o = 0.0
lim = 0
t0 = time.time()
solvers = {}
for case in range(n_cases):
    n_hor = int(rng.choice([6, 8, 10, 10, 10, 12, 15]))
    n_rob = int(rng.choice([9, 16, 25, 36, 49, 64]))
    kw = dict(spacing=float(rng.choice([0.8, 1.0, 1.3, 1.8, 2.5])), narrow=bool(rng.random() < 0.35),
              turn=bool(rng.random() < 0.5), chamfer=bool(rng.random() < 0.3),
              absent_frac=float(rng.choice([0, 0, 0.2])), speed=(0.0, float(rng.choice([3.0, 6.0, 9.0]))))
    rk4 = bool(rng.random() < 0.3)
    drag = tuple(rng.choice([0.0, 0.0, 0.1, 0.3], 3))
    prm = make_params(n_hor=n_hor, rk4=rk4, drag=drag, max_rows_static=18, poly_hor=int(rng.choice([2, 3, 4])))
    sn = problems.swarm_snapshot(prm, n_rob, seed=1000 + case, **kw)
    args = [sn[k] for k in K]
    key = (n_hor, rk4, drag, prm.poly_hor, n_rob)
    sol = lib.Solver(prm, n_rob, n_rob)  # fresh handle: cold start; second call below exercises the warm start
    for rep in range(2):
        g = sol.replan(*args)
        if rep == 0:
            o = orc.replan(prm, *args, n_threads=32)
        tot += n_rob
        bad = g["status"] != o["status"]
        lim += int((g["status"] == 1).sum())
        ok = (g["status"] == 0) & (o["status"] == 0)
        if ok.any():
            dt = np.abs(g["traj"] - o["traj"])[ok].reshape(ok.sum(), -1).max(1)
            do = np.abs(g["obj"] - o["obj"])[ok] / np.maximum(1, np.abs(o["obj"][ok]))
            # a different but equally good optimum (objective equal) is a tie, not an error
            tie = (dt > 1e-6) & (do < 1e-9)
            err = (dt > 1e-6) & ~tie
            worst_t = max(worst_t, float(dt[~tie].max()) if (~tie).any() else 0.0)
            worst_o = max(worst_o, float(do.max()))
            mism += int(err.sum())
            if err.any():
                print("TRAJ MISMATCH case", case, "rep", rep, "inst", np.where(ok)[0][err].tolist(), dt[err], do[err])
        if bad.any():
            mism += int(bad.sum())
            for gs, os_ in zip(g["status"][bad].tolist(), o["status"][bad].tolist()):
                kinds[(gs, os_)] = kinds.get((gs, os_), 0) + 1
            print("STATUS MISMATCH case", case, "rep", rep, dict(n_hor=n_hor, n_rob=n_rob, rk4=rk4, **kw),
                  "inst", np.where(bad)[0].tolist(), "gpu", g["status"][bad].tolist(), "oracle", o["status"][bad].tolist())
print("status mismatches by (device, oracle) status:", kinds, "| trajectory mismatches:", mism - sum(kinds.values()))
print(f"fuzz: {tot} instance-solves in {n_cases} cases, mismatches {mism}, LIMIT statuses {lim}, "
      f"worst |dtraj| {worst_t:.2e}, worst rel |dobj| {worst_o:.2e}, {time.time() - t0:.0f} s")
