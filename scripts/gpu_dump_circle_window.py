"""Fly the bench workload (1024 agents, circular exchange, H = 10) in closed loop on the GPU and save the solver inputs and the
per-instance statistics of a few rounds of the bench window (gpurun_out/circle_window.npz) for offline work on the slowest
instances (tests/wave_emu runs the device source on them, with the same warm-start chain).
usage: python scripts/gpu_dump_circle_window.py [FIRST=168] [ROUNDS=6]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_pkgs_amd import lib, swarm  # noqa: E402
from multi_agent_pkgs_amd.params import agile_params, agile_ref_config  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 168
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n_rob = 1024
prm = agile_params(10, max_rows_static=18)
sol = lib.Solver(prm, n_rob, n_rob)
rcfg = agile_ref_config()


def solve(inp, plans, has):
    return sol.replan(inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"], inp["b"], plans, has)


def ref_dev(ids, path, n_path, plans, has, vel_cap=None):
    full, _, pv = sol.reference(rcfg, ids, path, n_path, plans, has, vel_cap=vel_cap)
    return full, pv


loop = swarm.SwarmLoop(prm, swarm.default_swarm_config(), n_rob, solve=solve, reference=ref_dev, radius=max(22.0, n_rob / (2 * np.pi)))
save = {}
for r in range(first + rounds):
    rec = []
    out = loop.step(record=rec if r >= first else None)
    if r >= first:
        st = sol.last_stats(n_rob)
        k = r - first
        x = rec[0]
        for key in ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan"):
            save[f"r{k}_{key}"] = x[key]
        for key in ("status", "obj", "traj"):
            save[f"r{k}_{key}"] = out[key]
        for key in ("qp_iters", "nodes", "sweeps"):
            save[f"r{k}_{key}"] = st[key]
        o = np.argsort(-st["qp_iters"])[:6]
        print("round", r, "fails", int((out["status"] == 2).sum()), "slowest:", [(int(a), int(st["qp_iters"][a]), int(out["status"][a])) for a in o])
save["first"], save["rounds"] = first, rounds
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "circle_window.npz"), **save)
print("saved")
