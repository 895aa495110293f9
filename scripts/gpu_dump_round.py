"""Fly the bench's circle in closed loop on the GPU up to ROUND and save that round's solver inputs and results
(gpurun_out/round_<ROUND>.npz) for offline analysis. usage: python scripts/gpu_dump_round.py [ROUND=175] [AGENTS=1024]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_pkgs_amd import lib, swarm  # noqa: E402
from multi_agent_pkgs_amd.params import agile_params, agile_ref_config  # noqa: E402

rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 175
n_rob = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
prm = agile_params(10, max_rows_static=18)
sol = lib.Solver(prm, n_rob, n_rob)
rcfg = agile_ref_config()


def solve(inp, plans, has):
    return sol.replan(inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"], inp["b"], plans, has)


def ref_dev(ids, path, n_path, plans, has, vel_cap=None):
    full, _, pv = sol.reference(rcfg, ids, path, n_path, plans, has, vel_cap=vel_cap)
    return full, pv


loop = swarm.SwarmLoop(prm, swarm.default_swarm_config(), n_rob, solve=solve, radius=max(22.0, n_rob / (2 * np.pi)), reference=ref_dev)
for r in range(rnd + 1):
    rec = []
    out = loop.step(record=rec if r == rnd else None)
st = sol.last_stats(n_rob)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"round_{rnd}.npz"), status=out["status"], qp_iters=st["qp_iters"],
                    **{k: v for k, v in rec[0].items()})
print("round", rnd, "no solution:", int((out["status"] == 2).sum()), "iters max", int(st["qp_iters"].max()))
