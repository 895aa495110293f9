"""Fly a circular exchange in closed loop on the GPU and print, per round, what the solver had to do:
failed / limited instances, active-set operations, B&B nodes, host-path time, closest pair. Used to choose the
recorded window of the bench line (the rounds around the crossing).

usage: python scripts/gpu_flight_probe.py AGENTS HORIZON ROUNDS [RADIUS] > gpurun_out/flight_<agents>.jsonl
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_pkgs_amd import lib, swarm  # noqa: E402
from multi_agent_pkgs_amd.params import agile_params, agile_ref_config  # noqa: E402

n_rob, H, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
radius = float(sys.argv[4]) if len(sys.argv) > 4 else None
prm = agile_params(H, max_rows_static=18)
sol = lib.Solver(prm, n_rob, n_rob)
rcfg = agile_ref_config()


def solve(inp, plans, has):
    return sol.replan(inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"], inp["b"], plans, has)


def ref_dev(ids, path, n_path, plans, has, vel_cap=None):
    full, _, pv = sol.reference(rcfg, ids, path, n_path, plans, has, vel_cap=vel_cap)
    return full, pv


loop = swarm.SwarmLoop(prm, swarm.default_swarm_config(), n_rob, solve=solve, radius=radius, reference=ref_dev)
for r in range(rounds):
    t0 = time.perf_counter()
    out = loop.step()
    dt = time.perf_counter() - t0
    pos, dist, nfail = loop.shard.state()
    # closest pair (chunked)
    dmin = 1e9
    for a in range(0, n_rob, 512):
        d = np.linalg.norm(pos[a:a + 512, None, :] - pos[None, :, :], axis=2)
        d[np.arange(d.shape[0]), np.arange(a, a + d.shape[0])] = 1e9
        dmin = min(dmin, float(d.min()))
    print(json.dumps(dict(round=r, wall_ms=dt * 1e3, fail=int((out["status"] == 2).sum()), limit=int((out["status"] == 1).sum()),
                          it_max=int(out["qp_iters"].max()), it_mean=float(out["qp_iters"].mean()), nodes_max=int(out["nodes"].max()),
                          cand_max=int(out["cand"].max()), dist_goal_mean=float(dist.mean()), closest=dmin)), flush=True)
