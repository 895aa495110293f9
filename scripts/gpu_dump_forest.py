"""Fly BASELINE cfg 3 (256 agents through the pillar forest) in closed loop on the GPU and save, for a few rounds, the solver inputs
of the instances with the largest branch-and-bound trees (gpurun_out/forest_hard.npz) for offline work on the search
(tests/wave_emu runs the device source on them). usage: python scripts/gpu_dump_forest.py [LAST_ROUND=80] [TOP=6]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_pkgs_amd import lib, swarm  # noqa: E402
from multi_agent_pkgs_amd import scenarios as sc  # noqa: E402
from multi_agent_pkgs_amd.params import agile_params, agile_ref_config  # noqa: E402

last = int(sys.argv[1]) if len(sys.argv) > 1 else 80
top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n_rob = 256
prm = agile_params(10, max_rows_static=18)
sol = lib.Solver(prm, n_rob, n_rob)
rcfg = agile_ref_config()


def solve(inp, plans, has):
    return sol.replan(inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"], inp["b"], plans, has)


def ref_dev(ids, path, n_path, plans, has, vel_cap=None):
    full, _, pv = sol.reference(rcfg, ids, path, n_path, plans, has, vel_cap=vel_cap)
    return full, pv


raw, origin = sc.forest_for_circle(n_rob, seed=13)
loop = swarm.SwarmLoop(prm, swarm.default_swarm_config(), n_rob, solve=solve, reference=ref_dev)
assert loop.set_world(sc.inflate(raw), origin) == 0
keep = []
hist = []
prev_used = None
for r in range(last + 1):
    rec = []
    out = loop.step(record=rec)
    st = sol.last_stats(n_rob)
    hist.append((r, int(st["nodes"].max()), int(st["nodes"].sum()), int((st["nodes"] > 1).sum()), int(st["qp_iters"].max())))
    if r >= 40 and r % 8 == 0:
        order = np.argsort(-st["nodes"])[:top]
        for a in order:
            x = rec[0]
            keep.append(dict(round=r, agent=int(a), nodes=int(st["nodes"][a]), iters=int(st["qp_iters"][a]), status=int(out["status"][a]),
                             obj=float(out["obj"][a]), state=x["state"][a], ref=x["ref"][a], n_poly=int(x["n_poly"][a]), n_rows=x["n_rows"][a],
                             A=x["A"][a], b=x["b"][a], plans=x["plans"], has_plan=x["has_plan"], used=out["used"][a], traj=out["traj"][a]))
print("round, nodes max, nodes sum, instances with a tree, iters max")
for h in hist[::4]:
    print(h)
out = {}
for k, d in enumerate(keep):
    for key, v in d.items():
        out[f"c{k}_{key}"] = v
out["n"] = len(keep)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "forest_hard.npz"), **out)
print("saved", len(keep), "instances; node counts", [d["nodes"] for d in keep])
