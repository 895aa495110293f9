"""Fly BASELINE cfg 5 (4096 agents, forest + wall + forest, H = 15) for a few rounds on the GPU and save the inputs of the instances
with the largest branch-and-bound trees (gpurun_out/fwf_hard.npz). usage: python scripts/gpu_dump_fwf.py [LAST_ROUND=12] [TOP=4]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_pkgs_amd import lib, swarm  # noqa: E402
from multi_agent_pkgs_amd import scenarios as sc  # noqa: E402
from multi_agent_pkgs_amd.params import agile_params, agile_ref_config  # noqa: E402

last = int(sys.argv[1]) if len(sys.argv) > 1 else 12
top = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_y = n_z = 64
n_rob, N = n_y * n_z, 15
prm = agile_params(N, max_rows_static=18)
sol = lib.Solver(prm, n_rob, n_rob)
rcfg = agile_ref_config()


def solve(inp, plans, has):
    return sol.replan(inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"], inp["b"], plans, has)


def ref_dev(ids, path, n_path, plans, has, vel_cap=None):
    full, _, pv = sol.reference(rcfg, ids, path, n_path, plans, has, vel_cap=vel_cap)
    return full, pv


raw, origin = sc.forest_wall_forest(int(np.ceil((10 + 2.01 * n_y) / 30)), int(np.ceil((9 + 2.01 * n_z) / 15)), seed=0)
starts, goals = sc.lattice_scenario(n_y, n_z)
cfg = swarm.default_swarm_config()
cfg.grid_range[2], cfg.grid_z_min = 12.0, -6.0
loop = swarm.SwarmLoop(prm, cfg, n_rob, solve=solve, reference=ref_dev, starts=starts, goals=goals)
assert loop.set_world(sc.inflate(raw), origin) == 0
keep = []
for r in range(last + 1):
    rec = []
    out = loop.step(record=rec)
    st = sol.last_stats(n_rob)
    fl = sol.last_sweep_stats(n_rob)["flags"]
    print("round", r, "nodes max", int(st["nodes"].max()), "sum", int(st["nodes"].sum()), "trees", int((st["nodes"] > 1).sum()), "> 100 nodes", int((st["nodes"] > 100).sum()),
          "limit flags", int((fl & 1).sum()), "iters max", int(st["qp_iters"].max()), flush=True)
    if r >= 6 and r % 3 == 0:
        x = rec[0]
        for a in np.argsort(-st["nodes"])[:top]:
            nb = np.where(np.linalg.norm(x["plans"][:, 0, :3] - x["state"][a, :3], axis=1) < 12.0)[0]   # the neighbours that can matter
            keep.append(dict(round=r, agent=int(np.where(nb == a)[0][0]), nodes=int(st["nodes"][a]), iters=int(st["qp_iters"][a]), status=int(out["status"][a]),
                             obj=float(out["obj"][a]), state=x["state"][a], ref=x["ref"][a], n_poly=int(x["n_poly"][a]), n_rows=x["n_rows"][a],
                             A=x["A"][a], b=x["b"][a], plans=x["plans"][nb], has_plan=x["has_plan"][nb], used=out["used"][a], traj=out["traj"][a]))
o = {}
for k, d in enumerate(keep):
    for key, v in d.items():
        o[f"c{k}_{key}"] = v
o["n"] = len(keep)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "fwf_hard.npz"), **o)
print("saved", len(keep), [d["nodes"] for d in keep])
