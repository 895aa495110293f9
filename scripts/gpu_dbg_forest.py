"""Debug aid: fly cfg 3 on the device and dump the first round in which an agent centre sits in an obstacle voxel."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_pkgs_amd import lib, swarm, scenarios as sc
from multi_agent_pkgs_amd.params import agile_params, agile_ref_config
n = 256
prm = agile_params(10, max_rows_static=18)
sol = lib.Solver(prm, n, n)
rcfg = agile_ref_config()
def solve(inp, plans, has):
    return sol.replan(inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"], inp["b"], plans, has)
def ref_dev(ids, path, n_path, plans, has, vel_cap=None):
    full, _, pv = sol.reference(rcfg, ids, path, n_path, plans, has, vel_cap=vel_cap)
    return full, pv
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 13
raw, org = sc.forest_for_circle(n, seed=SEED)
occ = sc.inflate(raw)
loop = swarm.SwarmLoop(prm, swarm.default_swarm_config(), n, solve=solve, reference=ref_dev)
print("route failed", loop.set_world(occ, org))
hist = []
for r in range(130):
    rec = []
    out = loop.step(record=rec)
    pos, dist, nfail = loop.shard.state()
    hist.append((rec[0], out, pos.copy()))
    v = np.floor((pos - org) / 0.3).astype(int)
    hit = raw[v[:, 2], v[:, 1], v[:, 0]] >= 100
    hit_inf = occ[v[:, 2], v[:, 1], v[:, 0]] >= 100
    if r % 40 == 0 or hit.any():
        print(r, "fails", int((out["status"] == 2).sum()), "hit", np.where(hit)[0], "in inflated", int(hit_inf.sum()), "itmax", out["qp_iters"].max(), "nodes", out["nodes"].max())
    if hit.any():
        a = int(np.where(hit)[0][0])
        for back in (2, 1, 0):
            rc, ot, ps = hist[-1 - back]
            print("round", r - back, "agent", a, "pos", ps[a], "status", ot["status"][a], "n_poly", rc["n_poly"][a], "n_rows", rc["n_rows"][a], "used", ot["used"][a])
            print(" state", rc["state"][a]); print(" traj p", ot["traj"][a][:, :3])
            for j in range(rc["n_poly"][a]):
                rr = rc["n_rows"][a, j]; A = rc["A"][a, j, :rr]; b = rc["b"][a, j, :rr]
                print("  poly", j, "max viol of pos", (A @ ps[a] - b).max(), "max viol of traj pts", (ot["traj"][a][:, :3] @ A.T - b).max(axis=1).round(3))
                if back == 0: print(np.c_[A, b])
        # occupied voxels (inflated) strictly inside each polyhedron of the last round
        rc, ot, ps = hist[-1]
        zz, yy, xx = np.nonzero(occ >= 100)
        ctr = (np.stack([xx, yy, zz], 1) + 0.5) * 0.3 + org
        near = np.linalg.norm(ctr - ps[a], axis=1) < 6
        for j in range(rc["n_poly"][a]):
            rr = rc["n_rows"][a, j]; A = rc["A"][a, j, :rr]; b = rc["b"][a, j, :rr]
            inside = (ctr[near] @ A.T - b < -1e-9).all(axis=1)
            print("  poly", j, "inflated-occupied voxel centres strictly inside:", int(inside.sum()), ctr[near][inside][:5])
        break
