"""Development aid: two solver handles fly the same 48-agent forest (host-mirror loops); at the first round in which their
answers differ, the instances that differ are printed with status / objective / nodes / flags from both handles, and the
oracle's answer for them. usage: python scripts/gpu_debug_mirror.py [TRIES=6]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_pkgs_amd import lib, swarm  # noqa: E402
from multi_agent_pkgs_amd import scenarios as sc  # noqa: E402
from multi_agent_pkgs_amd.params import agile_params, agile_ref_config  # noqa: E402
from oracle import pyoracle  # noqa: E402

n_rob, N = 48, 10
prm = agile_params(N, max_rows_static=18)
rcfg = agile_ref_config()


def make():
    sol = lib.Solver(prm, n_rob, n_rob)
    last = {}

    def solve(inp, plans, has):
        out = sol.replan(inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"], inp["b"], plans, has)
        last["inp"], last["plans"], last["has"], last["out"] = inp, plans.copy(), has.copy(), out
        last["flags"] = sol.last_sweep_stats(n_rob)["flags"].copy()
        return out

    def ref_dev(ids, path, n_path, plans, has, vel_cap=None):
        full, _, pv = sol.reference(rcfg, ids, path, n_path, plans, has, vel_cap=vel_cap)
        return full, pv

    loop = swarm.SwarmLoop(prm, swarm.default_swarm_config(), n_rob, solve=solve, reference=ref_dev)
    raw, origin = sc.forest_for_circle(n_rob, seed=21)
    assert loop.set_world(sc.inflate(raw), origin) == 0
    return sol, loop, last


for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    (sa, la, ta), (sb, lb, tb) = make(), make()
    for r in range(40):
        oa, ob = la.step(), lb.step()
        d = np.abs(oa["traj"] - ob["traj"]).max(axis=(1, 2))
        bad = np.where((d > 1e-7) | (oa["status"] != ob["status"]))[0]
        if len(bad):
            print("try", t, "round", r, "differing agents", bad.tolist())
            inp = ta["inp"]
            same_inputs = all(np.array_equal(ta["inp"][k], tb["inp"][k]) for k in ("state", "ref", "A", "b", "n_rows", "n_poly")) and np.array_equal(ta["plans"], tb["plans"])
            print("  inputs identical:", same_inputs, "max input diff", max(float(np.abs(ta["inp"][k] - tb["inp"][k]).max()) for k in ("state", "ref", "A", "b")), float(np.abs(ta["plans"] - tb["plans"]).max()))
            o = pyoracle.replan(prm, inp["agent_id"][bad], inp["state"][bad], inp["ref"][bad], inp["n_poly"][bad], inp["n_rows"][bad], inp["A"][bad], inp["b"][bad], ta["plans"], ta["has"], n_threads=8)
            for i, a in enumerate(bad):
                print("  agent", a, "A: st", oa["status"][a], "obj %.9g" % oa["obj"][a], "nodes", oa["nodes"][a], "flags", hex(int(ta["flags"][a])),
                      "| B: st", ob["status"][a], "obj %.9g" % ob["obj"][a], "nodes", ob["nodes"][a], "flags", hex(int(tb["flags"][a])),
                      "| oracle st", o["status"][i], "obj %.9g" % o["obj"][i], "| traj diff A-B %.3g A-oracle %.3g B-oracle %.3g" % (d[a], np.abs(oa["traj"][a] - o["traj"][i]).max(), np.abs(ob["traj"][a] - o["traj"][i]).max()))
            break
    else:
        print("try", t, "no difference in 40 rounds")
    sa.close(), sb.close()
