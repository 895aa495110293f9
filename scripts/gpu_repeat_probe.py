"""Solve ONE recorded round of the forest scene (test_device_resident_loop_follows_the_host_mirror) many times with fresh handles:
a deterministic solver gives the same statuses every time.   usage: python scripts/gpu_repeat_probe.py [round] [repeats]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from multi_agent_pkgs_amd import lib as hdsm, scenarios as sc, swarm  # noqa: E402
from multi_agent_pkgs_amd.params import agile_params  # noqa: E402
import test_gpu_configs as T  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n_rob, N = 48, 10
prm = agile_params(N, max_rows_static=18)
os.environ["HDSM_SCANNER"] = "0"
sol, loop = T._device_loop(hdsm, prm, swarm.default_swarm_config(), n_rob)
raw, origin = sc.forest_for_circle(n_rob, seed=21)
assert loop.set_world(sc.inflate(raw), origin) == 0
recs = []
for r in range(rnd + 1):
    rec = []
    out = loop.step(record=rec)
    recs.append(rec[0])
KEYS = ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")
args = [recs[-1][k] for k in KEYS]
o = orc.replan(prm, *args, n_threads=16)
print("oracle status", np.bincount(o["status"], minlength=3).tolist())
for mode in ("0", "1", "2"):
    os.environ["HDSM_SCANNER"] = mode
    bad = 0
    for rep in range(reps):
        s2 = hdsm.Solver(prm, n_rob, n_rob)
        for rr in recs[:-1]:   # the rounds before: the handle's warm-start store is the one of the flight
            s2.replan(*[rr[k] for k in KEYS])
        g = s2.replan(*args)
        st = s2.last_stats(n_rob)
        d = np.nonzero(g["status"] != o["status"])[0]
        if len(d):
            bad += 1
            if bad <= 3:
                print("  scanner", mode, "rep", rep, "differs:", [(int(a), int(o["status"][a]), int(g["status"][a]), int(st["qp_iters"][a]), int(st["nodes"][a])) for a in d],
                      "oracle iters/nodes", [(int(o["qp_iters"][a]), int(o["nodes"][a])) for a in d])
        else:
            ok = o["status"] != 2
            dt = float(np.abs(g["traj"] - o["traj"])[ok].max())
            if dt > 1e-6:
                print("  scanner", mode, "rep", rep, "traj diff", dt)
    print("scanner", mode, ":", bad, "of", reps, "repeats differ from the oracle")
