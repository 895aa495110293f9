"""Development aid: one case of scripts/gpu_fuzz.py (same random sequence) on the GPU under the current HDSM_* environment, against the
oracle: statuses, node counts and flags of the instances that differ. usage: python scripts/gpu_fuzz_case.py CASE"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from multi_agent_pkgs_amd import lib  # noqa: E402
from multi_agent_pkgs_amd.params import make_params  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

target = int(sys.argv[1])
rng = np.random.default_rng(12345)
K = ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")
for case in range(target + 1):
    n_hor = int(rng.choice([6, 8, 10, 10, 10, 12, 15]))
    n_rob = int(rng.choice([9, 16, 25, 36, 49, 64]))
    kw = dict(spacing=float(rng.choice([0.8, 1.0, 1.3, 1.8, 2.5])), narrow=bool(rng.random() < 0.35), turn=bool(rng.random() < 0.5),
              chamfer=bool(rng.random() < 0.3), absent_frac=float(rng.choice([0, 0, 0.2])), speed=(0.0, float(rng.choice([3.0, 6.0, 9.0]))))
    rk4 = bool(rng.random() < 0.3)
    drag = tuple(rng.choice([0.0, 0.0, 0.1, 0.3], 3))
    ph = int(rng.choice([2, 3, 4]))
prm = make_params(n_hor=n_hor, rk4=rk4, drag=drag, max_rows_static=18, poly_hor=ph)
sn = problems.swarm_snapshot(prm, n_rob, seed=1000 + target, **kw)
args = [sn[k] for k in K]
o = orc.replan(prm, *args, n_threads=32, search=1)
sol = lib.Solver(prm, n_rob, n_rob)
for rep in range(2):
    g = sol.replan(*args)
    fl = sol.last_sweep_stats(n_rob)["flags"]
    bad = np.where(g["status"] != o["status"])[0]
    print("case", target, dict(n_hor=n_hor, n_rob=n_rob, poly_hor=ph), "rep", rep, {k: os.environ[k] for k in os.environ if k.startswith("HDSM_")},
          "differing", [(int(a), int(g["status"][a]), int(o["status"][a]), int(g["nodes"][a]), hex(int(fl[a]))) for a in bad])
