"""Would a portfolio of solver strategies on idle CUs cut the round time? Per recorded bench round, per instance:
active-set operation counts under several strategies; round cost ~ max over instances. Prints, per strategy, the mean
over rounds of that max, and the same for the per-instance minimum over strategies (the portfolio bound)."""
import os, sys, json, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run(tag, env, warm):
    code = f'''
import numpy as np, sys
sys.path.insert(0, {ROOT!r})
from multi_agent_pkgs_amd import lib
from multi_agent_pkgs_amd.params import agile_params
z = np.load({ROOT!r} + "/gpurun_out/rounds_cache.npz")
prm = agile_params(10, max_rows_static=18); prm.warm_start = {int(warm)}
n = int(z["n_rob"]); sol = lib.Solver(prm, n, n)
keys = ("agent_id","state","ref","n_poly","n_rows","A","b","plans","has_plan")
its = []
for r in range(z["state"].shape[0]):
    sol.replan(*[z[k][r] for k in keys]); its.append(sol.last_stats(n)["qp_iters"].copy())
np.save({ROOT!r} + "/gpurun_out/portfolio_{tag}.npy", np.array(its))
'''
    e = dict(os.environ); e.update(env)
    subprocess.check_call([sys.executable, "-c", code], env=e)
    return np.load(f"{ROOT}/gpurun_out/portfolio_{tag}.npy")

subprocess.check_call([sys.executable, "bench.py", "--no-cpu-baseline", "--cache", f"{ROOT}/gpurun_out/rounds_cache.npz"], cwd=ROOT, stdout=subprocess.DEVNULL)
S = {"warm": ({"HDSM_PRESWEEP": "0"}, True), "warm+presweep": ({"HDSM_PRESWEEP": "1"}, True),
     "cold": ({"HDSM_PRESWEEP": "0"}, False), "cold+presweep": ({"HDSM_PRESWEEP": "1"}, False),
     "warm tau0.2": ({"HDSM_PRESWEEP": "0", "HDSM_CAND_TAU": "0.2"}, True)}
res = {k: run(k.replace(" ", "_").replace("+", "_"), *v) for k, v in S.items()}
for k, v in res.items():
    print(f"{k:16s} mean over rounds of max ops {v[10:].max(axis=1).mean():6.1f}   mean ops {v[10:].mean():5.1f}")
allv = np.stack(list(res.values()))
print(f"{'portfolio(min)':16s} mean over rounds of max ops {allv.min(axis=0)[10:].max(axis=1).mean():6.1f}")
two = np.stack([res["warm"], res["warm+presweep"]]).min(axis=0)
print(f"{'warm|warm+pre':16s} mean over rounds of max ops {two[10:].max(axis=1).mean():6.1f}")
