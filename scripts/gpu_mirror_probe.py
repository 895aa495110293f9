"""Device-resident loop against the host mirror on the forest scene of test_device_resident_loop_follows_the_host_mirror:
first round in which the two flights differ, and which agents.   usage: python scripts/gpu_mirror_probe.py [rounds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from multi_agent_pkgs_amd import lib as hdsm, scenarios as sc, swarm  # noqa: E402
from multi_agent_pkgs_amd.params import agile_params  # noqa: E402
import test_gpu_configs as T  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_rob, N = 48, 10
prm = agile_params(N, max_rows_static=18)


def make():
    sol, loop = T._device_loop(hdsm, prm, swarm.default_swarm_config(), n_rob)
    raw, origin = sc.forest_for_circle(n_rob, seed=21)
    assert loop.set_world(sc.inflate(raw), origin) == 0
    return sol, loop


sol_h, host = make()
sol_d, dev_loop = make()
dsw = swarm.DeviceSwarm(dev_loop.shard, sol_d)
for r in range(rounds):
    out = host.step()
    dsw.round()
    plans, has, status, failed = dsw.download(states=False)
    bad = np.nonzero((has != host.has_plan) | (status != out["status"]))[0]
    dmax = float(np.abs(plans - host.plans_all).max())
    print("round", r, "status host", np.bincount(out["status"], minlength=3).tolist(), "device", np.bincount(status, minlength=3).tolist(), "max |plans diff| %.2e" % dmax,
          "differ:", [(int(a), int(out["status"][a]), int(status[a])) for a in bad][:8])
    if len(bad) or dmax > 1e-7:
        break
