#!/usr/bin/env python3
"""Measurement of the f1 kernel (k_reference): mean launch time (HIP events on the launch stream) and fraction of the
HBM roofline. Algorithmic bytes per agent (fp64): every other agent's position at all N+1 steps + own plan + path +
outputs = (n_rob - 1)(N + 1) 24 + (N + 1) 24 + pmax 24 + (N + 1) 48 + N 48 + 8."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems  # noqa: E402
from multi_agent_pkgs_amd import lib  # noqa: E402
from multi_agent_pkgs_amd.params import agile_params, agile_ref_config  # noqa: E402

out = []
dev = torch.device("cuda", 0)
for n_rob in (64, 256, 1024, 4096):
    N = 10 if n_rob < 4096 else 15
    prm = agile_params(N, max_rows_static=18)
    rcfg = agile_ref_config()
    sn = problems.swarm_snapshot(prm, n_rob, seed=3, spacing=1.5)
    path = np.zeros((n_rob, 3, 3))
    for k in range(n_rob):
        p0 = sn["state"][k, :3]
        path[k] = [p0, p0 + [3.0, 1.0, 0], p0 + [30.0, 5.0, 0]]
    sol = lib.Solver(prm, n_rob, n_rob, device=0)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
    d_id, d_path, d_np = t(sn["agent_id"], np.int32), t(path, np.float64), t(np.full(n_rob, 3), np.int32)
    d_plans, d_has = t(sn["plans"], np.float64), t(sn["has_plan"], np.uint8)
    d_full = torch.zeros((n_rob, N + 1, 6), dtype=torch.float64, device=dev)
    d_ref = torch.zeros((n_rob, N, 6), dtype=torch.float64, device=dev)
    d_pv = torch.zeros(n_rob, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream()
    run = lambda: sol.reference_device(rcfg, d_id, d_path, d_np, d_plans, d_has, d_full, d_ref, d_pv, stream=st)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record(st)
        run()
        b.record(st)
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    B = (n_rob - 1) * (N + 1) * 24 + (N + 1) * 24 + 3 * 24 + (N + 1) * 48 + N * 48 + 8
    ach = B * n_rob / (ms * 1e-3) / 1e9
    out.append({"kernel": "k_reference", "agents": n_rob, "horizon": N, "ms_per_launch": ms,
                "agents_per_s": n_rob / (ms * 1e-3), "algorithmic_bytes_per_agent": B,
                "roofline": {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0}})
    print(json.dumps(out[-1]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "f1_reference_bench.json"), "w"), indent=1)
