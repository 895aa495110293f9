"""Row f2 on the device vs the host: a batch of voxel decompositions (one per agent and new polyhedron) in the forest + wall +
forest world. usage: python scripts/bench_corridor.py [n_seeds = 16384] > gpurun_out/r02/f2_corridor.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_pkgs_amd import lib, swarm, scenarios as sc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
raw, origin = sc.forest_wall_forest(2, 2, seed=0)
occ = sc.inflate(raw)
wz, wy, wx = occ.shape
rng = np.random.default_rng(0)
ldim = (66, 66, 20)
off, seed = [], []
while len(off) < n:
    o = np.array([rng.integers(0, wx - 66), rng.integers(0, wy - 66), rng.integers(0, wz - 20)])
    s = np.array([rng.integers(20, 46), rng.integers(20, 46), rng.integers(5, 15)])
    g = o + s
    if occ[g[2], g[1], g[0]] < 100:
        off.append(o), seed.append(s)
off, seed = np.array(off, np.int32), np.array(seed, np.int32)
org = origin + off * 0.3
zero = np.zeros(n, np.int32)
var = np.full(n, -1, np.int32)
lib.poly_octa3d_batch(occ, ldim, off[:64], zero[:64], seed[:64], var[:64], org[:64])  # warm-up (module load)
t0 = time.perf_counter()
rows, n_rows, rc, cells = lib.poly_octa3d_batch(occ, ldim, off, zero, seed, var, org)
t_dev = time.perf_counter() - t0
# the two device forms on small batches (what one round of a swarm asks for): latency matters there, not seeds in flight
small = {}
lib.poly_octa3d_batch(occ, ldim, off[:64], zero[:64], seed[:64], var[:64], org[:64], wave=True)
for k in (1, 64, 256, 1024, 4096):
    if k > n:
        continue
    rec = {}
    for name, wave in (("thread_per_seed", False), ("wavefront_per_seed", True)):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            lib.poly_octa3d_batch(occ, ldim, off[:k], zero[:k], seed[:k], var[:k], org[:k], wave=wave)
            ts.append(time.perf_counter() - t0)
        rec[name + "_ms"] = min(ts) * 1e3
    small[str(k)] = rec
m = min(n, 512)
t0 = time.perf_counter()
for t in range(m):
    x0, y0, z0 = off[t]
    loc = occ[z0:z0 + 20, y0:y0 + 66, x0:x0 + 66].copy()
    swarm.poly_octa3d(loc, seed[t], n_it=42, res=0.3, mark=-1, origin=org[t], max_rows=32)
t_host = (time.perf_counter() - t0) / m
print(json.dumps({"what": "hdsm_poly_octa3d_batch (one thread per seed; includes H2D of the world grid, allocation, D2H) vs the host function, "
                          "one core, per call (includes the Python binding)",
                  "world_voxels": int(occ.size), "seeds": n, "device_batch_s": t_dev, "device_us_per_seed": t_dev / n * 1e6,
                  "host_us_per_seed_one_core": t_host * 1e6, "rows_mean": float(n_rows.mean()), "failed": int((rc != 0).sum()),
                  "cells_mean": float(cells.mean()),
                  "small_batches_incl_world_upload_ms": small}))
