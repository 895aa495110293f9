"""Phase profile of the cooperative voxel decomposition (-DCD_PROFILE build of corridor_kernels.hip, made by
scripts/gpu_corridor_profile.sh): cycles per phase, summed over the seeds of a batch, from lane 0's cycle counter.
usage: python scripts/corridor_profile.py <libcorr_prof.so> [n_seeds]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_pkgs_amd import scenarios as sc  # noqa: E402

L = C.CDLL(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
raw, origin = sc.forest_for_circle(256, seed=13)
occ = np.ascontiguousarray(sc.inflate(raw), dtype=np.int8)
wz, wy, wx = occ.shape
rng = np.random.default_rng(0)
ldim = np.array((66, 66, 20), np.int32)
off, seed = [], []
while len(off) < n:
    o = np.array([rng.integers(0, max(1, wx - 66)), rng.integers(0, max(1, wy - 66)), rng.integers(0, max(1, wz - 20))])
    s = np.array([rng.integers(20, 46), rng.integers(20, 46), rng.integers(5, 15)])
    g = o + s
    if (g < np.array([wx, wy, wz])).all() and occ[g[2], g[1], g[0]] < 100:
        off.append(o), seed.append(s)
off, seed = np.array(off, np.int32), np.array(seed, np.int32)
org = np.ascontiguousarray(origin + off * 0.3)
zero, var = np.zeros(n, np.int32), np.full(n, -1, np.int32)
wdim = np.array(occ.shape[::-1], np.int32)
rows = np.zeros((n, 32, 4))
n_rows, rc, cells = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
p = lambda a, t: a.ctypes.data_as(C.POINTER(t))


def run(k):
    t0 = time.perf_counter()
    r = L.hdsm_poly_octa3d_batch_wave(C.c_int32(0), C.c_int32(k), p(occ, C.c_int8), p(wdim, C.c_int32), p(ldim, C.c_int32), p(off, C.c_int32),
                                      p(zero, C.c_int32), p(seed, C.c_int32), p(var, C.c_int32), p(org, C.c_double), C.c_int32(42), C.c_double(0.3),
                                      p(rows, C.c_double), C.c_int32(32), p(n_rows, C.c_int32), p(rc, C.c_int32), p(cells, C.c_int32))
    assert r == 0, r
    return time.perf_counter() - t0


run(8)
out = (C.c_ulonglong * 16)()
L.hdsm_corridor_profile(out)
dt = run(n)
L.hdsm_corridor_profile(out)
names = ["seed search", "plane rows", "move loop", "rim / far write-out", "allowance", "edge state machine", "shape-aware side tests", "trial layers",
         "accept (copy, append, mark)", "rows", "world maps", "layer (0..3 inside)", "", "", "", "decompositions"]
v = [int(x) for x in out]
tot = sum(v[i] for i in (4, 5, 6, 7, 8, 9, 10, 11))
print(json.dumps({"seeds": n, "batch_s_incl_upload": dt, "rows_mean": float(n_rows[:n].mean()), "cells_mean": float(cells[:n].mean()), "failed": int((rc[:n] != 0).sum()),
                  "cycles_per_decomposition": {names[i]: v[i] / max(1, v[15]) for i in range(12)}, "cycles_per_decomposition_total": tot / max(1, v[15]),
                  "decompositions": v[15]}, indent=1))
