"""Row f2, a larger parity run than the -m gpu test: the cooperative device decomposition (one wavefront per seed, bit planes,
batched rim moves) and the one-thread-per-seed kernel against the host functions on random seeds of four worlds (two with a
potential field), for several n_it. usage: python scripts/gpu_decomp_fuzz.py [seeds per world and n_it = 2000] > profiles/r05_f2_parity.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import decomp_cases as dc  # noqa: E402
from multi_agent_pkgs_amd import lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(2026)
total = bad = chamfered = aware = failed_both = 0
t0 = time.time()
for n_it in (30, 42, 54, 66):
    for wname, potential in (("forest", False), ("fwf", False), ("forest", True), ("fwf", True)):
        occ2, origin = dc.world(wname, potential=potential, rng=rng)
        off, seed, ground, variant, org = dc.cases(occ2, origin, n, rng)
        rows, n_rows, rc, cells = lib.poly_octa3d_batch(occ2, dc.LDIM, off, ground, seed, variant, org, n_it=n_it, res=0.3, max_rows=32)
        w_rows, w_n, w_rc, w_cells = lib.poly_octa3d_batch(occ2, dc.LDIM, off, ground, seed, variant, org, n_it=n_it, res=0.3, max_rows=32, wave=True)
        for t in range(n):
            total += 1
            try:
                want, voxels, v = dc.host_answer(occ2, off[t], seed[t], ground[t], variant[t], org[t], n_it=n_it)
            except Exception:  # beyond the fixed workspace on the host: the device forms must say so too
                failed_both += 1
                if rc[t] == 0 or w_rc[t] == 0:
                    bad += 1
                continue
            ok = (rc[t] == 0 and w_rc[t] == 0 and n_rows[t] == len(want) and w_n[t] == len(want) and np.array_equal(rows[t, : n_rows[t]], want)
                  and np.array_equal(w_rows[t, : w_n[t]], want) and cells[t] == voxels and w_cells[t] == voxels)
            bad += 0 if ok else 1
            chamfered += len(want) > 6
            aware += v
print("voxel decomposition on the device against the host functions (rows bit for bit, row counts, voxel counts), %d seeds: n_it 30 / 42 / 54 / 66 x "
      "pillar forest, forest + wall + forest, each with and without a potential field; both device forms (one thread per seed, one wavefront "
      "per seed): %d mismatches; %d polyhedra with chamfers, %d shape-aware, %d beyond the fixed workspace in all three forms; %.0f s"
      % (total, bad, chamfered, aware, failed_both, time.time() - t0))
sys.exit(1 if bad else 0)
