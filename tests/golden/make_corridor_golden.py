"""Mint tests/golden/corridor_cases.npz: inputs and outputs of the reference's convex voxel decomposition.

How the expected rows were produced (this container only): convex_decomp_util/src/convex_decomp.cpp of the reference
was compiled where it lies, unmodified, into a scratch library OUTSIDE the repository, against a ~30-line stand-in for
the three Eigen / decomp_util headers it includes (it only needs fixed-size vectors with dot/+/-), plus a 15-line
extern "C" driver that calls convex_decomp_lib::GetPolyOcta3D / GetPolyOcta3DNew and flattens Polyhedron3D into rows
(n, n.p). Eigen is not installed here, so by the rules of this build a reference compiled against stand-in headers is
NOT an oracle/_ref and these vectors do not formally pin parity; they are the outputs that build produced, kept as
data. On 2 x 4059 random cases (five kinds of cluttered 66x66x20 grids, seeds anywhere, n_it in {6,12,42,60}) the
product code (csrc/corridor_host.cpp) reproduced rows AND marked grids bit for bit; 48 of those cases are kept here.

usage: REF_CD_SO=/path/to/libref_cd.so python tests/golden/make_corridor_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def world(rng, kind, dims=(66, 66, 20)):
    """Five kinds of cluttered local grids (int8 [nz][ny][nx]: 0 free, 40 potential field, 100 occupied)."""
    nx, ny, nz = dims
    g = np.zeros((nz, ny, nx), np.int8)
    if kind == 0:    # inflated pillar forest
        for _ in range(rng.integers(5, 60)):
            x, y = rng.integers(0, nx), rng.integers(0, ny)
            r = rng.integers(0, 3)
            g[:, max(0, y - r):y + r + 1, max(0, x - r):x + r + 1] = 100
    elif kind == 1:  # random boxes
        for _ in range(rng.integers(3, 30)):
            x, y, z = rng.integers(0, nx), rng.integers(0, ny), rng.integers(0, nz)
            a, b, c = rng.integers(1, 8, 3)
            g[z:z + c, y:y + b, x:x + a] = 100
    elif kind == 2:  # ground + wall with gaps
        g[:2] = 100
        x = rng.integers(10, nx - 10)
        g[:, :, x] = 100
        for _ in range(rng.integers(1, 5)):
            y, z = rng.integers(0, ny - 6), rng.integers(2, nz - 5)
            g[z:z + rng.integers(2, 6), y:y + rng.integers(2, 8), x] = 0
    elif kind == 3:  # diagonal rows of columns
        for _ in range(rng.integers(1, 6)):
            x, y = rng.integers(0, nx), rng.integers(0, ny)
            dx, dy = rng.integers(-2, 3), rng.integers(-2, 3)
            for t in range(rng.integers(4, 25)):
                xx, yy = x + t * dx, y + t * dy
                if 0 <= xx < nx and 0 <= yy < ny:
                    g[:, yy, xx] = 100
    else:            # forest with a potential-field halo
        for _ in range(rng.integers(5, 40)):
            x, y = rng.integers(0, nx), rng.integers(0, ny)
            sl = (slice(None), slice(max(0, y - 2), y + 3), slice(max(0, x - 2), x + 3))
            g[sl] = np.maximum(g[sl], 40)
            g[:, max(0, y - 1):y + 2, max(0, x - 1):x + 2] = 100
    return g


def main():
    ref = C.CDLL(os.environ["REF_CD_SO"])
    grids, seeds, nits, exp = [], [], [], {0: [], 1: []}
    marked = {0: [], 1: []}
    t = 0
    while len(grids) < 48:
        rng = np.random.default_rng(5000 + t)
        g = world(rng, t % 5)
        t += 1
        free = np.argwhere(g < 100)
        c = free[rng.integers(len(free))]
        seed = np.array([c[2], c[1], c[0]], np.int32)
        if seed.min() < 1 or seed[0] > 64 or seed[1] > 64 or seed[2] > 18:
            continue
        n_it = int(rng.choice([12, 42, 60]))
        org = np.array([-3.0, 1.5, -0.6])
        dim = np.array([66, 66, 20], np.int32)
        for which in (0, 1):
            gg = g.copy()
            rows = np.zeros((40, 4))
            n = ref.ref_poly(which, seed.ctypes.data_as(C.POINTER(C.c_int)), gg.ctypes.data_as(C.POINTER(C.c_byte)),
                             dim.ctypes.data_as(C.POINTER(C.c_int)), n_it, C.c_double(0.3), -3,
                             org.ctypes.data_as(C.POINTER(C.c_double)), rows.ctypes.data_as(C.POINTER(C.c_double)), 40)
            rows[n:] = np.nan
            exp[which].append(rows[:24].copy())
            marked[which].append(np.count_nonzero(gg == -3))
        grids.append(g), seeds.append(seed), nits.append(n_it)
    np.savez_compressed(os.path.join(HERE, "corridor_cases.npz"), grids=np.stack(grids), seeds=np.stack(seeds),
                        n_it=np.array(nits, np.int32), origin=np.array([-3.0, 1.5, -0.6]), res=0.3, mark=-3,
                        rows_octa3d=np.stack(exp[0]), rows_octa3d_new=np.stack(exp[1]),
                        cells_octa3d=np.array(marked[0]), cells_octa3d_new=np.array(marked[1]))
    print("wrote corridor_cases.npz:", len(grids), "cases")


if __name__ == "__main__":
    sys.exit(main())
