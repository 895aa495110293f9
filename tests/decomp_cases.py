"""Random voxel-decomposition cases (row f2) for the device tests and the CPU execution of the cooperative device source: local
grids cut out of an inflated pillar forest / forest + wall + forest (ground below, world border inside the window, narrowed
free space so that pinched seeds exist, optionally a potential field of values 1..99 around the obstacles), and the host
functions' answer for one case."""
import numpy as np

from multi_agent_pkgs_amd import scenarios as sc
from multi_agent_pkgs_amd import swarm

LDIM = (66, 66, 20)


def world(wname, potential=False, rng=None):
    raw, origin = sc.forest_for_circle(64, seed=3) if wname == "forest" else sc.forest_wall_forest(1, 1, seed=2)
    occ = sc.inflate(raw)
    occ2 = occ.copy()  # narrow the free space on half of the world: a second inflation
    occ2[:, :, : occ.shape[2] // 2] = sc.inflate(occ)[:, :, : occ.shape[2] // 2]
    if potential:  # free for the growth, "not empty" for the chamfer test of the shape-aware variant (CD:577-588)
        near = sc.inflate(occ2)
        occ2[(near >= 100) & (occ2 < 100) & (rng.random(occ2.shape) < 0.7)] = 40
    return occ2, origin


def cases(occ2, origin, n, rng):
    wz, wy, wx = occ2.shape
    off, seed, ground, variant, org = [], [], [], [], []
    while len(off) < n:
        o = np.array([rng.integers(-20, wx - 40), rng.integers(-20, wy - 40), rng.integers(-6, max(1, wz - 18))])
        s = np.array([rng.integers(1, 65), rng.integers(1, 65), rng.integers(1, 19)])
        gk = int(rng.integers(0, 8))
        gcell = o + s
        inside = (gcell >= 0).all() and gcell[0] < wx and gcell[1] < wy and gcell[2] < wz
        val = occ2[gcell[2], gcell[1], gcell[0]] if inside else 0
        if val >= 100 or s[2] < gk:
            continue
        off.append(o), seed.append(s), ground.append(gk), variant.append(int(rng.choice([-1, -1, 0, 1]))), org.append(origin + o * 0.3)
    return (np.array(off, np.int32), np.array(seed, np.int32), np.array(ground, np.int32), np.array(variant, np.int32), np.array(org))


def host_answer(occ2, off, seed, ground, variant, org, n_it=42):
    """hdsm_poly_octa3d / hdsm_poly_octa3d_new on the local grid of one case: (rows, voxels of the polyhedron)."""
    wz, wy, wx = occ2.shape
    loc = np.zeros((20, 66, 66), np.int8)
    x0, y0, z0 = off
    xs, ys, zs = max(0, -x0), max(0, -y0), max(0, -z0)
    xe, ye, ze = min(66, wx - x0), min(66, wy - y0), min(20, wz - z0)
    if xe > xs and ye > ys and ze > zs:
        loc[zs:ze, ys:ye, xs:xe] = occ2[z0 + zs:z0 + ze, y0 + ys:y0 + ye, x0 + xs:x0 + xe]
    loc[loc < 0] = 100
    loc[:ground] = 100
    s, v = seed, int(variant)
    if v < 0:  # AC:1385-1395
        o = lambda i, j, k: 0 <= i < 66 and 0 <= j < 66 and 0 <= k < 20 and loc[k, j, i] == 100
        v = int((o(s[0] - 1, s[1], s[2]) and o(s[0] + 1, s[1], s[2])) or (o(s[0], s[1] - 1, s[2]) and o(s[0], s[1] + 1, s[2]))
                or (o(s[0], s[1], s[2] - 1) and o(s[0], s[1], s[2] + 1)))
    want, gm = swarm.poly_octa3d(loc, s, n_it=n_it, res=0.3, mark=-1, origin=org, max_rows=32, shape_aware=bool(v))
    return want, int(np.count_nonzero(gm == -1)), v
