"""Harness of the f2 oracle (oracle/hdsm_oracle.c: orc_safe_corridor, a statement-by-statement restatement of
Agent::GenerateSafeCorridor, agent_class.cpp:1236-1447). TEST INFRASTRUCTURE.

What the oracle takes are the reference's own members before the call — poly_const_vec_, poly_seeds_, poly_used_idx_, traj_curr_,
path_curr_ with the current position in front, the agent's voxel grid — so this file only moves DATA: it reads the planner state
of an agent out of the product (hdsm_swarm_export_state: the plain struct hdsm_sw::AgentS of csrc/swarm_core.h, mirrored below
field by field), cuts the agent's local voxel grid out of the world the way the environment builder hands it to the planner
(environment_builder.cpp:58-67: origin = floor((position - range / 2) / voxel) * voxel, below the ground and outside knowledge =
unknown), and forms path_curr_ the way the path thread leaves it (the part of the global path ahead of the point the last
reference started from, AC:328-350, AC:1480-1495). The voxel decomposition the reference calls for a new seed is handed in as a
callback to the product's HOST functions hdsm_poly_octa3d / hdsm_poly_octa3d_new (pinned against recorded reference outputs in
tests/test_host.py); the corridor maintenance around it is what is being checked."""
import ctypes as C

import numpy as np

from multi_agent_pkgs_amd import lib as hdsm
from multi_agent_pkgs_amd.params import HdsmParams
from multi_agent_pkgs_amd.swarm import SwarmConfig

MAXH, MAXP, RSMAX, PATH_PTS = 16, 8, 32, 48   # hdsm.h: HDSM_MAX_HOR, HDSM_MAX_POLY, HDSM_MAX_ROWS_STATIC; swarm_core.h: PATH_PTS


class V3(C.Structure):
    _fields_ = [("v", C.c_double * 3)]


class Poly(C.Structure):
    _fields_ = [("rows", C.c_int32), ("pad", C.c_int32), ("A", (C.c_double * 3) * RSMAX), ("b", C.c_double * RSMAX), ("seed", V3)]


class AgentS(C.Structure):
    _fields_ = [("id", C.c_int32), ("n_path", C.c_int32), ("start", V3), ("goal", V3), ("path", V3 * PATH_PTS),
                ("state_curr", C.c_double * 9), ("has_traj", C.c_int32), ("n_ref", C.c_int32),
                ("traj_curr", (C.c_double * 9) * (MAXH + 1)), ("ctrl_curr", (C.c_double * 3) * MAXH),
                ("traj_ref", (C.c_double * 6) * (MAXH + 1)), ("n_poly", C.c_int32), ("increment", C.c_int32),
                ("polys", Poly * MAXP), ("poly_used", C.c_uint8 * MAXP), ("external_ref", C.c_int32), ("n_fail", C.c_int32),
                ("corridor_rc", C.c_int32), ("pad", C.c_int32), ("path_vel", C.c_double)]


def export_agents(shard):
    """The agent states of a SwarmShard (host mirror), plus its world."""
    L = hdsm.load()
    n = shard.n_local
    buf = (AgentS * max(n, 1))()
    n_local, n_rob, first = C.c_int32(), C.c_int32(), C.c_int32()
    prm, cfg = HdsmParams(), SwarmConfig()
    world = C.POINTER(C.c_int8)()
    wdim, worigin = (C.c_int32 * 3)(), (C.c_double * 3)()
    rc = L.hdsm_swarm_export_state(shard.h, buf, C.byref(n_local), C.byref(n_rob), C.byref(first), C.byref(prm), C.byref(cfg),
                                   C.byref(world), wdim, worigin)
    assert rc == 0 and n_local.value == n
    w = None
    if world:
        w = np.ctypeslib.as_array(world, shape=(wdim[2], wdim[1], wdim[0])).copy()
    return buf, prm, cfg, w, np.array(worigin[:])


def _on_segment(p, a, b):   # IsOnSegment, AC:1864-1884
    d1, d2, d12 = np.linalg.norm(p - a), np.linalg.norm(p - b), np.linalg.norm(a - b)
    return abs(d1 + d2 - d12) < 1e-6 and float(np.dot(p - a, p - b)) <= 0


def path_curr(ag):
    """path_curr_ with the current position pushed in front (AC:1286-1290)."""
    pts = np.array([[ag.path[i].v[k] for k in range(3)] for i in range(ag.n_path)])
    head = pts[0] if ag.n_ref == 0 else np.array([ag.traj_ref[0][k] for k in range(3)])
    start = 0
    for i in range(ag.n_path - 1):
        if _on_segment(head, pts[i], pts[i + 1]):
            start = i + 1
            break
    return np.vstack([[ag.state_curr[k] for k in range(3)], head, pts[start:]])


def local_grid(cfg, world, worigin, pos):
    """The agent's voxel grid as the environment builder cuts it (raw values: -1 unknown — also everything below the ground —,
    outside the world free) -> data [nz][ny][nx] int8, dim (x, y, z), origin."""
    vs = cfg.voxel_size
    dim = [int(np.floor(cfg.grid_range[k] / vs)) for k in range(3)]
    origin = np.array([np.floor((pos[k] - cfg.grid_range[k] / 2) / vs) * vs for k in range(3)])
    off = [int(round((origin[k] - worigin[k]) / vs)) for k in range(3)]
    ground_k = int(np.ceil((cfg.grid_z_min - origin[2]) / vs - 1e-9))
    data = np.zeros((dim[2], dim[1], dim[0]), np.int8)
    wz, wy, wx = world.shape
    lo = [max(0, -off[k]) for k in range(3)]
    hi = [min(dim[k], (wx, wy, wz)[k] - off[k]) for k in range(3)]
    if all(hi[k] > lo[k] for k in range(3)):
        data[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]] = world[lo[2] + off[2]:hi[2] + off[2], lo[1] + off[1]:hi[1] + off[1], lo[0] + off[0]:hi[0] + off[0]]
    data[data < 0] = -1
    if ground_k > 0:
        data[:min(ground_k, dim[2])] = -1
    return data, np.array(dim, np.int32), origin


DECOMP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int8), C.POINTER(C.c_int32), C.c_int32, C.c_double,
                        C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_int32))


def _decomp(ctx, seed, grid, dim, n_it, vs, mark, origin, use_new, rows, max_rows, n_rows):
    L = hdsm.load()
    fn = L.hdsm_poly_octa3d_new if use_new else L.hdsm_poly_octa3d
    if any(seed[k] < 0 or seed[k] >= dim[k] for k in range(3)):
        return -1   # (a seed outside the grid: the reference would index out of bounds; the product stops the agent's corridor)
    return fn(seed, grid, dim, C.c_int32(n_it), C.c_double(vs), C.c_int32(mark), origin, rows, C.c_int32(max_rows), n_rows)


_decomp_cb = DECOMP_FN(_decomp)


def oracle_corridor(orc_lib, prm, cfg, world, worigin, ag):
    """orc_safe_corridor on the state `ag` (before the corridor step) -> (rc, [(rows, A[r][3], b[r], seed[3]), ...])."""
    P, RS = prm.poly_hor, prm.max_rows_static
    n_prev = ag.n_poly
    prev_rows = np.array([ag.polys[i].rows for i in range(n_prev)], np.int32)
    prev_A = np.zeros((max(n_prev, 1), RS, 3))
    prev_b = np.zeros((max(n_prev, 1), RS))
    prev_seed = np.zeros((max(n_prev, 1), 3))
    for i in range(n_prev):
        r = ag.polys[i].rows
        prev_A[i, :r] = np.array([[ag.polys[i].A[q][k] for k in range(3)] for q in range(r)]).reshape(r, 3)
        prev_b[i, :r] = [ag.polys[i].b[q] for q in range(r)]
        prev_seed[i] = [ag.polys[i].seed.v[k] for k in range(3)]
    used = np.array([ag.poly_used[i] for i in range(max(n_prev, 1))], np.uint8)
    N = prm.n_hor
    traj = np.array([[ag.traj_curr[j][k] for k in range(3)] for j in range(N + 1)]) if ag.has_traj else np.zeros((0, 3))
    path = np.ascontiguousarray(path_curr(ag))
    grid, dim, origin = local_grid(cfg, world, worigin, [ag.state_curr[k] for k in range(3)])
    n_out = C.c_int32()
    out_rows = np.zeros(P + 1, np.int32)
    out_A, out_b, out_seed = np.zeros((P + 1, RS, 3)), np.zeros((P + 1, RS)), np.zeros((P + 1, 3))
    d, i32 = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    fn = orc_lib.orc_safe_corridor
    fn.restype = C.c_int
    fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, i32, d, d, d, C.POINTER(C.c_uint8), C.c_int32, d, C.c_int32, d,
                   C.POINTER(C.c_int8), i32, d, C.c_double, DECOMP_FN, C.c_void_p, i32, i32, d, d, d]
    rc = fn(P, cfg.n_it_decomp, cfg.use_cvx_new, RS, n_prev, prev_rows.ctypes.data_as(i32), prev_A.ctypes.data_as(d),
            prev_b.ctypes.data_as(d), prev_seed.ctypes.data_as(d), used.ctypes.data_as(C.POINTER(C.c_uint8)), len(traj),
            np.ascontiguousarray(traj).ctypes.data_as(d), len(path), path.ctypes.data_as(d), grid.ctypes.data_as(C.POINTER(C.c_int8)),
            dim.ctypes.data_as(i32), origin.ctypes.data_as(d), cfg.voxel_size, _decomp_cb, None, C.byref(n_out),
            out_rows.ctypes.data_as(i32), out_A.ctypes.data_as(d), out_b.ctypes.data_as(d), out_seed.ctypes.data_as(d))
    return rc, [(int(out_rows[i]), out_A[i, :out_rows[i]].copy(), out_b[i, :out_rows[i]].copy(), out_seed[i].copy()) for i in range(n_out.value)]


def product_corridor(ag):
    """The corridor an agent state holds: [(rows, A, b, seed), ...]."""
    out = []
    for i in range(ag.n_poly):
        r = ag.polys[i].rows
        out.append((r, np.array([[ag.polys[i].A[q][k] for k in range(3)] for q in range(r)]).reshape(r, 3),
                    np.array([ag.polys[i].b[q] for q in range(r)]), np.array([ag.polys[i].seed.v[k] for k in range(3)])))
    return out


def same_corridor(a, b):
    """Bit for bit: counts, rows in order, seeds."""
    if len(a) != len(b):
        return False
    for (ra, Aa, ba, sa), (rb, Ab, bb, sb) in zip(a, b):
        if ra != rb or not (np.array_equal(Aa, Ab) and np.array_equal(ba, bb) and np.array_equal(sa, sb)):
            return False
    return True
