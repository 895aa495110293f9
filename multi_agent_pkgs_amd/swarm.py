"""Closed-loop driver: shards of agents replanning in lock-step, one all-gather of the new plans per round.

Mirrors the sequencing of Agent::TrajPlanningIteration (agent_class.cpp:157-258) for a batch:

    prepare (host: corridor + reference)  ->  solve (device: planes + MIQP)  ->  commit (host: fallback,
    increment check, state advance)  ->  all-gather of the published plans (replaces the DDS all-to-all of
    agent_class.cpp:610-677)  ->  next round

The host-side steps live in libhdsm.so (csrc/swarm_host.cpp, include/hdsm_swarm.h). The solver is pluggable so
that the multi-process CPU tests can drive the same loop with the oracle standing in for the device.
"""
import ctypes as C

import numpy as np

from . import lib as _lib
from .params import HdsmParams


class SwarmConfig(C.Structure):
    _fields_ = [
        ("path_vel_min", C.c_double), ("path_vel_max", C.c_double), ("sens_dist", C.c_double),
        ("sens_pot", C.c_double), ("sens_other_agents", C.c_double), ("path_vel_dec", C.c_double),
        ("thresh_dist", C.c_double), ("voxel_size", C.c_double), ("grid_range", C.c_double * 3),
        ("grid_z_min", C.c_double), ("n_it_decomp", C.c_int32), ("step_plan", C.c_int32),
        ("use_cvx_new", C.c_int32), ("reserved0", C.c_int32),
    ]


def default_swarm_config():
    cfg = SwarmConfig()
    _lib.load().hdsm_swarm_default_config(C.byref(cfg))
    return cfg


from .scenarios import circle_scenario, lattice_scenario  # noqa: E402,F401  (scenario geometry lives in scenarios.py)


def lane_forest_scenario(n_y, n_z=1, pitch=2.01, length=96.01, y0=5.0, z0=1.5, voxel=0.3, seed=0,
                         density=0.1, inflate=0.3, pillar_radius=0.05, jitter=0.3):
    """A forest the straight paths of a line formation are collision-free in (f3, the path planner, is not built).

    Agents: the line formation of multi_agent_planner_long.launch.py:36-42 generalised to a y-z lattice (SURVEY.md
    section 8d, cfg 5): start = (0, y0 + pitch i, z0 + pitch j), goal = start + (length, 0, 0). Obstacles: full-height
    pillars of env_long_config.yaml's kind (radius 0.05 m, inflated by 0.3 m as the map builder does) in the two
    forest bands x in [3, 33] and [63, 93], `density` pillars per m^2 — but placed within `jitter` of the mid-lines
    BETWEEN the lanes, so that every lane centre keeps >= 0.2 m to the nearest occupied voxel (the shipped forest is
    random in y and relies on JPS to route around it). Own seeded PRNG.
    Returns starts [n][3], goals [n][3], occupancy int8 [nz][ny][nx], origin (3,)."""
    rng = np.random.default_rng(seed)
    starts = np.array([[0.0, y0 + pitch * i, z0 + pitch * j] for j in range(n_z) for i in range(n_y)])
    goals = starts + [length, 0.0, 0.0]
    origin = np.array([-3.0, 0.0, 0.0])
    hi = np.array([length + 6.0, y0 + pitch * n_y + 5.0, z0 + pitch * n_z + 3.0])
    dims = np.ceil((hi - origin) / voxel).astype(int)
    occ = np.zeros((dims[2], dims[1], dims[0]), np.int8)
    mids = y0 + pitch * (np.arange(-1, n_y) + 0.5)
    r = pillar_radius + inflate
    xc = (np.arange(dims[0]) + 0.5) * voxel + origin[0]
    yc = (np.arange(dims[1]) + 0.5) * voxel + origin[1]
    for x_lo, x_hi in ((3.0, 33.0), (63.0, 93.0)):
        for ym0 in mids:
            for px in rng.uniform(x_lo, x_hi, rng.poisson(density * (x_hi - x_lo) * pitch)):
                ym = ym0 + rng.uniform(-jitter, jitter)
                ix = np.nonzero(np.abs(xc - px) <= r)[0]
                iy = np.nonzero(np.abs(yc - ym) <= r)[0]
                for a in ix:
                    for b in iy:
                        if (xc[a] - px) ** 2 + (yc[b] - ym) ** 2 <= r * r:
                            occ[:, b, a] = 100
    return starts, goals, occ, origin


def shard_range(n_rob, rank, world):
    """Contiguous id blocks, n_rob/world per rank."""
    per = (n_rob + world - 1) // world
    first = min(rank * per, n_rob)
    return first, min(per, n_rob - first)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class SwarmShard:
    """Host planner state of agents [first_id, first_id + n_local)."""

    def __init__(self, prm: HdsmParams, cfg: SwarmConfig, n_rob, first_id, starts, goals):
        self.lib = _lib.load()
        self.prm, self.cfg = prm.copy(), cfg
        self.n_rob, self.first_id = int(n_rob), int(first_id)
        starts = np.ascontiguousarray(starts, dtype=np.float64)
        goals = np.ascontiguousarray(goals, dtype=np.float64)
        self.n_local = starts.shape[0]
        self.h = C.c_void_p()
        rc = self.lib.hdsm_swarm_create(C.byref(self.prm), C.byref(self.cfg), self.n_rob, self.first_id,
                                        self.n_local, _p(starts, C.c_double), _p(goals, C.c_double),
                                        C.byref(self.h))
        if rc:
            raise _lib.HdsmError(rc, "hdsm_swarm_create")
        N, P, RS, n = prm.n_hor, prm.poly_hor, prm.max_rows_static, self.n_local
        self.inp = dict(agent_id=np.zeros(n, np.int32), state=np.zeros((n, 9)), ref=np.zeros((n, N, 6)),
                        n_poly=np.zeros(n, np.int32), n_rows=np.zeros((n, P), np.int32),
                        A=np.zeros((n, P, RS, 3)), b=np.zeros((n, P, RS)))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.hdsm_swarm_destroy(self.h)
            self.h = None

    def set_world(self, occupancy, origin=(0.0, 0.0, 0.0)):
        """occupancy int8 [nz][ny][nx] at cfg.voxel_size (>= 100 occupied, already inflated), or None for free space."""
        if occupancy is None:
            rc = self.lib.hdsm_swarm_set_world(self.h, None, _p(np.zeros(3, np.int32), C.c_int32), _p(np.zeros(3), C.c_double))
        else:
            occ = np.ascontiguousarray(occupancy, dtype=np.int8)
            dim = np.asarray(occ.shape[::-1], dtype=np.int32)
            org = np.asarray(origin, dtype=np.float64)
            rc = self.lib.hdsm_swarm_set_world(self.h, occ.ctypes.data_as(C.POINTER(C.c_int8)), _p(dim, C.c_int32),
                                               _p(org, C.c_double))
        if rc:
            raise _lib.HdsmError(rc, "hdsm_swarm_set_world")

    def prepare_corridor(self):
        """GenerateSafeCorridor alone (hdsm_swarm_prepare_corridor): the reference's order when the reference trajectory is
        generated elsewhere — corridor from the previous reference first."""
        rc = self.lib.hdsm_swarm_prepare_corridor(self.h)
        if rc:
            raise _lib.HdsmError(rc, "hdsm_swarm_prepare_corridor")

    def prepare(self, plans_all, has_plan):
        i = self.inp
        d, i32, u8 = C.c_double, C.c_int32, C.c_uint8
        plans_all = np.ascontiguousarray(plans_all, dtype=np.float64)
        has_plan = np.ascontiguousarray(has_plan, dtype=np.uint8)
        rc = self.lib.hdsm_swarm_prepare(self.h, _p(plans_all, d), _p(has_plan, u8), _p(i["agent_id"], i32),
                                         _p(i["state"], d), _p(i["ref"], d), _p(i["n_poly"], i32),
                                         _p(i["n_rows"], i32), _p(i["A"], d), _p(i["b"], d))
        if rc:
            raise _lib.HdsmError(rc, "hdsm_swarm_prepare")
        return i

    def reference_inputs(self, pmax=3):
        """Polyline each agent's reference will be sampled along this round (for hdsm_reference, f1)."""
        path = np.zeros((self.n_local, pmax, 3))
        n_path = np.zeros(self.n_local, np.int32)
        rc = self.lib.hdsm_swarm_reference_inputs_n(self.h, C.c_int32(pmax), _p(path, C.c_double), _p(n_path, C.c_int32))
        if rc:
            raise _lib.HdsmError(rc, "hdsm_swarm_reference_inputs_n")
        return path, n_path

    def vel_cap(self):
        """vel_cap for hdsm_reference: the voxel / potential-field term of ComputePathVelocity on the current world."""
        cap = np.zeros(self.n_local)
        rc = self.lib.hdsm_swarm_vel_cap(self.h, _p(cap, C.c_double))
        if rc:
            raise _lib.HdsmError(rc, "hdsm_swarm_vel_cap")
        return cap

    def route(self):
        """Global paths on the world given to set_world (hdsm_swarm_route); returns the number of agents without a route."""
        nf = C.c_int32(0)
        rc = self.lib.hdsm_swarm_route(self.h, C.byref(nf))
        if rc:
            raise _lib.HdsmError(rc, "hdsm_swarm_route")
        return nf.value

    def set_paths(self, paths, n_path):
        paths = np.ascontiguousarray(paths, dtype=np.float64)
        n_path = np.ascontiguousarray(n_path, dtype=np.int32)
        rc = self.lib.hdsm_swarm_set_paths(self.h, _p(paths, C.c_double), _p(n_path, C.c_int32), C.c_int32(paths.shape[1]))
        if rc:
            raise _lib.HdsmError(rc, "hdsm_swarm_set_paths")

    def get_paths(self, pmax=64):
        paths = np.zeros((self.n_local, pmax, 3))
        n_path = np.zeros(self.n_local, np.int32)
        rc = self.lib.hdsm_swarm_get_paths(self.h, C.c_int32(pmax), _p(paths, C.c_double), _p(n_path, C.c_int32))
        if rc:
            raise _lib.HdsmError(rc, "hdsm_swarm_get_paths")
        return paths, n_path

    def corridor_errors(self):
        codes = np.zeros(self.n_local, np.int32)
        n = self.lib.hdsm_swarm_corridor_errors(self.h, _p(codes, C.c_int32))
        return n, codes

    def set_reference(self, ref_full, path_vel):
        ref_full = np.ascontiguousarray(ref_full, dtype=np.float64)
        path_vel = np.ascontiguousarray(path_vel, dtype=np.float64)
        rc = self.lib.hdsm_swarm_set_reference(self.h, _p(ref_full, C.c_double), _p(path_vel, C.c_double))
        if rc:
            raise _lib.HdsmError(rc, "hdsm_swarm_set_reference")

    def commit(self, out):
        N = self.prm.n_hor
        plans_local = np.zeros((self.n_local, N + 1, 9))
        has_local = np.zeros(self.n_local, np.uint8)
        d, i32, u8 = C.c_double, C.c_int32, C.c_uint8
        traj = np.ascontiguousarray(out["traj"], dtype=np.float64)
        ctrl = np.ascontiguousarray(out["ctrl"], dtype=np.float64)
        used = np.ascontiguousarray(out["used"], dtype=np.uint8)
        status = np.ascontiguousarray(out["status"], dtype=np.int32)
        rc = self.lib.hdsm_swarm_commit(self.h, _p(traj, d), _p(ctrl, d), _p(used, u8), _p(status, i32),
                                        _p(plans_local, d), _p(has_local, u8))
        if rc:
            raise _lib.HdsmError(rc, "hdsm_swarm_commit")
        return plans_local, has_local

    def state(self):
        pos = np.zeros((self.n_local, 3))
        dist = np.zeros(self.n_local)
        nfail = np.zeros(self.n_local, np.int32)
        self.lib.hdsm_swarm_state(self.h, _p(pos, C.c_double), _p(dist, C.c_double), _p(nfail, C.c_int32))
        return pos, dist, nfail


class SwarmLoop:
    """One rank of the lock-step loop. `solve(inputs, plans_all, has_plan) -> out dict` is the device solver
    (or, in CPU tests, the oracle); `allgather(local_array) -> full array` exchanges the shards."""

    def __init__(self, prm, cfg, n_rob, rank=0, world=1, solve=None, allgather=None, radius=None, reference=None,
                 starts=None, goals=None):
        """reference(agent_id, path, n_path, plans_all, has_plan) -> (ref_full, path_vel): when given, the reference
        trajectory comes from it (the device kernel of row f1, or the oracle) instead of the host code.
        starts / goals [n_rob][3]: explicit scenario (default: the circular exchange)."""
        self.prm, self.n_rob, self.rank, self.world = prm, n_rob, rank, world
        self.reference = reference
        self.has_world = False
        self.pmax = 3  # points of the reference polyline handed to `reference` (16 after route())
        if starts is None:
            starts, goals = circle_scenario(n_rob, radius)
        starts, goals = np.asarray(starts, dtype=np.float64), np.asarray(goals, dtype=np.float64)
        self.first, self.n_local = shard_range(n_rob, rank, world)
        sl = slice(self.first, self.first + self.n_local)
        self.shard = SwarmShard(prm, cfg, n_rob, self.first, starts[sl], goals[sl])
        self.solve, self.allgather = solve, allgather
        N = prm.n_hor
        self.plans_all = np.zeros((n_rob, N + 1, 9))
        self.has_plan = np.zeros(n_rob, np.uint8)
        self.round_idx = 0

    def set_world(self, occupancy, origin, route=True):
        """Occupied world for the corridor generator (f2) and, with route=True, global paths from the built-in router."""
        self.shard.set_world(occupancy, origin)
        self.has_world = occupancy is not None
        if route:
            failed = self.shard.route()
            self.pmax = 32
            return failed
        return 0

    def step(self, record=None):
        if self.reference is not None:
            self.shard.prepare_corridor()  # AC:165 before AC:171
            path, n_path = self.shard.reference_inputs(self.pmax)
            ids = np.arange(self.first, self.first + self.n_local, dtype=np.int32)
            ref_full, pv = self.reference(ids, path, n_path, self.plans_all, self.has_plan, **({"vel_cap": self.shard.vel_cap()} if self.has_world else {}))
            self.shard.set_reference(ref_full, pv)
        inputs = self.shard.prepare(self.plans_all, self.has_plan)
        if record is not None:
            record.append(dict({k: v.copy() for k, v in inputs.items()}, plans=self.plans_all.copy(),
                               has_plan=self.has_plan.copy()))
        out = self.solve(inputs, self.plans_all, self.has_plan)
        plans_local, has_local = self.shard.commit(out)
        if self.world > 1:
            per = (self.n_rob + self.world - 1) // self.world
            pad = per - self.n_local  # equal-size shards for the collective
            pl = np.concatenate([plans_local, np.zeros((pad,) + plans_local.shape[1:])]) if pad else plans_local.copy()
            hl = np.concatenate([has_local, np.zeros(pad, np.uint8)]) if pad else has_local
            # ONE collective per round: the has_plan flag travels inside the record (first entry NaN = no plan), exactly like
            # hdsm_publish_device / hdsm_exchange_device do on the device
            pl[hl == 0, 0, 0] = np.nan
            full = self.allgather(pl)[: self.n_rob]
            self.has_plan = (~np.isnan(full[:, 0, 0])).astype(np.uint8)
            full[self.has_plan == 0, 0, 0] = 0.0
            self.plans_all = full
        else:
            self.plans_all, self.has_plan = plans_local, has_local
        self.round_idx += 1
        return out


def poly_octa3d(grid, seed, n_it=42, res=0.3, mark=-1, origin=(0.0, 0.0, 0.0), max_rows=18, shape_aware=False):
    """Convex voxel decomposition around `seed` (hdsm_poly_octa3d = GetPolyOcta3D of the reference; shape_aware:
    hdsm_poly_octa3d_new = GetPolyOcta3DNew).
    grid: int8 [nz][ny][nx] (x fastest), < 100 free, >= 100 occupied; modified in place (taken voxels = mark).
    Returns rows [k][4] = (n, n . p), n . x <= n . p."""
    grid = np.ascontiguousarray(grid, dtype=np.int8)
    nz, ny, nx = grid.shape
    seed = np.asarray(seed, dtype=np.int32)
    dim = np.asarray([nx, ny, nz], dtype=np.int32)
    org = np.asarray(origin, dtype=np.float64)
    rows = np.zeros((max_rows, 4))
    n = C.c_int32(0)
    fn = _lib.load().hdsm_poly_octa3d_new if shape_aware else _lib.load().hdsm_poly_octa3d
    rc = fn(_p(seed, C.c_int32), grid.ctypes.data_as(C.POINTER(C.c_int8)), _p(dim, C.c_int32),
                                      C.c_int32(int(n_it)), C.c_double(float(res)), C.c_int32(int(mark)), _p(org, C.c_double),
                                      _p(rows, C.c_double), C.c_int32(int(max_rows)), C.byref(n))
    if rc:
        raise _lib.HdsmError(rc, "hdsm_poly_octa3d")
    return rows[: n.value].copy(), grid


def torch_allgather(group=None):
    """all-gather of a numpy shard through torch.distributed (gloo on CPU, RCCL via 'nccl' on GPUs)."""
    import torch
    import torch.distributed as dist

    def fn(local):
        t = torch.from_numpy(np.ascontiguousarray(local))
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        world = dist.get_world_size(group)
        full = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(full, t, group=group)
        return full.cpu().numpy()

    return fn


class DeviceSwarm:
    """The device-resident closed loop of one shard (hdsm_dswarm_*): corridor -> reference -> replan -> commit -> exchange on one
    stream. Built from a SwarmShard that has been set up (world, paths) and possibly flown, and an hdsm Solver."""

    def __init__(self, shard, solver, world_size=1, device=0):
        self.lib, self.shard, self.solver = _lib.load(), shard, solver
        self.h = C.c_void_p()
        self.world_size = int(world_size)
        rc = self.lib.hdsm_dswarm_create(shard.h, solver.h, C.c_int32(device), C.c_int32(world_size), C.byref(self.h))
        if rc:
            self.lib.hdsm_dswarm_last_error.restype = C.c_char_p
            raise _lib.HdsmError(rc, self.lib.hdsm_dswarm_last_error().decode())
        self.per = (shard.n_rob + self.world_size - 1) // self.world_size

    def upload_plans(self, plans_all, has_plan):
        plans_all = np.ascontiguousarray(plans_all, dtype=np.float64)
        has_plan = np.ascontiguousarray(has_plan, dtype=np.uint8)
        rc = self.lib.hdsm_dswarm_upload_plans(self.h, _p(plans_all, C.c_double), _p(has_plan, C.c_uint8))
        if rc:
            raise _lib.HdsmError(rc, "hdsm_dswarm_upload_plans")

    def round(self, comm=None, stream=None):
        sp = C.c_void_p(stream.cuda_stream if stream is not None else 0)
        rc = self.lib.hdsm_dswarm_round(self.h, comm.h if comm is not None else None, sp)
        if rc:
            self.lib.hdsm_dswarm_last_error.restype = C.c_char_p
            raise _lib.HdsmError(rc, self.lib.hdsm_dswarm_last_error().decode())

    def download(self, states=True):
        """Synchronises; returns (plans_all, has_plan, status of the last round, instances without solution so far) and, with
        states=True, copies the agent states back into the host mirror."""
        N, G = self.shard.prm.n_hor, self.per * self.world_size
        plans = np.zeros((G, N + 1, 9))
        has = np.zeros(G, np.uint8)
        status = np.zeros(self.shard.n_local, np.int32)
        failed = C.c_int32(0)
        rc = self.lib.hdsm_dswarm_download(self.h, self.shard.h if states else None, _p(plans, C.c_double), _p(has, C.c_uint8),
                                           _p(status, C.c_int32), C.byref(failed))
        if rc:
            self.lib.hdsm_dswarm_last_error.restype = C.c_char_p
            raise _lib.HdsmError(rc, self.lib.hdsm_dswarm_last_error().decode())
        return plans, has, status, failed.value

    PHASES = ("k_corridor", "k_vel_cap", "hdsm_reference_device", "k_keep_free", "hdsm_replan_device", "k_commit", "exchange")

    def _err(self, rc):
        self.lib.hdsm_dswarm_last_error.restype = C.c_char_p
        return _lib.HdsmError(rc, self.lib.hdsm_dswarm_last_error().decode())

    def set_phase_timing(self, on=True):
        """hdsm_dswarm_set_phase_timing: HIP events between the launches of the following rounds (see phase_ms)."""
        rc = self.lib.hdsm_dswarm_set_phase_timing(self.h, C.c_int32(1 if on else 0))
        if rc:
            raise self._err(rc)

    def phase_ms(self):
        """Milliseconds of the last timed round per phase (PHASES order); synchronises with that round."""
        ms = (C.c_float * 7)()
        rc = self.lib.hdsm_dswarm_last_phase_ms(self.h, ms)
        if rc:
            raise self._err(rc)
        return dict(zip(self.PHASES, [float(x) for x in ms]))

    def cache_stats(self):
        """hdsm_dswarm_cache_stats: what the device corridor's polyhedron cache did since the dswarm was created."""
        out = (C.c_int64 * 4)()
        rc = self.lib.hdsm_dswarm_cache_stats(self.h, out)
        if rc:
            raise self._err(rc)
        return {"asked": int(out[0]), "hits_same_grid": int(out[1]), "hits_interior": int(out[2]), "cache_on": bool(out[3])}

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.hdsm_dswarm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
