"""ctypes binding of libhdsm.so (the C ABI of include/hdsm.h).

There is deliberately no fallback: if the shared library has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C multi_agent_pkgs_amd/csrc``) importing the library raises,
and creating a solver on a machine without a HIP device raises :class:`HdsmError` (HDSM_ERR_NO_DEVICE).
"""
import ctypes as C
import os

import numpy as np

from .params import HdsmParams

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("HDSM_LIBRARY") or os.path.join(_HERE, "libhdsm.so")  # (HDSM_LIBRARY: a development build, e.g. -DCD_PROFILE)

HDSM_OK, HDSM_ERR_BAD_ARG, HDSM_ERR_NO_DEVICE, HDSM_ERR_DEVICE, HDSM_ERR_CAPACITY, HDSM_ERR_COMM = 0, -1, -2, -3, -4, -5
HDSM_FLAG_NODE_LIMIT, HDSM_FLAG_ITER_LIMIT, HDSM_FLAG_TIME_LIMIT, HDSM_FLAG_STAGING_OVERFLOW = 1, 2, 4, 8
HDSM_COMM_ID_BYTES = 128

EXPORTS = ("hdsm_version", "hdsm_last_error", "hdsm_default_params", "hdsm_create", "hdsm_destroy",
           "hdsm_replan", "hdsm_replan_device", "hdsm_solve", "hdsm_tasc_planes", "hdsm_last_stats",
           "hdsm_reset_warm_start", "hdsm_host_register", "hdsm_host_unregister", "hdsm_last_sweep_stats", "hdsm_set_kernel_timing", "hdsm_last_kernel_ms", "hdsm_comm_unique_id", "hdsm_comm_create", "hdsm_comm_info",
           "hdsm_comm_destroy", "hdsm_publish_device", "hdsm_exchange_device", "hdsm_reference", "hdsm_reference_device", "hdsm_poly_octa3d", "hdsm_poly_octa3d_new", "hdsm_poly_octa3d_batch",
           "hdsm_poly_octa3d_device", "hdsm_poly_octa3d_scratch_bytes", "hdsm_poly_octa3d_batch_wave", "hdsm_poly_octa3d_device_wave", "hdsm_corridor_last_error",
           "hdsm_swarm_set_world", "hdsm_swarm_set_paths", "hdsm_swarm_route", "hdsm_swarm_get_paths",
           "hdsm_swarm_reference_inputs_n", "hdsm_swarm_corridor_errors", "hdsm_swarm_yaw", "hdsm_swarm_view", "hdsm_swarm_record_solve_ms", "hdsm_swarm_shutdown", "hdsm_swarm_prepare_corridor", "hdsm_swarm_vel_cap",
           "hdsm_dswarm_create", "hdsm_dswarm_upload_plans", "hdsm_dswarm_round", "hdsm_dswarm_download", "hdsm_dswarm_destroy", "hdsm_dswarm_last_error",
           "hdsm_dswarm_set_phase_timing", "hdsm_dswarm_last_phase_ms", "hdsm_dswarm_cache_stats",
           "hdsm_stats_create", "hdsm_stats_destroy", "hdsm_stats_add", "hdsm_stats_add_state", "hdsm_stats_add_latency",
           "hdsm_stats_shutdown", "hdsm_map_preprocess", "hdsm_map_preprocess_device", "hdsm_map_last_error")


class HdsmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hdsm error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load libhdsm.so; raises if it is missing (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(f"{SO_PATH} not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        # One ROCm runtime per process. libhdsm.so links /opt/rocm's libamdhip64 / librccl; PyTorch bundles its own copies under
        # the SAME sonames, and whichever is mapped first serves both. A process that also uses torch (device tensors, streams)
        # must let torch map its copies first, or torch no longer finds the GPU ("No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(SO_PATH)
        lib.hdsm_last_error.restype = C.c_char_p
        lib.hdsm_version.restype = C.c_int32
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise HdsmError(rc, load().hdsm_last_error().decode())


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def host_register(arr):
    """hdsm_host_register on a C-contiguous numpy array that the caller keeps alive and reuses every round (page-locked,
    mapped: DMA without staging; output arrays of replan() are then written by the device). Undo with host_unregister()."""
    assert arr.flags["C_CONTIGUOUS"] and arr.nbytes > 0
    lib = load()
    lib.hdsm_host_register.argtypes = [C.c_void_p, C.c_size_t]
    _check(lib.hdsm_host_register(C.c_void_p(arr.ctypes.data), arr.nbytes))
    return arr


def host_unregister(arr):
    lib = load()
    lib.hdsm_host_unregister.argtypes = [C.c_void_p]
    _check(lib.hdsm_host_unregister(C.c_void_p(arr.ctypes.data)))


class Solver:
    """One hdsm handle (= the persistent GRBModel of one planner thread, but batched)."""

    def __init__(self, prm: HdsmParams, max_instances: int, n_rob_max: int, device: int = 0):
        self.lib = load()
        self.prm = prm.copy()
        self.max_instances, self.n_rob_max, self.device = int(max_instances), int(n_rob_max), int(device)
        self.h = C.c_void_p()
        _check(self.lib.hdsm_create(C.byref(self.prm), self.max_instances, self.n_rob_max, self.device,
                                    C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.hdsm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-pointer entry point (PCIe inclusive) -------------------------------------------------------
    def replan(self, agent_id, state, ref, n_poly, n_rows, A, b, plans, has_plan, out=None, stats=True):
        N, P = self.prm.n_hor, self.prm.poly_hor
        agent_id, n_poly, n_rows = _i32(agent_id), _i32(n_poly), _i32(n_rows)
        state, ref, A, b, plans, has_plan = _f64(state), _f64(ref), _f64(A), _f64(b), _f64(plans), _u8(has_plan)
        n_inst, n_rob = state.shape[0], plans.shape[0]
        if out is None:
            out = dict(traj=np.zeros((n_inst, N + 1, 9)), ctrl=np.zeros((n_inst, N, 3)),
                       used=np.zeros((n_inst, P), dtype=np.uint8), status=np.zeros(n_inst, dtype=np.int32),
                       obj=np.zeros(n_inst))
        d, i, u = C.c_double, C.c_int32, C.c_uint8
        _check(self.lib.hdsm_replan(self.h, n_inst, n_rob, _p(agent_id, i), _p(state, d), _p(ref, d),
                                    _p(n_poly, i), _p(n_rows, i), _p(A, d), _p(b, d), _p(plans, d),
                                    _p(has_plan, u), _p(out["traj"], d), _p(out["ctrl"], d),
                                    _p(out["used"], u), _p(out["status"], i), _p(out["obj"], d)))
        if stats:
            out.update(self.last_stats(n_inst))
        return out

    # ---- device-pointer entry point: torch tensors (already resident in HBM), async on `stream` ---------
    def replan_device(self, agent_id, state, ref, n_poly, n_rows, A, b, plans, has_plan, traj, ctrl, used,
                      status, obj, stream=None):
        """All arguments are CUDA(HIP) torch tensors with the dtypes/layouts of include/hdsm.h."""
        n_inst, n_rob = state.shape[0], plans.shape[0]
        for t in (agent_id, state, ref, n_poly, n_rows, A, b, plans, has_plan, traj, ctrl, used, status, obj):
            assert t.is_cuda and t.is_contiguous()
        sp = C.c_void_p(stream.cuda_stream if stream is not None else 0)
        vp = lambda t: C.c_void_p(t.data_ptr())
        _check(self.lib.hdsm_replan_device(self.h, n_inst, n_rob, vp(agent_id), vp(state), vp(ref), vp(n_poly),
                                           vp(n_rows), vp(A), vp(b), vp(plans), vp(has_plan), vp(traj),
                                           vp(ctrl), vp(used), vp(status), vp(obj), sp))

    def solve(self, state, ref, n_poly, n_rows, A, b, out=None):
        """Level 1 (hdsm_solve): fully formed per-step polyhedra poly_const_final_vec_[N][<=P]."""
        N, P = self.prm.n_hor, self.prm.poly_hor
        state, ref, A, b = _f64(state), _f64(ref), _f64(A), _f64(b)
        n_poly, n_rows = _i32(n_poly), _i32(n_rows)
        n_inst, r_max = state.shape[0], A.shape[3]
        if out is None:
            out = dict(traj=np.zeros((n_inst, N + 1, 9)), ctrl=np.zeros((n_inst, N, 3)),
                       used=np.zeros((n_inst, P), dtype=np.uint8), status=np.zeros(n_inst, dtype=np.int32),
                       obj=np.zeros(n_inst))
        d, i, u = C.c_double, C.c_int32, C.c_uint8
        _check(self.lib.hdsm_solve(self.h, n_inst, r_max, _p(state, d), _p(ref, d), _p(n_poly, i), _p(n_rows, i),
                                   _p(A, d), _p(b, d), _p(out["traj"], d), _p(out["ctrl"], d), _p(out["used"], u),
                                   _p(out["status"], i), _p(out["obj"], d)))
        return out

    def reference(self, cfg, agent_id, path, n_path, plans, has_plan, vel_cap=None):
        """Next row f1 (hdsm_reference): path_vel + sampled reference for every instance. path [n_inst][pmax][3]."""
        N = self.prm.n_hor
        agent_id, n_path = _i32(agent_id), _i32(n_path)
        path, plans, has_plan = _f64(path), _f64(plans), _u8(has_plan)
        n_inst, pmax, n_rob = path.shape[0], path.shape[1], plans.shape[0]
        ref_full, ref, pv = np.zeros((n_inst, N + 1, 6)), np.zeros((n_inst, N, 6)), np.zeros(n_inst)
        cap = _f64(vel_cap) if vel_cap is not None else None
        d, i, u = C.c_double, C.c_int32, C.c_uint8
        _check(self.lib.hdsm_reference(self.h, C.byref(cfg), n_inst, n_rob, _p(agent_id, i), _p(path, d), _p(n_path, i),
                                       pmax, _p(cap, d) if cap is not None else None, _p(plans, d), _p(has_plan, u),
                                       _p(ref_full, d), _p(ref, d), _p(pv, d)))
        return ref_full, ref, pv

    def reference_device(self, cfg, agent_id, path, n_path, plans, has_plan, ref_full, ref, path_vel, vel_cap=None,
                         stream=None):
        """hdsm_reference_device on CUDA/HIP torch tensors (asynchronous on `stream`)."""
        n_inst, pmax, n_rob = path.shape[0], path.shape[1], plans.shape[0]
        vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        sp = C.c_void_p(stream.cuda_stream if stream is not None else 0)
        _check(self.lib.hdsm_reference_device(self.h, C.byref(cfg), n_inst, n_rob, vp(agent_id), vp(path), vp(n_path), pmax,
                                              vp(vel_cap), vp(plans), vp(has_plan), vp(ref_full), vp(ref), vp(path_vel), sp))

    def tasc_planes(self, agent_id, state, plans, has_plan):
        N = self.prm.n_hor
        agent_id, state, plans, has_plan = _i32(agent_id), _f64(state), _f64(plans), _u8(has_plan)
        n_inst, n_rob = state.shape[0], plans.shape[0]
        planes = np.zeros((n_inst, N, n_rob, 4))
        _check(self.lib.hdsm_tasc_planes(self.h, n_inst, n_rob, _p(agent_id, C.c_int32), _p(state, C.c_double),
                                         _p(plans, C.c_double), _p(has_plan, C.c_uint8),
                                         _p(planes, C.c_double)))
        return planes

    def reset_warm_start(self):
        _check(self.lib.hdsm_reset_warm_start(self.h))

    def last_sweep_stats(self, n_inst):
        st = dict(sphere_records=np.zeros(n_inst, dtype=np.int32), pairs=np.zeros(n_inst, dtype=np.int32),
                  flags=np.zeros(n_inst, dtype=np.uint32))
        _check(self.lib.hdsm_last_sweep_stats(self.h, n_inst, _p(st["sphere_records"], C.c_int32), _p(st["pairs"], C.c_int32),
                                              _p(st["flags"], C.c_uint32)))
        return st

    def set_kernel_timing(self, on=True):
        _check(self.lib.hdsm_set_kernel_timing(self.h, C.c_int32(1 if on else 0)))

    def last_kernel_ms(self):
        ms = C.c_float(0.0)
        _check(self.lib.hdsm_last_kernel_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def last_stats(self, n_inst):
        st = {k: np.zeros(n_inst, dtype=np.int32) for k in ("qp_iters", "nodes", "sweeps", "cand")}
        _check(self.lib.hdsm_last_stats(self.h, n_inst, _p(st["qp_iters"], C.c_int32), _p(st["nodes"], C.c_int32),
                                        _p(st["sweeps"], C.c_int32), _p(st["cand"], C.c_int32)))
        return st


def comm_unique_id():
    """hdsm_comm_unique_id (rank 0): 128 opaque bytes the launcher hands to every rank."""
    buf = (C.c_uint8 * HDSM_COMM_ID_BYTES)()
    _check(load().hdsm_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """One RCCL communicator of the per-round plan exchange (hdsm_comm_* / hdsm_exchange_device)."""

    def __init__(self, solver, unique_id, rank, world):
        self.lib = load()
        self.h = C.c_void_p()
        buf = (C.c_uint8 * HDSM_COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(self.lib.hdsm_comm_create(solver.h, buf, int(rank), int(world), C.byref(self.h)))
        r, w = C.c_int32(-1), C.c_int32(-1)
        _check(self.lib.hdsm_comm_info(self.h, C.byref(r), C.byref(w)))
        self.rank, self.world, self.solver = r.value, w.value, solver

    def publish_device(self, traj, has_local, plans_local, n_local=None, stream=None):
        per = plans_local.shape[0]
        n_local = per if n_local is None else int(n_local)
        sp = C.c_void_p(stream.cuda_stream if stream is not None else 0)
        _check(self.lib.hdsm_publish_device(self.solver.h, per, n_local, C.c_void_p(traj.data_ptr()),
                                            C.c_void_p(has_local.data_ptr()), C.c_void_p(plans_local.data_ptr()), sp))

    def exchange_device(self, plans_local, plans_all, has_all, stream=None):
        """ONE all-gather: plans_local [per][N+1][9] of every rank -> plans_all [world*per][N+1][9], has_all [world*per]."""
        per = plans_local.shape[0]
        assert plans_all.shape[0] == per * self.world and has_all.shape[0] == per * self.world
        sp = C.c_void_p(stream.cuda_stream if stream is not None else 0)
        _check(self.lib.hdsm_exchange_device(self.h, per, C.c_void_p(plans_local.data_ptr()), C.c_void_p(plans_all.data_ptr()),
                                             C.c_void_p(has_all.data_ptr()), sp))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.hdsm_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def map_preprocess(cfg, grids, device=0):
    """hdsm_map_preprocess: grids int8 [n][nz][ny][nx] (-1 unknown, 0 free, 100 occupied) -> same shape, after
    SetUncertainToUnknown, InflateObstacles and CreatePotentialField (next row f4). Runs on the GPU."""
    import numpy as np
    L = load()
    g = np.ascontiguousarray(grids, dtype=np.int8)
    assert g.ndim == 4
    out = np.empty_like(g)
    dim = np.asarray(g.shape[:0:-1], dtype=np.int32)
    rc = L.hdsm_map_preprocess(C.c_int32(device), C.byref(cfg), C.c_int32(g.shape[0]), dim.ctypes.data_as(C.POINTER(C.c_int32)),
                               g.ctypes.data_as(C.POINTER(C.c_int8)), out.ctypes.data_as(C.POINTER(C.c_int8)))
    if rc:
        L.hdsm_map_last_error.restype = C.c_char_p
        raise HdsmError(rc, L.hdsm_map_last_error().decode())
    return out


def map_preprocess_device(cfg, d_in, d_out, d_scratch, stream=None, device=0):
    """Device-pointer variant on torch tensors: d_in/d_out int8 [n][nz][ny][nx], d_scratch uint8 with >= 2 * d_in.numel()."""
    import numpy as np
    L = load()
    dim = np.asarray(tuple(d_in.shape)[:0:-1], dtype=np.int32)
    assert d_scratch.numel() >= 2 * d_in.numel() and d_in.is_contiguous() and d_out.is_contiguous()
    rc = L.hdsm_map_preprocess_device(C.c_int32(device), C.byref(cfg), C.c_int32(d_in.shape[0]),
                                      dim.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(d_in.data_ptr()),
                                      C.c_void_p(d_out.data_ptr()), C.c_void_p(d_scratch.data_ptr()),
                                      C.c_void_p(stream.cuda_stream if stream is not None else 0))
    if rc:
        L.hdsm_map_last_error.restype = C.c_char_p
        raise HdsmError(rc, L.hdsm_map_last_error().decode())


def poly_octa3d_batch(world, ldim, off, ground_k, seed, variant, origin, n_it=42, res=0.3, max_rows=32, device=0, wave=False):
    """hdsm_poly_octa3d_batch (row f2 on the device): world int8 [wz][wy][wx]; off/seed [n][3], ground_k/variant [n], origin [n][3].
    Returns rows [n][max_rows][4], n_rows [n], rc [n], cells [n]. wave=True: hdsm_poly_octa3d_batch_wave (one wavefront per seed)."""
    L = load()
    world = np.ascontiguousarray(world, dtype=np.int8)
    wdim = np.asarray(world.shape[::-1], dtype=np.int32)
    ldim = np.asarray(ldim, dtype=np.int32)
    off, seed = _i32(off), _i32(seed)
    ground_k, variant = _i32(ground_k), _i32(variant)
    origin = _f64(origin)
    n = off.shape[0]
    rows = np.zeros((n, max_rows, 4))
    n_rows, rc, cells = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    i32, d = C.c_int32, C.c_double
    r = (L.hdsm_poly_octa3d_batch_wave if wave else L.hdsm_poly_octa3d_batch)(C.c_int32(device), C.c_int32(n), world.ctypes.data_as(C.POINTER(C.c_int8)), _p(wdim, i32), _p(ldim, i32),
                                 _p(off, i32), _p(ground_k, i32), _p(seed, i32), _p(variant, i32), _p(origin, d), C.c_int32(n_it),
                                 C.c_double(res), _p(rows, d), C.c_int32(max_rows), _p(n_rows, i32), _p(rc, i32), _p(cells, i32))
    if r:
        L.hdsm_corridor_last_error.restype = C.c_char_p
        raise HdsmError(r, L.hdsm_corridor_last_error().decode())
    return rows, n_rows, rc, cells
