"""ctypes loader of the CPU execution of the DEVICE kernel source (tests/wave_emu/libhdsm_wave_emu.so). TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhdsm_wave_emu.so")
_lib = None
MAXNV = 48


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = C.CDLL(_SO)
        _lib.wave_last_error.restype = C.c_char_p
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def new_warm_store(n_inst):
    """The handle's warm-start store: pass the same array to consecutive replans of the same instances."""
    return np.zeros((n_inst, MAXNV + 2), dtype=np.int32)


def replan(prm, agent_id, state, ref, n_poly, n_rows, A, b, plans, has_plan, warm=None, bounds_min=256, threads=64, cmax=0, split_budget=0):
    N, P = prm.n_hor, prm.poly_hor
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    agent_id, n_poly, n_rows = i32(agent_id), i32(n_poly), i32(n_rows)
    state, ref, A, b, plans = f64(state), f64(ref), f64(A), f64(b), f64(plans)
    has_plan = np.ascontiguousarray(has_plan, dtype=np.uint8)
    n_inst, n_rob = state.shape[0], plans.shape[0]
    out = dict(traj=np.zeros((n_inst, N + 1, 9)), ctrl=np.zeros((n_inst, N, 3)), used=np.zeros((n_inst, P), dtype=np.uint8),
               status=np.zeros(n_inst, dtype=np.int32), obj=np.zeros(n_inst), qp_iters=np.zeros(n_inst, dtype=np.int32),
               nodes=np.zeros(n_inst, dtype=np.int32), sweeps=np.zeros(n_inst, dtype=np.int32), cand=np.zeros(n_inst, dtype=np.int32),
               flags=np.zeros(n_inst, dtype=np.uint32))
    d, i, u = C.c_double, C.c_int32, C.c_uint8
    if warm is not None:
        assert warm.dtype == np.int32 and warm.flags.c_contiguous and warm.shape == (n_inst, MAXNV + 2)
    rc = lib().wave_replan(C.byref(prm), n_inst, n_rob, _p(agent_id, i), _p(state, d), _p(ref, d), _p(n_poly, i), _p(n_rows, i),
                           _p(A, d), _p(b, d), _p(plans, d), _p(has_plan, u), _p(out["traj"], d), _p(out["ctrl"], d),
                           _p(out["used"], u), _p(out["status"], i), _p(out["obj"], d), _p(out["qp_iters"], i), _p(out["nodes"], i),
                           _p(out["sweeps"], i), _p(out["cand"], i), _p(out["flags"], C.c_uint32),
                           _p(warm, i) if warm is not None else None, C.c_int32(bounds_min), C.c_int32(threads), C.c_int32(cmax), C.c_int32(split_budget))
    if rc == -100:
        raise RuntimeError("wavefront emulation: " + lib().wave_last_error().decode())
    assert rc == 0, rc
    return out


def solve(prm, state, ref, n_poly, n_rows, A, b, threads=64):
    """Level 1 through the product's host-side split + the device source."""
    N, P = prm.n_hor, prm.poly_hor
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    state, ref, A, b, n_poly, n_rows = f64(state), f64(ref), f64(A), f64(b), i32(n_poly), i32(n_rows)
    n_inst, r_max = state.shape[0], A.shape[3]
    out = dict(traj=np.zeros((n_inst, N + 1, 9)), ctrl=np.zeros((n_inst, N, 3)), used=np.zeros((n_inst, P), dtype=np.uint8),
               status=np.zeros(n_inst, dtype=np.int32), obj=np.zeros(n_inst), qp_iters=np.zeros(n_inst, dtype=np.int32),
               nodes=np.zeros(n_inst, dtype=np.int32))
    d, i, u = C.c_double, C.c_int32, C.c_uint8
    rc = lib().wave_solve(C.byref(prm), n_inst, r_max, _p(state, d), _p(ref, d), _p(n_poly, i), _p(n_rows, i), _p(A, d), _p(b, d),
                          _p(out["traj"], d), _p(out["ctrl"], d), _p(out["used"], u), _p(out["status"], i), _p(out["obj"], d),
                          C.c_int32(threads), _p(out["qp_iters"], i), _p(out["nodes"], i))
    if rc == -100:
        raise RuntimeError("wavefront emulation: " + lib().wave_last_error().decode())
    out["rc"] = rc
    return out


_lib_t2 = None


def poly_octa3d_batch(world, ldim, off, ground_k, seed, variant, origin, n_it=42, res=0.3, max_rows=32, short_batches=False):
    """The cooperative voxel decomposition (corridor_wave.h: one wavefront per seed) run on the CPU; arguments and results of
    multi_agent_pkgs_amd.lib.poly_octa3d_batch(..., wave=True). short_batches: the build whose rim-move batches hold two turns."""
    global _lib_t2
    L = lib()
    if short_batches:
        if _lib_t2 is None:
            _lib_t2 = C.CDLL(os.path.join(_HERE, "libhdsm_corridor_emu_t2.so"))
            _lib_t2.wave_last_error.restype = C.c_char_p
        L = _lib_t2
    world = np.ascontiguousarray(world, dtype=np.int8)
    wdim = np.asarray(world.shape[::-1], dtype=np.int32)
    ldim = np.asarray(ldim, dtype=np.int32)
    i32a = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    off, seed, ground_k, variant = i32a(off), i32a(seed), i32a(ground_k), i32a(variant)
    origin = np.ascontiguousarray(origin, dtype=np.float64)
    n = off.shape[0]
    rows = np.zeros((n, max_rows, 4))
    n_rows, rc, cells = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    i32, d = C.c_int32, C.c_double
    r = L.wave_poly_octa3d_batch(C.c_int32(n), world.ctypes.data_as(C.POINTER(C.c_int8)), _p(wdim, i32), _p(ldim, i32), _p(off, i32),
                                 _p(ground_k, i32), _p(seed, i32), _p(variant, i32), _p(origin, d), C.c_int32(n_it), C.c_double(res),
                                 _p(rows, d), C.c_int32(max_rows), _p(n_rows, i32), _p(rc, i32), _p(cells, i32))
    if r == -101:
        raise RuntimeError("wave emulation: a store outside the LDS the launch asks for (guard zone overwritten)")
    if r:
        raise RuntimeError("wave emulation: %s" % L.wave_last_error().decode())
    return rows, n_rows, rc, cells
