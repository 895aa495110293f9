"""ctypes loader of oracle/libhdsm_cpu_port.so: the product's algorithm (lazy closed-form planes behind a sphere prefilter, normalised pick
rule, warm start) as plain C on host cores — bench.py's `cpu_baseline_warm` leg. BENCH / TEST INFRASTRUCTURE, never loaded by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhdsm_cpu_port.so")
WARM_STRIDE = 3 * 16 + 1   # ON + 1
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = [os.path.join(_HERE, f) for f in ("hdsm_cpu_port.c", "hdsm_oracle.c", "hdsm_oracle.h")]
        if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libhdsm_cpu_port.so"])
        _lib = C.CDLL(_SO)
    return _lib


def new_warm_store(n_inst):
    return np.zeros((n_inst, WARM_STRIDE), np.int32)


def replan(prm, agent_id, state, ref, n_poly, n_rows, A, b, plans, has_plan, warm=None, n_threads=1):
    """cpu_port_replan with the layouts of hdsm_replan. `warm`: new_warm_store(n_inst), carried from call to call (None: cold)."""
    f64, i32, u8 = np.float64, np.int32, np.uint8
    agent_id, n_poly, n_rows = (np.ascontiguousarray(x, dtype=i32) for x in (agent_id, n_poly, n_rows))
    state, ref, A, b, plans = (np.ascontiguousarray(x, dtype=f64) for x in (state, ref, A, b, plans))
    has_plan = np.ascontiguousarray(has_plan, dtype=u8)
    n_inst, n_rob, N, P = state.shape[0], plans.shape[0], prm.n_hor, prm.poly_hor
    out = dict(traj=np.zeros((n_inst, N + 1, 9)), ctrl=np.zeros((n_inst, N, 3)), used=np.zeros((n_inst, P), dtype=u8),
               status=np.zeros(n_inst, dtype=i32), obj=np.zeros(n_inst), qp_iters=np.zeros(n_inst, dtype=i32))
    fb = C.c_int32(0)
    d, i, u = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    p = lambda a, t: a.ctypes.data_as(t)  # noqa: E731
    rc = lib().cpu_port_replan(C.byref(prm), C.c_int32(n_inst), C.c_int32(n_rob), p(agent_id, i), p(state, d), p(ref, d), p(n_poly, i), p(n_rows, i),
                               p(A, d), p(b, d), p(plans, d), p(has_plan, u), p(warm, i) if warm is not None else None, p(out["traj"], d),
                               p(out["ctrl"], d), p(out["used"], u), p(out["status"], i), p(out["obj"], d), p(out["qp_iters"], i), C.byref(fb),
                               C.c_int32(int(n_threads)))
    assert rc == 0
    out["fallbacks"] = fb.value
    return out


def replay(prm, rec, n_warm, n_threads):
    """cpu_port_replay on a list of recorded rounds (dicts with the arrays of hdsm_replan, the same agents in every round): the first
    n_warm rounds build the warm-start stores, the rest is timed. Returns (outputs stacked per round, seconds of the timed rounds)."""
    f64, i32, u8 = np.float64, np.int32, np.uint8
    st = lambda k, t: np.ascontiguousarray(np.stack([x[k] for x in rec]), dtype=t)  # noqa: E731
    agent_id, n_poly, n_rows = st("agent_id", i32), st("n_poly", i32), st("n_rows", i32)
    state, ref, A, b, plans, has = st("state", f64), st("ref", f64), st("A", f64), st("b", f64), st("plans", f64), st("has_plan", u8)
    R, n_inst, n_rob, N, P = state.shape[0], state.shape[1], plans.shape[1], prm.n_hor, prm.poly_hor
    out = dict(traj=np.zeros((R, n_inst, N + 1, 9)), ctrl=np.zeros((R, n_inst, N, 3)), used=np.zeros((R, n_inst, P), dtype=u8),
               status=np.zeros((R, n_inst), dtype=i32), obj=np.zeros((R, n_inst)), qp_iters=np.zeros((R, n_inst), dtype=i32))
    fb, secs = C.c_int32(0), C.c_double(0.0)
    d, i, u = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    p = lambda a, t: a.ctypes.data_as(t)  # noqa: E731
    rc = lib().cpu_port_replay(C.byref(prm), C.c_int32(R), C.c_int32(n_warm), C.c_int32(n_inst), C.c_int32(n_rob), p(agent_id, i), p(state, d),
                               p(ref, d), p(n_poly, i), p(n_rows, i), p(A, d), p(b, d), p(plans, d), p(has, u), p(out["traj"], d), p(out["ctrl"], d),
                               p(out["used"], u), p(out["status"], i), p(out["obj"], d), p(out["qp_iters"], i), C.byref(fb), C.c_int32(int(n_threads)),
                               C.byref(secs))
    assert rc == 0
    out["fallbacks"] = fb.value
    return out, secs.value
