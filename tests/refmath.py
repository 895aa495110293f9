"""Independent numpy restatement of the hot path's mathematics, used to CERTIFY oracle and GPU results.

Written separately from oracle/hdsm_oracle.c (different language, different variable ordering, no shared
code): it rebuilds the optimisation problem literally from the reference's model —
variables/bounds/dynamics of Agent::CreateGurobiModel (agent_class.cpp:2071-2167), objective and corridor
rows of Agent::SolveOptimizationProblem (agent_class.cpp:858-1023) — and checks a candidate solution with
solver-independent KKT certificates (stationarity via non-negative least squares on the active rows).
"""
import numpy as np
from scipy.optimize import nnls

ABSENT = 1e20


def step9(prm, x, u):
    """One step of the 9-state model: literal Euler / RK4 of ModelODE (agent_class.cpp:2115-2167)."""
    D = np.array([prm.drag[0], prm.drag[1], prm.drag[2]])

    def f(x):
        return np.concatenate([x[3:6], x[6:9] - D * x[3:6], u])

    dt = prm.dt
    k1 = f(x)
    if not prm.rk4:
        return x + dt * k1
    k2 = f(x + dt / 2 * k1)
    k3 = f(x + dt / 2 * k2)
    k4 = f(x + dt * k3)
    return x + dt * ((k1 + 2 * k2 + 2 * k3 + k4) / 6)


def rollout(prm, state, ctrl):
    N = prm.n_hor
    traj = np.zeros((N + 1, 9))
    traj[0] = state
    for i in range(N):
        traj[i + 1] = step9(prm, traj[i], np.asarray(ctrl[i], dtype=float))
    return traj


def affine_maps(prm, state):
    """traj.flatten() = T0 + T @ ctrl.flatten() with ctrl laid out [N][3] (the ABI layout)."""
    N = prm.n_hor
    T0 = rollout(prm, state, np.zeros((N, 3))).reshape(-1)
    T = np.zeros(((N + 1) * 9, 3 * N))
    zero_state = np.zeros(9)
    base = rollout(prm, zero_state, np.zeros((N, 3))).reshape(-1)
    for k in range(3 * N):
        e = np.zeros(3 * N)
        e[k] = 1.0
        T[:, k] = rollout(prm, zero_state, e.reshape(N, 3)).reshape(-1) - base
    return T0, T


def objective(prm, traj, ctrl, ref):
    """Literal objective (agent_class.cpp:870-883, 2098)."""
    N = prm.n_hor
    J = prm.r_u * float(np.sum(np.asarray(ctrl) ** 2))
    for i in range(1, N + 1):
        w = prm.r_n if i == N else prm.r_x
        for k in range(6):
            J += w[k] * (traj[i][k] - ref[i - 1][k]) ** 2
    return J


def quad_form(prm, state, ref):
    """J(u) = 1/2 u'Hu + g'u + f0 in the ABI's ctrl ordering."""
    N = prm.n_hor
    T0, T = affine_maps(prm, state)
    n = 3 * N
    H = 2 * prm.r_u * np.eye(n)
    g = np.zeros(n)
    f0 = 0.0
    for i in range(1, N + 1):
        w = prm.r_n if i == N else prm.r_x
        for k in range(6):
            if w[k] == 0:
                continue
            row = T[9 * i + k]
            e0 = T0[9 * i + k] - ref[i - 1][k]
            H += 2 * w[k] * np.outer(row, row)
            g += 2 * w[k] * e0 * row
            f0 += w[k] * e0 * e0
    return H, g, f0, T0, T


def linear_rows(prm, state, planes_by_point, tol_fixed=1e-6):
    """All rows of the model in u-space.

    planes_by_point: dict m -> array [K,4] of rows n.p_m <= c (corridor + neighbour rows that the
    chosen assignment imposes on point m, m = 0..N).
    Returns (Aeq, beq, Ain, bin, fixed_ok).
    """
    N = prm.n_hor
    T0, T = affine_maps(prm, state)
    Aeq, beq, Ain, bin_ = [], [], [], []
    for k in range(3, 9):  # v_N = a_N = 0 (agent_class.cpp:2078-2081)
        Aeq.append(T[9 * N + k])
        beq.append(-T0[9 * N + k])
    n = 3 * N
    for i in range(N):  # input box (agent_class.cpp:2185-2186)
        for ax in range(3):
            e = np.zeros(n)
            e[3 * i + ax] = 1
            if abs(prm.u_ub[ax]) < ABSENT:
                Ain.append(e.copy()), bin_.append(prm.u_ub[ax])
            if abs(prm.u_lb[ax]) < ABSENT:
                Ain.append(-e), bin_.append(-prm.u_lb[ax])
    for i in range(1, N):  # state boxes on x_1..x_{N-1} (agent_class.cpp:2084, 2179-2184)
        for k in range(3, 9):
            if abs(prm.x_ub[k]) < ABSENT:
                Ain.append(T[9 * i + k]), bin_.append(prm.x_ub[k] - T0[9 * i + k])
            if abs(prm.x_lb[k]) < ABSENT:
                Ain.append(-T[9 * i + k]), bin_.append(T0[9 * i + k] - prm.x_lb[k])
    fixed_ok = True
    for m, rows in planes_by_point.items():
        rows = np.asarray(rows, dtype=float).reshape(-1, 4)
        for r in rows:
            a = r[0] * T[9 * m + 0] + r[1] * T[9 * m + 1] + r[2] * T[9 * m + 2]
            c = r[3] - (r[0] * T0[9 * m + 0] + r[1] * T0[9 * m + 1] + r[2] * T0[9 * m + 2])
            if m == 0:
                fixed_ok &= bool(-c <= tol_fixed)
                continue
            Ain.append(a), bin_.append(c)
    return (np.array(Aeq), np.array(beq), np.array(Ain).reshape(-1, n), np.array(bin_), fixed_ok)


def rows_for_assignment(N, polys, assign, common=None):
    """dict point m -> rows [K,4] imposed by assignment (polys[i][j] = (A,b)) plus common rows."""
    out = {m: [] for m in range(N + 1)}
    for i in range(N):
        blocks = []
        if assign is not None and assign[i] is not None and assign[i] >= 0:
            A, b = polys[i][assign[i]]
            blocks.append(np.hstack([np.asarray(A, float).reshape(-1, 3), np.asarray(b, float).reshape(-1, 1)]))
        if common is not None and common[i] is not None and len(common[i]):
            blocks.append(np.asarray(common[i], float).reshape(-1, 4))
        for blk in blocks:
            out[i].append(blk)
            out[i + 1].append(blk)
    return {m: (np.vstack(v) if v else np.zeros((0, 4))) for m, v in out.items()}


def kkt_certificate(H, g, Aeq, beq, Ain, bin_, u, act_tol=1e-7):
    """Solver-independent optimality certificate of a strictly convex QP.

    Returns dict(primal_eq, primal_in, stationarity, n_active). Stationarity is the residual of
        H u + g + Aeq' nu + Aact' lam = 0,  lam >= 0
    minimised over (nu free, lam >= 0) — a non-negative least-squares problem."""
    u = np.asarray(u, float).reshape(-1)
    grad = H @ u + g
    r_eq = float(np.max(np.abs(Aeq @ u - beq))) if len(beq) else 0.0
    slack = bin_ - Ain @ u if len(bin_) else np.zeros(0)
    r_in = float(max(0.0, -slack.min())) if len(slack) else 0.0
    act = np.where(slack <= act_tol)[0] if len(slack) else np.zeros(0, int)
    cols = [Ain[act].T] if len(act) else []
    if len(beq):
        cols += [Aeq.T, -Aeq.T]
    if cols:
        M = np.hstack(cols)
        scale = np.linalg.norm(M, axis=0)
        scale[scale == 0] = 1
        lam, rnorm = nnls(M / scale, -grad, maxiter=20 * M.shape[1] + 200)
        stat = float(rnorm)
    else:
        stat = float(np.linalg.norm(grad))
    return dict(primal_eq=r_eq, primal_in=r_in, stationarity=stat, n_active=int(len(act)),
                grad_norm=float(np.linalg.norm(grad)))


def certify(prm, state, ref, ctrl, planes_by_point, act_tol=1e-7):
    H, g, f0, T0, T = quad_form(prm, state, ref)
    Aeq, beq, Ain, bin_, fixed_ok = linear_rows(prm, state, planes_by_point)
    cert = kkt_certificate(H, g, Aeq, beq, Ain, bin_, np.asarray(ctrl).reshape(-1), act_tol)
    cert["fixed_ok"] = fixed_ok
    u = np.asarray(ctrl).reshape(-1)
    cert["obj"] = float(0.5 * u @ H @ u + g @ u + f0)
    return cert


def tasc_plane_algebraic(prm, c, o):
    """The trig-free form used by the HIP kernel: s = r / sqrt(1 - nz^2 + (r/h)^2 nz^2)."""
    c, o = np.asarray(c, float), np.asarray(o, float)
    d = o - c
    nrm = np.linalg.norm(d)
    if nrm == 0:
        return np.zeros(4)
    nh = d / nrm
    r, h = prm.drone_radius, prm.drone_z_offset
    k = r / h
    s = r / np.sqrt(1 - nh[2] ** 2 + k * k * nh[2] ** 2)
    q = (c + o) / 2 - min(2 * s, nrm) / 2 * nh
    c1 = np.array([nh[1], -nh[0], 0.0])
    c2 = np.array([-nh[2], 0.0, nh[0]])
    nf = prm.plane_perturb * (c1 + c2) + prm.plane_perturb * c2 + nh
    return np.array([nf[0], nf[1], nf[2], nf @ q])
