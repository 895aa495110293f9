#!/usr/bin/env python3
"""Generates tests/golden/*.npz — run in the BUILD container only (needs scipy; the GPU box only reads the
committed files).

The reference holds no golden vectors for this path (multi_agent_planner has no tests and
its solve runs inside Gurobi, which is absent). These fixtures therefore pin the ORACLE against independent
mathematics, not against the reference's own outputs:

  qp_scipy.npz      fixed-assignment QPs built by tests/refmath.py (numpy restatement of the reference model)
                    and solved with scipy.optimize.minimize(trust-constr): a different algorithm (interior
                    point / SQP family) in a different code base. Stored: inputs, scipy's controls + objective.
  miqp_small.npz    tiny MIQPs (N = 4, P = 3) whose optimum was found by EXHAUSTIVE enumeration of all P^N
                    assignments with scipy solving every leaf: the combinatorial answer is pinned too.
  miqp_enum.npz     what SURVEY.md section 8c-2 asks for: the COMBINATORIAL part pinned independently. Six cases with N = 6,
                    P = 3 in which every one of the 3^6 = 729 assignments is resolved, and three cases with N = 10, P = 4 in
                    which every ADMISSIBLE assignment is (depth-first over the steps; a prefix is dropped only on a HiGHS
                    proof that its rows are infeasible, never on an objective bound). "Resolved" = solved by scipy with a
                    KKT certificate from tests/refmath.py (solver-independent proof of optimality of a strictly convex QP)
                    or proved infeasible by the phase-1 LP. Separating planes come from refmath's algebraic form, not from
                    the oracle. Stored: the snapshot (so that the device can replan it), every feasible leaf with its
                    objective, the optimum and the runner-up.
  planes.npz        separating planes for hand-computable geometry (agents on the x axis, vertical stacking)
                    from the closed-form ellipsoid support function, evaluated with mpmath-free exact algebra.
  circle.npz        start/goal of the circle launch file for the shipped n=10, R=22, c=(18,15) (known answers
                    quoted in SURVEY.md section 8d).
"""
import itertools
import os
import sys

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np
from scipy.optimize import LinearConstraint, linprog, minimize

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import problems  # noqa: E402
import refmath as rm  # noqa: E402
from multi_agent_pkgs_amd.params import agile_params, make_params  # noqa: E402


def scipy_qp(H, g, f0, Aeq, beq, Ain, bin_):
    n = len(g)
    cons = []
    if len(beq):
        cons.append(LinearConstraint(Aeq, beq, beq))
    if len(bin_):
        cons.append(LinearConstraint(Ain, -np.inf, bin_))
    fun = lambda u: 0.5 * u @ H @ u + g @ u + f0
    jac = lambda u: H @ u + g
    hess = lambda u: H
    res = minimize(fun, np.zeros(n), jac=jac, hess=hess, constraints=cons, method="trust-constr",
                   options=dict(gtol=1e-11, xtol=1e-13, barrier_tol=1e-12, maxiter=5000))
    return res.x, float(res.fun), res


def slsqp_qp(H, g, f0, Aeq, beq, Ain, bin_):
    """Fast leaf solver for the enumeration; a result is only accepted with a KKT certificate (refmath)."""
    cons = []
    if len(beq):
        cons.append(dict(type="eq", fun=lambda u: Aeq @ u - beq, jac=lambda u: Aeq))
    if len(bin_):
        cons.append(dict(type="ineq", fun=lambda u: bin_ - Ain @ u, jac=lambda u: -Ain))
    res = minimize(lambda u: 0.5 * u @ H @ u + g @ u + f0, np.zeros(len(g)), jac=lambda u: H @ u + g,
                   constraints=cons, method="SLSQP", options=dict(ftol=1e-14, maxiter=500))
    u = res.x
    feas = (len(bin_) == 0 or (Ain @ u - bin_).max() < 1e-7) and (len(beq) == 0 or np.abs(Aeq @ u - beq).max() < 1e-7)
    if not feas:
        return None
    cert = rm.kkt_certificate(H, g, Aeq, beq, Ain, bin_, u, act_tol=1e-6)
    if cert["stationarity"] > 1e-5 * max(1.0, cert["grad_norm"]):
        return None
    return u, float(0.5 * u @ H @ u + g @ u + f0)


def prm_fields(prm):
    return dict(n_hor=prm.n_hor, poly_hor=prm.poly_hor, rk4=prm.rk4, dt=prm.dt, drag=list(prm.drag))


def make_qp_cases():
    out = {}
    k = 0
    for (N, rk4, drag, seed, kw) in [(6, False, (0, 0, 0), 1, {}), (6, True, (0.1, 0.1, 0.2), 2, dict(turn=True)),
                                     (8, False, (0, 0, 0), 3, dict(narrow=True)), (5, False, (0.3, 0, 0), 4, {})]:
        prm = make_params(n_hor=N, rk4=rk4, drag=drag, max_rows_static=18)
        sn = problems.swarm_snapshot(prm, 4, seed, **kw)
        from oracle import pyoracle as orc
        for a in range(4):
            planes, valid = orc.tasc_planes(prm, a, sn["state"][a], sn["plans"], sn["has_plan"])
            common = [planes[i][valid[i] > 0] for i in range(N)]
            polys = [sn["polys"][a][: prm.poly_hor]] * N
            assign = [0] * N
            rows = rm.rows_for_assignment(N, polys, assign, common)
            H, g, f0, T0, T = rm.quad_form(prm, sn["state"][a], sn["ref"][a])
            Aeq, beq, Ain, bin_, fixed_ok = rm.linear_rows(prm, sn["state"][a], rows)
            u, f, res = scipy_qp(H, g, f0, Aeq, beq, Ain, bin_)
            viol = max(0.0, float((Ain @ u - bin_).max())) if len(bin_) else 0.0
            if not fixed_ok or viol > 1e-7 or np.abs(Aeq @ u - beq).max() > 1e-7:
                continue  # infeasible / not converged: not a golden case
            out[f"c{k}_N"] = N
            out[f"c{k}_rk4"] = int(rk4)
            out[f"c{k}_drag"] = np.array(drag, float)
            out[f"c{k}_state"] = sn["state"][a]
            out[f"c{k}_ref"] = sn["ref"][a]
            out[f"c{k}_polyA"] = polys[0][0][0]
            out[f"c{k}_polyb"] = polys[0][0][1]
            out[f"c{k}_common"] = np.concatenate([np.hstack([np.full((len(c), 1), i), c]) for i, c in enumerate(common)]
                                                 ) if sum(len(c) for c in common) else np.zeros((0, 5))
            out[f"c{k}_u"] = u.reshape(N, 3)
            out[f"c{k}_obj"] = f
            k += 1
    out["n_cases"] = k
    np.savez(os.path.join(HERE, "qp_scipy.npz"), **out)
    print("qp_scipy.npz:", k, "cases")


def make_miqp_cases():
    out = {}
    k = 0
    N, P = 4, 3
    for seed in (11, 12, 13, 14, 15, 16):
        prm = make_params(n_hor=N, poly_hor=P, max_rows_static=18)
        sn = problems.swarm_snapshot(prm, 3, seed, narrow=True, turn=True, box_half=1.2, speed=(2.0, 6.0))
        a = 0
        from oracle import pyoracle as orc
        planes, valid = orc.tasc_planes(prm, a, sn["state"][a], sn["plans"], sn["has_plan"])
        common = [planes[i][valid[i] > 0] for i in range(N)]
        plist = sn["polys"][a][:P]
        if len(plist) < 2:
            continue
        polys = [plist] * N
        H, g, f0, T0, T = rm.quad_form(prm, sn["state"][a], sn["ref"][a])
        best = (np.inf, None, None)
        second = np.inf
        for assign in itertools.product(range(len(plist)), repeat=N):
            rows = rm.rows_for_assignment(N, polys, list(assign), common)
            Aeq, beq, Ain, bin_, fixed_ok = rm.linear_rows(prm, sn["state"][a], rows)
            if not fixed_ok:
                continue
            sol = slsqp_qp(H, g, f0, Aeq, beq, Ain, bin_)
            if sol is None:  # infeasible leaf, or SLSQP gave up: phase-1 LP decides, the slow solver finishes
                lp = linprog(np.zeros(len(g)), A_ub=Ain, b_ub=bin_, A_eq=Aeq, b_eq=beq, bounds=(None, None),
                             method="highs")
                if lp.status != 0:
                    continue  # infeasible
                u, f, res = scipy_qp(H, g, f0, Aeq, beq, Ain, bin_)
                if (Ain @ u - bin_).max() > 1e-6 or np.abs(Aeq @ u - beq).max() > 1e-6:
                    raise RuntimeError("feasible leaf not solved")
            else:
                u, f = sol
            if f < best[0] - 1e-9:
                second = best[0]
                best = (f, u, assign)
            elif f < second and abs(f - best[0]) > 1e-6:
                second = f
        if best[1] is None:
            continue
        out[f"m{k}_state"] = sn["state"][a]
        out[f"m{k}_ref"] = sn["ref"][a]
        out[f"m{k}_npoly"] = len(plist)
        for j, (A, b) in enumerate(plist):
            out[f"m{k}_A{j}"] = A
            out[f"m{k}_b{j}"] = b
        out[f"m{k}_common"] = np.concatenate([np.hstack([np.full((len(c), 1), i), c]) for i, c in enumerate(common)])
        out[f"m{k}_u"] = best[1].reshape(N, 3)
        out[f"m{k}_obj"] = best[0]
        out[f"m{k}_second"] = second
        k += 1
    out["n_cases"] = k
    out["N"], out["P"] = N, P
    np.savez(os.path.join(HERE, "miqp_small.npz"), **out)
    print("miqp_small.npz:", k, "cases")


def _u_rows(T0, T, m, rows):
    """rows [K][4] on point m -> (A [K][n], b [K]) in u-space."""
    rows = np.asarray(rows, float).reshape(-1, 4)
    A = rows[:, 0:1] * T[9 * m + 0] + rows[:, 1:2] * T[9 * m + 1] + rows[:, 2:3] * T[9 * m + 2]
    b = rows[:, 3] - (rows[:, 0] * T0[9 * m + 0] + rows[:, 1] * T0[9 * m + 1] + rows[:, 2] * T0[9 * m + 2])
    return A, b


def enumerate_case(prm, sn, a, prune_prefixes):
    """Every assignment of polyhedra to steps for agent `a` of snapshot `sn`, resolved (see the module docstring).
    Returns None when a leaf can be neither certified nor proved infeasible (the seed is then not used)."""
    N, P = prm.n_hor, prm.poly_hor
    plist = sn["polys"][a][:P]
    m = len(plist)
    n_rob = sn["plans"].shape[0]
    own = sn["plans"][a]
    common = []
    for i in range(N):
        rows = [rm.tasc_plane_algebraic(prm, own[i + 1, :3], sn["plans"][k, i + 1, :3]) for k in range(n_rob)
                if k != a and sn["has_plan"][k]]
        common.append(np.array(rows).reshape(-1, 4))
    state, ref = sn["state"][a], sn["ref"][a]
    H, g, f0, T0, T = rm.quad_form(prm, state, ref)
    Aeq, beq, Ab, bb, ok0 = rm.linear_rows(prm, state, rm.rows_for_assignment(N, [plist] * N, None, common))
    if not ok0:
        return None
    blocks = {}
    for i in range(N):
        for j, (A, b) in enumerate(plist):
            rows = np.hstack([np.asarray(A, float).reshape(-1, 3), np.asarray(b, float).reshape(-1, 1)])
            ok = True
            parts = []
            for e in (0, 1):
                Au, bu = _u_rows(T0, T, i + e, rows)
                if i + e == 0:
                    ok = bool((-bu <= 1e-6).all())   # rows on the pinned p_0 only gate the choice (feas_tol_fixed)
                else:
                    parts.append((Au, bu))
            blocks[i, j] = (np.vstack([p_[0] for p_ in parts]), np.concatenate([p_[1] for p_ in parts]), ok)
    n = 3 * N

    def rows_of(assign):
        As, bs = [Ab], [bb]
        for i, j in enumerate(assign):
            As.append(blocks[i, j][0]), bs.append(blocks[i, j][1])
        return np.vstack(As), np.concatenate(bs)

    def lp_feasible(assign):
        Ain, bin_ = rows_of(assign)
        lp = linprog(np.zeros(n), A_ub=Ain, b_ub=bin_, A_eq=Aeq, b_eq=beq, bounds=(None, None), method="highs")
        if lp.status == 0:
            return True
        if lp.status == 2:
            return False
        raise RuntimeError("phase-1 LP undecided")

    leaves, stats = [], dict(lp_infeasible=0, gated=0, slow=0, prefixes_pruned=0)

    def leaf(assign):
        if not all(blocks[i, j][2] for i, j in enumerate(assign)):
            stats["gated"] += 1
            return True
        Ain, bin_ = rows_of(assign)
        sol = slsqp_qp(H, g, f0, Aeq, beq, Ain, bin_)
        if sol is None:
            if not lp_feasible(assign):
                stats["lp_infeasible"] += 1
                return True
            u, f, res = scipy_qp(H, g, f0, Aeq, beq, Ain, bin_)
            cert = rm.kkt_certificate(H, g, Aeq, beq, Ain, bin_, u, act_tol=1e-6)
            if (Ain @ u - bin_).max() > 1e-7 or np.abs(Aeq @ u - beq).max() > 1e-7 or cert["stationarity"] > 1e-5 * max(1.0, cert["grad_norm"]):
                return False     # feasible but not certified: the case is not used
            stats["slow"] += 1
            sol = (u, f)
        leaves.append((sol[1], tuple(assign), sol[0]))
        return True

    def rec(prefix):
        if len(prefix) == N:
            return leaf(prefix)
        for j in range(m):
            nxt = prefix + [j]
            if not blocks[len(prefix), j][2]:
                stats["gated"] += m ** (N - len(nxt))
                continue
            if prune_prefixes and len(nxt) < N and not lp_feasible(nxt):
                stats["prefixes_pruned"] += 1
                stats["lp_infeasible"] += m ** (N - len(nxt))
                continue
            if not rec(nxt):
                return False
        return True

    if not rec([]):
        return None
    if not leaves:
        return None
    leaves.sort(key=lambda t: t[0])
    best = leaves[0]
    others = [t[0] for t in leaves if abs(t[0] - best[0]) > 1e-6 * max(1.0, abs(best[0]))]
    return dict(state=state, ref=ref, plans=sn["plans"], has_plan=sn["has_plan"], agent=a, npoly=m,
                polyA=np.array([p_[0] for p_ in plist]), polyb=np.array([p_[1] for p_ in plist]),
                common=np.concatenate([np.hstack([np.full((len(c), 1), i), c]) for i, c in enumerate(common)]),
                u=best[2].reshape(N, 3), obj=best[0], second=min(others) if others else np.inf,
                leaf_obj=np.array([t[0] for t in leaves]), leaf_assign=np.array([t[1] for t in leaves], np.int8),
                counts=np.array([m ** N, len(leaves), stats["lp_infeasible"], stats["gated"], stats["slow"], stats["prefixes_pruned"]]))


def _enum_job(job):
    N, P, seed, kw, prune = job
    prm = make_params(n_hor=N, poly_hor=P, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 3, seed, **kw)
    a = 0
    if len(sn["polys"][a][:P]) < min(P, 3) or any(len(pb[1]) != 6 for pb in sn["polys"][a][:P]):
        return None
    try:
        return enumerate_case(prm, sn, a, prune)
    except RuntimeError:
        return None


def make_miqp_enum_cases():
    from multiprocessing import Pool
    kw6 = dict(narrow=True, turn=True, box_half=1.2, speed=(2.0, 6.0), spacing=1.2)
    kw10 = dict(narrow=True, turn=True, box_half=1.6, speed=(3.0, 7.0), spacing=1.4)
    jobs = [(6, 3, s_, kw6, False) for s_ in range(20, 64)] + [(10, 4, s_, kw10, True) for s_ in range(50, 66)]
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        res = pool.map(_enum_job, jobs, chunksize=1)
    out = {}
    want = {6: 6, 10: 3}
    have = {6: 0, 10: 0}
    k = 0
    for job, r in zip(jobs, res):
        N = job[0]
        # keep cases with a real combinatorial choice: several feasible leaves with distinct objectives and an optimum that
        # uses more than one polyhedron
        if r is None or have[N] >= want[N] or len(r["leaf_obj"]) < 8 or not np.isfinite(r["second"]):
            continue
        if len(set(r["leaf_assign"][0].tolist())) < 2:
            continue
        for key, v in r.items():
            out[f"e{k}_{key}"] = v
        out[f"e{k}_N"], out[f"e{k}_P"] = job[0], job[1]
        print(f"miqp_enum case {k}: N={N} P={job[1]} seed={job[2]} counts(total, feasible, lp_infeasible, gated, slow, pruned)={r['counts'].tolist()} "
              f"obj={r['obj']:.6f} second={r['second']:.6f} assign={r['leaf_assign'][0].tolist()}")
        have[N] += 1
        k += 1
    assert have == want, have
    out["n_cases"] = k
    np.savez_compressed(os.path.join(HERE, "miqp_enum.npz"), **out)
    print("miqp_enum.npz:", k, "cases")


def make_planes():
    """Hand-computable planes. r = drone_radius, h = drone_z_offset, p = plane_perturb = 0.1.
    (1) agents on the x axis, c = (0,0,0), o = (d,0,0), d > 2r:  n = (1,0,0), s = r,
        q = (d/2 - r, 0, 0), n_f = n + p*(c1 + c2) + p*c2 with c1 = n x z = (0,-1,0), c2 = n x y = (0,0,1)
        -> n_f = (1, -p, 2p),  rhs = d/2 - r.
    (2) vertical stacking, o = c + (0,0,d): n = (0,0,1), s = h, c1 = 0, c2 = (-1,0,0)
        -> n_f = (-2p, 0, 1), rhs = n_f . (c + (0,0,d/2 - min(2h,d)/2)).
    (3) closer than 2 s: the plane passes through c (rhs = n_f . c)."""
    cases = []
    r, h, p = 0.25, 0.4, 0.1
    c = np.array([1.0, 2.0, 1.5])
    for d in (3.0, 0.7):
        o = c + [d, 0, 0]
        nf = np.array([1.0, -p, 2 * p])
        back = min(2 * r, d) / 2
        q = (c + o) / 2 - back * np.array([1.0, 0, 0])
        cases.append((c, o, np.append(nf, nf @ q)))
    for d in (2.0, 0.5):
        o = c + [0, 0, d]
        nf = np.array([-2 * p, 0.0, 1.0])
        back = min(2 * h, d) / 2
        q = (c + o) / 2 - back * np.array([0, 0, 1.0])
        cases.append((c, o, np.append(nf, nf @ q)))
    np.savez(os.path.join(HERE, "planes.npz"), r=r, h=h, p=p, c=np.array([x[0] for x in cases]),
             o=np.array([x[1] for x in cases]), plane=np.array([x[2] for x in cases]))
    print("planes.npz:", len(cases), "cases")


def make_circle():
    # multi_agent_planner_circle.launch.py:36-44, shipped values n=10, R=22, centre (18, 15), z = 1.5
    n, R, cx, cy = 10, 22.0, 18.0, 15.0
    starts = np.array([[cx + R * np.cos(2 * np.pi * k / n), cy + R * np.sin(2 * np.pi * k / n), 1.5] for k in range(n)])
    goals = starts[(np.arange(n) + n // 2) % n]
    np.savez(os.path.join(HERE, "circle.npz"), n=n, R=R, cx=cx, cy=cy, starts=starts, goals=goals)
    print("circle.npz: start_0", starts[0], "start_1", starts[1], "goal_0", goals[0])


if __name__ == "__main__":
    which = sys.argv[1:] or ["planes", "circle", "qp", "miqp"]
    if "planes" in which:
        make_planes()
    if "circle" in which:
        make_circle()
    if "qp" in which:
        make_qp_cases()
    if "miqp" in which:
        make_miqp_cases()
    if "enum" in which:
        make_miqp_enum_cases()
