"""Seeded synthetic problem generators for the parity tests (shared by CPU and GPU tests).

A "swarm snapshot" has exactly the structure of one replan round of the reference: every agent k holds the
plan it published last round (plans_all[k] = traj_curr_, agent_class.cpp:645-677), its current state is
that plan's state 1 (agent_class.cpp:233-238), it tracks a reference sampled along a straight path at
path_vel*dt spacing with the backward-pointing velocity reference of agent_class.cpp:1527-1547, and its
static corridor is a chain of overlapping axis-aligned boxes (what GetPolyOcta3D yields in free space,
SURVEY.md App. D.2) optionally cut by chamfer rows.
"""
import numpy as np


def box_rows(lo, hi):
    """6 rows (A x <= b) of the axis-aligned box [lo, hi], row order of convex_decomp.cpp:361-373."""
    A = np.array([[0, -1, 0], [1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1.0]])
    b = np.array([-lo[1], hi[0], hi[1], -lo[0], hi[2], -lo[2]], dtype=float)
    return A, b


def ref_from_path(p0, direction, path_vel, dt, N):
    """N reference rows: positions every path_vel*dt along `direction` starting AT p0 (ref_0 is the
    sampling start point, agent_class.cpp:1625); velocity rows point backwards (agent_class.cpp:1533)."""
    direction = np.asarray(direction, float)
    direction = direction / np.linalg.norm(direction)
    pts = np.array([p0 + direction * path_vel * dt * i for i in range(N + 1)])
    ref = np.zeros((N, 6))
    ref[:, :3] = pts[:N]
    for i in range(N):
        dvec = pts[i] - pts[i + 1]
        nrm = np.linalg.norm(dvec)
        ref[i, 3:] = path_vel * dvec / nrm if nrm > 1e-2 else 0.0
    return ref


def pack_static(polys_per_inst, P, RS):
    n_inst = len(polys_per_inst)
    n_poly = np.zeros(n_inst, dtype=np.int32)
    n_rows = np.zeros((n_inst, P), dtype=np.int32)
    A = np.zeros((n_inst, P, RS, 3))
    b = np.zeros((n_inst, P, RS))
    for k, polys in enumerate(polys_per_inst):
        n_poly[k] = len(polys)
        for j, (Aj, bj) in enumerate(polys[:P]):
            r = len(bj)
            assert r <= RS
            n_rows[k, j] = r
            A[k, j, :r] = Aj
            b[k, j, :r] = bj
    return n_poly, n_rows, A, b


def swarm_snapshot(prm, n_rob, seed, spacing=2.0, speed=(0.0, 6.0), box_half=2.25, narrow=False,
                   turn=False, absent_frac=0.0, chamfer=False, first_round=False):
    """Build one replan round for n_rob agents. Returns a dict of arrays in the ABI layouts."""
    rng = np.random.default_rng(seed)
    N, P, RS, dt = prm.n_hor, prm.poly_hor, prm.max_rows_static, prm.dt
    side = int(np.ceil(np.sqrt(n_rob)))
    plans = np.zeros((n_rob, N + 1, 9))
    state = np.zeros((n_rob, 9))
    ref = np.zeros((n_rob, N, 6))
    polys_all = []
    for k in range(n_rob):
        gx, gy = k % side, k // side
        pos = np.array([gx * spacing, gy * spacing, 1.5]) + rng.uniform(-0.3, 0.3, 3) * [1, 1, 0.5]
        ang = rng.uniform(0, 2 * np.pi)
        v = rng.uniform(*speed)
        vel = np.array([np.cos(ang), np.sin(ang), rng.uniform(-0.05, 0.05)]) * v
        acc = rng.uniform(-1, 1, 3) * [1, 1, 0.2]
        # last round's plan: rollout with small random jerks (dynamically consistent Euler/RK4 is not needed
        # for the neighbour buffer: only positions are read, agent_class.cpp:1147-1149)
        x = np.concatenate([pos - vel * dt, vel, acc])
        for i in range(N + 1):
            plans[k, i] = x
            j = rng.uniform(-5, 5, 3) * [1, 1, 0.2]
            x = np.concatenate([x[:3] + dt * x[3:6], x[3:6] + dt * x[6:9], x[6:9] + dt * j])
        if first_round:
            state[k] = np.concatenate([pos, np.zeros(6)])
        else:
            state[k] = plans[k, 1]
        p0 = state[k, :3]
        dirv = vel / (np.linalg.norm(vel) + 1e-9) if v > 0.3 else np.array([np.cos(ang), np.sin(ang), 0.0])
        path_vel = rng.uniform(4.5, 9.0)
        r = ref_from_path(p0, dirv, path_vel, dt, N)
        if turn:  # L-shaped path: after a few samples continue at 90 degrees
            kturn = int(rng.integers(1, max(2, N - 2)))
            perp = np.array([-dirv[1], dirv[0], 0.0])
            pts = [p0 + dirv * path_vel * dt * i for i in range(kturn + 1)]
            for i in range(kturn + 1, N + 1):
                pts.append(pts[kturn] + perp * path_vel * dt * (i - kturn))
            pts = np.array(pts)
            r[:, :3] = pts[:N]
            for i in range(N):
                dvec = pts[i] - pts[i + 1]
                r[i, 3:] = path_vel * dvec / np.linalg.norm(dvec)
        ref[k] = r
        # corridor: chain of boxes seeded along the reference, first one around p0
        hw = np.array([box_half, box_half, 1.2]) * (0.45 if narrow else 1.0)
        seeds = [p0.copy()]
        for i in range(N):
            pt = r[i, :3]
            if np.any(np.abs(pt - seeds[-1]) > hw * 0.75) and len(seeds) < P:
                seeds.append(pt.copy())
        polys = []
        for sd in seeds:
            A, b = box_rows(sd - hw, sd + hw)
            if chamfer:  # cut one corner with an un-normalised integer-slope row like GetPolyOcta3D emits
                nrm = np.array([rng.integers(1, 4), rng.integers(-3, 4), 0.0])
                corner = sd + hw * np.sign(nrm + 1e-9) * [1, 1, 0]
                A = np.vstack([A, nrm])
                b = np.append(b, nrm @ (corner - 0.35 * hw * np.sign(nrm + 1e-9) * [1, 1, 0]))
            polys.append((A, b))
        polys_all.append(polys)
    has_plan = np.ones(n_rob, dtype=np.uint8)
    if first_round:
        has_plan[:] = 0
    elif absent_frac > 0:
        has_plan[rng.random(n_rob) < absent_frac] = 0
    n_poly, n_rows, A, b = pack_static(polys_all, P, RS)
    return dict(agent_id=np.arange(n_rob, dtype=np.int32), state=state, ref=ref, n_poly=n_poly,
                n_rows=n_rows, A=A, b=b, plans=plans, has_plan=has_plan, polys=polys_all)


def circle_states(n, R=22.0, cx=18.0, cy=15.0, z=1.5):
    """start/goal of multi_agent_planner_circle.launch.py:36-44 (restated, the launch file needs ROS)."""
    starts, goals = [], []
    for k in range(n):
        a = 2 * np.pi * k / n
        starts.append([cx + R * np.cos(a), cy + R * np.sin(a), z])
    for k in range(n):
        goals.append(starts[(k + n // 2) % n])
    return np.array(starts), np.array(goals)


def level1_from_snapshot(prm, sn, planes_fn):
    """Fully formed per-step polyhedra poly_const_final_vec_[N][<=P] of a swarm snapshot: the static rows followed
    by the neighbour planes, appended to EVERY polyhedron (AddHyperplane, agent_class.cpp:1217-1234).
    planes_fn(agent) -> (planes[N][n_rob][4], valid[N][n_rob])."""
    N, P = prm.n_hor, prm.poly_hor
    n_inst = sn["state"].shape[0]
    n_rob = sn["plans"].shape[0]
    r_max = prm.max_rows_static + n_rob
    n_poly = np.zeros((n_inst, N), np.int32)
    n_rows = np.zeros((n_inst, N, P), np.int32)
    A = np.zeros((n_inst, N, P, r_max, 3))
    b = np.zeros((n_inst, N, P, r_max))
    for a in range(n_inst):
        planes, valid = planes_fn(a)
        for i in range(N):
            rows = planes[i][valid[i] > 0]
            n_poly[a, i] = min(P, len(sn["polys"][a]))
            for j, (Aj, bj) in enumerate(sn["polys"][a][:P]):
                r = len(bj) + len(rows)
                n_rows[a, i, j] = r
                A[a, i, j, :r] = np.vstack([Aj, rows[:, :3]])
                b[a, i, j, :r] = np.concatenate([bj, rows[:, 3]])
    return n_poly, n_rows, A, b


def constant_row_case(prm, oracle, mstep, delta):
    """One agent flying along +x in a big box; the polyhedron of step mstep - 1 carries one more row that the input-independent
    position p_mstep (Euler model, jerk inputs: p_1 and p_2 are fixed by the current state) violates by `delta`."""
    N, P = prm.n_hor, prm.poly_hor
    state = np.zeros((1, 9))
    state[0, :3] = (1.0, 2.0, 1.5)
    state[0, 3:6] = (2.0, 0.5, 0.0)
    state[0, 6:9] = (0.3, 0.0, 0.0)
    ref = ref_from_path(state[0, :3], np.array([1.0, 0.2, 0.0]) / np.hypot(1.0, 0.2), 2.0, prm.dt, N)[None]
    free = oracle.rollout(prm, state[0], np.zeros((N, 3)))       # p_1, p_2 of ANY trajectory
    nrm = state[0, 3:6] / np.linalg.norm(state[0, 3:6])
    Ab, bb = box_rows(np.array([-50.0, -50.0, -50.0]), np.array([50.0, 50.0, 50.0]))
    r_max = len(bb) + 1
    n_poly = np.ones((1, N), np.int32)
    n_rows = np.zeros((1, N, P), np.int32)
    A, b = np.zeros((1, N, P, r_max, 3)), np.zeros((1, N, P, r_max))
    for i in range(N):
        n_rows[0, i, 0] = len(bb)
        A[0, i, 0, :len(bb)], b[0, i, 0, :len(bb)] = Ab, bb
    i = mstep - 1
    A[0, i, 0, len(bb)], b[0, i, 0, len(bb)] = nrm, nrm @ free[mstep, :3] - delta
    n_rows[0, i, 0] = len(bb) + 1
    assert nrm @ free[i, :3] < b[0, i, 0, len(bb)] - 1e-3       # the other end point of the segment is well inside
    return state, ref, n_poly, n_rows, A, b
