"""CPU tests of the oracle (oracle/hdsm_oracle.c): golden fixtures, analytic known answers, KKT certificates,
exhaustive enumeration. The reference holds no vectors for this path (parity unpinned), so the
oracle is pinned against independent mathematics: tests/refmath.py (numpy restatement of the reference model,
separate code) + scipy solutions committed under tests/golden/ (generator: tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

import problems
import refmath as rm
from multi_agent_pkgs_amd.params import agile_params, default_params, make_params

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BIG = (np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1.0]]), np.full(6, 1e3))


def common_from_table(tab, N):
    return [tab[tab[:, 0] == i][:, 1:] for i in range(N)]


# ------------------------------------------------------------------------------------------- dynamics / objective
@pytest.mark.parametrize("rk4,drag", [(False, (0, 0, 0)), (True, (0, 0, 0)), (True, (0.2, 0.1, 0.4)), (False, (0.3, 0.3, 0.3))])
def test_rollout_and_objective_match_numpy_restatement(oracle, rk4, drag):
    prm = make_params(n_hor=9, rk4=rk4, drag=drag)
    rng = np.random.default_rng(3)
    state = rng.normal(size=9)
    ctrl = rng.uniform(-60, 60, size=(9, 3))
    ref = rng.normal(size=(9, 6))
    t_o = oracle.rollout(prm, state, ctrl)
    t_n = rm.rollout(prm, state, ctrl)
    assert np.abs(t_o - t_n).max() < 1e-12
    assert abs(oracle.objective(prm, t_o, ctrl, ref) - rm.objective(prm, t_n, ctrl, ref)) < 1e-9 * (1 + abs(rm.objective(prm, t_n, ctrl, ref)))


def test_euler_last_two_inputs_do_not_move_terminal_position(oracle):
    """with forward Euler dp_N/du_{N-1} = dp_N/du_{N-2} = 0."""
    prm = agile_params(10)
    z = np.zeros((10, 3))
    base = oracle.rollout(prm, np.zeros(9), z)
    for k in (8, 9):
        u = z.copy()
        u[k] = 1.0
        assert np.abs(oracle.rollout(prm, np.zeros(9), u)[10, :3] - base[10, :3]).max() == 0.0
    u = z.copy()
    u[7] = 1.0
    assert np.abs(oracle.rollout(prm, np.zeros(9), u)[10, :3]).max() > 0


# ------------------------------------------------------------------------------------------- separating planes
def test_planes_golden_hand_computed(oracle):
    g = np.load(os.path.join(GOLD, "planes.npz"))
    prm = agile_params(10, drone_radius=float(g["r"]), drone_z_offset=float(g["h"]), plane_perturb=float(g["p"]))
    for c, o, want in zip(g["c"], g["o"], g["plane"]):
        got = oracle.tasc_plane(prm, c, o)
        assert np.abs(got - want).max() < 1e-12
        assert np.abs(rm.tasc_plane_algebraic(prm, c, o) - want).max() < 1e-12


def test_planes_literal_chain_equals_closed_form(oracle):
    rng = np.random.default_rng(0)
    for r, h in [(0.25, 0.25), (0.25, 0.6), (0.125, 0.05)]:
        prm = agile_params(10, drone_radius=r, drone_z_offset=h)
        for _ in range(200):
            c = rng.normal(size=3) * 3
            o = c + rng.normal(size=3) * rng.choice([0.2, 1.0, 5.0])
            assert np.abs(oracle.tasc_plane(prm, c, o) - rm.tasc_plane_algebraic(prm, c, o)).max() < 1e-12
    # coincident agents: Eigen normalized() keeps the zero vector -> the row is 0 . p <= 0
    assert np.abs(oracle.tasc_plane(prm, [1, 2, 3], [1, 2, 3])).max() == 0.0


def test_plane_keeps_own_previous_position_feasible(oracle):
    """n_f . (c - q) = (min(2s,|d|) - |d|)/2 <= 0: the plane never cuts off the point it was built around."""
    prm = agile_params(10)
    rng = np.random.default_rng(1)
    for _ in range(100):
        c = rng.normal(size=3)
        o = c + rng.normal(size=3) * rng.choice([0.1, 2.0])
        pl = oracle.tasc_plane(prm, c, o)
        assert pl[:3] @ c - pl[3] <= 1e-12


# ------------------------------------------------------------------------------------------- QP: known answers
def test_unconstrained_tracking_closed_form(oracle):
    """With every bound absent except the terminal equalities, u = argmin over the null space: compare with
    the KKT solve of the numpy restatement."""
    prm = make_params(n_hor=8, max_vel=1e100, max_acc_xy=1e100, min_acc_xy=-1e100, max_acc_z=1e100,
                      min_acc_z=-1e100, max_jerk=1e100)
    state = np.array([0.3, -0.2, 1.5, 1.0, 0.5, 0.0, 0.2, 0.0, -0.1])
    ref = problems.ref_from_path(state[:3], [1, 0.3, 0], 5.0, prm.dt, 8)
    cor = oracle.Corridor([[BIG]] * 8)
    r = oracle.miqp(prm, state, ref, cor)
    H, g, f0, T0, T = rm.quad_form(prm, state, ref)
    Aeq, beq, Ain, bin_, _ = rm.linear_rows(prm, state, {})
    n = len(g)
    K = np.block([[H, Aeq.T], [Aeq, np.zeros((6, 6))]])
    sol = np.linalg.solve(K, np.concatenate([-g, beq]))
    assert r["status"] == 0
    assert np.abs(r["ctrl"].reshape(-1) - sol[:n]).max() < 1e-8
    assert abs(r["obj"] - (0.5 * sol[:n] @ H @ sol[:n] + g @ sol[:n] + f0)) < 1e-8


def test_passing_agents_point_mirror_symmetry(oracle):
    """Two agents passing each other with point-mirrored inputs (reflection about (0, 0, 1.5)) must produce
    point-mirrored optimal plans: the plane construction is odd in the normal (n -> -n flips n_f, including
    the perturbation terms of agent_class.cpp:1173-1200), dynamics, bounds and costs are symmetric."""
    prm = agile_params(8)
    N = 8
    plans = np.zeros((2, N + 1, 9))
    for i in range(N + 1):
        plans[0, i, :3] = [-2 + 0.5 * i, -0.3, 1.4]
        plans[0, i, 3] = 5.0
    centre = np.array([0, 0, 1.5])
    plans[1, :, :3] = 2 * centre - plans[0, :, :3]
    plans[1, :, 3:] = -plans[0, :, 3:]
    state = plans[:, 1].copy()
    # both references run along the line between the two plans: tracking pulls the agents onto the planes
    ref0 = problems.ref_from_path(np.array([state[0, 0], 0.0, 1.5]), [1, 0, 0], 5.0, prm.dt, N)
    ref1 = ref0.copy()
    ref1[:, :3] = 2 * centre - ref0[:, :3]
    ref1[:, 3:] = -ref0[:, 3:]
    A, b = BIG
    n_poly, n_rows, As, bs = problems.pack_static([[(A, b)], [(A, b)]], prm.poly_hor, prm.max_rows_static)
    o = oracle.replan(prm, [0, 1], state, np.stack([ref0, ref1]), n_poly, n_rows, As, bs, plans, [1, 1])
    assert (o["status"] == 0).all()
    mirrored = -o["traj"][0].copy()
    mirrored[:, :3] += 2 * centre
    assert np.abs(o["traj"][1] - mirrored).max() < 1e-8
    assert abs(o["obj"][0] - o["obj"][1]) < 1e-9 * max(1, abs(o["obj"][0]))
    # and the separating planes actually matter in this geometry
    free = oracle.replan(prm, [0, 1], state, np.stack([ref0, ref1]), n_poly, n_rows, As, bs, plans, [0, 0])
    assert np.abs(free["traj"][0] - o["traj"][0]).max() > 1e-3


# ------------------------------------------------------------------------------------------- golden: scipy QPs
def test_fixed_assignment_qps_match_scipy_golden(oracle):
    g = np.load(os.path.join(GOLD, "qp_scipy.npz"))
    n = int(g["n_cases"])
    assert n >= 8
    for k in range(n):
        N = int(g[f"c{k}_N"])
        prm = make_params(n_hor=N, rk4=bool(g[f"c{k}_rk4"]), drag=tuple(g[f"c{k}_drag"]), max_rows_static=18)
        common = common_from_table(g[f"c{k}_common"], N)
        poly = (g[f"c{k}_polyA"], g[f"c{k}_polyb"])
        cor = oracle.Corridor([[poly]] * N, common)
        r = oracle.qp_fixed(prm, g[f"c{k}_state"], g[f"c{k}_ref"], cor, [0] * N)
        assert r["status"] == 0
        # scipy's interior-point answer is good to ~1e-6 relative on the objective; the oracle must not be worse
        assert r["obj"] <= float(g[f"c{k}_obj"]) + 1e-6 * max(1.0, abs(float(g[f"c{k}_obj"])))
        assert abs(r["obj"] - float(g[f"c{k}_obj"])) < 1e-5 * max(1.0, abs(float(g[f"c{k}_obj"])))
        assert np.abs(r["traj"] - rm.rollout(prm, g[f"c{k}_state"], g[f"c{k}_u"])).max() < 1e-4


def test_small_miqps_match_exhaustive_scipy_golden(oracle):
    path = os.path.join(GOLD, "miqp_small.npz")
    if not os.path.exists(path):
        pytest.skip("miqp_small.npz not generated")
    g = np.load(path)
    N, P = int(g["N"]), int(g["P"])
    prm = make_params(n_hor=N, poly_hor=P, max_rows_static=18)
    assert int(g["n_cases"]) >= 2
    for k in range(int(g["n_cases"])):
        polys = [(g[f"m{k}_A{j}"], g[f"m{k}_b{j}"]) for j in range(int(g[f"m{k}_npoly"]))]
        cor = oracle.Corridor([polys] * N, common_from_table(g[f"m{k}_common"], N))
        r = oracle.miqp(prm, g[f"m{k}_state"], g[f"m{k}_ref"], cor)
        assert r["status"] == 0
        want = float(g[f"m{k}_obj"])
        assert abs(r["obj"] - want) < 1e-5 * max(1.0, abs(want))
        if float(g[f"m{k}_second"]) - want > 1e-3 * max(1.0, abs(want)):  # unique optimum: trajectories agree
            assert np.abs(r["traj"] - rm.rollout(prm, g[f"m{k}_state"], g[f"m{k}_u"])).max() < 1e-4


# ------------------------------------------------------------------------------------------- KKT certificates
@pytest.mark.parametrize("kw", [dict(seed=1), dict(seed=2, turn=True), dict(seed=3, narrow=True, turn=True),
                                dict(seed=5, chamfer=True, narrow=True, turn=True), dict(seed=6, first_round=True)])
def test_oracle_results_carry_kkt_certificates(oracle, kw):
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 9, **kw)
    o = oracle.replan(prm, *[sn[k] for k in ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")])
    N = prm.n_hor
    checked = 0
    for a in range(9):
        if o["status"][a] != 0:
            continue
        planes, valid = oracle.tasc_planes(prm, a, sn["state"][a], sn["plans"], sn["has_plan"])
        common = [planes[i][valid[i] > 0] for i in range(N)]
        polys = [sn["polys"][a][: prm.poly_hor]] * N
        # recover a feasible assignment from the returned trajectory (ties are fine: any containing polyhedron)
        assign = []
        for i in range(N):
            js = [j for j, (A, b) in enumerate(polys[i])
                  if (A @ o["traj"][a, i, :3] - b).max() <= (1e-6 if i == 0 else 1e-8) and (A @ o["traj"][a, i + 1, :3] - b).max() <= 1e-8]
            assert js, "returned trajectory violates the corridor"
            assign.append(js[0])
        cert = rm.certify(prm, sn["state"][a], sn["ref"][a], o["ctrl"][a], rm.rows_for_assignment(N, polys, assign, common))
        assert cert["primal_eq"] < 1e-9 and cert["primal_in"] < 1e-8
        assert abs(cert["obj"] - o["obj"][a]) < 1e-8 * max(1, abs(cert["obj"]))
        assert cert["stationarity"] < 1e-7 * max(1.0, cert["grad_norm"]), cert
        assert np.abs(rm.rollout(prm, sn["state"][a], o["ctrl"][a]) - o["traj"][a]).max() < 1e-12
        checked += 1
    assert checked >= 3


# ------------------------------------------------------------------------------------------- branch and bound
def test_branch_and_bound_equals_enumeration(oracle):
    prm = make_params(n_hor=5, poly_hor=3, max_rows_static=18)
    n_cmp = 0
    for seed in range(20, 32):
        sn = problems.swarm_snapshot(prm, 4, seed, narrow=True, turn=True, box_half=1.3, speed=(1.0, 6.0))
        for a in range(4):
            planes, valid = oracle.tasc_planes(prm, a, sn["state"][a], sn["plans"], sn["has_plan"])
            common = [planes[i][valid[i] > 0] for i in range(5)]
            cor = oracle.Corridor([sn["polys"][a][:3]] * 5, common)
            r1 = oracle.miqp(prm, sn["state"][a], sn["ref"][a], cor)
            r2 = oracle.miqp_enum(prm, sn["state"][a], sn["ref"][a], cor)
            assert r1["status"] == r2["status"]
            if r1["status"] == 0:
                assert abs(r1["obj"] - r2["obj"]) < 1e-7 * max(1, abs(r2["obj"]))
                if r2["runner_up"] - r2["obj"] > 1e-6 * max(1, abs(r2["obj"])):
                    assert np.abs(r1["traj"] - r2["traj"]).max() < 1e-6
                n_cmp += 1
    assert n_cmp >= 20


def test_level1_literal_equals_level2_common_rows(oracle):
    """Appending the neighbour planes to EVERY polyhedron (what AddHyperplane does, agent_class.cpp:1217-1234) and
    treating them as rows common to the step give the same optimum."""
    prm = make_params(n_hor=6, poly_hor=3, max_rows_static=18)
    N, P = 6, 3
    sn = problems.swarm_snapshot(prm, 6, 41, narrow=True, turn=True, spacing=1.6)
    args = [sn[k] for k in ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")]
    o2 = oracle.replan(prm, *args)
    r_max = prm.max_rows_static + 6
    n_poly = np.zeros((6, N), np.int32)
    n_rows = np.zeros((6, N, P), np.int32)
    A = np.zeros((6, N, P, r_max, 3))
    b = np.zeros((6, N, P, r_max))
    for a in range(6):
        planes, valid = oracle.tasc_planes(prm, a, sn["state"][a], sn["plans"], sn["has_plan"])
        for i in range(N):
            rows = planes[i][valid[i] > 0]
            n_poly[a, i] = min(P, len(sn["polys"][a]))
            for j, (Aj, bj) in enumerate(sn["polys"][a][:P]):
                r = len(bj) + len(rows)
                n_rows[a, i, j] = r
                A[a, i, j, :r] = np.vstack([Aj, rows[:, :3]])
                b[a, i, j, :r] = np.concatenate([bj, rows[:, 3]])
    o1 = oracle.solve(prm, sn["state"], sn["ref"], n_poly, n_rows, A, b)
    assert (o1["status"] == o2["status"]).all()
    ok = o2["status"] == 0
    assert ok.sum() >= 3
    assert np.abs(o1["obj"] - o2["obj"])[ok].max() < 1e-7 * np.abs(o2["obj"][ok]).max()
    assert np.abs(o1["traj"] - o2["traj"])[ok].max() < 1e-6


def test_infeasible_and_outputs_untouched(oracle):
    prm = agile_params(6, max_rows_static=18)
    A, b = problems.box_rows(np.array([5, 5, 0.0]), np.array([6, 6, 3.0]))  # current position is outside
    n_poly, n_rows, As, bs = problems.pack_static([[(A, b)]], prm.poly_hor, prm.max_rows_static)
    state = np.array([[0, 0, 1.5, 0, 0, 0, 0, 0, 0.0]])
    ref = problems.ref_from_path(state[0, :3], [1, 0, 0], 5.0, prm.dt, 6)[None]
    o = oracle.replan(prm, [0], state, ref, n_poly, n_rows, As, bs, np.zeros((1, 7, 9)), [0])
    assert o["status"][0] == 2 and (o["traj"] == 0).all()


def test_default_config_is_solved(oracle):
    prm = default_params(9, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 6, 77, speed=(0, 4))
    o = oracle.replan(prm, *[sn[k] for k in ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")])
    assert (o["status"] == 0).sum() >= 4


# ---- next row f4: map pre-processing (orc_map_preprocess) --------------------------------------------------------
def test_map_preprocess_single_obstacle_matches_the_formulas(oracle):
    """One occupied voxel in a free grid: after InflateObstacles + CreatePotentialField the value of a voxel depends on
    its offset n to the NEAREST inflated voxel only; recomputed here from CreateMask's formulas (voxel_grid.cpp:192-226)
    with numpy: member iff |hypot(n) - 1| * res < dist, value int8(100 (1 - hypot(n) / (rn + 1))^pow)."""
    from multi_agent_pkgs_amd.params import default_map_config
    cfg = default_map_config()
    g = np.zeros((1, 24, 40, 40), np.int8)
    g[0, 12, 20, 20] = 100
    out = oracle.map_preprocess(cfg, g)[0]
    res = cfg.voxel_size
    zz, yy, xx = np.meshgrid(np.arange(24) - 12, np.arange(40) - 20, np.arange(40) - 20, indexing="ij")
    hyp = np.hypot(np.hypot(xx, yy), zz)
    infl = (np.abs(hyp - 1) * res < cfg.inflation_dist) & (hyp > 0) | (hyp == 0)
    assert ((out == 100) == infl).all() and infl.sum() == 27
    # potential: max over inflated voxels u of the mask value at (v - u)
    rn = int(np.ceil(cfg.potential_dist / res))
    want = np.where(infl, 100, 0).astype(np.int64)
    occ = np.argwhere(infl)
    for dz in range(-rn, rn + 1):
        for dy in range(-rn, rn + 1):
            for dx in range(-rn, rn + 1):
                h = np.hypot(np.hypot(dx, dy), dz)
                if abs(h - 1) * res >= cfg.potential_dist:
                    continue
                val = 100.0 * (1 - h / (rn + 1)) ** cfg.potential_pow
                if not val > 1e-3:
                    continue
                v = occ + [dz, dy, dx]
                ok = (v >= 0).all(1) & (v < [24, 40, 40]).all(1)
                v = v[ok]
                want[v[:, 0], v[:, 1], v[:, 2]] = np.maximum(want[v[:, 0], v[:, 1], v[:, 2]], int(val))
    assert (out == want).all()


def test_map_preprocess_unknown_voxels(oracle):
    """SetUncertainToUnknown (map_builder.cpp:331-362): an interior unknown voxel makes its non-occupied cube
    neighbours unknown; unknown voxels receive no potential but are overwritten by the inflation."""
    from multi_agent_pkgs_amd.params import default_map_config
    cfg = default_map_config()
    g = np.zeros((1, 12, 16, 16), np.int8)
    g[0, 6, 8, 8] = -1       # interior unknown voxel
    g[0, 6, 8, 9] = 100      # an occupied neighbour: stays occupied
    g[0, 0, 0, 0] = -1       # unknown voxel on the border: does not spread (loop bounds cube .. dim - cube)
    out = oracle.map_preprocess(cfg, g)[0]
    assert out[6, 8, 9] == 100 and out[6, 8, 8] == 100      # the unknown voxel itself is within the inflation
    assert out[5, 7, 7] == -1 and out[7, 9, 7] == -1         # cube neighbours out of the inflation's reach: unknown
    assert out[0, 0, 0] == -1 and out[1, 1, 1] == 0          # border unknown voxel did not spread
    assert out[6, 8, 11] > 0 and out[6, 8, 11] < 100         # known voxel near the obstacle: potential


# ------------------------------------------------------------------------------------------- verification mode
def test_hinted_search_returns_the_same_optimum_and_never_trusts_the_hint(oracle):
    """orc_replan_hinted (a claimed objective as the initial cut-off): with the TRUE optimum as the claim the answer is the
    one of the plain search, found with fewer nodes; a claim below the optimum finds nothing; a claim above it is ignored."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 25, seed=35, spacing=1.0, narrow=True, turn=True)
    args = [sn[k] for k in ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")]
    plain = oracle.replan(prm, *args, n_threads=4)
    ok = plain["status"] == 0
    assert ok.sum() >= 10 and plain["nodes"][ok].max() > 3
    hint = np.where(ok, plain["obj"], np.nan)
    hinted = oracle.replan(prm, *args, n_threads=4, obj_hint=hint)
    assert (hinted["status"] == plain["status"]).all()
    assert np.abs(hinted["traj"] - plain["traj"])[ok].max() < 1e-9
    assert (hinted["nodes"] <= plain["nodes"]).all()
    low = oracle.replan(prm, *args, n_threads=4, obj_hint=np.where(ok, plain["obj"] * (1 - 1e-3) - 1e-3, np.nan))
    assert (low["status"][ok] == 2).all()
    high = oracle.replan(prm, *args, n_threads=4, obj_hint=np.where(ok, plain["obj"] * 2 + 1, np.nan))
    assert (high["status"] == plain["status"]).all() and np.abs(high["traj"] - plain["traj"])[ok].max() < 1e-9


def test_both_search_orders_of_the_oracle_agree(oracle):
    """orc_replan_ex search = 1 (branch on the most infeasible uncontained step, leaves at any depth) against the default
    step-ordered search on tight snapshots with real trees: same statuses, trajectories, objectives and poly_used."""
    for prm, n_rob, seed, kw in ((agile_params(10, max_rows_static=18), 25, 35, dict(spacing=1.0, narrow=True, turn=True)),
                                 (make_params(n_hor=12, rk4=True, drag=(0.1, 0.0, 0.3), max_rows_static=18, poly_hor=3), 16, 7,
                                  dict(spacing=1.3, narrow=True, turn=True, chamfer=True))):
        sn = problems.swarm_snapshot(prm, n_rob, seed=seed, **kw)
        args = [sn[k] for k in ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")]
        a = oracle.replan(prm, *args, n_threads=4)
        b = oracle.replan(prm, *args, n_threads=4, search=1)
        assert (a["status"] == b["status"]).all() and (a["status"] != 1).all()
        ok = a["status"] == 0
        assert ok.sum() >= 5 and a["nodes"][ok].max() > 20
        assert np.abs(a["traj"] - b["traj"])[ok].max() < 1e-8
        assert (np.abs(a["obj"] - b["obj"])[ok] / np.maximum(1, np.abs(a["obj"][ok]))).max() < 1e-9


# ------------------------------------------------------------------------------------------- enumerated MIQPs
def enum_cases():
    path = os.path.join(GOLD, "miqp_enum.npz")
    g = np.load(path)
    for k in range(int(g["n_cases"])):
        yield k, {key[len(f"e{k}_"):]: g[key] for key in g.files if key.startswith(f"e{k}_")}


def enum_snapshot(c):
    """The replan inputs of an enumerated case in the ABI layouts (one instance: the agent the case was built for)."""
    N, P = int(c["N"]), int(c["P"])
    prm = make_params(n_hor=N, poly_hor=P, max_rows_static=18)
    a = int(c["agent"])
    polys = [(c["polyA"][j], c["polyb"][j]) for j in range(int(c["npoly"]))]
    n_poly, n_rows, A, b = problems.pack_static([polys], P, prm.max_rows_static)
    args = (np.array([a], np.int32), c["state"][None], c["ref"][None], n_poly, n_rows, A, b, c["plans"], c["has_plan"])
    return prm, polys, args


def test_enumerated_miqps_pin_the_combinatorial_part(oracle):
    """tests/golden/miqp_enum.npz: six N = 6 / P = 3 cases with all 729 assignments resolved and three N = 10 / P = 4 cases with
    every admissible assignment resolved — scipy leaf solutions with KKT certificates, infeasibility by LP proofs, planes from
    refmath (generator: tests/golden/make_golden.py enum). The oracle's branch and bound (both search orders), its own
    enumerator, and its fixed-assignment QP on the recorded leaves must reproduce them."""
    n6 = n10 = 0
    rng = np.random.default_rng(0)
    for k, c in enum_cases():
        prm, polys, args = enum_snapshot(c)
        N = prm.n_hor
        total, feasible = int(c["counts"][0]), int(c["counts"][1])
        assert total == len(polys) ** N and feasible == len(c["leaf_obj"]) and feasible + int(c["counts"][2]) + int(c["counts"][3]) == total
        want, second = float(c["obj"]), float(c["second"])
        cor = oracle.Corridor([polys] * N, common_from_table(c["common"], N))
        r = oracle.miqp(prm, c["state"], c["ref"], cor)
        assert r["status"] == 0 and abs(r["obj"] - want) < 1e-6 * max(1.0, abs(want)), (k, r["obj"], want)
        unique = second - want > 1e-3 * max(1.0, abs(want))
        if unique:
            assert np.abs(r["traj"] - rm.rollout(prm, c["state"], c["u"])).max() < 1e-4, k
        for search in (0, 1):   # level 2 from the snapshot: the oracle's own planes this time, both search orders
            o = oracle.replan(prm, *args, search=search)
            assert o["status"][0] == 0 and abs(o["obj"][0] - want) < 1e-6 * max(1.0, abs(want)), (k, search)
        leaves = range(feasible) if N == 6 else rng.choice(feasible, min(feasible, 40), replace=False)
        for t in leaves:
            q = oracle.qp_fixed(prm, c["state"], c["ref"], cor, c["leaf_assign"][t].astype(np.int32))
            assert q["status"] == 0 and abs(q["obj"] - c["leaf_obj"][t]) < 1e-6 * max(1.0, abs(c["leaf_obj"][t])), (k, t)
        if N == 6:
            e = oracle.miqp_enum(prm, c["state"], c["ref"], cor)
            assert e["status"] == 0 and abs(e["obj"] - want) < 1e-6 * max(1.0, abs(want)) and e["nodes"] == total
            n6 += 1
        else:
            n10 += 1
    assert n6 >= 6 and n10 >= 3


@pytest.mark.parametrize("mstep", [1, 2])
def test_rows_on_input_independent_positions_are_judged_with_feas_tol_fixed(oracle, mstep):
    """Under Euler p_1 and p_2 do not depend on the inputs (jerk inputs reach the position after three steps): a row there is a
    constant. Within feas_tol_fixed (Gurobi's FeasibilityTol, 1e-6) it holds and changes nothing; beyond it nothing satisfies it."""
    prm = make_params(n_hor=8, poly_hor=2, max_rows_static=18)
    base = None
    for delta, want in ((-1.0, 0), (1e-8, 0), (1e-7, 0), (1e-5, 2)):
        o = oracle.solve(prm, *problems.constant_row_case(prm, oracle, mstep, delta))
        assert o["status"][0] == want, (mstep, delta, o["status"])
        if want == 0:
            base = o["traj"].copy() if base is None else base
            assert np.abs(o["traj"] - base).max() < 1e-12
    # with RK4 every position depends on the inputs: the same row is an ordinary constraint (tolerance solver_tol) that the
    # solver satisfies by moving p_mstep
    prm4 = make_params(n_hor=8, poly_hor=2, max_rows_static=18, rk4=True)
    args = problems.constant_row_case(prm4, oracle, mstep, 1e-5)
    o = oracle.solve(prm4, *args)
    assert o["status"][0] == 0
    A, b = args[4], args[5]
    r = int(args[3][0, mstep - 1, 0]) - 1
    assert A[0, mstep - 1, 0, r] @ o["traj"][0, mstep, :3] - b[0, mstep - 1, 0, r] < 1e-8


def test_cpu_port_of_the_product_algorithm_matches_the_oracle(oracle):
    """oracle/hdsm_cpu_port.c (bench.py's cpu_baseline_warm leg: lazy closed-form planes, sphere prefilter, normalised pick rule, warm
    start) gives the oracle's answers — cold, and warm-started from its own previous working sets on a second, shifted snapshot."""
    from oracle import pycpuport as port
    prm = agile_params(10, max_rows_static=18)
    keys = ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")
    for kw in (dict(seed=3, turn=True), dict(seed=11, spacing=1.3), dict(seed=8, narrow=True, turn=True, spacing=1.5)):
        sn = problems.swarm_snapshot(prm, 16, **kw)
        args = [sn[k] for k in keys]
        o = oracle.replan(prm, *args, n_threads=8)
        store = port.new_warm_store(16)
        for rep in range(2):   # second call: warm-started from the sets the first one stored (a wrong guess must not matter either)
            g = port.replan(prm, *args, warm=store, n_threads=4)
            assert (g["status"] == o["status"]).all(), (kw, rep, g["status"], o["status"])
            ok = o["status"] != 2
            assert np.abs(g["traj"] - o["traj"])[ok].max() < 1e-7 and np.abs(g["obj"] - o["obj"])[ok].max() < 1e-6 * max(1.0, np.abs(o["obj"][ok]).max())
        assert (store[:, 0] > 0).any()
