"""CPU tests of the DEVICE source of the solver kernel. tests/wave_emu compiles multi_agent_pkgs_amd/csrc/hdsm_core.h +
hdsm_wave_gi.h in DEVICE mode with g++ against a stand-in for <hip/hip_runtime.h> and runs one workgroup = one wavefront as 64
fibers in lockstep (every DPP move, v_readlane, v_permlane32_swap, ballot and wsync() is a rendezvous of all lanes, and a
cross-lane operation reached by only some of the lanes is an error). What the -m gpu tests check on hardware for the
register-resident active set — split rows, Householder add / drop, warm start and certificates, conflict learning, the
sweeps on packed positions, the sphere prefilter — is checked here against the oracle without a GPU."""
import numpy as np
import pytest

import problems
from multi_agent_pkgs_amd.params import agile_params, make_params

ARG_KEYS = ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")


@pytest.fixture(scope="module")
def wave():
    from wave_emu import pywave
    pywave.lib()
    return pywave


def compare(e, o, tol=1e-8):
    assert (e["status"] == o["status"]).all(), (e["status"].tolist(), o["status"].tolist())
    ok = o["status"] != 2
    if ok.any():
        assert np.abs(e["traj"] - o["traj"])[ok].max() < tol
        assert (np.abs(e["obj"] - o["obj"])[ok] / np.maximum(1, np.abs(o["obj"][ok]))).max() < 1e-8


CASES = [dict(n_rob=12, seed=1), dict(n_rob=12, seed=2, turn=True), dict(n_rob=12, seed=3, narrow=True, turn=True),
         dict(n_rob=16, seed=4, spacing=1.0), dict(n_rob=12, seed=5, chamfer=True, narrow=True, turn=True),
         dict(n_rob=12, seed=6, first_round=True), dict(n_rob=24, seed=8, absent_frac=0.3, spacing=1.5)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_device_source_matches_oracle_h10(wave, oracle, case):
    """n = 30: the kernel whose rows of J are split over two lanes (NV = 32)."""
    prm = agile_params(10, max_rows_static=18)
    case = dict(case)
    sn = problems.swarm_snapshot(prm, case.pop("n_rob"), case.pop("seed"), **case)
    args = [sn[k] for k in ARG_KEYS]
    compare(wave.replan(prm, *args), oracle.replan(prm, *args, n_threads=8))


@pytest.mark.parametrize("n_hor,rk4,drag", [(15, False, (0, 0, 0)), (9, True, (0.1, 0.1, 0.3)), (12, True, (0, 0, 0)), (7, False, (0.2, 0.1, 0))])
def test_device_source_matches_oracle_other_configs(wave, oracle, n_hor, rk4, drag):
    """Other horizons (n > 30: one lane per row, NV = 48; n < 30: padded factors), RK4 and drag."""
    prm = make_params(n_hor=n_hor, rk4=rk4, drag=drag, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 10, seed=100 + n_hor, turn=True)
    args = [sn[k] for k in ARG_KEYS]
    compare(wave.replan(prm, *args), oracle.replan(prm, *args, n_threads=8))


def test_warm_start_and_gridlock_exit_do_not_change_the_answer(wave, oracle):
    """Consecutive replans on one warm-start store (what a handle keeps between launches): the second replan of the SAME
    inputs is seeded with the first one's working sets and must return the same results; a store filled by a DIFFERENT
    snapshot (a wrong guess) must not change the results either. The infeasible instances of this tight snapshot are all
    gridlocked — a neighbour's plane cuts off everything p_1 can reach under the jerk limit — and end on the reachability
    test of their first sweep without a single active-set operation; the store remembers them (certificate bit, no rows) and
    the next replan sweeps first."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 16, seed=4, spacing=1.0)
    args = [sn[k] for k in ARG_KEYS]
    o = oracle.replan(prm, *args, n_threads=8)
    bad = o["status"] == 2
    assert bad.any() and (~bad).any()
    store = wave.new_warm_store(16)
    cold = wave.replan(prm, *args, warm=store)
    compare(cold, o)
    assert (cold["qp_iters"][bad] == 0).all() and (cold["sweeps"][bad] == 1).all()
    assert (store[bad, 0] == (1 << 30)).all() and (store[~bad, 0] > 0).all() and (store[~bad, 0] < (1 << 30)).all()
    warm = wave.replan(prm, *args, warm=store)
    compare(warm, o)
    assert (warm["qp_iters"][bad] == 0).all() and (warm["sweeps"][bad] == 1).all()
    other = problems.swarm_snapshot(prm, 16, seed=2, turn=True)
    wrong = wave.replan(prm, *[other[k] for k in ARG_KEYS], warm=store)          # store still holds snapshot 4's sets
    compare(wrong, oracle.replan(prm, *[other[k] for k in ARG_KEYS], n_threads=8))


def test_infeasibility_certificates_are_handed_over(wave, oracle):
    """An instance whose ROOT relaxation is infeasible without being gridlocked (here: it starts far above the velocity limit
    and the jerk limit cannot bring it back inside the state box in time — no neighbour involved) still goes through the dual
    method; its certificate (working set + the row that could not join it) seeds the next replan, which must come to the
    same verdict in about as many operations as the certificate has rows (with the normalised pick rule the cold proof of this
    case is itself only four operations long)."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 1, seed=11)                                # alone: no neighbour rows at all
    k = 0
    sn["state"] = sn["state"].copy()
    sn["state"][k, 3] = 40.0                                                     # v_x far above max_vel
    args = [sn[key] for key in ARG_KEYS]
    o = oracle.replan(prm, *args, n_threads=8)
    assert o["status"][k] == 2
    store = wave.new_warm_store(1)
    cold = wave.replan(prm, *args, warm=store)
    compare(cold, o)
    assert cold["qp_iters"][k] > 0 and cold["nodes"][k] == 1
    assert (store[k, 0] & (1 << 30)) != 0 and (store[k, 0] & 0xffff) > 0          # certificate with its rows
    warm = wave.replan(prm, *args, warm=store)
    compare(warm, o)
    assert 0 < warm["qp_iters"][k] <= max(cold["qp_iters"][k], int(store[k, 0] & 0xffff))


def test_sphere_prefilter_stages_the_same_rows(wave, oracle):
    """With the sphere records (swarms >= bounds_min agents) the sweeps skip whole neighbours and the automatic pre-sweep is
    left to the instances whose warm start holds neighbour rows; the answer is that of the step-by-step test."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 40, seed=8, absent_frac=0.2, spacing=1.5)
    args = [sn[k] for k in ARG_KEYS]
    plain = wave.replan(prm, *args, bounds_min=10 ** 6)
    pre = wave.replan(prm, *args, bounds_min=1)
    compare(pre, oracle.replan(prm, *args, n_threads=8))
    assert np.array_equal(pre["status"], plain["status"]) and np.abs(pre["traj"] - plain["traj"]).max() < 1e-12


def test_branch_and_bound_with_conflict_learning(wave, oracle):
    """Corridors that force real branching (narrow boxes, turning paths): the device's lazy B&B with conflict learning
    returns the oracle's optimum (the oracle enumerates depth-first without learning)."""
    prm = agile_params(10, max_rows_static=18)
    nodes = 0
    for seed in (3, 5, 21):
        sn = problems.swarm_snapshot(prm, 10, seed, narrow=True, turn=True, chamfer=(seed % 2 == 1))
        args = [sn[k] for k in ARG_KEYS]
        e = wave.replan(prm, *args)
        compare(e, oracle.replan(prm, *args, n_threads=8))
        nodes += int(e["nodes"].sum())
    assert nodes > 3 * 10


@pytest.mark.parametrize("budget", [1, 2, 5])
def test_subtree_splitting_gives_the_unsplit_answer(wave, oracle, budget):
    """The split launch of hdsm_api.hip, run here one workgroup after the other on the device source: pass 1 with a node budget —
    an instance that exceeds it writes a hand-over record (open levels, staged rows, conflicts) and queues one item per open child —
    then every item (set-up, record, snapshot of its level, the search continues inside the child's subtree, pruning against the
    instance's shared incumbent), then the merge. With budgets of one, two and five nodes the hand-over happens at different depths
    of the dive (items on one level, on several levels, with and without an incumbent); the answer is the oracle's every time."""
    prm = agile_params(10, max_rows_static=18)
    handed = 0
    for seed in (3, 5):
        sn = problems.swarm_snapshot(prm, 10, seed, narrow=True, turn=True, chamfer=(seed % 2 == 1))
        args = [sn[k] for k in ARG_KEYS]
        plain = wave.replan(prm, *args)
        e = wave.replan(prm, *args, split_budget=budget)
        compare(e, oracle.replan(prm, *args, n_threads=8))
        assert np.array_equal(e["status"], plain["status"]) and np.abs(e["traj"] - plain["traj"]).max() < 1e-9
        handed += int((plain["nodes"] > budget).sum())
    assert handed >= 3


def test_node_budget_is_reported(wave, oracle):
    """hdsm_params.max_nodes = 1 on corridors that need branching: the instances that would branch come back as HDSM_LIMIT or
    HDSM_NO_SOLUTION with HDSM_FLAG_NODE_LIMIT, never as a wrong optimum; the others are untouched."""
    prm = agile_params(10, max_rows_static=18, max_nodes=1)
    full = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 10, 3, narrow=True, turn=True)
    args = [sn[k] for k in ARG_KEYS]
    e, o = wave.replan(prm, *args), oracle.replan(full, *args, n_threads=8)
    limited = (e["flags"] & 1) != 0
    assert limited.any()
    assert (e["status"][limited] != 0).all()
    same = ~limited
    assert (e["status"][same] == o["status"][same]).all()
    ok = same & (o["status"] == 0)
    assert np.abs(e["traj"][ok] - o["traj"][ok]).max() < 1e-8


@pytest.mark.parametrize("threads,cmax", [(64, 0), (128, 256)], ids=["one-wave", "two-waves-small-lds"])
def test_closed_loop_with_a_persistent_warm_start_store(wave, oracle, threads, cmax):
    """Consecutive rounds of a closed loop (host mirror around the solve): the device source, warm-started every round from
    the store it filled the round before (working sets moved one step towards the present, gridlock / certificate marks),
    returns what the cold-started oracle returns on the same inputs, round after round. With two wavefronts (the shape of the
    four-per-CU kernel, small LDS layout) wave 1 runs the first staging sweep — around the own previous plan — WHILE wave 0 installs
    the guess (round 5); one wavefront keeps the sequential order."""
    from multi_agent_pkgs_amd import swarm
    n = 10
    prm = agile_params(10, max_rows_static=18)
    store = wave.new_warm_store(n)
    seen = []

    def solve(inp, plans, has):
        args = [inp[k] for k in ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b")] + [plans, has]
        g = wave.replan(prm, *args, warm=store, threads=threads, cmax=cmax)
        o = oracle.replan(prm, *args, n_threads=8)
        compare(g, o, tol=1e-7)
        seen.append((int(g["qp_iters"].sum()), int((g["status"] == 2).sum())))
        return g

    loop = swarm.SwarmLoop(prm, swarm.default_swarm_config(), n, solve=solve, radius=4.0)   # tight: the ring gridlocks early on
    for _ in range(20):
        loop.step()
    assert len(seen) == 20 and (store[:, 0] != 0).any()
    assert sum(s[1] for s in seen) >= 8 and seen[-1][1] == 0                              # instances without a solution, then recovery
    # warm rounds need fewer operations than the cold first one did per agent on a comparable problem
    assert min(s[0] for s in seen[1:]) < seen[0][0]


@pytest.mark.parametrize("case", [dict(n_rob=12, seed=2, turn=True), dict(n_rob=12, seed=5, chamfer=True, narrow=True, turn=True),
                                  dict(n_rob=16, seed=4, spacing=1.0)], ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_four_wavefront_workgroup_matches_oracle(wave, oracle, case):
    """The product's default launch shape: 256 threads = four wavefronts per instance. Wave 0 iterates, the other three wait at
    the workgroup barrier and share the set-up, the sweeps and the leaf test (the emulator runs 256 fibers: cross-lane
    operations meet within a wavefront, __syncthreads() across the workgroup)."""
    prm = agile_params(10, max_rows_static=18)
    case = dict(case)
    sn = problems.swarm_snapshot(prm, case.pop("n_rob"), case.pop("seed"), **case)
    args = [sn[k] for k in ARG_KEYS]
    compare(wave.replan(prm, *args, threads=256), oracle.replan(prm, *args, n_threads=8))


def test_helper_wavefronts_share_the_staged_row_scan(wave, oracle):
    """A dense neighbourhood (more than 256 staged rows): wave 0 wakes the three helper wavefronts for every violation scan
    (select(): `mw`, helper_loop). Same answers as the oracle and as the one-wavefront shape; H = 15 takes the NV = 48 kernel
    through the same paths."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 100, seed=9, spacing=1.5)
    sub = np.arange(0, 100, 8)
    args = [sn[k] if k in ("plans", "has_plan") else sn[k][sub] for k in ARG_KEYS]
    four, one = wave.replan(prm, *args, threads=256), wave.replan(prm, *args, threads=64)
    two = wave.replan(prm, *args, threads=128)   # the three-workgroups-per-CU shape: ONE helper wavefront
    assert four["cand"].max() > 256
    o = oracle.replan(prm, *args, n_threads=8)
    compare(four, o)
    compare(two, o)
    compare(one, o)
    prm15 = agile_params(15, max_rows_static=18)
    sn = problems.swarm_snapshot(prm15, 10, seed=115, turn=True)
    args = [sn[k] for k in ARG_KEYS]
    o15 = oracle.replan(prm15, *args, n_threads=8)
    compare(wave.replan(prm15, *args, threads=256), o15)
    compare(wave.replan(prm15, *args, threads=128), o15)   # the two-per-CU shape of the H > 10 kernel


@pytest.mark.parametrize("kw", [dict(seed=41, narrow=True, turn=True, spacing=1.6), dict(seed=43, chamfer=True, turn=True)])
def test_level1_through_the_device_source(wave, oracle, kw):
    """hdsm_solve: the host-side split (common suffix = the planes AddHyperplane appended) + the device source with explicit
    neighbour rows (no sweeps over plans) vs the oracle's LITERAL level-1 semantics, on both launch shapes."""
    prm = make_params(n_hor=6, poly_hor=3, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 8, **kw)
    n_poly, n_rows, A, b = problems.level1_from_snapshot(
        prm, sn, lambda a: oracle.tasc_planes(prm, a, sn["state"][a], sn["plans"], sn["has_plan"]))
    o = oracle.solve(prm, sn["state"], sn["ref"], n_poly, n_rows, A, b, n_threads=8)
    for threads in (64, 256):
        e = wave.solve(prm, sn["state"], sn["ref"], n_poly, n_rows, A, b, threads=threads)
        assert e["rc"] == 0
        compare(e, o, tol=1e-7)


def test_device_source_reproduces_the_enumerated_miqps(wave):
    """tests/golden/miqp_enum.npz (every admissible assignment resolved by scipy + KKT certificates, see test_oracle.py): the
    device source's lazy branch and bound on the same snapshots — NV = 32 with n = 18 (padded factor) and n = 30."""
    from test_oracle import enum_cases, enum_snapshot
    import refmath as rm
    for k, c in enum_cases():
        prm, polys, args = enum_snapshot(c)
        w = wave.replan(prm, *args)
        want = float(c["obj"])
        assert w["status"][0] == 0 and abs(w["obj"][0] - want) < 1e-6 * max(1.0, abs(want)), (k, w["obj"], want)
        if float(c["second"]) - want > 1e-3 * max(1.0, abs(want)):
            assert np.abs(w["traj"][0] - rm.rollout(prm, c["state"], c["u"])).max() < 1e-4, k


def test_staging_overflow_is_exact_or_flagged(wave, oracle):
    """With room for only 16 staged neighbour rows the solver must tighten its staging radius and still return the exact
    optimum, or report LIMIT / NO_SOLUTION — never a wrong 'optimal'."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 25, seed=9, spacing=1.2)
    args = [sn[k] for k in ARG_KEYS]
    e = wave.replan(prm, *args, cmax=16)
    o = oracle.replan(prm, *args, n_threads=8)
    exact = e["status"] == 0
    assert exact.sum() >= 1 and (e["status"] != 0).sum() >= 1   # both outcomes occur with so little room
    assert (o["status"][exact] == 0).all()
    assert np.abs(e["traj"] - o["traj"])[exact].max() < 1e-8
    assert (e["sweeps"] >= 1).all()
    lim = e["status"] == 1
    assert ((e["flags"][lim] & 8) != 0).all()                   # HDSM_FLAG_STAGING_OVERFLOW says why


def test_lazy_rows_are_verified(wave):
    """Every accepted solution went through at least one full sweep of the neighbour buffer."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 16, seed=2, turn=True)
    e = wave.replan(prm, *[sn[k] for k in ARG_KEYS])
    assert (e["sweeps"][e["status"] == 0] >= 1).all()


def test_level1_rejects_step_dependent_static_polyhedra(wave):
    prm = make_params(n_hor=6, poly_hor=3, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 2, seed=1)
    n_poly = np.full((2, 6), 2, np.int32)
    n_rows = np.full((2, 6, 3), 6, np.int32)
    A = np.zeros((2, 6, 3, 8, 3))
    b = np.zeros((2, 6, 3, 8))
    for i in range(6):
        for j in range(2):
            Aj, bj = problems.box_rows(np.array([-1.0 - j, -1, 0]), np.array([1.0 + i, 1 + j, 3]))  # grows with the step
            A[:, i, j, :6], b[:, i, j, :6] = Aj, bj
    e = wave.solve(prm, sn["state"], sn["ref"], n_poly, n_rows, A, b)
    assert e["rc"] == -1  # HDSM_ERR_BAD_ARG


@pytest.mark.parametrize("rule", ["0", "1"])
def test_either_branching_rule_reaches_the_same_optimum(wave, oracle, rule, monkeypatch):
    """The branching step (first uncontained segment in time / most infeasible one, HDSM_BRANCH_RULE) only shapes the search
    tree: on corridors that force real branching (narrow boxes, turning paths) both rules must return the oracle's optimum,
    and the default rule must not need more nodes in total than the other."""
    monkeypatch.setenv("HDSM_BRANCH_RULE", rule)
    prm = agile_params(10, max_rows_static=18)
    nodes = 0
    for seed in (3, 5, 21, 22):
        sn = problems.swarm_snapshot(prm, 12, seed, narrow=True, turn=True, chamfer=(seed % 2 == 1))
        args = [sn[k] for k in ARG_KEYS]
        e = wave.replan(prm, *args)
        compare(e, oracle.replan(prm, *args, n_threads=8))
        nodes += int(e["nodes"].sum())
    assert nodes > 4 * 12                      # the cases do branch
    seen = test_either_branching_rule_reaches_the_same_optimum.__dict__.setdefault("nodes", {})
    seen[rule] = nodes
    if len(seen) == 2:
        assert seen["1"] <= seen["0"], seen


def test_rows_on_input_independent_positions_are_judged_with_feas_tol_fixed(wave, oracle):
    """The device source on rows that act on p_1 / p_2 (constants under Euler): feas_tol_fixed decides, as in the oracle."""
    from multi_agent_pkgs_amd.params import make_params
    prm = make_params(n_hor=8, poly_hor=2, max_rows_static=18)
    for mstep in (1, 2):
        for delta, want in ((1e-8, 0), (1e-7, 0), (1e-5, 2)):
            args = problems.constant_row_case(prm, oracle, mstep, delta)
            for threads in (64, 128):
                e = wave.solve(prm, *args, threads=threads)
                assert e["status"][0] == want, (mstep, delta, threads, e["status"])
            if want == 0:
                compare(e, oracle.solve(prm, *args))


def test_small_lds_layout_of_the_four_per_cu_kernel(wave, oracle):
    """Shm<32, 256, SMALL> (k_replan_quad: 4 polyhedra of <= 20 rows, 512-neighbour chunks, 256 staged rows), two wavefronts."""
    prm = agile_params(10, max_rows_static=18)
    for kw in (dict(seed=3, turn=True), dict(seed=8, chamfer=True, narrow=True, turn=True, spacing=1.4)):
        sn = problems.swarm_snapshot(prm, 12, **kw)
        args = [sn[k] for k in ARG_KEYS]
        compare(wave.replan(prm, *args, threads=128, cmax=256), oracle.replan(prm, *args, n_threads=8))


@pytest.mark.parametrize("wname,n_it,short", [("forest", 42, False), ("fwf", 42, False), ("forest", 54, False), ("fwf", 30, False), ("fwf", 42, True), ("forest", 80, False)])
def test_cooperative_voxel_decomposition_matches_the_host_bit_for_bit(wave, wname, n_it, short):
    """Row f2, the form the device-resident loop runs (corridor_wave.h: one wavefront per seed, the world under the overlay as
    bit maps in LDS, a layer grown as bit planes — ballots and v_readlane instead of the cell deques) executed on the CPU
    against the host functions hdsm_poly_octa3d / hdsm_poly_octa3d_new (the serial statement-by-statement form): rows, row
    counts and the number of voxels taken must be IDENTICAL, in worlds with a potential field (values 1..99) too. The rim moves
    of a layer are made in batches (a closed-form schedule, one lane per move); this build also makes them one after the other
    and fails the decomposition where the two differ (-DCD_CHECK_BATCH). short: batches of two turns instead of sixteen, so that
    every layer continues over several batches; n_it = 80: more turns than the bit planes hold — the plain form on lane 0."""
    import decomp_cases as dc
    rng = np.random.default_rng(100 + n_it)
    occ2, origin = dc.world(wname, potential=True, rng=rng)
    off, seed, ground, variant, org = dc.cases(occ2, origin, 50, rng)
    rows, n_rows, rc, cells = wave.poly_octa3d_batch(occ2, dc.LDIM, off, ground, seed, variant, org, n_it=n_it, res=0.3, max_rows=32, short_batches=short)
    chamfered = 0
    for t in range(len(off)):
        try:
            want, voxels, _ = dc.host_answer(occ2, off[t], seed[t], ground[t], variant[t], org[t], n_it=n_it)
        except Exception:  # (n_it = 80: a polyhedron beyond the fixed workspace — both forms must say so)
            assert n_it > 54 and rc[t] != 0, (wname, t)
            continue
        assert rc[t] == 0 and n_rows[t] == len(want), (wname, t)
        assert np.array_equal(rows[t, : n_rows[t]], want), (wname, t)
        assert cells[t] == voxels, (wname, t)
        chamfered += len(want) > 6
    assert chamfered >= 3


def _with_contained_copies(sn, P, RS):
    """Every instance gets, next to each of its polyhedra, a COPY with its first row pulled in by 0.05 (contained in the original),
    up to P polyhedra: [P0, P1] -> [P0, P1, P0', P1']. The MIQP optimum cannot change: whatever lies in a copy lies in its original."""
    out = dict(sn)
    n_poly, n_rows, A, b = sn["n_poly"].copy(), sn["n_rows"].copy(), sn["A"].copy(), sn["b"].copy()
    for a in range(len(n_poly)):
        k0 = int(n_poly[a])
        for j in range(k0):
            if n_poly[a] >= P:
                break
            jn = int(n_poly[a])
            r = int(n_rows[a, j])
            A[a, jn, :r], b[a, jn, :r] = A[a, j, :r], b[a, j, :r]
            b[a, jn, 0] -= 0.05 * np.linalg.norm(A[a, j, 0])
            n_rows[a, jn] = r
            n_poly[a] = jn + 1
    out.update(n_poly=n_poly, n_rows=n_rows, A=A, b=b)
    return out


def test_dominated_polyhedra_are_taken_out_of_the_choice(wave, oracle, monkeypatch):
    """Round 6 (hdsm_core.h, dominated_mask): a polyhedron contained in another polyhedron of the instance is neither a container nor a
    choice. Instances whose corridors force branching, each polyhedron accompanied by a slightly smaller copy of itself: the answer is
    the oracle's answer for the ORIGINAL polyhedra (and for the padded ones: the oracle does not know the rule), with the rule and
    without it (HDSM_DOMINANCE=0), and with the rule the trees are no larger than those of the original instances — without it every
    copy multiplies them. A corridor of ONE polyhedron and its copy is a pure QP: one node (the copy is dominated, the survivor's rows
    are assigned to every uncontained step at once)."""
    prm = make_params(n_hor=10, max_rows_static=18, poly_hor=4)
    tot = {"orig": 0, "on": 0, "off": 0}
    for seed in (3, 5, 21):
        sn = problems.swarm_snapshot(prm, 10, seed, narrow=True, turn=True, chamfer=(seed % 2 == 1))
        sn["n_poly"] = np.minimum(sn["n_poly"], 2)          # two polyhedra + their two copies
        pad = _with_contained_copies(sn, prm.poly_hor, prm.max_rows_static)
        assert (pad["n_poly"] == 2 * sn["n_poly"]).all()
        o = oracle.replan(prm, *[sn[k] for k in ARG_KEYS], n_threads=8)
        compare(oracle.replan(prm, *[pad[k] for k in ARG_KEYS], n_threads=8), o)   # (the copies change nothing: the oracle agrees)
        e0 = wave.replan(prm, *[sn[k] for k in ARG_KEYS])
        compare(e0, o)
        e1 = wave.replan(prm, *[pad[k] for k in ARG_KEYS])
        compare(e1, o)
        monkeypatch.setenv("HDSM_DOMINANCE", "0")
        e2 = wave.replan(prm, *[pad[k] for k in ARG_KEYS])
        monkeypatch.delenv("HDSM_DOMINANCE")
        compare(e2, o)
        # a copy is never the polyhedron an answer names (its original contains the segment too, and comes first)
        assert not e1["used"][:, 2:].any() or (e1["nodes"] == 1).all()
        tot["orig"] += int(e0["nodes"].sum())
        tot["on"] += int(e1["nodes"].sum())
        tot["off"] += int(e2["nodes"].sum())
    assert tot["orig"] > 3 * 10, tot                         # the cases do branch
    assert tot["on"] <= tot["orig"] and tot["off"] > tot["on"], tot
    # one polyhedron + its copy: no tree at all
    sn = problems.swarm_snapshot(prm, 10, 5, narrow=True, turn=True, chamfer=True)
    sn["n_poly"] = np.minimum(sn["n_poly"], 1)
    pad = _with_contained_copies(sn, prm.poly_hor, prm.max_rows_static)
    o = oracle.replan(prm, *[sn[k] for k in ARG_KEYS], n_threads=8)
    e = wave.replan(prm, *[pad[k] for k in ARG_KEYS])
    compare(e, o)
    assert (e["nodes"] == 1).all(), e["nodes"].tolist()
    # ... and so is a corridor that comes with one polyhedron (step-by-step branching with one child per level was a node per step)
    e = wave.replan(prm, *[sn[k] for k in ARG_KEYS])
    compare(e, o)
    assert (e["nodes"] == 1).all(), e["nodes"].tolist()
