"""CPU tests of the KERNEL LOGIC: multi_agent_pkgs_amd/csrc/hdsm_core.h compiled for the host (tests/emu/),
i.e. the statements the HIP kernel executes for staging rows, verification sweeps and the lazy
branch-and-bound state machine, checked against the oracle without a GPU. (The device build swaps the inner
active-set iteration for the register-resident wave version of hdsm_wave_gi.h; that one is covered by the
-m gpu tests and, on the CPU, by tests/test_wave_emu.py, which runs the device source itself in lockstep fibers.)"""
import numpy as np
import pytest

import problems
from multi_agent_pkgs_amd.params import agile_params, make_params

ARG_KEYS = ("agent_id", "state", "ref", "n_poly", "n_rows", "A", "b", "plans", "has_plan")


@pytest.fixture(scope="module")
def emu():
    from emu import pyemu
    pyemu.lib()
    return pyemu


def compare(e, o, tol=1e-8):
    assert (e["status"] == o["status"]).all(), (e["status"].tolist(), o["status"].tolist())
    ok = o["status"] != 2
    if ok.any():
        assert np.abs(e["traj"] - o["traj"])[ok].max() < tol
        assert (np.abs(e["obj"] - o["obj"])[ok] / np.maximum(1, np.abs(o["obj"][ok]))).max() < 1e-8


CASES = [dict(n_rob=16, seed=1), dict(n_rob=16, seed=2, turn=True), dict(n_rob=12, seed=3, narrow=True, turn=True),
         dict(n_rob=16, seed=4, spacing=1.0), dict(n_rob=12, seed=5, chamfer=True, narrow=True, turn=True),
         dict(n_rob=16, seed=6, first_round=True), dict(n_rob=36, seed=8, absent_frac=0.3, spacing=1.5)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_emu_matches_oracle_h10(emu, oracle, case):
    prm = agile_params(10, max_rows_static=18)
    case = dict(case)
    sn = problems.swarm_snapshot(prm, case.pop("n_rob"), case.pop("seed"), **case)
    args = [sn[k] for k in ARG_KEYS]
    compare(emu.replan(prm, *args), oracle.replan(prm, *args, n_threads=8))


@pytest.mark.parametrize("n_hor,rk4,drag", [(15, False, (0, 0, 0)), (9, True, (0.1, 0.1, 0.3)), (12, True, (0, 0, 0)), (7, False, (0.2, 0.1, 0))])
def test_emu_matches_oracle_other_configs(emu, oracle, n_hor, rk4, drag):
    prm = make_params(n_hor=n_hor, rk4=rk4, drag=drag, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 12, seed=100 + n_hor, turn=True)
    args = [sn[k] for k in ARG_KEYS]
    compare(emu.replan(prm, *args), oracle.replan(prm, *args, n_threads=8))


def test_staging_overflow_is_exact_or_flagged(emu, oracle):
    """With room for only 16 staged neighbour rows the solver must tighten its staging radius and still return the
    exact optimum, or report LIMIT / NO_SOLUTION — never a wrong 'optimal'."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 25, seed=9, spacing=1.2)
    args = [sn[k] for k in ARG_KEYS]
    e = emu.replan(prm, *args, cmax=16)
    o = oracle.replan(prm, *args, n_threads=8)
    exact = e["status"] == 0
    assert exact.sum() >= 1 and (e["status"] != 0).sum() >= 1   # both outcomes occur with so little room
    assert (o["status"][exact] == 0).all()
    assert np.abs(e["traj"] - o["traj"])[exact].max() < 1e-8
    assert (e["sweeps"] >= 1).all()


def test_lazy_rows_are_verified(emu):
    """Every accepted solution went through at least one full verification sweep of the neighbour buffer."""
    prm = agile_params(10, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 16, seed=2, turn=True)
    e = emu.replan(prm, *[sn[k] for k in ARG_KEYS])
    assert (e["sweeps"][e["status"] == 0] >= 1).all()


@pytest.mark.parametrize("kw", [dict(seed=41, narrow=True, turn=True, spacing=1.6), dict(seed=42, spacing=1.2), dict(seed=43, chamfer=True, turn=True)])
def test_level1_split_and_solve_match_literal_oracle(emu, oracle, kw):
    """hdsm_solve's host-side split (common suffix = the planes AddHyperplane appended) + kernel logic vs the
    oracle's LITERAL level-1 semantics (every row is a choice row)."""
    prm = make_params(n_hor=6, poly_hor=3, max_rows_static=18)
    sn = problems.swarm_snapshot(prm, 8, **kw)
    n_poly, n_rows, A, b = problems.level1_from_snapshot(
        prm, sn, lambda a: oracle.tasc_planes(prm, a, sn["state"][a], sn["plans"], sn["has_plan"]))
    e = emu.solve(prm, sn["state"], sn["ref"], n_poly, n_rows, A, b)
    assert e["rc"] == 0
    o = oracle.solve(prm, sn["state"], sn["ref"], n_poly, n_rows, A, b, n_threads=8)
    compare(e, o, tol=1e-7)


def test_level1_rejects_step_dependent_static_polyhedra(emu):
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
    e = emu.solve(prm, sn["state"], sn["ref"], n_poly, n_rows, A, b)
    assert e["rc"] == -1  # HDSM_ERR_BAD_ARG


@pytest.mark.parametrize("rule", ["0", "1"])
def test_either_branching_rule_reaches_the_same_optimum(emu, oracle, rule, monkeypatch):
    """The branching step (first uncontained segment in time / most infeasible one, HDSM_BRANCH_RULE) only shapes the
    search tree: on corridors that force real branching (narrow boxes, turning paths) both rules must return the
    oracle's optimum, and the default rule must not need more nodes in total than the other."""
    monkeypatch.setenv("HDSM_BRANCH_RULE", rule)
    prm = agile_params(10, max_rows_static=18)
    nodes = 0
    for seed in (3, 5, 21, 22):
        sn = problems.swarm_snapshot(prm, 12, seed, narrow=True, turn=True, chamfer=(seed % 2 == 1))
        args = [sn[k] for k in ARG_KEYS]
        e = emu.replan(prm, *args)
        compare(e, oracle.replan(prm, *args, n_threads=8))
        nodes += int(e["nodes"].sum())
    assert nodes > 4 * 12                      # the cases do branch
    test_either_branching_rule_reaches_the_same_optimum.nodes = getattr(test_either_branching_rule_reaches_the_same_optimum, "nodes", {})
    test_either_branching_rule_reaches_the_same_optimum.nodes[rule] = nodes
    seen = test_either_branching_rule_reaches_the_same_optimum.nodes
    if len(seen) == 2:
        assert seen["1"] <= seen["0"], seen
