"""CPU tests of the host side: the C ABI surface, parameter plumbing, scenario geometry, the swarm host code
(corridor / reference / fallback logic of include/hdsm_swarm.h) and its closed loop driven by the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from multi_agent_pkgs_amd import lib, swarm
from multi_agent_pkgs_amd.params import HdsmParams, agile_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hdsm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = lib.load()
    names = declared_functions("hdsm.h") + declared_functions("hdsm_swarm.h") + declared_functions("hdsm_stats.h")
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), n
    assert set(lib.EXPORTS) <= set(names)
    assert L.hdsm_version() >> 16 == 1


def test_default_params_match_python_mirror():
    L = lib.load()
    p = HdsmParams()
    L.hdsm_default_params(C.byref(p), 10)
    q = agile_params(10)
    for name, _ in HdsmParams._fields_:
        a, b = getattr(p, name), getattr(q, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        else:
            assert a == b, name


def test_no_device_means_error_not_fallback():
    """There is no GPU in the build container: creating a solver must FAIL (HDSM_ERR_NO_DEVICE), not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.HdsmError) as e:
        lib.Solver(agile_params(10), 4, 4)
    assert e.value.code == lib.HDSM_ERR_NO_DEVICE


def test_circle_scenario_known_answers():
    g = np.load(os.path.join(GOLD, "circle.npz"))
    starts, goals = swarm.circle_scenario(10, radius=22.0)
    assert np.abs(starts - g["starts"]).max() < 1e-12 and np.abs(goals - g["goals"]).max() < 1e-12
    # start_0 = (40, 15, 1.5), start_1 = (35.7984, 27.9313, 1.5), goal_0 = start_5 = (-4, 15, 1.5)
    assert np.allclose(starts[0], [40, 15, 1.5]) and np.allclose(goals[0], [-4, 15, 1.5])
    assert np.allclose(starts[1], [35.7984, 27.9313, 1.5], atol=1e-4)
    assert swarm.shard_range(1024, 3, 8) == (384, 128) and swarm.shard_range(10, 3, 4) == (9, 1)


def _first_round(n=1, **cfgkw):
    prm = agile_params(10, max_rows_static=18)
    cfg = swarm.default_swarm_config()
    for k, v in cfgkw.items():
        setattr(cfg, k, v)
    starts = np.array([[9.95, 9.95, 3.15]] * n)
    goals = starts + [30.0, 0, 0]
    sh = swarm.SwarmShard(prm, cfg, n, 0, starts, goals)
    inp = sh.prepare(np.zeros((n, 11, 9)), np.zeros(n, np.uint8))
    return prm, sh, inp


def test_free_space_polyhedron_matches_reference_golden():
    """SURVEY.md 8c-5 (obtained by running the reference's convex_decomp.cpp): free 66x66x20 grid @0.3 m, origin 0,
    seed voxel (33,33,10), n_it = 42 -> box x,y in [7.8, 12.3], z in [0.9, 5.4]; row order -y,+x,+y,-x,+z,-z."""
    prm, sh, inp = _first_round(grid_z_min=-100.0)
    # agent at (9.95, 9.95, 3.15): local grid origin = floor((p - range/2)/0.3)*0.3 = (-0.3+..): use the first box
    A, b = inp["A"][0, 0, :6], inp["b"][0, 0, :6]
    assert inp["n_rows"][0, 0] == 6
    assert np.array_equal(A, [[0, -1, 0], [1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1]])
    lo = np.array([-b[3], -b[0], -b[5]])
    hi = np.array([b[1], b[2], b[4]])
    assert np.allclose(hi - lo, [4.5, 4.5, 4.5])           # 15 voxels per axis = 7 layers per face
    assert np.all(lo <= [9.95, 9.95, 3.15]) and np.all(hi >= [9.95, 9.95, 3.15])
    assert np.allclose((lo / 0.3).round() * 0.3, lo)       # faces sit on the voxel lattice
    # ground clipping: with the grid floor at z = 0 a seed at z = 1.5 cannot grow below the ground
    prm2, sh2, inp2 = _first_round(grid_z_min=0.0)
    assert -inp2["b"][0, 0, 5] >= 0.0


def test_first_round_inputs_follow_the_reference_conventions():
    prm, sh, inp = _first_round()
    ref = inp["ref"][0]
    assert np.allclose(ref[0, :3], [9.95, 9.95, 3.15])               # ref_0 is the sampling start (AC:1625)
    assert np.allclose(np.diff(ref[:, 0]), 9.0 * prm.dt)             # nobody around: path_vel_max * dt spacing
    assert np.allclose(ref[:, 3], -9.0) and np.allclose(ref[:, 4:], 0)  # backward velocity reference (AC:1533)
    assert 2 <= inp["n_poly"][0] <= 4                                  # chain of boxes along the path


def test_closed_loop_with_oracle_swaps_eight_agents_without_collision(oracle):
    prm = agile_params(10, max_rows_static=18)
    cfg = swarm.default_swarm_config()

    def solve(inp, plans, has):
        return oracle.replan(prm, inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"],
                             inp["A"], inp["b"], plans, has, n_threads=8)

    loop = swarm.SwarmLoop(prm, cfg, 8, solve=solve)
    min_sep = 1e9
    for r in range(100):
        loop.step()
        pos, dist, nfail = loop.shard.state()
        D = np.linalg.norm(pos[:, None] - pos[None], axis=2) + np.eye(8) * 1e9
        min_sep = min(min_sep, D.min())
    assert nfail.sum() == 0
    assert dist.max() < 0.2            # everybody arrived at the antipodal point
    assert min_sep > 2 * prm.drone_radius


def test_fallback_shifts_previous_plan():
    """AC:1000-1019: on failure the previous plan loses its first state and the last one is duplicated."""
    prm = agile_params(10, max_rows_static=18)
    cfg = swarm.default_swarm_config()
    sh = swarm.SwarmShard(prm, cfg, 1, 0, np.array([[0, 0, 1.5]]), np.array([[30, 0, 1.5]]))
    sh.prepare(np.zeros((1, 11, 9)), np.zeros(1, np.uint8))
    traj = np.arange(99, dtype=float).reshape(1, 11, 9)
    ok = dict(traj=traj, ctrl=np.arange(30, dtype=float).reshape(1, 10, 3), used=np.array([[1, 0, 0, 0]], np.uint8),
              status=np.array([0], np.int32))
    plans, has = sh.commit(ok)
    assert has[0] == 1 and np.array_equal(plans[0], traj[0])
    sh.prepare(plans, has)
    bad = dict(ok, status=np.array([2], np.int32), traj=np.zeros_like(traj))
    plans2, has2 = sh.commit(bad)
    assert has2[0] == 1
    assert np.array_equal(plans2[0, :10], traj[0, 1:]) and np.array_equal(plans2[0, 10], traj[0, 10])
    assert sh.state()[2][0] == 1


def test_reference_oracle_equals_host_restatement_in_closed_loop(oracle):
    """Row f1: the oracle's restatement of GenerateReferenceTrajectory (oracle/hdsm_oracle.c, orc_reference) and
    the host code in csrc/swarm_host.cpp are two independent restatements of AC:1449-1553 / 1591-1663 / 1769-1817:
    flying the same swarm with either must give the same references, round after round."""
    from multi_agent_pkgs_amd.params import agile_ref_config
    prm = agile_params(10, max_rows_static=18)
    rcfg = agile_ref_config()

    def solve(inp, plans, has):
        return oracle.replan(prm, inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"],
                             inp["b"], plans, has, n_threads=8)

    seen = []

    def ref_fn(ids, path, n_path, plans, has):
        full, ref, pv = oracle.reference(prm, rcfg, ids, path, n_path, plans, has)
        seen.append((full.copy(), pv.copy()))
        return full, pv

    a = swarm.SwarmLoop(prm, swarm.default_swarm_config(), 12, solve=solve)
    b = swarm.SwarmLoop(prm, swarm.default_swarm_config(), 12, solve=solve, reference=ref_fn)
    slowed = False
    for r in range(40):
        a.step()
        b.step()
        assert np.abs(a.shard.inp["ref"] - b.shard.inp["ref"]).max() < 1e-9, r
        slowed |= bool((seen[-1][1] < 8.99).any())
    assert np.abs(a.plans_all - b.plans_all).max() < 1e-7
    assert slowed                                   # the neighbour speed modulation was active at some point
    assert np.allclose(seen[0][1], 9.0)             # first round: nobody has a plan -> path_vel_max


# ---- next row f2, first piece: the convex voxel decomposition (hdsm_poly_octa3d) ---------------------------------
def _free_grid():
    return np.zeros((20, 66, 66), np.int8)  # [z][y][x]: 66 x 66 x 20 voxels of 0.3 m, the local grid of the reference


def test_poly_octa3d_reproduces_the_reference_outputs():
    """outputs of convex_decomp_lib::GetPolyOcta3D itself (grid 66x66x20, res 0.3, origin 0,
    seed voxel (33,33,10), CONV = -1), rows n.x <= n.p, chamfers first, then -y +x +y -x +z -z."""
    from multi_agent_pkgs_amd.swarm import poly_octa3d
    box = lambda lo, hi, zlo, zhi: [[0, -1, 0, -lo], [1, 0, 0, hi], [0, 1, 0, hi], [-1, 0, 0, -lo], [0, 0, 1, zhi], [0, 0, -1, -zlo]]
    rows, _ = poly_octa3d(_free_grid(), (33, 33, 10), n_it=42)
    assert np.allclose(rows, box(7.8, 12.3, 0.9, 5.4), atol=1e-12)
    rows, _ = poly_octa3d(_free_grid(), (33, 33, 10), n_it=60)      # z clipped to voxels 1..18
    assert np.allclose(rows, box(6.9, 13.2, 0.3, 5.7), atol=1e-12)
    rows, _ = poly_octa3d(_free_grid(), (33, 33, 3), n_it=42)
    assert np.allclose(rows, box(7.8, 12.3, 0.3, 3.3), atol=1e-12)
    g = _free_grid()
    g[:, 35, 37] = 100                                               # one full-height occupied column at (37, 35)
    rows, _ = poly_octa3d(g, (33, 33, 10), n_it=42)
    want = [[3, 1, 0, 43.8]] + box(7.8, 12.3, 0.9, 5.4)
    want[2][3] = 12.0                                                # the +x face is pulled in
    assert np.allclose(rows, want, atol=1e-12)
    assert np.isclose(rows[0, :3] @ [11.1, 10.5, 2.25], 43.8)       # the chamfer passes through p = (11.1, 10.5, 2.25)
    g = _free_grid()
    for t in range(12):
        g[:, 30 + t, 36 + t] = 100                                   # occupied diagonal columns (36+t, 30+t)
    rows, _ = poly_octa3d(g, (33, 33, 10), n_it=42)
    assert np.allclose(rows, [[2, -1, 0, 12.3]] + box(7.8, 12.3, 0.9, 5.4), atol=1e-12)


def test_poly_octa3d_new_known_answer_and_fixture():
    """The shape-aware variant (hdsm_poly_octa3d_new = GetPolyOcta3DNew): SURVEY.md section 8c-5's diagonal-columns case gives
    the chamfer (2,-1,0; 12.45) (half a voxel further out than the original's 12.3), and both variants reproduce, row by
    row and bit for bit, the 48 recorded cases of tests/golden/corridor_cases.npz (see make_corridor_golden.py for
    how those were produced and what they do and do not pin)."""
    from multi_agent_pkgs_amd.swarm import poly_octa3d
    g = _free_grid()
    for t in range(12):
        g[:, 30 + t, 36 + t] = 100
    rows, _ = poly_octa3d(g, (33, 33, 10), n_it=42, shape_aware=True)
    assert np.allclose(rows[0], [2, -1, 0, 12.45], atol=1e-12) and len(rows) == 7
    z = np.load(os.path.join(GOLD, "corridor_cases.npz"))
    differ = 0
    for k in range(z["grids"].shape[0]):
        for aware, key in ((False, "octa3d"), (True, "octa3d_new")):
            want = z["rows_" + key][k]
            want = want[~np.isnan(want[:, 0])]
            rows, gm = poly_octa3d(z["grids"][k].copy(), z["seeds"][k], n_it=int(z["n_it"][k]), res=float(z["res"]),
                                   mark=int(z["mark"]), origin=z["origin"], max_rows=32, shape_aware=aware)
            assert rows.shape == want.shape and np.array_equal(rows, want), (k, key)
            assert np.count_nonzero(gm == int(z["mark"])) == int(z["cells_" + key][k]), (k, key)
        a, b = z["rows_octa3d"][k], z["rows_octa3d_new"][k]
        differ += not np.array_equal(np.nan_to_num(a), np.nan_to_num(b))
    assert differ >= 20  # the cases really exercise the difference between the two variants


def test_poly_octa3d_properties_in_random_forests():
    """Size-independent properties on pillar forests with walls: the seed lies inside, no occupied voxel centre lies
    strictly inside, at most 18 rows (what GetPolyOcta3D can emit: 12 edges + 6 faces), only free voxels are taken."""
    from multi_agent_pkgs_amd.swarm import poly_octa3d
    rng = np.random.default_rng(5)
    res = 0.3
    for case in range(120):
        g = _free_grid()
        for _ in range(int(rng.integers(5, 120))):
            x, y = rng.integers(1, 65, 2)
            w = int(rng.integers(1, 3))
            g[:, y:y + w, x:x + w] = 100
        if rng.random() < 0.3:
            g[:int(rng.integers(3, 20)), int(rng.integers(5, 60)), 10:50] = 100
        while True:
            s = (int(rng.integers(5, 60)), int(rng.integers(5, 60)), int(rng.integers(2, 18)))
            if g[s[2], s[1], s[0]] == 0:
                break
        g0 = g.copy()
        rows, gm = poly_octa3d(g, s, n_it=int(rng.choice([24, 42, 60])), res=res)
        A, b = rows[:, :3], rows[:, 3]
        assert 6 <= len(rows) <= 18
        assert (A @ ((np.array(s) + 0.5) * res) - b <= 1e-9).all()
        zz, yy, xx = np.nonzero(g0 >= 100)
        ctr = (np.stack([xx, yy, zz], 1) + 0.5) * res
        assert not (ctr @ A.T - b < -1e-9).all(axis=1).any(), case
        assert ((gm == -1) <= (g0 < 100)).all() and gm[s[2], s[1], s[0]] == -1


def test_poly_octa3d_capacity_and_arguments():
    from multi_agent_pkgs_amd import lib
    from multi_agent_pkgs_amd.swarm import poly_octa3d
    with pytest.raises(lib.HdsmError) as e:
        poly_octa3d(_free_grid(), (33, 33, 10), max_rows=4)
    assert e.value.code == lib.HDSM_ERR_CAPACITY
    with pytest.raises(lib.HdsmError) as e:
        poly_octa3d(_free_grid(), (70, 33, 10))
    assert e.value.code == lib.HDSM_ERR_BAD_ARG


def test_empty_world_grid_gives_the_free_space_corridors(oracle):
    """hdsm_swarm_set_world with an all-free grid: the voxel decomposition on each agent's local grid must produce the
    polyhedra of the free-space closed form, round after round (8 agents, oracle as the solver)."""
    prm = agile_params(10, max_rows_static=18)

    def cpu(inp, plans, has):
        return oracle.replan(prm, inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"],
                             inp["b"], plans, has, n_threads=8)

    la = swarm.SwarmLoop(prm, swarm.default_swarm_config(), 8, solve=cpu)
    lb = swarm.SwarmLoop(prm, swarm.default_swarm_config(), 8, solve=cpu)
    lb.shard.set_world(np.zeros((30, 200, 200), np.int8), origin=(-12.0, -15.0, 0.0))  # covers the 44 m circle
    ra, rb = [], []
    for r in range(30):
        la.step(record=ra)
        lb.step(record=rb)
        for k in ("n_poly", "n_rows", "A", "b"):
            assert np.array_equal(ra[-1][k], rb[-1][k]) or np.allclose(ra[-1][k], rb[-1][k], atol=1e-12), (r, k)
    assert np.abs(la.plans_all - lb.plans_all).max() < 1e-9


def test_set_world_argument_checks():
    prm = agile_params(10, max_rows_static=18)
    starts, goals = swarm.circle_scenario(4)
    sh = swarm.SwarmShard(prm, swarm.default_swarm_config(), 4, 0, starts, goals)
    with pytest.raises(lib.HdsmError) as e:   # origin must be a multiple of the voxel size (local grids register with it)
        sh.set_world(np.zeros((4, 4, 4), np.int8), origin=(0.1, 0.0, 0.0))
    assert e.value.code == lib.HDSM_ERR_BAD_ARG
    sh.set_world(np.zeros((4, 4, 4), np.int8), origin=(-0.3, 0.6, 0.0))
    sh.set_world(None)                         # back to free space


# ---- next row f3, ROS-free half: timing records and the shutdown report --------------------------------------------
def test_shutdown_statistics_files_and_report(tmp_path):
    """hdsm_stats_shutdown = Agent::OnShutdown (AC:2446-2466): comp_time_*_<id>.csv hold std::fixed values each followed by a
    comma on one line (AC:1948-1952), state_hist_<id>.csv one 'stamp,s0,...,s8' line per record (AC:1978-1991),
    com_latency_<id>.csv one line per OTHER agent (AC:2049-2059); the report is what the reference prints to std::cout
    (default ostream formatting = %g)."""
    L = lib.load()
    L.hdsm_stats_create.restype = C.c_void_p
    st = C.c_void_p(L.hdsm_stats_create(2, 4))
    vals = {0: [1.5, 0.25, 3.0], 1: [0.0, 0.0], 2: [12.3456789, 0.001], 3: [20.0], 4: [21.5], 5: [100.125, 7.0]}
    for kind, vs in vals.items():
        for v in vs:
            assert L.hdsm_stats_add(st, kind, C.c_double(v)) == 0
    states = [[1, 2, 3, 3, 4, 0, 0, 0, 0], [1.5, 2, 3, 0, 0, 12, 0.1, 0.2, 0.3]]
    for k, s_ in enumerate(states):
        arr = (C.c_double * 9)(*s_)
        assert L.hdsm_stats_add_state(st, C.c_double(100.5 + k), arr, 9) == 0
    for frm, lat in ((0, 1.0), (0, 3.0), (1, 2.0), (3, 10.0)):
        assert L.hdsm_stats_add_latency(st, frm, C.c_double(lat)) == 0
    assert L.hdsm_stats_add_latency(st, 2, C.c_double(1.0)) == lib.HDSM_ERR_BAD_ARG   # no subscription to oneself (AC:617)
    buf = C.create_string_buffer(4096)
    n = L.hdsm_stats_shutdown(st, str(tmp_path).encode(), 1, buf, 4096)
    text = buf.value.decode()
    assert n == len(text)
    names = ["sc", "tasc", "opt", "tot", "tot_wall", "path"]
    want = ""
    for kind, nm in enumerate(names):
        f = tmp_path / f"comp_time_{nm}_2.csv"
        assert f.read_text() == "".join("%f," % v for v in vals[kind])
        v = vals[kind]
        want += f"comp_time_{nm}_2.csv: \nmean: {'%g' % (sum(v) / len(v))}\nmax: {'%g' % max(v)}\nmin: {'%g' % min(v)}\n"
    assert (tmp_path / "state_hist_2.csv").read_text() == "".join(
        "%f," % (100.5 + k) + ",".join("%f" % x for x in s_) + "\n" for k, s_ in enumerate(states))
    assert (tmp_path / "com_latency_2.csv").read_text() == "1.000000,3.000000,\n2.000000,\n10.000000,\n"
    want += "\nvelocity for agent: 2\nmean: %g\nmax: %g\n" % ((5.0 + 12.0) / 2, 12.0)
    want += "communication latency (ms) for agent 2: mean: %g max: %g\n" % ((2.0 + 2.0 + 10.0) / 3, 10.0)
    assert text == want
    L.hdsm_stats_destroy(st)


def test_swarm_keeps_the_planner_records(oracle, tmp_path):
    """The host mirror books one record per replan round and agent (AC:193-245) and hdsm_swarm_shutdown writes them."""
    prm = agile_params(10, max_rows_static=18)

    def cpu(inp, plans, has):
        return oracle.replan(prm, inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"], inp["b"],
                             plans, has, n_threads=4)

    loop = swarm.SwarmLoop(prm, swarm.default_swarm_config(), 4, solve=cpu)
    L = lib.load()
    for r in range(5):
        loop.shard.prepare(loop.plans_all, loop.has_plan)
        out = cpu(loop.shard.inp, loop.plans_all, loop.has_plan)
        assert L.hdsm_swarm_record_solve_ms(loop.shard.h, C.c_double(0.5 + r)) == 0
        loop.plans_all, loop.has_plan = loop.shard.commit(out)
    buf = C.create_string_buffer(4096)
    assert L.hdsm_swarm_shutdown(loop.shard.h, 1, str(tmp_path).encode(), 1, buf, 4096) > 0
    opt = (tmp_path / "comp_time_opt_1.csv").read_text()
    assert opt == "".join("%f," % (0.5 + r) for r in range(5))
    assert (tmp_path / "comp_time_tasc_1.csv").read_text() == "0.000000," * 5
    hist = (tmp_path / "state_hist_1.csv").read_text().strip().split("\n")
    assert len(hist) == 5 and all(len(row.split(",")) == 10 for row in hist)
    assert "comp_time_sc_1.csv: " in buf.value.decode() and "velocity for agent: 1" in buf.value.decode()


# ---- row f1, map-dependent half: ComputePathVelocity's voxel term and KeepOnlyFreeReference on the host mirror --------
def _py_raycast(val, dim, start, end, max_dist):
    """voxel_grid_util::Raycast (raycast.cpp:22-183) statement by statement; val(i,j,k) raw voxel, -1 outside."""
    import math
    mod = lambda v, m: math.fmod(math.fmod(v, m) + m, m)

    def intbound(s, ds):
        if ds < 0:
            return intbound(-s, -ds)
        s = mod(s, 1)
        return (1 - s) / ds if ds != 0 else math.inf

    sg = lambda x: 0 if x == 0 else (-1 if x < 0 else 1)
    x, y, z = (int(math.floor(c)) for c in start)
    ex, ey, ez = (int(math.floor(c)) for c in end)
    d = [end[k] - start[k] for k in range(3)]
    st = [sg(ex - x), sg(ey - y), sg(ez - z)]
    tm = [intbound(start[k], d[k]) for k in range(3)]
    td = [(st[k] / d[k]) if d[k] != 0 else math.nan for k in range(3)]
    out, hit = [], None
    if st == [0, 0, 0]:
        return [list(end), list(start)], None
    tmax = 0.0
    inside = lambda i, j, k: 0 <= i < dim[0] and 0 <= j < dim[1] and 0 <= k < dim[2]
    while True:
        real = [start[k] + min(1.0, tmax) * d[k] for k in range(3)]
        if inside(x, y, z):
            if val(x, y, z) == 100 and tmax <= 1:
                hit = real
                out.append(real)
                break
            out.append(real)
            if (x - start[0]) ** 2 + (y - start[1]) ** 2 + (z - start[2]) ** 2 > max_dist ** 2:
                break
        if tmax >= 1:
            break
        if (tm[0] < tm[1] and st[0] != 0) or st[1] == 0:
            ax = 0 if ((tm[0] < tm[2] and st[0] != 0) or st[2] == 0) else 2
        else:
            ax = 1 if ((tm[1] < tm[2] and st[1] != 0) or st[2] == 0) else 2
        tmax = tm[ax]
        if ax == 0:
            x += st[0]
        elif ax == 1:
            y += st[1]
        else:
            z += st[2]
        tm[ax] += td[ax]
    return out, hit


def test_voxel_velocity_cap_and_keep_only_free_follow_the_reference_statements():
    """hdsm_swarm_vel_cap = the voxel term of Agent::ComputePathVelocity (AC:1709-1766) and hdsm_swarm_set_reference's
    KeepOnlyFreeReference (AC:1665-1693), against a statement-by-statement Python rendering of the reference (Raycast,
    GetVelocityLimit, the world-metres-vs-voxel-units distance of AC:1739) on worlds with a potential field."""
    import math
    from multi_agent_pkgs_amd import scenarios as sc
    rng = np.random.default_rng(8)
    prm = agile_params(10, max_rows_static=18)
    cfg = swarm.default_swarm_config()
    vs, vmin, vmax, sd, sp = 0.3, cfg.path_vel_min, cfg.path_vel_max, cfg.sens_dist, cfg.sens_pot
    lowered = 0
    for case in range(12):
        raw = np.zeros((30, 120, 120), np.int8)
        for _ in range(60):
            i, j = rng.integers(5, 115, 2)
            raw[:, j, i] = 100
        occ = sc.inflate(raw)
        halo = sc.inflate(occ, inflation_dist=0.6)
        world = np.where(occ >= 100, 100, np.where(halo >= 100, int(rng.integers(20, 90)), 0)).astype(np.int8)  # a crude potential field
        origin = np.array([-3.0, -6.0, -0.9])
        starts = np.array([[rng.uniform(6, 24), rng.uniform(3, 24), 1.5] for _ in range(6)])
        goals = starts + rng.uniform(-9, 9, (6, 3)) * [1, 1, 0.05]
        sh = swarm.SwarmShard(prm, cfg, 6, 0, starts, goals)
        sh.set_world(world, origin)
        cap = sh.vel_cap()
        for k in range(6):
            go = np.floor((starts[k] - np.array([10.0, 10.0, 3.0])) / vs) * vs
            off = np.round((go - origin) / vs).astype(int)
            gk = int(math.ceil((0.0 - go[2]) / vs - 1e-9))

            def val(i, j, kk):
                if not (0 <= i < 66 and 0 <= j < 66 and 0 <= kk < 20):
                    return -1
                if kk < gk:
                    return -1
                g = (i + off[0], j + off[1], kk + off[2])
                if not (0 <= g[0] < 120 and 0 <= g[1] < 120 and 0 <= g[2] < 30):
                    return 0
                return int(world[g[2], g[1], g[0]])

            pts = [starts[k], goals[k]]
            loc = [(p - go) / vs for p in pts]
            want = vmax
            visited, hit = _py_raycast(val, (66, 66, 20), list(loc[0]), list(loc[1]), float(np.linalg.norm(loc[0] - loc[1])))
            lim = lambda o, d: vmin + (vmax - vmin) * (1 - (min(max(o, 0), 100) / 100) ** sp * (1 / math.exp(sd * d)))
            if hit is None:
                for pt in visited + [list(loc[0])]:
                    v = val(int(pt[0]), int(pt[1]), int(pt[2]))
                    v = 100 if v == -1 else v
                    want = min(want, lim(v, float(np.linalg.norm(pts[0] - np.array(pt))) * vs))
            else:
                want = min(want, lim(val(int(hit[0]), int(hit[1]), int(hit[2])), float(np.linalg.norm(loc[0] - np.array(hit)))))
            assert abs(cap[k] - want) < 1e-9, (case, k, cap[k], want)
            lowered += want < vmax - 1e-6
        # KeepOnlyFreeReference: a straight reference of 11 points along the path; the first point in an occupied / unknown voxel
        # stops it (AC:1677-1688) and the velocity references are rebuilt on the result (AC:1527-1547)
        ref = np.zeros((6, 11, 6))
        for k in range(6):
            ref[k, :, :3] = starts[k] + np.linspace(0, 1, 11)[:, None] * (goals[k] - starts[k])
        sh.set_reference(ref, np.full(6, 5.0))
        sh.prepare(np.zeros((6, 11, 9)), np.zeros(6, np.uint8))
        got = sh.inp["ref"]
        for k in range(6):
            go = np.floor((starts[k] - np.array([10.0, 10.0, 3.0])) / vs) * vs
            pts = ref[k, :, :3].copy()
            for i in range(1, 11):
                c = ((pts[i] - go) / vs).astype(int)
                g = c + np.round((go - origin) / vs).astype(int)
                inside = (c >= 0).all() and c[0] < 66 and c[1] < 66 and c[2] < 20
                v = -1 if (not inside or c[2] < int(math.ceil((0.0 - go[2]) / vs - 1e-9))) else (
                    int(world[g[2], g[1], g[0]]) if (0 <= g[0] < 120 and 0 <= g[1] < 120 and 0 <= g[2] < 30) else 0)
                if v in (-1, 100):
                    pts[i:] = pts[i - 1]
                    break
            assert np.abs(got[k, :, :3] - pts[:10]).max() < 1e-12, (case, k)
    assert lowered > 10


def test_ros_node_type_checks_against_rclcpp_shaped_headers():
    """Row f3: ros/hdsm_agent_node.cpp compiled (-Wall -Wextra) against tests/ros_shim — rclcpp::Node with declare_parameter /
    create_publisher / create_subscription / create_wall_timer / now(), multi_agent_planner_msgs::msg::Trajectory with the fields
    of Trajectory.msg:1-11 and State.msg:1-8 — and linked with libhdsm.so into a program that hosts two nodes. Without a GPU the
    node's constructor must fail LOUDLY on hdsm_create (no CPU fallback); the exchange itself is a -m gpu test."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = os.path.join(root, "tests", "ros_shim")
    subprocess.check_call(["make", "-C", shim, "-s", "-B"])
    r = subprocess.run([os.path.join(shim, "two_nodes"), "3"], capture_output=True, text=True, timeout=120)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 7 and "hdsm_create" in r.stdout and "no HIP device" in r.stdout, r.stdout + r.stderr


def test_yaw_follows_the_reference_with_the_p_controller_of_the_reference():
    """hdsm_swarm_yaw = Agent::ComputeYawAngle (AC:1025-1051): yaw += k_p * wrap(atan2(dy, dx) - yaw) * dt towards reference point
    yaw_idx, frozen while that point is within sqrt(0.1) m. Checked against a literal Python rendering over a short flight (the
    oracle stands in for the device solver), including the wrap across +-pi; hdsm_swarm_view hands back what the rviz publishers show."""
    import ctypes as C
    import math
    from multi_agent_pkgs_amd import swarm
    from multi_agent_pkgs_amd.params import agile_params
    from oracle import pyoracle as orc
    prm = agile_params(10, max_rows_static=18)
    n = 4
    starts = np.array([[0.0, 0.0, 1.5], [10.0, 0.0, 1.5], [0.0, 8.0, 1.5], [10.0, 8.0, 1.5]])
    goals = np.array([[10.0, 0.0, 1.5], [0.0, 0.1, 1.5], [0.0, -8.0, 1.5], [10.0, 8.2, 1.5]])   # +x, -x (wrap), -y, nearly at rest

    def cpu(inp, plans, has):
        return orc.replan(prm, inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"], inp["b"], plans, has, n_threads=4)

    loop = swarm.SwarmLoop(prm, swarm.default_swarm_config(), n, solve=cpu, starts=starts, goals=goals)
    L = loop.shard.lib
    yaw_idx, k_p = 3, 1.0
    want = np.zeros(n)
    got = np.zeros(n)
    N = prm.n_hor
    for r in range(12):
        inputs = loop.shard.prepare(loop.plans_all, loop.has_plan)
        out = cpu(inputs, loop.plans_all, loop.has_plan)
        # the literal rendering, from what the rviz view shows BEFORE the commit
        for k in range(n):
            tr, pos, n_ref = np.zeros((N + 1, 3)), np.zeros(3), C.c_int32()
            assert L.hdsm_swarm_view(loop.shard.h, k, None, None, tr.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n_ref), None, 0, None, None, None,
                                     None, None, None, pos.ctypes.data_as(C.POINTER(C.c_double))) == 0
            if n_ref.value > yaw_idx:
                v = tr[yaw_idx] - pos
                if v @ v > 0.1:
                    err = math.atan2(v[1], v[0]) - want[k]
                    if err > math.pi:
                        err -= 2 * math.pi
                    elif err < -math.pi:
                        err += 2 * math.pi
                    want[k] = want[k] + k_p * err * prm.dt
        assert L.hdsm_swarm_yaw(loop.shard.h, yaw_idx, C.c_double(k_p), got.ctypes.data_as(C.POINTER(C.c_double))) == 0
        assert np.array_equal(got, want), (r, got, want)
        plans_local, has_local = loop.shard.commit(out)
        loop.plans_all[:], loop.has_plan[:] = plans_local, has_local
    assert abs(got[0]) < 0.05                             # flying along +x: (almost) no error to correct
    assert abs(got[1]) > 0.5                              # towards -x: turning (through the wrap branch) towards +-pi
    assert got[2] < -0.5                                  # towards -y
    assert abs(got[3]) < 1.6                              # (goal 0.2 m away: the yaw freezes once the reference point is within sqrt(0.1) m)


def test_closed_form_of_the_increment_check_stays_within_1e_10_of_the_literal_walk():
    """k_commit (device loop) first takes the minimum distance of the increment check's samples (AC:569-585: a walk in 1-cm steps
    along the reference) in closed form and walks literally only when the decision hangs on less than 1e-9 m. The closed form
    must then be within that margin of the literal walk: random references at swarm-scale coordinates, points near and far,
    segments shorter than a step, segments whose length is a multiple of the step."""
    import ctypes as C
    from multi_agent_pkgs_amd import lib
    L = lib.load()
    rng = np.random.default_rng(3)
    d = C.POINTER(C.c_double)
    worst = 0.0
    for case in range(400):
        n_ref = int(rng.integers(2, 12))
        base = rng.uniform(-300, 300, 3)
        steps = rng.uniform(-1.0, 1.0, (n_ref - 1, 3)) * rng.choice([0.002, 0.3, 0.9, 1.5])
        if case % 7 == 0:
            steps = np.tile(np.array([[0.05, 0.0, 0.0]]), (n_ref - 1, 1))       # |segment| / 0.01 an integer
        ref = np.ascontiguousarray(np.vstack([base, base + np.cumsum(steps, 0)]))
        pt = np.ascontiguousarray(ref[rng.integers(0, n_ref)] + rng.normal(0, rng.choice([1e-3, 0.05, 2.0]), 3))
        lit, cf = np.zeros(n_ref - 1), np.zeros(n_ref - 1)
        assert L.hdsm_internal_increment_minima(ref.ctypes.data_as(d), n_ref, pt.ctypes.data_as(d), lit.ctypes.data_as(d), cf.ctypes.data_as(d)) == 0
        worst = max(worst, float(np.abs(lit - cf).max()))
    assert worst < 1e-10, worst
