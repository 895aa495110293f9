"""CPU tests of the scenario pieces for BASELINE configs 3 and 5 (multi_agent_pkgs_amd/scenarios.py, harness code) and of
the host mirror's router: known answers from the reference's launch files, the shipped forest + wall + forest instance
(tests/golden/env_long_occupancy.npz, minted from the reference's env_long_config.yaml), the literal inflation loops of the
oracle, and size-independent properties of the routes."""
import os

import numpy as np
import pytest

from multi_agent_pkgs_amd import scenarios as sc
from multi_agent_pkgs_amd import swarm
from multi_agent_pkgs_amd.params import agile_params

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_lattice_scenario_known_answers():
    """multi_agent_planner_long.launch.py:36-42 (n_rob = 10): start_i = (0, 5 + 2.01 i, 0), goal_i = start_i + (96.01, 0, 0)."""
    s, g = sc.lattice_scenario(10)
    assert np.allclose(s[0], [0, 5.0, 0]) and np.allclose(s[9], [0, 5.0 + 9 * 2.01, 0]) and np.allclose(g - s, [96.01, 0, 0])
    s, g = sc.lattice_scenario(64, 64)
    assert s.shape == (4096, 3) and np.allclose(s[64], [0, 5.0, 2.01]) and np.allclose(s[4095], [0, 5 + 63 * 2.01, 63 * 2.01])


def test_forest_wall_forest_reproduces_the_shipped_wall_and_density():
    """The wall (x = 48, 0.3 m thick, fifteen gaps, generate_random_grid.py:60-82) is deterministic: the generator must give the
    shipped instance's wall voxel for voxel, after EnvironmentBuilder::AddObstacles' two-voxels-per-axis marking. The cylinders
    are random (own PRNG): same bands, comparable number of occupied columns."""
    z = np.load(os.path.join(GOLD, "env_long_occupancy.npz"))
    ref = np.unpackbits(z["packed"])[: np.prod(z["shape"])].reshape(z["shape"]).astype(bool)
    occ, origin = sc.forest_wall_forest(1, 1, seed=0)
    mine = occ >= 100
    assert mine.shape == ref.shape and np.allclose(origin, z["origin"])
    for x in range(150, 170):
        if ref[:, :, x].sum() > 2000:                               # the two voxel planes of the wall
            assert np.array_equal(mine[:, :, x], ref[:, :, x]), x
    assert ref[:, :, 158].sum() > 4000 and (~ref[:, :, 158]).sum() > 300   # ... with its gaps
    for lo, hi in ((8, 114), (206, 314)):                           # the two forest bands, x in [3, 33] and [63, 93]
        for sel in (lambda o: o.any(axis=0), lambda o: o.all(axis=0), lambda o: o[20]):   # touched / full-height columns, the slice z = 0
            a, b = sel(ref)[:, lo:hi].sum(), sel(mine)[:, lo:hi].sum()
            assert 0.65 * a < b < 1.5 * a, (lo, a, b)
    cols = mine.any(axis=0)
    assert not cols[:, 116:150].any() and not cols[:, 170:204].any()   # nothing between forests and wall
    big, _ = sc.forest_wall_forest(2, 3, seed=1)                    # tiles repeat the cross-section
    assert big.shape == (150, 200, 334) and np.array_equal(big[:50, :100, 158], occ[:, :, 158]) and np.array_equal(big[100:, 100:, 158], occ[:, :, 158])


def test_pillar_forest_follows_add_obstacles():
    """EnvironmentBuilder::AddObstacles (environment_builder.cpp:189-231) with env_default_config.yaml: 180 pillars of
    0.1 x 0.1 x 10 m, centres on whole metres relative to origin_obst (the integer division), voxelised by AddObstacle."""
    occ, origin = sc.pillar_forest([0.0, 0.0, -8.0], [40.0, 40.0, 20.0], [6.5, 6.5, 0.0], [30.0, 30.0, 0.0], 180, seed=13)
    assert occ.shape == (67, 134, 134)
    zz, yy, xx = np.nonzero(occ)
    assert zz.min() == 10 and zz.max() == 43                         # z in [3, 13] m of the grid -> voxels floor(3/0.3)..floor(13/0.3)
    cols = occ.any(axis=0)
    ys, xs = np.nonzero(cols)
    # every occupied column belongs to a pillar whose centre (6.5 + k) m lies within 0.05 m of the voxel's extent
    for x, y in zip(xs, ys):
        for v in (x, y):
            lo, hi = v * 0.3 - 0.05, (v + 1) * 0.3 + 0.05
            k = np.arange(0, 31)
            assert ((6.5 + k >= lo - 1e-9) & (6.5 + k <= hi + 1e-9)).any(), (x, y)
    assert 100 < cols.sum() <= 2 * 2 * 180


def test_inflate_equals_the_literal_loops(oracle):
    """scenarios.inflate (numpy, harness) against the oracle's literal InflateObstacles (voxel_grid.cpp:249-276)."""
    rng = np.random.default_rng(2)
    g = np.zeros((12, 30, 30), np.int8)
    g[rng.integers(0, 12, 40), rng.integers(0, 30, 40), rng.integers(0, 30, 40)] = 100
    for dist in (0.3, 0.45, 0.6, 0.9):
        from multi_agent_pkgs_amd.params import default_map_config
        want = oracle.map_preprocess(default_map_config(voxel_size=0.3, inflation_dist=dist, potential_dist=0.0, potential_pow=1), g[None])[0]
        got = sc.inflate(g, inflation_dist=dist)
        assert np.array_equal(got >= 100, want >= 100), dist


def test_router_gives_collision_free_bounded_polylines():
    """hdsm_swarm_route: every route starts at the start and ends at the goal, has at most PATH_PTS points, and every point of
    every segment (sampled at a quarter voxel) lies in a free voxel of the inflated world; in an empty world it is the
    straight segment."""
    prm = agile_params(10, max_rows_static=18)
    cfg = swarm.default_swarm_config()
    raw, origin = sc.forest_for_circle(32, seed=5)
    occ = sc.inflate(raw)
    starts, goals = sc.circle_scenario(32)
    sh = swarm.SwarmShard(prm, cfg, 32, 0, starts, goals)
    sh.set_world(occ, origin)
    assert sh.route() == 0
    paths, n = sh.get_paths()
    bent = 0
    for k in range(32):
        p = paths[k, : n[k]]
        assert 2 <= n[k] <= 48 and np.allclose(p[0], starts[k]) and np.allclose(p[-1], goals[k])
        bent += n[k] > 2
        vox_of = lambda q: tuple(np.floor((q - origin) / 0.3).astype(int)[::-1])
        ends_free = occ[vox_of(starts[k])] < 100 and occ[vox_of(goals[k])] < 100
        for si, (a, b) in enumerate(zip(p[:-1], p[1:])):
            if not ends_free and si in (0, n[k] - 2):
                continue  # a start / goal inside the inflation margin can only be left through it
            m = int(np.ceil(np.linalg.norm(b - a) / 0.075)) + 1
            pts = a + np.linspace(0, 1, m)[:, None] * (b - a)
            v = np.floor((pts - origin) / 0.3).astype(int)
            assert (occ[v[:, 2], v[:, 1], v[:, 0]] < 100).all(), k
            assert (pts[:, 2] > 0).all() and (pts[:, 2] < 1.5 + 3.0 + 0.31).all()      # altitude band of the local grid
    assert bent >= 16                                                 # the forest really is in the way
    empty = swarm.SwarmShard(prm, cfg, 4, 0, starts[:4], goals[:4])
    empty.set_world(np.zeros_like(occ), origin)
    assert empty.route() == 0
    paths, n = empty.get_paths()
    assert (n == 2).all()


def test_walk_shortcut_leaves_the_corridor_unchanged(oracle, monkeypatch):
    """The device corridor walk skips the inside tests of samples that provably stay inside a kept polyhedron
    (swarm_kernels.hip corridor_step_wave); swarm_core.h carries a scalar twin of that shortcut behind
    HDSM_FAST_WALK_HOST. In a forest the path slides along polyhedron faces (slack ~ 0, rate ~ 0), where the reference's
    sample-by-sample outcome depends on the last bit: the shortcut must leave those to the regular loop. Flown twice,
    the corridor inputs of every round are bit-identical."""
    from oracle import pyoracle as orc
    n, rounds = 48, 18
    prm = agile_params(10, max_rows_static=18)
    raw, org = sc.forest_for_circle(n, seed=21)
    occ = sc.inflate(raw)

    def cpu(inp, plans, has):
        return orc.replan(prm, inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"], inp["b"], plans, has, n_threads=8)

    def fly(mode):
        monkeypatch.setenv("HDSM_FAST_WALK_HOST", mode)
        loop = swarm.SwarmLoop(prm, swarm.default_swarm_config(), n, solve=cpu)
        loop.set_world(occ, org)
        out = []
        for _ in range(rounds):
            rec = []
            loop.step(record=rec)
            out.append({k: rec[0][k].copy() for k in ("n_poly", "n_rows", "A", "b")})
        return out

    plain, fast = fly("0"), fly("1")
    for r in range(rounds):
        for k in plain[r]:
            assert np.array_equal(plain[r][k], fast[r][k]), (r, k)


def test_walk_jump_in_free_space_leaves_the_corridor_unchanged(oracle, monkeypatch):
    """Empty world: the shortcut does not even generate the skipped samples one by one (swarm_core.h walk_jump: k samples along
    the segment in one step). The position then differs from the sample-by-sample walk by accumulated rounding only, far
    below the shortcut's margin: the boxes the corridor generator produces — seeds are voxel indices — must be identical.
    (23 agents: no path is parallel to a grid axis. On an axis-parallel path that starts on the voxel lattice EVERY tenth
    sample lies exactly on a voxel face, and which side it is counted on hangs on the last bit of the accumulated position in
    any implementation, the reference's included; such a path is not a test of anything.)"""
    from oracle import pyoracle as orc
    n, rounds = 23, 60
    prm = agile_params(10, max_rows_static=18)

    def cpu(inp, plans, has):
        return orc.replan(prm, inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"], inp["b"], plans, has, n_threads=8)

    def fly(mode):
        monkeypatch.setenv("HDSM_FAST_WALK_HOST", mode)
        loop = swarm.SwarmLoop(prm, swarm.default_swarm_config(), n, solve=cpu, radius=12.0)
        out = []
        for _ in range(rounds):
            rec = []
            loop.step(record=rec)
            out.append({k: rec[0][k].copy() for k in ("n_poly", "n_rows", "A", "b")})
        return out

    plain, fast = fly("0"), fly("1")
    new_boxes = 0
    for r in range(rounds):
        for k in plain[r]:
            assert np.array_equal(plain[r][k], fast[r][k]), (r, k)
        if r:
            new_boxes += int((plain[r]["b"] != plain[r - 1]["b"]).any(axis=(1, 2)).sum())
    assert new_boxes > n          # the corridors really moved on during the flight


def test_corridor_maintenance_matches_the_reference_restatement(oracle):
    """Row f2, the half around the voxel decomposition: GenerateSafeCorridor's keep-last / keep-used / path walk / seed logic
    (agent_class.cpp:1236-1447). The host mirror's corridor step (csrc/swarm_core.h, the source the device loop runs too) against
    oracle/hdsm_oracle.c: orc_safe_corridor — written from the reference text — on every agent of every round of a flight through
    the pillar forest: same polyhedra in the same order, same rows, same seeds, bit for bit."""
    import corridor_oracle as co
    from oracle import pyoracle as orc
    n, rounds = 24, 14
    prm = agile_params(10, max_rows_static=18)
    raw, org = sc.forest_for_circle(n, seed=21)
    occ = sc.inflate(raw)

    def cpu(inp, plans, has):
        return orc.replan(prm, inp["agent_id"], inp["state"], inp["ref"], inp["n_poly"], inp["n_rows"], inp["A"], inp["b"], plans, has, n_threads=8)

    loop = swarm.SwarmLoop(prm, swarm.default_swarm_config(), n, solve=cpu)
    loop.set_world(occ, org)
    new_polys = kept = 0
    for r in range(rounds):
        pre, prm_s, cfg_s, world, worigin = co.export_agents(loop.shard)
        loop.shard.prepare_corridor()
        post = co.export_agents(loop.shard)[0]
        for a in range(n):
            rc, want = co.oracle_corridor(orc.lib(), prm_s, cfg_s, world, worigin, pre[a])
            got = co.product_corridor(post[a])
            assert rc == 0 and post[a].corridor_rc == 0, (r, a, rc, post[a].corridor_rc)
            assert co.same_corridor(got, want), (r, a, [g[0] for g in got], [w[0] for w in want])
            kept += sum(1 for g in got if any(np.array_equal(g[3], h[3]) for h in co.product_corridor(pre[a])))
            new_polys += len(got)
        loop.step()
    assert new_polys - kept > n and kept > n   # polyhedra were generated AND carried over


def test_cylinders_against_the_reference_run():
    """The reference's own generator, RUN (env_builder/scripts/shapes.py imported in the build container by
    tests/golden/make_shapes_golden.py; the fixture holds data only): the 90 cylinder centres its RNG draws for the first forest block
    of cfg 5's world (generate_random_grid.py:93-100, Python `random` seed 0) and the voxels Cylinder.occupy_voxels marks for each.
    scenarios.cylinder_voxels - what forest_wall_forest lists per pillar - on the SAME centres: every voxel of the reference is
    there; the differences are the ones its docstring names (a few extra columns at the corners of the bounding square, a pillar one
    voxel taller where the reference's last z sample falls short of the clipped top). The pillar POSITIONS of scenarios.py come from
    numpy's generator, not Python's: worlds are equivalent in distribution, not voxel for voxel (not required)."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shapes_cylinders.npz"))
    c, cnt, v = z["centres"], z["counts"], z["voxels"]
    nx, ny, nz = (int(x) for x in z["grid_size"])
    assert (nx, ny, nz) == (333, 100, 50) and len(c) == 90   # (the reference truncates 100 / 0.3: 333 columns; scenarios.py lists 334)
    off = np.r_[0, np.cumsum(cnt)]
    extra_cols = cols_ref_total = taller = 0
    for n in range(len(c)):
        vv = v[off[n]:off[n + 1]]
        ref_cols = {(int(i), int(j)) for i, j in vv[:, :2]}
        cols, k0, k1 = sc.cylinder_voxels(c[n][0], c[n][1], c[n][2], -6.0, 9.0, nx, ny, nz)
        assert ref_cols <= set(cols), (n, ref_cols, cols)                         # nothing the reference lists is missing
        # every listed column spans the same z range in the reference: a full prism
        for col in ref_cols:
            ks = np.sort(vv[(vv[:, 0] == col[0]) & (vv[:, 1] == col[1]), 2])
            assert ks[0] == k0 and ks[-1] in (k1 - 1, k1 - 2) and len(ks) == ks[-1] - ks[0] + 1, (n, col, ks[0], ks[-1], k0, k1)
            taller += int(ks[-1] == k1 - 2)
        extra_cols += len(set(cols) - ref_cols)
        cols_ref_total += len(ref_cols)
    assert extra_cols <= 0.05 * cols_ref_total and taller <= 0.1 * cols_ref_total, (extra_cols, taller, cols_ref_total)
