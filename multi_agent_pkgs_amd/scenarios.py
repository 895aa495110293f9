"""Scenario generators for BASELINE.json's configurations: start / goal layouts and occupied worlds.

These restate the ARITHMETIC of the reference's scenario files (launch files and world generators are harness code, not
the hot path; they cannot be imported here — `launch_ros` is absent — and must not travel to the GPU box):

  circle_scenario          multi_agent_planner/launch/multi_agent_planner_circle.launch.py:25-44
  lattice_scenario         multi_agent_planner/launch/multi_agent_planner_long.launch.py:24-42 (a line of agents along y),
                           generalised to a y-z lattice for 4096 agents (cfg 5)
  pillar_forest            env_builder/src/environment_builder.cpp:189-231 (AddObstacles) with the parameters of
                           env_builder/config/env_default_config.yaml:9-13, voxelised by voxel_grid.cpp:314-332
  forest_wall_forest       env_builder/scripts/generate_random_grid.py:57-115 (+ shapes.py Wall / Cylinder / RandomVolume),
                           then EnvironmentBuilder::AddObstacles on the voxel list it writes
  inflate                  what mapping_util's MapBuilder does to every grid before the planner sees it
                           (map_builder.cpp:209-216 -> VoxelGrid::InflateObstacles, voxel_grid.cpp:249-276); numpy here, the
                           device version is row f4 (hdsm_map_preprocess)

Randomness: the reference uses glibc rand() (C++) and Python's random (scripts); bit-identical worlds are not required
 — numpy's default_rng with a recorded seed is used, the DETERMINISTIC parts (wall and gaps, obstacle
voxelisation, pillar positions on the integer-metre lattice) follow the reference exactly and are tested against the
shipped instance (tests/golden/env_long_occupancy.npz).

Occupancy arrays are int8 [nz][ny][nx] (x fastest), 100 = occupied, 0 = free, as hdsm_swarm_set_world takes them.
"""
import numpy as np

VOX = 0.3


def circle_scenario(n, radius=None, cx=18.0, cy=15.0, z=1.5):
    """start/goal of multi_agent_planner_circle.launch.py:36-44: agent k starts at angle 2 pi k / n, its goal is the start
    of agent (k + n//2) mod n. The shipped radius (22 m) is kept while the chord between neighbours stays >= 1 m; larger
    swarms use R = n / (2 pi)."""
    if radius is None:
        radius = max(22.0, n / (2 * np.pi))
    ang = 2 * np.pi * np.arange(n) / n
    starts = np.stack([cx + radius * np.cos(ang), cy + radius * np.sin(ang), np.full(n, z)], axis=1)
    goals = starts[(np.arange(n) + n // 2) % n].copy()
    return starts, goals


def lattice_scenario(n_y, n_z=1, pitch=2.01, length=96.01, x0=0.0, y0=5.0, z0=0.0):
    """multi_agent_planner_long.launch.py:36-42: start_i = (0, 5 + 2.01 i, 0), goal_i = start_i + (96.01, 0, 0); rows of
    the same line stacked in z with the same pitch for swarms larger than one line (agent index = j * n_y + i)."""
    starts = np.array([[x0, y0 + pitch * i, z0 + pitch * j] for j in range(n_z) for i in range(n_y)], dtype=np.float64)
    goals = starts + [length, 0.0, 0.0]
    return starts, goals


def _add_obstacle(occ, center_local, size, vox=VOX):
    """voxel_grid_util::AddObstacle (voxel_grid.cpp:314-332): voxels floor((c - s/2)/vox) .. floor((c + s/2)/vox)."""
    nz, ny, nx = occ.shape
    lo = np.floor((np.asarray(center_local) - np.asarray(size) / 2) / vox).astype(int)
    hi = np.floor((np.asarray(center_local) + np.asarray(size) / 2) / vox).astype(int)
    lo = np.maximum(lo, 0)
    hi = np.minimum(hi, [nx - 1, ny - 1, nz - 1])
    if (lo <= hi).all():
        occ[lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = 100


def pillar_forest(origin_grid, dimension_grid, origin_obst, range_obst, n_obst, size_obst=(0.1, 0.1, 10.0), seed=13, vox=VOX):
    """EnvironmentBuilder::AddObstacles (environment_builder.cpp:189-231) for random positions:
        centre_k = ((rand() % int((range_k + 0.02) * 100)) / 100) + origin_obst_k - origin_grid_k      (grid-local metres)
    The division is an INTEGER division in the reference, so pillar centres sit on whole metres relative to origin_obst.
    Defaults of env_default_config.yaml: 180 pillars of 0.1 x 0.1 x 10 m on 30 x 30 m (0.2 per m^2), grid 40 x 40 x 20 m at
    z in [-8, 12]. Returns (occupancy int8 [nz][ny][nx], origin (3,))."""
    origin_grid = np.asarray(origin_grid, dtype=np.float64)
    dims = np.ceil(np.asarray(dimension_grid, dtype=np.float64) / vox - 1e-9).astype(int)  # environment_builder.cpp:146-151
    occ = np.zeros((dims[2], dims[1], dims[0]), np.int8)
    rng = np.random.default_rng(seed)
    for _ in range(int(n_obst)):
        c = np.zeros(3)
        for k in range(3):
            m = int((range_obst[k] + 0.02) * 100)
            c[k] = (int(rng.integers(0, 2 ** 31 - 1)) % m) // 100 + origin_obst[k] - origin_grid[k]
        _add_obstacle(occ, c, size_obst, vox)
    return occ, origin_grid


def forest_for_circle(n_rob, radius=None, cx=18.0, cy=15.0, density=180.0 / 900.0, seed=13, vox=VOX):
    """The forest of env_default_config.yaml scaled to a circle of n_rob agents (cfg 3): the shipped
    world is 40 x 40 m with a 30 x 30 m forest inside a ring of radius 22 m; for a ring of radius R the forest square keeps
    the same proportion (side = 30 R / 22, centred on the ring) and the same density (0.2 pillars per m^2)."""
    if radius is None:
        radius = max(22.0, n_rob / (2 * np.pi))
    side = float(np.floor(30.0 * radius / 22.0))
    margin = 4.0 + 10.0  # ring + half a local grid
    lo = np.array([cx - radius - margin, cy - radius - margin, -8.0])
    lo = np.floor(lo / vox) * vox  # local grids register with multiples of the voxel size
    dimension = [2 * (radius + margin), 2 * (radius + margin), 20.0]
    origin_obst = [cx - side / 2, cy - side / 2, 0.0]
    n_obst = int(round(density * side * side))
    return pillar_forest(lo, dimension, origin_obst, [side, side, 0.0], n_obst, seed=seed, vox=vox)


# generate_random_grid.py:66-81: (rel_origin_y, rel_origin_z, length, height) of the fifteen square gaps of the wall
WALL_GAPS = ((2.5, 5, 1.5, 2), (5, 7.5, 3, 3), (7.5, 5, 3, 2), (10, 9.5, 2, 3), (12.5, 4, 2, 2), (15, 12.5, 2, 2),
             (17.5, 3, 2, 1.5), (20, 8.5, 3, 1.5), (22.5, 12, 1.5, 2), (25, 5.5, 2, 2), (27, 4, 1.2, 2), (30, 7, 2, 2),
             (32.5, 12, 3, 1.5), (35, 9, 2, 2), (37.5, 7, 1.5, 1.5))


def _listed_to_occupied(listed):
    """EnvironmentBuilder::AddObstacles on the voxel list generate_random_grid.py writes: a listed voxel is given by its LOWER
    CORNER and the obstacle size is 0.01 m, so floor((c -+ 0.005)/vox) marks voxels {i-1, i} on every axis."""
    occ = listed.copy()
    for ax in range(3):
        sh = np.zeros_like(occ)
        sl_dst = [slice(None)] * 3
        sl_src = [slice(None)] * 3
        sl_dst[ax], sl_src[ax] = slice(0, -1), slice(1, None)
        sh[tuple(sl_dst)] = occ[tuple(sl_src)]
        occ |= sh
    return occ


def cylinder_voxels(cx, cy, cz, z_lo, z_hi, nx, ny, nz, vox=VOX):
    """Voxel columns (i, j) and the z range [k0, k1) a vertical cylinder of radius 0.05 m and height 20 m centred at (cx, cy, cz) lists
    (Cylinder.occupy_voxels, shapes.py:243-264, clipped by its RandomVolume): the mesh circle of radius 0.05 m is taken at its four
    bounding-square corners. Against the reference's own run on its own 90 centres (tests/golden/shapes_cylinders.npz, minted by
    tests/golden/make_shapes_golden.py): no voxel the reference lists is missing; 4 of ~150 columns are extra (a corner of the
    square lies 0.07 m from the axis, the circle 0.05 m) and 3 of 90 pillars are one voxel taller (the reference's last z sample lies
    0.05 m below the clipped top) - tests/test_scenarios.py::test_cylinders_against_the_reference_run."""
    k0 = max(0, int(np.floor((max(z_lo, cz - 10.0) - z_lo) / vox)))
    k1 = min(nz, int(np.ceil((min(z_hi, cz + 10.0) - z_lo) / vox)))
    cols = []
    for sx in (-0.05, 0.05):                  # the mesh circle of radius 0.05 m
        for sy in (-0.05, 0.05):
            i, j = int(np.floor((cx + sx) / vox)), int(np.floor((cy + sy) / vox))
            if 0 <= i < nx and 0 <= j < ny and (i, j) not in cols:
                cols.append((i, j))
    return cols, k0, k1


def forest_wall_forest(tiles_y=1, tiles_z=1, seed=0, n_cyl=(90, 180), vox=VOX):
    """generate_random_grid.py:57-115: a 100 x 30 x 15 m grid at origin (0, 0, -6); a wall at x = 48 (0.3 m thick, meshed at
    vox/2) with fifteen square gaps; 90 vertical cylinders of radius 0.05 m in x in [3, 33] and 180 in x in [63, 93], full
    height. `tiles_y`, `tiles_z` repeat the 30 m x 15 m cross-section (wall pattern and pillar density) so that a y-z
    lattice of agents fits (cfg 5: "widen y/z to fit the lattice"). Returns (occupancy, origin)."""
    nx, ny1, nz1 = 334, 100, 50                      # ceil(100/0.3), ceil(30/0.3), ceil(15/0.3)
    ny, nz = ny1 * tiles_y, nz1 * tiles_z
    listed = np.zeros((nz, ny, nx), bool)
    # wall (shapes.py Wall.occupy_voxels :399-420): mesh samples every vox/2 along y, z and through the thickness; a voxel is
    # listed when at least one sample in it is not inside a gap (Wall.includes :365-382). The float arithmetic of the script
    # is kept literally (np.arange sample positions, int() truncation) because it decides voxels at gap borders.
    mesh = vox / 2
    ys = np.arange(0.0, 100.0, mesh)                 # default wall length / height: 100 m, clipped by the grid
    zs = -6.0 + np.arange(0.0, 100.0, mesh)
    jj = ((ys - 0.0) / vox).astype(int)              # to_voxel_coordinates (:100-114), truncation
    kk = ((zs - (-6.0)) / vox).astype(int)
    ys, jj = ys[jj < ny1], jj[jj < ny1]
    zs, kk = zs[kk < nz1], kk[kk < nz1]
    in_gap = np.zeros((len(zs), len(ys)), bool)
    for (gy, gz, length, height) in WALL_GAPS:
        oy, oz = ys - (0.0 + 1.0 * gy), zs - (-6.0 + 1.0 * gz)
        in_gap |= np.outer((0 <= oz) & (oz < height), (0 <= oy) & (oy < length))
    wall = np.zeros((nz1, ny1), bool)
    np.logical_or.at(wall, (kk[:, None].repeat(len(ys), 1), jj[None, :].repeat(len(zs), 0)), ~in_gap)
    ix_wall = 159                                     # x samples 47.7 and 47.85 -> voxel 159
    listed[:, :, ix_wall] = np.tile(wall, (tiles_z, tiles_y))
    # RandomVolume([[x0, 0, -6], [30, 30, 15]]).add_random_cylinders(n, direction +z, radius 0.05, height 20) (shapes.py:423-446):
    # the cylinder's CENTRE is uniform in the 30 x 30 x 15 m box, it reaches 10 m up and down and is clipped by the box — so a pillar
    # spans the full height only when its centre lies in z in [-1, 4]. With tiles in z the box grows and the count is scaled so that
    # a horizontal slice meets about as many pillars as in the shipped world.
    rng = np.random.default_rng(seed)
    z_lo, z_hi = -6.0, -6.0 + 15.0 * tiles_z
    scale = tiles_y * (1.0 if tiles_z == 1 else 0.75 * tiles_z)
    for (x_lo, cnt) in ((3.0, n_cyl[0]), (63.0, n_cyl[1])):
        for _ in range(int(round(cnt * scale))):
            cx, cy, cz = rng.uniform(x_lo, x_lo + 30.0), rng.uniform(0.0, 30.0 * tiles_y), rng.uniform(z_lo, z_hi)
            cols, k0, k1 = cylinder_voxels(cx, cy, cz, z_lo, z_hi, nx, ny, nz, vox)
            for (i, j) in cols:
                listed[k0:k1, j, i] = True
    occ = _listed_to_occupied(listed)
    return (occ.astype(np.int8) * 100), np.array([0.0, 0.0, -6.0])


def inflate(occ, inflation_dist=0.3, vox=VOX):
    """VoxelGrid::InflateObstacles (voxel_grid.cpp:249-276) with the mask of CreateMask(inflation_dist, 1) (:192-226): with
    rn = ceil(dist / vox), every offset n in [-rn, rn]^3 with |hypot(n) - 1| * vox < dist (one voxel is taken off the distance)
    and 100 * (1 - hypot(n) / (rn + 1)) > 1e-3 — for the shipped 0.3 m that is the whole 3 x 3 x 3 cube around an occupied voxel.
    numpy harness version; the device version (bit-exact against the literal loops) is hdsm_map_preprocess (row f4)."""
    src = occ >= 100
    out = src.copy()
    if not inflation_dist > 0:
        return np.where(out, np.int8(100), np.int8(0)).astype(np.int8)
    rn = int(np.ceil(inflation_dist / vox))
    nz, ny, nx = occ.shape
    for dx in range(-rn, rn + 1):
        for dy in range(-rn, rn + 1):
            for dz in range(-rn, rn + 1):
                h3 = np.hypot(np.hypot(dx, dy), dz)
                if abs(h3 - 1) * vox >= inflation_dist or not (100.0 * (1 - h3 / (rn + 1)) > 1e-3):
                    continue
                zs, zd = slice(max(0, -dz), nz - max(0, dz)), slice(max(0, dz), nz - max(0, -dz))
                ys, yd = slice(max(0, -dy), ny - max(0, dy)), slice(max(0, dy), ny - max(0, -dy))
                xs, xd = slice(max(0, -dx), nx - max(0, dx)), slice(max(0, dx), nx - max(0, -dx))
                out[zd, yd, xd] |= src[zs, ys, xs]
    return np.where(out, np.int8(100), np.int8(0)).astype(np.int8)
