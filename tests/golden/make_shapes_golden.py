#!/usr/bin/env python3
"""Mint tests/golden/shapes_cylinders.npz by RUNNING the reference's own environment generator (build container only).

The one piece of the reference that runs here unmodified is Python: env_builder/scripts/shapes.py (numpy + matplotlib are in the
image). This script imports it from /root/reference and executes what generate_random_grid.py:93-112 executes for the first forest
block of BASELINE cfg 5's world — RandomVolume([[3, 0, -6], [30, 30, 15]], seed 0).add_random_cylinders(90, +z, r = 0.05,
h = 20) voxelised into VoxelGrid((100, 30, 15), 0.3, (0, 0, -6)) — and stores DATA only: the 90 cylinder centres the reference's
RNG draws (Python `random`, seed 0) and, per cylinder, the voxels its Cylinder.occupy_voxels marks. Nothing of the reference's
source travels. tests/test_scenarios.py compares multi_agent_pkgs_amd/scenarios.py against it.

usage (container with /root/reference): python tests/golden/make_shapes_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/env_builder/scripts")
import matplotlib
matplotlib.use("Agg")
from shapes import RandomVolume, VoxelGrid  # noqa: E402  (the reference's module)

HERE = os.path.dirname(os.path.abspath(__file__))
seed = 0
centres, counts, voxels = [], [], []
rv = RandomVolume([[3, 0, -6], [30, 30, 15]], seed)
rv.add_random_cylinders(90, direction_range=[[0.0, 0.0, 1.0], [0.0, 0.0, 1.0]], radius_range=[0.05, 0.05], height_range=[20.0, 20.0])
for cyl in rv.shapes:
    vg = VoxelGrid((100, 30, 15), 0.3, (0, 0, -6))
    cyl.occupy_voxels(vg, container_volume=rv)
    nx, ny, nz = vg.grid_size
    occ = np.array(vg.data, dtype=bool).reshape(nz, ny, nx)   # index = i + nx j + nx ny k
    kk, jj, ii = np.nonzero(occ)
    centres.append(np.asarray(cyl.axis_origin, float))
    counts.append(len(ii))
    voxels.append(np.stack([ii, jj, kk], 1).astype(np.int32))
np.savez_compressed(os.path.join(HERE, "shapes_cylinders.npz"), centres=np.array(centres), counts=np.array(counts, np.int32),
                    voxels=np.concatenate(voxels), grid_size=np.array(VoxelGrid((100, 30, 15), 0.3, (0, 0, -6)).grid_size, np.int32),
                    what="reference run: env_builder/scripts/shapes.py, RandomVolume([[3,0,-6],[30,30,15]], 0).add_random_cylinders(90, +z, 0.05, 20) "
                         "-> per cylinder the voxels (i, j, k) Cylinder.occupy_voxels marks in VoxelGrid((100,30,15), 0.3, (0,0,-6))")
print("cylinders", len(centres), "voxels", int(sum(counts)), "grid", VoxelGrid((100, 30, 15), 0.3, (0, 0, -6)).grid_size)
