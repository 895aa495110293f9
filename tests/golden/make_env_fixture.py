"""Mint tests/golden/env_long_occupancy.npz from the reference's shipped forest+wall+forest world.

Input: env_builder/config/env_long_config.yaml of the reference (a DATA file: grid geometry + the list of 26 130 obstacle
positions written by env_builder/scripts/generate_random_grid.py). The occupancy is what EnvironmentBuilder::AddObstacles
(environment_builder.cpp:189-231) makes of it with voxel_grid_util::AddObstacle (voxel_grid.cpp:314-332): every listed
position is the lower corner of a voxel and the obstacle size is 0.01 m, so floor((c -+ 0.005) / vox) marks the two
voxels either side of that corner on every axis. Runs in the build container only (reads /root/reference).

usage: python tests/golden/make_env_fixture.py
"""
import os

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    d = yaml.safe_load(open("/root/reference/env_builder/config/env_long_config.yaml"))["env_builder_node"]["ros__parameters"]
    vox = float(d["vox_size"])
    dim = np.ceil(np.array(d["dimension_grid"]) / vox).astype(int)  # environment_builder.cpp:146-151
    p = np.array(d["position_obst_vec"]).reshape(-1, 3)
    size = np.array(d["size_obst"])
    lo = np.floor((p - size / 2) / vox).astype(int)
    hi = np.floor((p + size / 2) / vox).astype(int)
    occ = np.zeros((dim[2], dim[1], dim[0]), bool)
    for a, b in zip(lo, hi):
        a, b = np.maximum(a, 0), np.minimum(b, dim - 1)
        occ[a[2]:b[2] + 1, a[1]:b[1] + 1, a[0]:b[0] + 1] = True
    np.savez_compressed(os.path.join(HERE, "env_long_occupancy.npz"), packed=np.packbits(occ), shape=np.array(occ.shape),
                        origin=np.array(d["origin_grid"]), vox=vox, n_listed=len(p))
    print("occupied voxels", int(occ.sum()), "grid [z][y][x]", occ.shape)


if __name__ == "__main__":
    main()
