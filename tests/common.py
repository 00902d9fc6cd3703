"""Shared helpers for the test-suite (oracle-side scene set-up)."""
import numpy as np

HOME = np.array([0, -0.785, 0, -2.356, 0, 1.571, 0.785])


def table_pose(p, z_extra=0.0, half=0.0155):
    """A deterministic, well separated cube pose on the table top (z = 0.4)."""
    xs = [0.40, 0.50, 0.45, 0.38]
    ys = [-0.30, -0.10, 0.12, 0.33]
    return np.array([xs[p], ys[p], 0.4 + half + z_extra, 1.0, 0.0, 0.0, 0.0])


def init_oracle_env(env, nprops, sizes=None, yaw=None, z_extra=0.0):
    q = env.arr("qpos")
    q[:7] = HOME
    for p in range(nprops):
        half = 0.0155 if sizes is None else float(sizes[p][2])
        pose = table_pose(p, z_extra, half)
        if yaw is not None:
            pose[3:7] = [np.cos(yaw[p] / 2), 0, 0, np.sin(yaw[p] / 2)]
        q[15 + 7 * p: 22 + 7 * p] = pose
    env.arr("qvel")[:] = 0
    return q[:43].copy()
