"""Batched cube placement on the HOST: a conservative closed-form sampler for scenes that must be
identical on the oracle and on the device without a device (tests, __graft_entry__.smoke, the
bench's cpu_baseline leg).  It is NOT what the library does any more: ``mre_place_props`` follows
``PropPlacer.__call__`` (reference environment/prop_initializer.py:164-283) through the narrow phase
on the device -- reject while the prop has a detected contact with any geom but the table, robot geoms
included (tests/test_gpu_env.py checks it against the oracle draw for draw).

Reference semantics per prop: sample position ~ U(workspace), yaw = pi*U(0,1) about z
(tasks/rearrangement.py:190-206), ``physics.forward()``, reject when any non-table
contact involving the prop is detected (``dist <= 0.6`` is always true for a detected
contact, prop_initializer.py:134-138) -- and because every cube carries margin = 0.15
(environment/props.py:239) "detected" means "closer than 0.15 m to any other geom".
Then settle with the robot frozen (>= 0.3 s, :240-258).

Here the geometric test is done in closed form on the host (cube-cube: centre distance
against the bounding-sphere sum + margin, a conservative superset of MuJoCo's SAT test;
the arm at home is > 0.4 m above the workspace) with a counter-based RNG keyed by the
global env id, so results do not depend on how envs are sharded over ranks.  The
settle phase runs on the GPU (BatchedPhysics.step with the freeze flag).
"""
from __future__ import annotations

import numpy as np

from . import rng as _rng

PROP_MARGIN = 0.15  # environment/props.py:239


def sample_poses(seed: int, env_ids, nprops, half_sizes, ws_min, ws_max, max_attempts: int = 1000,
                 max_rounds: int = 50):
    """Returns (pose[N,4,7] (x,y,z,qw,qx,qy,qz), ok[N]); unplaced slots are NaN.

    When a prop cannot be placed within ``max_attempts`` (the first cubes can block the
    0.2 m x 0.8 m workspace) the reference raises RuntimeError and the caller's episode is
    abandoned (transporter_network_data_generation.py:137-139); a batch cannot raise per
    env, so such envs re-draw their whole scene in a further round (new RNG stream)."""
    env_ids = np.asarray(env_ids, np.int64)
    nprops = np.asarray(nprops, np.int64)
    half = np.asarray(half_sizes, np.float64).reshape(len(env_ids), 4, 3)
    pose = np.full((len(env_ids), 4, 7), np.nan)
    ok = np.zeros(len(env_ids), bool)
    for rnd in range(max_rounds):
        idx = np.nonzero(~ok)[0]
        if len(idx) == 0:
            break
        p_, ok_ = _sample_round(seed + 7919 * rnd, env_ids[idx], nprops[idx], half[idx], ws_min, ws_max,
                                max_attempts)
        pose[idx], ok[idx] = p_, ok_
    return pose, ok


def _sample_round(seed, env_ids, nprops, half_sizes, ws_min, ws_max, max_attempts):
    env_ids = np.asarray(env_ids, np.int64)
    N = len(env_ids)
    nprops = np.asarray(nprops, np.int64)
    half = np.asarray(half_sizes, np.float64).reshape(N, 4, 3)
    rb = np.linalg.norm(half, axis=2)                      # bounding radius of each cube
    lo, hi = np.asarray(ws_min, np.float64), np.asarray(ws_max, np.float64)
    pose = np.full((N, 4, 7), np.nan)
    placed = np.zeros((N, 4), bool)
    ok = np.ones(N, bool)
    for p in range(4):
        todo = (nprops > p)
        pending = todo.copy()
        for att in range(max_attempts):
            if not pending.any():
                break
            idx = np.nonzero(pending)[0]
            u = _rng.uniform(seed, env_ids[idx], [p * max_attempts + att], 4)[0]
            pos = lo + (hi - lo) * u[:, :3]
            yaw = np.pi * u[:, 3]
            good = np.ones(len(idx), bool)
            for o in range(p):
                d = np.linalg.norm(pos - pose[idx, o, :3], axis=1)
                lim = rb[idx, p] + rb[idx, o] + PROP_MARGIN
                good &= ~(placed[idx, o] & (d <= lim))
            sel = idx[good]
            pose[sel, p, :3] = pos[good]
            pose[sel, p, 3] = np.cos(yaw[good] / 2)
            pose[sel, p, 4:6] = 0.0
            pose[sel, p, 6] = np.sin(yaw[good] / 2)
            placed[sel, p] = True
            pending[sel] = False
        ok &= ~pending
    return pose, ok


def write_poses(qpos: np.ndarray, pose: np.ndarray, nprops) -> np.ndarray:
    """Insert sampled poses into qpos rows [N, >=43] (free-joint coordinates 15+7p..)."""
    q = qpos.copy()
    for p in range(4):
        m = np.asarray(nprops) > p
        q[m, 15 + 7 * p: 22 + 7 * p] = pose[m, p]
    return q
