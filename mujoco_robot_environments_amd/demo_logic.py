"""Batched demonstration logic around the step: vectorised
restatement of ``sort_colours`` / ``prop_pick`` / ``prop_place`` of the reference
(tasks/rearrangement.py:579-751) over thousands of environments.  Pure numpy functions of the
gathered cube poses, so they are testable without a GPU and give the same answer however the
env axis is sharded (every random draw is keyed by the global env id and a per-env call counter).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import rng
from .model import compile as _compile

PLACE_CLEARANCE = 0.05   # contact.dist <= 0.05 rejects a place pose (tasks/rearrangement.py:623)
MAX_PLACE_ATTEMPTS = 10000  # tasks/rearrangement.py:629


def quat_to_yaw_deg(quat_wxyz: np.ndarray) -> np.ndarray:
    """atan2(R[1,0], R[0,0]) in degrees for quaternions [..., 4] (w, x, y, z)."""
    w, x, y, z = np.moveaxis(np.asarray(quat_wxyz, np.float64), -1, 0)
    n = np.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    r10 = 2 * (x * y + w * z)
    r00 = w * w + x * x - y * y - z * z
    return np.rad2deg(np.arctan2(r10, r00))


def grasp_quat(rz_deg: np.ndarray) -> np.ndarray:
    """mju_mat2Quat(R.from_euler('xyz', [0, 180, rz])) for an array of yaw angles:
    Rz(rz) Ry(180 deg) -> (w, x, y, z) through the same matrix->quaternion routine."""
    rz = np.deg2rad(np.atleast_1d(np.asarray(rz_deg, np.float64)))
    out = np.zeros((len(rz), 4))
    ry = np.array([[-1.0, 0, 0], [0, 1.0, 0], [0, 0, -1.0]])
    for i, a in enumerate(rz):
        c, s = np.cos(a), np.sin(a)
        out[i] = _compile.m2q(np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]]) @ ry)
    return out


def target_bounds(task_cfg, colours: Sequence[Sequence[str]], nmax: int = 4) -> Tuple[np.ndarray, np.ndarray]:
    """Per (env, prop) xy bounds of the colour's target zone (get_location_bounds, :703-710)."""
    n = len(colours)
    lo = np.zeros((n, nmax, 3))
    hi = np.zeros((n, nmax, 3))
    cache: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
    for col, zone in task_cfg.colour_target_map.items():
        t = task_cfg.target_locations[zone]
        cache[col] = (np.array([t["location"][0] - t["size"][0] / 2, t["location"][1] - t["size"][1] / 2, 0.4]),
                      np.array([t["location"][0] + t["size"][0] / 2, t["location"][1] + t["size"][1] / 2, 0.4]))
    for i, cols in enumerate(colours):
        for p, c in enumerate(cols):
            lo[i, p], hi[i, p] = cache[c]
    return lo, hi


def first_misplaced(pos: np.ndarray, nprops: np.ndarray, lo: np.ndarray, hi: np.ndarray):
    """Index of the first cube (prop order) outside its target, -1 if all are inside (:727-749)."""
    x, y = pos[..., 0], pos[..., 1]
    inside = (lo[..., 0] <= x) & (x <= hi[..., 0]) & (lo[..., 1] <= y) & (y <= hi[..., 1])
    active = np.arange(pos.shape[1])[None, :] < np.asarray(nprops)[:, None]
    bad = active & ~inside
    idx = np.where(bad.any(axis=1), bad.argmax(axis=1), -1)
    return idx


def batched_pick_pose(pos: np.ndarray, quat: np.ndarray, which: np.ndarray) -> np.ndarray:
    """prop_pick (:579-595) for the selected cube of every env: position = cube position, yaw folded
    with min(|yaw|, |yaw| - 90)."""
    n = len(which)
    sel = np.maximum(which, 0)
    p = pos[np.arange(n), sel]
    yaw = np.abs(quat_to_yaw_deg(quat[np.arange(n), sel]))
    rz = np.minimum(yaw, yaw - 90.0)
    return np.concatenate([p, grasp_quat(rz)], axis=1)


def batched_place_pose(seed: int, env_ids: np.ndarray, call_counts: np.ndarray, pos: np.ndarray,
                       nprops: np.ndarray, half: np.ndarray, which: np.ndarray, lo: np.ndarray,
                       hi: np.ndarray, need: np.ndarray):
    """prop_place (:597-665): uniform samples in [lo, hi] of the selected cube's zone, rejected while
    another cube is closer than 0.05 m; draws keyed by (seed+1, global env id, call*10000 + attempt).
    Returns (pose[N,7], ok[N])."""
    n = len(which)
    sel = np.maximum(which, 0)
    rb = np.sqrt(2.0) * half.max(axis=2)        # footprint circum-radius of a cube lying on the table
    qhome = _compile.m2q(np.array([[-1.0, 0, 0], [0, 1.0, 0], [0, 0, -1.0]]))
    out = np.zeros((n, 7))
    out[:, 3:] = qhome
    ok = ~need
    pending = need.copy()
    others = np.arange(pos.shape[1])[None, :] < np.asarray(nprops)[:, None]
    others[np.arange(n), sel] = False
    l0, h0 = lo[np.arange(n), sel], hi[np.arange(n), sel]
    for att in range(MAX_PLACE_ATTEMPTS):
        idx = np.nonzero(pending)[0]
        if len(idx) == 0:
            break
        u = _uniform_rows(seed + 1, env_ids[idx], call_counts[idx] * MAX_PLACE_ATTEMPTS + att)
        cand = l0[idx] + (h0[idx] - l0[idx]) * u
        d = np.linalg.norm(cand[:, None, :] - pos[idx], axis=2)
        lim = rb[idx, sel[idx]][:, None] + rb[idx] + PLACE_CLEARANCE
        good = ~((d <= lim) & others[idx]).any(axis=1)
        acc = idx[good]
        out[acc, :3] = cand[good]
        ok[acc] = True
        pending[acc] = False
    return out, ok


def _uniform_rows(seed: int, env_ids: np.ndarray, ticks: np.ndarray) -> np.ndarray:
    """rng.uniform for per-row (env, tick) pairs -> [n, 3] (vectorised form of the scalar calls)."""
    e = np.asarray(env_ids, np.uint64)[:, None]
    t = np.asarray(ticks, np.uint64)[:, None]
    c = np.arange(3, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        key = rng._mix(np.uint64(seed) + rng._G * e)
        key = rng._mix(key ^ (t * rng._M1))
        key = rng._mix(key ^ (c * rng._M2))
    return (key >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def batched_sort_colours(task_cfg, seed: int, env_ids: np.ndarray, call_counts: np.ndarray,
                         prop_pose: np.ndarray, nprops: np.ndarray, half: np.ndarray,
                         colours: List[List[str]]):
    """sort_colours (:700-751) for every env.  Returns (in_progress[N], pick[N,7], place[N,7],
    failed[N]) where failed marks envs whose place sampling exhausted its attempts (the reference
    raises there and the caller abandons the episode)."""
    pos = prop_pose[..., :3].astype(np.float64)
    quat = prop_pose[..., 3:7].astype(np.float64)
    lo, hi = target_bounds(task_cfg, colours, pos.shape[1])
    which = first_misplaced(pos, nprops, lo, hi)
    need = which >= 0
    pick = batched_pick_pose(pos, quat, which)
    place, ok = batched_place_pose(seed, np.asarray(env_ids), np.asarray(call_counts), pos, np.asarray(nprops),
                                   np.asarray(half, np.float64), which, lo, hi, need)
    failed = need & ~ok
    return need & ok, pick, place, failed, which
