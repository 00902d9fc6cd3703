"""Counter-based action stream keyed by (seed, global env id, tick, channel).

The reference draws nothing on this path (actions come from its caller); the
benchmark / parity configs of SURVEY.md section 8(d) need CPU and GPU -- and any
sharding of the env axis over ranks -- to see identical numbers, so the stream is
a pure function of its key (splitmix64 finaliser), never of call order.
"""
from __future__ import annotations

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def _mix(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (x + _G)
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        x = x ^ (x >> np.uint64(31))
    return x


def uniform(seed: int, env_ids, ticks, channels: int) -> np.ndarray:
    """U[0,1) of shape [len(ticks), len(env_ids), channels] (float64)."""
    e = np.asarray(env_ids, np.uint64)[None, :, None]
    t = np.asarray(ticks, np.uint64)[:, None, None]
    c = np.arange(channels, dtype=np.uint64)[None, None, :]
    with np.errstate(over="ignore"):
        key = _mix(np.uint64(seed) + _G * e)
        key = _mix(key ^ (t * _M1))
        key = _mix(key ^ (c * _M2))
    return (key >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def random_actions(seed: int, env_ids, ticks, scale: float = 1.0) -> np.ndarray:
    """ctrl[T, N, 8]: arm torques U(-range, range)*scale (motor.yaml:2-5: 87 / 12 Nm),
    finger command U(0, 255) (min_max.yaml:3-4)."""
    u = uniform(seed, env_ids, ticks, 8)
    rng = np.array([87.0, 87.0, 87.0, 87.0, 12.0, 12.0, 12.0])
    out = np.empty_like(u)
    out[..., :7] = (2.0 * u[..., :7] - 1.0) * rng * scale
    out[..., 7] = u[..., 7] * 255.0
    return out


def prop_params(seed: int, env_ids, min_size: float = 0.015, max_size: float = 0.016):
    """Per-env cube count 2 + (env_id mod 3) (SURVEY 8(d) cfg 2) and half sizes
    U(min,max) with all three equal (colour_splitter.yaml:3-4, props.py:280)."""
    e = np.asarray(env_ids, np.int64)
    nprops = (2 + (e % 3)).astype(np.int32)
    u = uniform(seed ^ 0x5A5A, e, [0], 4)[0]
    s = min_size + (max_size - min_size) * u
    return nprops, np.repeat(s[:, :, None], 3, axis=2).astype(np.float64)
