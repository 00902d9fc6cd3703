"""Env-axis sharding over the GPUs of one node.

Environments are independent (no cross-env term anywhere in reference
models/robot_arm.py:61-94 or the physics), so rank r simply owns the contiguous block
of global env ids [r*n, (r+1)*n); every random draw is keyed by the GLOBAL id
(rng.py), so results do not depend on the number of ranks.  No collective runs during
a rollout; the only exchange is the end-of-rollout gather of final (qpos, qvel, status)
via torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in CPU tests).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_env_ids(total_envs: int, rank: int, world: int) -> np.ndarray:
    if total_envs % world:
        raise ValueError(f"{total_envs} envs do not split evenly over {world} ranks")
    n = total_envs // world
    return np.arange(rank * n, (rank + 1) * n)


def pack_final_state(qpos: np.ndarray, qvel: np.ndarray, status: np.ndarray) -> torch.Tensor:
    """[n, 43+39+1] fp32 rows (status carried as an exact small integer)."""
    return torch.from_numpy(np.concatenate([qpos.astype(np.float32), qvel.astype(np.float32),
                                            status.astype(np.float32)[:, None]], axis=1))


def gather_final_state(local: torch.Tensor) -> torch.Tensor:
    """all_gather_into_tensor along the env axis; row order = global env id."""
    if not dist.is_initialized():
        return local
    out = torch.empty((dist.get_world_size() * local.shape[0], local.shape[1]), dtype=local.dtype,
                      device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def unpack_final_state(t: torch.Tensor) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    a = t.cpu().numpy()
    return a[:, :43], a[:, 43:82], a[:, 82].astype(np.uint32)
