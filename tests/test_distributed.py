"""world_size-2 gloo test of the N>1 path (runs on CPU): env-axis sharding keyed by global
env id and the end-of-rollout gather reproduce the single-rank result bit for bit."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_rollout(env_ids):
    """Stand-in for the per-rank device rollout: any pure function of the global env id
    (here the action stream itself) -- what is under test is sharding + gather order."""
    from mujoco_robot_environments_amd import rng
    a = rng.random_actions(9, env_ids, np.arange(3))          # [3, n, 8]
    qpos = np.tile(a[0], (1, 6))[:, :43]
    qvel = np.tile(a[1], (1, 5))[:, :39]
    status = (np.asarray(env_ids) % 5).astype(np.uint32)
    return qpos, qvel, status


def _worker(rank, world, port, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mujoco_robot_environments_amd import distributed as D
    ids = D.shard_env_ids(total, rank, world)
    local = D.pack_final_state(*_fake_rollout(ids))
    allt = D.gather_final_state(local)
    if rank == 0:
        torch.save(allt, out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gather_equals_single_rank(tmp_path):
    from mujoco_robot_environments_amd import distributed as D
    total, world = 64, 2
    out = str(tmp_path / "gathered.pt")
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, total, out), nprocs=world, join=True)
    got = torch.load(out)
    ref = D.pack_final_state(*_fake_rollout(np.arange(total)))
    assert torch.equal(got, ref)
    qp, qv, st = D.unpack_final_state(got)
    assert qp.shape == (total, 43) and qv.shape == (total, 39)
    assert np.array_equal(st, np.arange(total) % 5)


def test_shard_ids_partition():
    from mujoco_robot_environments_amd import distributed as D
    ids = np.concatenate([D.shard_env_ids(32768, r, 8) for r in range(8)])
    assert np.array_equal(ids, np.arange(32768))
    assert len(D.shard_env_ids(32768, 3, 8)) == 4096
