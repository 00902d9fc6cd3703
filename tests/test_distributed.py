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


def _oracle_rollout(env_ids, ticks=6, seed=4):
    """The bench's workload on the CPU oracle for the given GLOBAL env ids: per-env cube count and sizes, start poses and
    the action stream are all keyed by the global id, exactly as bench.py / the device rollout key them."""
    from mujoco_robot_environments_amd import placement, rng
    from mujoco_robot_environments_amd.model import compile as MC
    from oracle import oracle as O
    A = MC.compile_scene()
    om = O.Model(MC.to_blob(A))
    ids = np.asarray(env_ids)
    nprops, sizes = rng.prop_params(seed, ids)
    pose, ok = placement.sample_poses(seed, ids, nprops, sizes, (0.35, -0.4, 0.43), (0.55, 0.4, 0.435))
    assert ok.all()
    acts = rng.random_actions(seed, ids, np.arange(ticks), scale=0.3).astype(np.float32).astype(np.float64)
    qpos, qvel = np.zeros((len(ids), 43)), np.zeros((len(ids), 39))
    for i in range(len(ids)):
        e = O.Env(om, int(nprops[i]), sizes[i])
        e.set_solver("Newton")
        q = e.arr("qpos")
        q[:7] = A["home_qpos"]
        for p in range(int(nprops[i])):
            q[15 + 7 * p: 22 + 7 * p] = pose[i, p]
        e.forward()
        for t in range(ticks):
            e.arr("ctrl")[:] = acts[t, i]
            e.step(5)
        qpos[i], qvel[i] = e.arr("qpos")[:43], e.arr("qvel")[:39]
    return qpos, qvel, np.zeros(len(ids), np.uint32)


def _worker_oracle(rank, world, port, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mujoco_robot_environments_amd import distributed as D
    local = D.pack_final_state(*_oracle_rollout(D.shard_env_ids(total, rank, world)))
    allt = D.gather_final_state(local)
    if rank == 0:
        torch.save(allt, out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_physics_rollout_equals_single_rank(tmp_path):
    """The same with a REAL rollout behind it: two gloo ranks each step their shard of 8 envs of the bench workload on the
    CPU oracle (per-env scene, start poses and actions drawn by global env id) and gather; one process steps all 16.
    Bit for bit: nothing an env sees depends on which rank owns it or on how many ranks there are."""
    from mujoco_robot_environments_amd import distributed as D
    from oracle import oracle as O
    O.build()
    total, world = 16, 2
    out = str(tmp_path / "gathered_physics.pt")
    port = 29500 + ((os.getpid() + 137) % 1000)
    mp.spawn(_worker_oracle, args=(world, port, total, out), nprocs=world, join=True)
    got = torch.load(out)
    ref = D.pack_final_state(*_oracle_rollout(np.arange(total)))
    assert torch.equal(got, ref)
    qp, qv, _ = D.unpack_final_state(got)
    assert np.isfinite(qp).all() and np.abs(qv[:, :7]).max() > 1e-3     # the arms moved
