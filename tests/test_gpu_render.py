"""GPU tests of the batched overhead camera (mre_render) against the numpy
ray-casting oracle (oracle/render_oracle.py), at the reference's resolution (480 x 640, fovy 61,
config/arena/cameras/transporter_data_collection.yaml)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAM_POS = np.array([0.7, 0.0, 1.3])
CAM_QUAT = np.array([0.707, 0.0, 0.0, -0.707])
H, W, FOVY = 480, 640, 61.0


def _scene(N, seed=0):
    import bench
    from mujoco_robot_environments_amd import rng
    from mujoco_robot_environments_amd.physics import BatchedPhysics
    ids = np.arange(N)
    phys = BatchedPhysics(N)
    nprops, sizes = bench.setup_envs(phys, seed, ids)
    u = rng.uniform(seed + 77, ids, [0], 12)[0]
    prop_rgb = (u.reshape(N, 4, 3) * 255).astype(np.uint8)
    geom_rgb = np.linspace(0.2, 0.9, 60).reshape(20, 3).astype(np.float32)
    phys.set_render_colours(prop_rgb, geom_rgb)
    return phys, nprops, sizes, prop_rgb, geom_rgb


def _cam():
    from mujoco_robot_environments_amd.model import compile as MC
    return MC.q2m(CAM_QUAT / np.linalg.norm(CAM_QUAT))


def test_images_match_the_ray_casting_oracle(compiled_model):
    from oracle import render_oracle as RO
    A, _ = compiled_model
    N = 3
    phys, nprops, sizes, prop_rgb, geom_rgb = _scene(N)
    # move the arm over the table in env 1 so that robot hulls are in view
    qp = phys.qpos().copy()
    qp[1, :7] = [0.3, 0.4, 0.0, -1.6, 0.0, 2.0, 0.8]
    phys.set_state(qp, phys.qvel())
    Rc = _cam()
    rgb, depth, seg = phys.render(CAM_POS, Rc, FOVY, H, W)
    rgb, depth, seg = rgb.cpu().numpy(), depth.cpu().numpy(), seg.cpu().numpy()
    qpos = phys.qpos()
    for i in range(N):
        o_rgb, o_depth, o_seg = RO.render(A, qpos[i, :43], int(nprops[i]), sizes[i], prop_rgb[i], geom_rgb,
                                          CAM_POS, Rc, FOVY, H, W)
        same = seg[i] == o_seg
        assert same.mean() > 0.9995, (i, same.mean())       # silhouettes may flip a pixel (fp32 vs fp64)
        assert np.abs(depth[i][same] - o_depth[same]).max() < 2e-5
        drgb = np.abs(rgb[i][same].astype(int) - o_rgb[same].astype(int))
        assert (drgb <= 1).mean() > 0.9999 and drgb.max() <= 3, (i, drgb.max())
        for p in range(int(nprops[i])):
            # every cube on the table is in view (unless the arm of env 1 hangs over it)
            assert (seg[i] == 12 + p).sum() > 100 or i == 1, (i, p, (seg[i] == 12 + p).sum(), qpos[i, 15 + 7 * p: 18 + 7 * p])
    assert (seg[1] >= 2).any() and ((seg[1] >= 2) & (seg[1] <= 11)).sum() > 500  # robot hulls visible in env 1
    phys.close()


def test_render_respects_mask_and_leaves_the_state_alone(compiled_model):
    import torch
    N = 4
    phys, nprops, sizes, prop_rgb, geom_rgb = _scene(N, seed=3)
    before = (phys.qpos().copy(), phys.qvel().copy())
    Rc = _cam()
    _, d_all, _ = phys.render(CAM_POS, Rc, FOVY, H, W, rgb=False, seg=False)
    n = phys.num_envs
    sentinel = torch.full((n, H, W), -7.0, dtype=torch.float32, device=phys.device)
    from mujoco_robot_environments_amd import lib as L
    from mujoco_robot_environments_amd.physics import _ptr
    cp = np.ascontiguousarray(CAM_POS, np.float32); cm = np.ascontiguousarray(Rc, np.float32).reshape(9)
    mask = np.array([1, 0, 1, 0], np.uint8)
    L.check(L.lib().mre_render(phys._h, _ptr(cp), _ptr(cm), FOVY, H, W, None, _ptr(sentinel), None, _ptr(mask)), "mre_render")
    phys.sync()
    out = sentinel.cpu().numpy()
    assert (out[1] == -7).all() and (out[3] == -7).all()
    assert np.array_equal(out[0], d_all[0].cpu().numpy()) and np.array_equal(out[2], d_all[2].cpu().numpy())
    assert np.array_equal(before[0], phys.qpos()) and np.array_equal(before[1], phys.qvel())
    # argument checks
    with pytest.raises(L.MreError):
        L.check(L.lib().mre_render(phys._h, _ptr(cp), _ptr(cm), FOVY, H, 642, None, _ptr(sentinel), None, None), "mre_render")
    phys.close()


def test_depth_back_projects_to_the_cube_positions(compiled_model):
    """pixel_2_world of the reference (tasks/rearrangement.py:500-530): depth at a cube's pixel,
    pushed back through the pinhole model, lands on the cube's top face."""
    N = 64
    phys, nprops, sizes, prop_rgb, geom_rgb = _scene(N, seed=5)
    Rc = _cam()
    _, depth, seg = phys.render(CAM_POS, Rc, FOVY, H, W, rgb=False)
    depth, seg = depth.cpu().numpy(), seg.cpu().numpy()
    qpos = phys.qpos()
    f = 0.5 * H / np.tan(np.deg2rad(FOVY) / 2)
    for i in range(N):
        for p in range(int(nprops[i])):
            c = qpos[i, 15 + 7 * p: 18 + 7 * p].astype(float)
            c[2] += sizes[i, p, 2]   # centre of the top face (what an overhead ray through its pixel hits)
            cc = Rc.T @ (c - CAM_POS)
            u = int(round((W - 1) / 2 + f * cc[0] / -cc[2])); v = int(round((H - 1) / 2 - f * cc[1] / -cc[2]))
            assert seg[i, v, u] == 12 + p
            cam = np.array([(u - (W - 1) / 2) / f, -(v - (H - 1) / 2) / f, -1.0]) * depth[i, v, u]
            world = CAM_POS + Rc @ cam
            assert abs(world[2] - c[2]) < 2e-3     # top face (cube may be tilted by < 1 mm)
            assert np.abs(world[:2] - c[:2]).max() < 3e-3              # within a pixel (1.6 mm) of the centre
    phys.close()
