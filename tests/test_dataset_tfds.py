"""The shards against TFDS itself.  tensorflow_datasets is not installed on the build image nor on the GPU boxes of
this pipeline, so this test SKIPS there; it is the check to run wherever `tfds.builder_from_directory` -- the
reference's consumer of TFDSBackendWriter's output -- is importable."""
import collections

import numpy as np
import pytest

tfds = pytest.importorskip("tensorflow_datasets")


def test_builder_from_directory_reads_the_shards(tmp_path):
    from mujoco_robot_environments_amd import dataset as D
    TimeStep = collections.namedtuple("TimeStep", ["step_type", "reward", "discount", "observation"])
    H, W, N = 12, 16, 3

    class Env:
        num_envs = N

        def get_camera_metadata(self):
            return {"intrinsics": {"fx": -405.0, "fy": 405.0, "cx": 7.5, "cy": 5.5},
                    "extrinsics": {"x": 0.7, "y": 0.0, "z": 1.3, "qx": 0.0, "qy": 0.0, "qz": -0.707, "qw": 0.707}}
    rs = np.random.RandomState(0)

    def ts():
        return TimeStep(0, 0.0, 0.0, {"overhead_camera/rgb": rs.randint(0, 255, (N, H, W, 3)).astype(np.uint8),
                                     "overhead_camera/depth": rs.rand(N, H, W).astype(np.float32)})
    w = D.EpisodeWriter(str(tmp_path), "colour_splitter_test", H, W, max_episodes_per_file=2)
    first = ts()
    acts = []
    with D.BatchedEpisodeLogger(Env(), w) as log:
        log.reset(first)
        for _ in range(2):
            a = {"pose": rs.rand(N, 7), "pixel_coords": rs.randint(0, 600, (N, 2)), "gripper_rot": 0.0}
            acts.append(a)
            log.step(a, ts())
    w.close()
    builder = tfds.builder_from_directory(str(tmp_path))
    assert builder.info.splits["train"].num_examples == N
    eps = list(tfds.as_numpy(builder.as_dataset(split="train", shuffle_files=False)))
    assert len(eps) == N
    for i, e in enumerate(eps):
        steps = list(e["steps"])
        assert len(steps) == 3
        assert np.array_equal(steps[0]["observation"]["overhead_camera/rgb"], first.observation["overhead_camera/rgb"][i])
        assert np.allclose(steps[0]["action"]["pose"], acts[0]["pose"][i], atol=1e-6)
        assert steps[0]["is_first"] and steps[-1]["is_last"] and not steps[1]["is_last"]
        assert abs(float(e["intrinsics"]["fx"]) + 405.0) < 1e-4
