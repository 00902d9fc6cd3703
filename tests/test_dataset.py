"""Episode shards: the writer of mujoco_robot_environments_amd/dataset.py with the
feature keys of the reference's ds_config (transporter_network_data_generation.py:56-86).  CPU only."""
import collections
import os

import numpy as np

from mujoco_robot_environments_amd import dataset as D


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert D.crc32c(b"") == 0x00000000
    assert D.crc32c(b"123456789") == 0xE3069283
    assert D.crc32c(bytes(32)) == 0x8A9136AA
    assert D.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert D.crc32c(bytes(range(32))) == 0x46DD794E


def test_example_wire_format_known_bytes():
    """tf.train.Example with one int64 feature {"a": [1, 300]}: bytes worked out by hand from the proto
    definitions (Example.features = 1, Features.feature = 1 (map entry: key = 1, value = 2),
    Feature.int64_list = 3, Int64List.value = 1 packed)."""
    got = D.encode_example({"a": np.array([1, 300])})
    packed = bytes([0x01, 0xAC, 0x02])
    int64_list = bytes([0x0A, len(packed)]) + packed
    feature = bytes([0x1A, len(int64_list)]) + int64_list
    entry = bytes([0x0A, 1]) + b"a" + bytes([0x12, len(feature)]) + feature
    features = bytes([0x0A, len(entry)]) + entry
    assert got == bytes([0x0A, len(features)]) + features
    back = D.decode_example(got)
    assert back["a"].tolist() == [1, 300]


def test_batched_logger_round_trip(tmp_path):
    TimeStep = collections.namedtuple("TimeStep", ["step_type", "reward", "discount", "observation"])
    H, W, N = 12, 16, 5

    class Env:
        num_envs = N

        def get_camera_metadata(self):
            return {"intrinsics": {"fx": -405.0, "fy": 405.0, "cx": 7.5, "cy": 5.5},
                    "extrinsics": {"x": 0.0, "y": 0.0, "z": 0.0, "qx": 0.1, "qy": 0.2, "qz": 0.3, "qw": 0.9}}

    rs = np.random.RandomState(0)

    def ts():
        return TimeStep(0, 0.0, 0.0, {"overhead_camera/rgb": rs.randint(0, 255, (N, H, W, 3)).astype(np.uint8),
                                     "overhead_camera/depth": rs.rand(N, H, W).astype(np.float32)})
    env = Env()
    w = D.EpisodeWriter(str(tmp_path), "colour_splitter_test", H, W, max_episodes_per_file=2)
    seen = []
    with D.BatchedEpisodeLogger(env, w) as log:
        t0 = ts()
        log.reset(t0)
        seen.append(t0)
        for k in range(4):
            act = {"pose": rs.rand(N, 7), "pixel_coords": rs.randint(0, 600, (N, 2)), "gripper_rot": 0.0}
            t = ts()
            active = np.arange(N) != 3 if k >= 2 else None   # env 3 finishes early (task complete)
            log.step(act, t, active)
            seen.append((act, t, active))
    info = w.close()
    assert info["splits"][0]["shardLengths"] == ["2", "2", "1"] and info["version"] == "0.0.1"
    sizes = [os.path.getsize(os.path.join(tmp_path, p)) for p in os.listdir(tmp_path) if "tfrecord" in p]
    assert int(info["splits"][0]["numBytes"]) == sum(sizes) - 16 * N   # payloads without the 16 framing bytes per record
    files = sorted(p for p in os.listdir(tmp_path) if "tfrecord" in p)
    assert files == [f"colour_splitter_test-train.tfrecord-{k:05d}-of-00003" for k in range(3)]
    eps = list(D.read_episodes(str(tmp_path), "colour_splitter_test"))
    assert len(eps) == N
    for i, e in enumerate(eps):
        T = 5 if i != 3 else 3
        s = e["steps"]
        assert s["observation"]["overhead_camera/rgb"].shape == (T, H, W, 3)
        assert s["is_first"].tolist() == [True] + [False] * (T - 1) and s["is_last"].tolist() == [False] * (T - 1) + [True]
        assert np.array_equal(s["observation"]["overhead_camera/rgb"][0], seen[0].observation["overhead_camera/rgb"][i])
        act, t, _ = seen[1]
        assert np.allclose(s["action"]["pose"][0], act["pose"][i], atol=1e-6)
        assert s["action"]["pixel_coords"][0].tolist() == act["pixel_coords"][i].tolist()
        assert np.array_equal(s["observation"]["overhead_camera/depth"][1], t.observation["overhead_camera/depth"][i])
        assert s["observation"]["overhead_camera/rgb"].dtype == np.uint8 and s["action"]["pixel_coords"].dtype == np.int32
        assert s["is_first"].dtype == np.bool_ and s["reward"].dtype == np.float64
        assert abs(e["intrinsics"]["fx"] + 405.0) < 1e-6      # episode_metadata_info entries are top-level features
        assert abs(e["extrinsics"]["qw"] - 0.9) < 1e-6


def test_metadata_files_follow_the_tfds_schema(tmp_path):
    """features.json / dataset_info.json as tfds writes them (proto3 JSON of feature.proto / dataset_info.proto):
    spelled out by hand for the reference's ds_config (transporter_network_data_generation.py:56-86)."""
    import json
    w = D.EpisodeWriter(str(tmp_path), "colour_splitter_x", 480, 640, max_episodes_per_file=10)
    w.close()
    f = json.load(open(tmp_path / "features.json"))
    pkg = "tensorflow_datasets.core.features."
    assert f["pythonClassName"] == pkg + "features_dict.FeaturesDict"
    top = f["featuresDict"]["features"]
    assert sorted(top) == ["extrinsics", "intrinsics", "steps"]       # rlds_base.build_info: {steps, **episode_metadata}
    assert top["steps"]["pythonClassName"] == pkg + "dataset_feature.Dataset" and top["steps"]["sequence"]["length"] == "-1"
    step = top["steps"]["sequence"]["feature"]["featuresDict"]["features"]
    assert sorted(step) == ["action", "discount", "is_first", "is_last", "is_terminal", "observation", "reward"]
    assert step["observation"]["featuresDict"]["features"]["overhead_camera/rgb"] == {
        "pythonClassName": pkg + "tensor_feature.Tensor",
        "tensor": {"shape": {"dimensions": ["480", "640", "3"]}, "dtype": "uint8", "encoding": "none"}}
    assert step["observation"]["featuresDict"]["features"]["overhead_camera/depth"]["tensor"] == {
        "shape": {"dimensions": ["480", "640"]}, "dtype": "float32", "encoding": "none"}
    act = step["action"]["featuresDict"]["features"]
    assert act["pose"]["tensor"] == {"shape": {"dimensions": ["7"]}, "dtype": "float64", "encoding": "none"}
    assert act["pixel_coords"]["tensor"]["dtype"] == "int32"
    assert act["gripper_rot"] == {"pythonClassName": pkg + "scalar.Scalar",
                                  "tensor": {"shape": {}, "dtype": "float64", "encoding": "none"}}
    assert step["is_terminal"]["tensor"]["dtype"] == "bool" and step["reward"]["tensor"]["dtype"] == "float64"
    assert sorted(top["extrinsics"]["featuresDict"]["features"]) == ["qw", "qx", "qy", "qz", "x", "y", "z"]
    keys = {k: (shape, dtype, seq) for k, shape, dtype, seq in D.feature_leaves(f)}
    assert keys["steps/observation/overhead_camera/rgb"] == ((480, 640, 3), "uint8", True)
    assert keys["intrinsics/fx"] == ((), "float64", False) and len(keys) == 21
    info = json.load(open(tmp_path / "dataset_info.json"))
    assert info["name"] == "colour_splitter_x" and info["fileFormat"] == "tfrecord"
    assert info["splits"] == [{"name": "train", "shardLengths": [], "numBytes": "0",
                               "filepathTemplate": "{DATASET}-{SPLIT}.{FILEFORMAT}-{SHARD_X_OF_Y}"}]


def test_uint8_tensor_goes_to_an_int64_list_like_tfds():
    """tfds.features.Tensor(dtype=uint8) with the default Encoding.NONE: one varint per byte in an int64_list
    (values >= 128 take two bytes).  Key order inside the map is sorted, as TFDS's serializer emits it."""
    img = np.array([0, 127, 128, 255], np.uint8)
    got = D.encode_example({"x": img})
    packed = bytes([0x00, 0x7F, 0x80, 0x01, 0xFF, 0x01])
    int64_list = bytes([0x0A, len(packed)]) + packed
    feature = bytes([0x1A, len(int64_list)]) + int64_list
    entry = bytes([0x0A, 1]) + b"x" + bytes([0x12, len(feature)]) + feature
    features = bytes([0x0A, len(entry)]) + entry
    assert got == bytes([0x0A, len(features)]) + features
    assert D.decode_example(got)["x"].tolist() == [0, 127, 128, 255]
    big = np.random.RandomState(1).randint(0, 256, 100000).astype(np.uint8)
    assert np.array_equal(D.decode_example(D.encode_example({"i": big}))["i"], big)
    neg = np.array([-1, 5, 1 << 40])
    assert D.decode_example(D.encode_example({"n": neg}))["n"].tolist() == neg.tolist()


def test_tfrecord_detects_corruption(tmp_path):
    p = tmp_path / "x.tfrecord"
    with open(p, "wb") as f:
        D.write_record(f, b"hello world")
    assert list(D.read_records(str(p))) == [b"hello world"]
    raw = bytearray(open(p, "rb").read())
    raw[14] ^= 1
    open(p, "wb").write(bytes(raw))
    try:
        list(D.read_records(str(p)))
        raise AssertionError("corruption not detected")
    except ValueError:
        pass
