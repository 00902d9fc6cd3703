"""RLDS episode shards in TFDS's on-disk format from batched rollouts -- the on-disk side of the reference's data
generation (transporter_network_data_generation.py:56-111: ``tfds.rlds.rlds_base.DatasetConfig`` +
``envlogger.EnvLogger`` with a ``TFDSBackendWriter``), without envlogger / TensorFlow.

What ``TFDSBackendWriter(data_directory, split_name, max_episodes_per_file, ds_config)`` leaves in ``data_directory``,
and what is written here (all restated from the public TFDS / RLDS / TFRecord / protobuf formats; TensorFlow is
absent from this image, so ``tests/test_dataset_tfds.py`` -- ``tfds.builder_from_directory`` on a written
directory -- skips here and runs wherever tensorflow_datasets is importable):

* ``<name>-<split>.tfrecord-XXXXX-of-NNNNN``: TFRecord files (length, masked CRC-32C of the length, payload, masked
  CRC-32C of the payload), ``max_episodes_per_file`` episodes each (config/dataset/default.yaml:3), one
  ``tf.train.Example`` per EPISODE.  Feature keys are the ``/``-joined paths of ``rlds_base.build_info``'s feature
  tree -- ``steps`` is a ``tfds.features.Dataset`` of the step fields, the ``episode_metadata_info`` entries sit at
  the TOP level beside it -- and every leaf is serialised the way TFDS's example serializer does it: a step field
  is ONE flat list over all steps (leading step axis, then the tensor's own shape), integers AND uint8 AND bool go to
  ``int64_list`` (the default ``Encoding.NONE`` of ``tfds.features.Tensor``: one varint per pixel byte), float32 AND
  float64 go to ``float_list`` (the Example proto has no double list):

      steps/observation/overhead_camera/rgb     int64_list   T * H * W * 3
      steps/observation/overhead_camera/depth   float_list   T * H * W
      steps/action/pose                         float_list   T * 7
      steps/action/pixel_coords                 int64_list   T * 2
      steps/action/gripper_rot, steps/reward, steps/discount          float_list   T
      steps/is_first, steps/is_last, steps/is_terminal                int64_list   T
      intrinsics/{fx,fy,cx,cy}, extrinsics/{x,y,z,qx,qy,qz,qw}        float_list   1   (calibration_metadata, :88-95)

* ``features.json``: the feature tree as TFDS serialises it (proto3 JSON of ``feature.proto``: ``pythonClassName`` +
  one of ``featuresDict`` / ``sequence`` / ``tensor``; int64 fields as strings).
* ``dataset_info.json``: proto3 JSON of ``DatasetInfo`` (name, version 0.0.1 -- TFDSBackendWriter's default --,
  ``fileFormat``, one split with ``shardLengths`` / ``numBytes`` as strings and the standard ``filepathTemplate``).

``read_episodes`` parses a directory back BY ITS features.json (it is a generic reader of this format, not a mirror
of the writer), and tests/test_dataset.py round-trips through it.
"""
from __future__ import annotations

import json
import os
import struct
from typing import Dict, Iterator, List, Optional

import numpy as np

# ------------------------------------------------------------------ CRC-32C (Castagnoli), TFRecord mask
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.zeros((8, 256), np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            t[0, i] = c
        for k in range(1, 8):
            t[k] = (t[k - 1] >> np.uint32(8)) ^ t[0, t[k - 1] & np.uint32(0xFF)]
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    """CRC-32C, slicing-by-8 over numpy words for the bulk (images are megabytes)."""
    t = _crc_table()
    crc = 0xFFFFFFFF
    n = len(data)
    mv = memoryview(data)
    # bulk: process 8 bytes per step in Python is still slow for MBs; use a vectorised fold over
    # independent 4 KiB blocks is not possible for a CRC without combine -- so fall back to the C
    # helper of libmre.so when it is there
    fast = _native_crc()
    if fast is not None:
        return fast(data)
    t0 = t[0]
    for b in mv.tobytes():
        crc = int(t0[(crc ^ b) & 0xFF]) ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


_NATIVE = False


def _native_crc():
    """mre_crc32c of the C-ABI library (hardware CRC32 instruction), if the library is built."""
    global _NATIVE
    if _NATIVE is False:
        _NATIVE = None
        try:
            import ctypes as C
            so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmre.so")
            if os.path.exists(so):
                from . import lib as _lib
                L = _lib.lib()
                L.mre_crc32c.restype = C.c_uint32
                L.mre_crc32c.argtypes = [C.c_char_p, C.c_size_t]
                _NATIVE = lambda d: int(L.mre_crc32c(bytes(d), len(d)))  # noqa: E731
        except Exception:
            _NATIVE = None
    return _NATIVE


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def write_record(f, payload: bytes) -> None:
    head = struct.pack("<Q", len(payload))
    f.write(head)
    f.write(struct.pack("<I", _masked_crc(head)))
    f.write(payload)
    f.write(struct.pack("<I", _masked_crc(payload)))


def read_records(path: str) -> Iterator[bytes]:
    with open(path, "rb") as f:
        while True:
            head = f.read(8)
            if len(head) < 8:
                return
            (n,) = struct.unpack("<Q", head)
            (c1,) = struct.unpack("<I", f.read(4))
            if c1 != _masked_crc(head):
                raise ValueError("TFRecord: bad length CRC")
            payload = f.read(n)
            (c2,) = struct.unpack("<I", f.read(4))
            if c2 != _masked_crc(payload):
                raise ValueError("TFRecord: bad payload CRC")
            yield payload


# ------------------------------------------------------------------ tf.train.Example wire format
def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:   # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _pack_varints(a: np.ndarray) -> bytes:
    """Packed varints of non-negative integers < 2^14 (pixel bytes, flags, pixel coordinates): vectorised."""
    a = a.reshape(-1).astype(np.uint16)
    two = a >= 128
    out = np.empty(a.size + int(two.sum()), np.uint8)
    pos = np.arange(a.size) + np.concatenate([[0], np.cumsum(two)[:-1]]) if a.size else np.zeros(0, np.int64)
    out[pos] = np.where(two, (a & 0x7F) | 0x80, a).astype(np.uint8)
    out[pos[two] + 1] = (a[two] >> 7).astype(np.uint8)
    return out.tobytes()


def _feature(value) -> bytes:
    """One tf.train.Feature: bytes_list for a list of bytes, float_list for floating arrays, int64_list otherwise."""
    if isinstance(value, (list, tuple)) and value and isinstance(value[0], (bytes, bytearray)):
        body = b"".join(_ld(1, bytes(v)) for v in value)
        return _ld(1, body)                                   # bytes_list
    a = np.asarray(value)
    if a.dtype.kind == "f":
        return _ld(2, _ld(1, a.astype("<f4").tobytes()))      # float_list, packed
    # fast path (at most two varint bytes per value): uint8 / bool arrays by type, anything else by its value range
    if a.size and (a.dtype in (np.uint8, np.bool_) or (a.min() >= 0 and a.max() < (1 << 14))):
        return _ld(3, _ld(1, _pack_varints(a)))               # int64_list, packed
    packed = b"".join(_varint(int(v) & 0xFFFFFFFFFFFFFFFF) for v in a.reshape(-1))
    return _ld(3, _ld(1, packed))                             # int64_list, packed


def encode_example(features: Dict[str, object]) -> bytes:
    entries = b""
    for k in sorted(features):
        entry = _ld(1, k.encode()) + _ld(2, _feature(features[k]))
        entries += _ld(1, entry)                              # map<string, Feature> entry
    return _ld(1, entries)                                    # Example.features


def _read_varint(b: bytes, i: int):
    n, s = 0, 0
    while True:
        c = b[i]
        i += 1
        n |= (c & 0x7F) << s
        if not c & 0x80:
            return n, i
        s += 7


def _fields(b: bytes):
    i = 0
    while i < len(b):
        key, i = _read_varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 2:
            n, i = _read_varint(b, i)
            yield f, b[i:i + n]
            i += n
        elif wt == 0:
            v, i = _read_varint(b, i)
            yield f, v
        elif wt == 5:
            yield f, b[i:i + 4]
            i += 4
        else:
            raise ValueError("unsupported wire type")


def _unpack_varints(v: bytes) -> np.ndarray:
    b = np.frombuffer(v, np.uint8)
    if b.size == 0:
        return np.zeros(0, np.int64)
    last = (b & 0x80) == 0                       # final byte of every varint
    if last.all():
        return b.astype(np.int64)
    start = np.concatenate([[0], np.nonzero(last)[0][:-1] + 1])
    length = np.nonzero(last)[0] + 1 - start
    if length.max() <= 2:                        # (values < 2^14: pixel bytes and coordinates)
        lo = (b[start] & 0x7F).astype(np.int64)
        hi = np.where(length == 2, b[np.minimum(start + 1, b.size - 1)].astype(np.int64) << 7, 0)
        return lo + hi
    vals, i = [], 0
    while i < len(v):
        n, i = _read_varint(v, i)
        vals.append(n - (1 << 64) if n >= (1 << 63) else n)
    return np.asarray(vals, np.int64)


def decode_example(payload: bytes) -> Dict[str, object]:
    out = {}
    for f, feats in _fields(payload):
        if f != 1:
            continue
        for f2, entry in _fields(feats):
            key, val = None, None
            for f3, x in _fields(entry):
                if f3 == 1:
                    key = x.decode()
                elif f3 == 2:
                    for f4, lst in _fields(x):
                        if f4 == 1:
                            val = [bytes(v) for _, v in _fields(lst)]
                        elif f4 == 2:
                            val = np.concatenate([np.frombuffer(v, "<f4") for _, v in _fields(lst)] or [np.zeros(0, "<f4")])
                        elif f4 == 3:
                            val = np.concatenate([_unpack_varints(v) for _, v in _fields(lst)] or [np.zeros(0, np.int64)])
            out[key] = val
    return out


# ------------------------------------------------------------------ TFDS metadata
_FEATURES_PKG = "tensorflow_datasets.core.features."


def _tensor_feature(shape, dtype: str) -> dict:
    """features.json node of tfds.features.Tensor(shape, dtype) / tfds.features.Scalar(dtype) (shape ())."""
    cls = "scalar.Scalar" if len(shape) == 0 else "tensor_feature.Tensor"
    sh = {"dimensions": [str(int(d)) for d in shape]} if len(shape) else {}
    return {"pythonClassName": _FEATURES_PKG + cls, "tensor": {"shape": sh, "dtype": dtype, "encoding": "none"}}


def _dict_feature(children: dict) -> dict:
    return {"pythonClassName": _FEATURES_PKG + "features_dict.FeaturesDict", "featuresDict": {"features": children}}


def rlds_features(height: int, width: int) -> dict:
    """features.json of ``rlds_base.build_info(ds_config)`` for the reference's ds_config
    (transporter_network_data_generation.py:56-86): FeaturesDict{steps: Dataset(FeaturesDict{observation, action,
    reward, discount, is_first, is_last, is_terminal}), **episode_metadata_info}."""
    h, w = int(height), int(width)
    step = _dict_feature({
        "observation": _dict_feature({"overhead_camera/rgb": _tensor_feature((h, w, 3), "uint8"),
                                      "overhead_camera/depth": _tensor_feature((h, w), "float32")}),
        "action": _dict_feature({"pose": _tensor_feature((7,), "float64"),
                                 "pixel_coords": _tensor_feature((2,), "int32"),
                                 "gripper_rot": _tensor_feature((), "float64")}),
        "reward": _tensor_feature((), "float64"), "discount": _tensor_feature((), "float64"),
        "is_first": _tensor_feature((), "bool"), "is_last": _tensor_feature((), "bool"),
        "is_terminal": _tensor_feature((), "bool")})
    return _dict_feature({
        "steps": {"pythonClassName": _FEATURES_PKG + "dataset_feature.Dataset",
                  "sequence": {"feature": step, "length": "-1"}},
        "intrinsics": _dict_feature({k: _tensor_feature((), "float64") for k in ("fx", "fy", "cx", "cy")}),
        "extrinsics": _dict_feature({k: _tensor_feature((), "float64") for k in ("x", "y", "z", "qx", "qy", "qz", "qw")})})


def feature_leaves(node: dict, prefix: str = "", in_sequence: bool = False):
    """(example key, shape tuple, dtype, inside a Dataset / Sequence) of every tensor leaf of a features.json tree."""
    if "featuresDict" in node:
        for k, v in node["featuresDict"]["features"].items():
            yield from feature_leaves(v, f"{prefix}/{k}" if prefix else k, in_sequence)
    elif "sequence" in node:
        yield from feature_leaves(node["sequence"]["feature"], prefix, True)
    elif "tensor" in node:
        dims = node["tensor"].get("shape", {}).get("dimensions", [])
        yield prefix, tuple(int(d) for d in dims), node["tensor"]["dtype"], in_sequence
    else:
        raise ValueError(f"features.json: unsupported feature at '{prefix}': {sorted(node)}")


# ------------------------------------------------------------------ the writer
class EpisodeWriter:
    """Shard writer mirroring ``TFDSBackendWriter(data_directory, split_name, max_episodes_per_file,
    ds_config)`` (transporter_network_data_generation.py:103-111)."""

    VERSION = "0.0.1"   # TFDSBackendWriter's default `version`

    def __init__(self, data_directory: str, name: str, height: int, width: int, split_name: str = "train",
                 max_episodes_per_file: int = 10):
        self.dir, self.name, self.split = data_directory, name, split_name
        self.h, self.w = int(height), int(width)
        self.max_per_file = int(max_episodes_per_file)
        os.makedirs(self.dir, exist_ok=True)
        self._shards: List[int] = []
        self._file = None
        self._in_file = 0
        self._episodes = 0
        self._bytes = 0

    def features(self) -> dict:
        return rlds_features(self.h, self.w)

    def _tmp_path(self, k: int) -> str:
        return os.path.join(self.dir, f"{self.name}-{self.split}.tfrecord-{k:05d}.tmp")

    def _roll(self):
        if self._file is not None:
            self._file.close()
            self._shards.append(self._in_file)
        self._file = open(self._tmp_path(len(self._shards)), "wb")
        self._in_file = 0

    def write_episode(self, steps: List[dict], metadata: dict) -> None:
        """steps: RLDS steps, each {"observation": {...}, "action": {...} or None (last step), "reward",
        "discount", "is_first", "is_last", "is_terminal"}."""
        if self._file is None or self._in_file >= self.max_per_file:
            self._roll()
        T = len(steps)
        zero_act = {"pose": np.zeros(7), "pixel_coords": np.zeros(2, np.int64), "gripper_rot": 0.0}
        acts = [s.get("action") or zero_act for s in steps]
        rgb, depth = [], []
        for s in steps:
            o = s["observation"]
            r = np.asarray(o["overhead_camera/rgb"], np.uint8)
            d = np.asarray(o["overhead_camera/depth"], np.float32)
            assert r.shape == (self.h, self.w, 3) and d.shape == (self.h, self.w), (r.shape, d.shape)
            rgb.append(r.reshape(-1))
            depth.append(d.reshape(-1))
        feats = {
            "steps/observation/overhead_camera/rgb": np.concatenate(rgb) if rgb else np.zeros(0, np.uint8),
            "steps/observation/overhead_camera/depth": np.concatenate(depth) if depth else np.zeros(0, np.float32),
            "steps/action/pose": np.concatenate([np.asarray(a["pose"], np.float64).reshape(7) for a in acts]),
            "steps/action/pixel_coords": np.concatenate([np.asarray(a["pixel_coords"], np.int64).reshape(2) for a in acts]),
            "steps/action/gripper_rot": np.asarray([float(a["gripper_rot"]) for a in acts]),
            "steps/reward": np.asarray([float(s.get("reward", 0.0)) for s in steps]),
            "steps/discount": np.asarray([float(s.get("discount", 0.0)) for s in steps]),
            "steps/is_first": np.asarray([int(bool(s.get("is_first", k == 0))) for k, s in enumerate(steps)], np.int64),
            "steps/is_last": np.asarray([int(bool(s.get("is_last", k == T - 1))) for k, s in enumerate(steps)], np.int64),
            "steps/is_terminal": np.asarray([int(bool(s.get("is_terminal", False))) for s in steps], np.int64),
        }
        for grp in ("intrinsics", "extrinsics"):       # episode_metadata_info entries: top-level features
            for k, v in metadata[grp].items():
                feats[f"{grp}/{k}"] = np.asarray([float(v)])
        payload = encode_example(feats)
        write_record(self._file, payload)
        self._bytes += len(payload)
        self._in_file += 1
        self._episodes += 1

    def dataset_info(self) -> dict:
        """dataset_info.json: proto3 JSON of tfds' DatasetInfo message (int64 fields are strings)."""
        return {"name": self.name, "version": self.VERSION, "fileFormat": "tfrecord", "moduleName": "",
                "description": "RLDS episodes of RearrangementEnv (overhead camera, scripted pick / place actions)",
                "splits": [{"name": self.split, "shardLengths": [str(n) for n in self._shards],
                            "numBytes": str(self._bytes),
                            "filepathTemplate": "{DATASET}-{SPLIT}.{FILEFORMAT}-{SHARD_X_OF_Y}"}]}

    def close(self) -> dict:
        if self._file is not None:
            self._file.close()
            self._shards.append(self._in_file)
            self._file = None
        n = len(self._shards)
        for k in range(n):
            os.replace(self._tmp_path(k), os.path.join(self.dir, f"{self.name}-{self.split}.tfrecord-{k:05d}-of-{n:05d}"))
        info = self.dataset_info()
        with open(os.path.join(self.dir, "dataset_info.json"), "w") as f:
            json.dump(info, f, indent=2)
        with open(os.path.join(self.dir, "features.json"), "w") as f:
            json.dump(self.features(), f, indent=4)
        return info


_NP_DTYPE = {"uint8": np.uint8, "int32": np.int32, "int64": np.int64, "bool": np.bool_, "float32": np.float32,
             "float64": np.float64}


def _nest(tree: dict, key: str, leaf_names, value):
    """Put `value` at the path of `key` in a nested dict; `leaf_names` are the feature names on the way (a name may
    contain '/' itself, e.g. overhead_camera/rgb)."""
    node = tree
    for name in leaf_names[:-1]:
        node = node.setdefault(name, {})
    node[leaf_names[-1]] = value


def _leaf_paths(node: dict, path=()):
    if "featuresDict" in node:
        for k, v in node["featuresDict"]["features"].items():
            yield from _leaf_paths(v, path + (k,))
    elif "sequence" in node:
        yield from _leaf_paths(node["sequence"]["feature"], path)
    else:
        yield path


def read_episodes(data_directory: str, name: Optional[str] = None, split_name: str = "train") -> Iterator[dict]:
    """Generic reader of a TFDS directory of this kind: dataset_info.json names the dataset, its split and shard
    count, features.json says how every Example key is typed and shaped.  Yields nested dicts shaped like the
    feature tree ({"steps": {...arrays with a leading step axis...}, "intrinsics": {...}, "extrinsics": {...}})."""
    with open(os.path.join(data_directory, "features.json")) as f:
        feat = json.load(f)
    with open(os.path.join(data_directory, "dataset_info.json")) as f:
        info = json.load(f)
    name = name or info["name"]
    split = next(sp for sp in info["splits"] if sp["name"] == split_name)
    n = len(split["shardLengths"])
    leaves = list(feature_leaves(feat))
    paths = list(_leaf_paths(feat))
    for k in range(n):
        p = os.path.join(data_directory, f"{name}-{split_name}.{info['fileFormat']}-{k:05d}-of-{n:05d}")
        count = 0
        for rec in read_records(p):
            e = decode_example(rec)
            out: dict = {}
            for (key, shape, dtype, seq), path in zip(leaves, paths):
                v = np.asarray(e[key])
                v = v.reshape((-1,) + shape) if seq else v.reshape(shape)
                _nest(out, key, path, v.astype(_NP_DTYPE[dtype]))
            count += 1
            yield out
        if count != int(split["shardLengths"][k]):
            raise ValueError(f"{p}: {count} records, dataset_info.json says {split['shardLengths'][k]}")


class BatchedEpisodeLogger:
    """The EnvLogger of the batched env: collects (observation, action) of every env step by step and
    writes one episode per env -- ``with BatchedEpisodeLogger(env, writer) as log: log.reset(ts);
    log.step(action, ts)``.  Observations may be CUDA tensors (env render=True) or numpy arrays."""

    def __init__(self, env, writer: EpisodeWriter, env_mask: Optional[np.ndarray] = None):
        self.env, self.writer = env, writer
        self.mask = np.ones(env.num_envs, bool) if env_mask is None else np.asarray(env_mask, bool)
        self._steps: List[List[dict]] = [[] for _ in range(env.num_envs)]
        self._meta = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.flush()

    @staticmethod
    def _np(x):
        return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)

    def _obs(self, ts, i):
        o = ts.observation
        return {"overhead_camera/rgb": self._np(o["overhead_camera/rgb"][i]),
                "overhead_camera/depth": self._np(o["overhead_camera/depth"][i])}

    def reset(self, ts) -> None:
        self._meta = self.env.get_camera_metadata()   # calibration_metadata on the FIRST step (:88-95)
        # an env whose reset() failed (PropPlacer found no pose: the reference's reset() raises and the loop drops the
        # episode, transporter_network_data_generation.py:137-139) logs nothing
        failed = getattr(self.env, "placement_failed", None)
        if failed is not None:
            self.mask = self.mask & ~np.asarray(failed, bool)
        for i in np.nonzero(self.mask)[0]:
            self._steps[i] = [{"observation": self._obs(ts, i), "action": None, "reward": 0.0, "discount": 0.0,
                               "is_first": True, "is_last": False, "is_terminal": False}]

    def step(self, action: dict, ts, active: Optional[np.ndarray] = None) -> None:
        """RLDS convention: the action is stored with the step it was taken FROM; the new timestep opens
        the next step."""
        act = self.mask if active is None else (self.mask & np.asarray(active, bool))
        pose, pix = np.asarray(action["pose"]), np.asarray(action["pixel_coords"])
        for i in np.nonzero(act)[0]:
            self._steps[i][-1]["action"] = {"pose": pose[i], "pixel_coords": pix[i],
                                            "gripper_rot": float(np.broadcast_to(action["gripper_rot"], (self.env.num_envs,))[i])}
            self._steps[i].append({"observation": self._obs(ts, i), "action": None, "reward": float(np.broadcast_to(ts.reward, (self.env.num_envs,))[i]),
                                   "discount": float(np.broadcast_to(ts.discount, (self.env.num_envs,))[i]),
                                   "is_first": False, "is_last": False, "is_terminal": False})

    def flush(self) -> None:
        for i in np.nonzero(self.mask)[0]:
            if self._steps[i]:
                self._steps[i][-1]["is_last"] = True
                self.writer.write_episode(self._steps[i], self._meta)
                self._steps[i] = []
