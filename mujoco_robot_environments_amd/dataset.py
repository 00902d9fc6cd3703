"""RLDS-style episode shards from batched rollouts -- the on-disk side of the reference's data
generation (transporter_network_data_generation.py:56-111: ``tfds.rlds.rlds_base.DatasetConfig`` +
``envlogger.EnvLogger`` with a ``TFDSBackendWriter``), without envlogger / TensorFlow.

What is written: TFRecord files (``<name>-train.tfrecord-00000-of-NNNNN``) whose records are
``tf.train.Example`` protos, one EPISODE per record, with the RLDS step fields flattened under
``steps/`` exactly as the reference's ``ds_config`` names them:

    steps/observation/overhead_camera/rgb     bytes_list: one raw uint8 [H, W, 3] buffer per step
    steps/observation/overhead_camera/depth   float_list: H * W floats per step, steps concatenated
    steps/action/pose                         float_list: 7 per step (float64 in the reference; the
                                              Example proto only has float32 lists -- TFDS stores
                                              float64 tensors the same way)
    steps/action/pixel_coords                 int64_list: 2 per step
    steps/action/gripper_rot                  float_list: 1 per step
    steps/reward, steps/discount              float_list: 1 per step
    steps/is_first, steps/is_last, steps/is_terminal   int64_list: 1 per step
    episode_metadata/intrinsics/{fx,fy,cx,cy}, episode_metadata/extrinsics/{x,y,z,qx,qy,qz,qw}
                                              float_list: 1 each (calibration_metadata, :88-95)

plus ``features.json`` (the feature spec above with shapes / dtypes) and ``dataset_info.json`` (name,
split, shard lengths).  ``max_episodes_per_file`` (config/dataset/default.yaml:3) cuts the shards.
The TFRecord framing (length, masked CRC-32C of the length, payload, masked CRC-32C of the payload)
and the proto wire format are restated from their public specifications; TensorFlow is not present
here, so byte-compatibility with ``tfds.builder_from_directory`` is untested -- ``read_episodes`` in
this module parses the files back and tests/test_dataset.py round-trips them.
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


def _feature(value) -> bytes:
    if isinstance(value, (list, tuple)) and value and isinstance(value[0], (bytes, bytearray)):
        body = b"".join(_ld(1, bytes(v)) for v in value)
        return _ld(1, body)                                   # bytes_list
    a = np.asarray(value)
    if a.dtype.kind == "f":
        return _ld(2, _ld(1, a.astype("<f4").tobytes()))      # float_list, packed
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
                            vals = []
                            for _, v in _fields(lst):
                                i = 0
                                while i < len(v):
                                    n, i = _read_varint(v, i)
                                    vals.append(n - (1 << 64) if n >= (1 << 63) else n)
                            val = np.asarray(vals, np.int64)
            out[key] = val
    return out


# ------------------------------------------------------------------ the writer
class EpisodeWriter:
    """Shard writer mirroring ``TFDSBackendWriter(data_directory, split_name, max_episodes_per_file,
    ds_config)`` (transporter_network_data_generation.py:103-111)."""

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

    def features(self) -> dict:
        h, w = self.h, self.w
        return {
            "steps": {
                "observation": {"overhead_camera/rgb": {"shape": [h, w, 3], "dtype": "uint8"},
                                "overhead_camera/depth": {"shape": [h, w], "dtype": "float32"}},
                "action": {"pose": {"shape": [7], "dtype": "float64"},
                           "pixel_coords": {"shape": [2], "dtype": "int32"},
                           "gripper_rot": {"shape": [], "dtype": "float64"}},
                "reward": {"shape": [], "dtype": "float64"}, "discount": {"shape": [], "dtype": "float64"},
                "is_first": {"shape": [], "dtype": "bool"}, "is_last": {"shape": [], "dtype": "bool"},
                "is_terminal": {"shape": [], "dtype": "bool"}},
            "episode_metadata": {"intrinsics": {k: {"shape": [], "dtype": "float64"} for k in ("fx", "fy", "cx", "cy")},
                                 "extrinsics": {k: {"shape": [], "dtype": "float64"}
                                                for k in ("x", "y", "z", "qx", "qy", "qz", "qw")}}}

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
            rgb.append(r.tobytes())
            depth.append(d.reshape(-1))
        feats = {
            "steps/observation/overhead_camera/rgb": rgb,
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
        for grp in ("intrinsics", "extrinsics"):
            for k, v in metadata[grp].items():
                feats[f"episode_metadata/{grp}/{k}"] = np.asarray([float(v)])
        write_record(self._file, encode_example(feats))
        self._in_file += 1
        self._episodes += 1

    def close(self) -> dict:
        if self._file is not None:
            self._file.close()
            self._shards.append(self._in_file)
            self._file = None
        n = len(self._shards)
        for k in range(n):
            os.replace(self._tmp_path(k), os.path.join(self.dir, f"{self.name}-{self.split}.tfrecord-{k:05d}-of-{n:05d}"))
        info = {"name": self.name, "splits": [{"name": self.split, "shard_lengths": self._shards,
                                               "num_examples": self._episodes}],
                "file_format": "tfrecord", "record": "tf.train.Example, one episode per record, RLDS step fields under steps/"}
        with open(os.path.join(self.dir, "dataset_info.json"), "w") as f:
            json.dump(info, f, indent=1)
        with open(os.path.join(self.dir, "features.json"), "w") as f:
            json.dump(self.features(), f, indent=1)
        return info


def read_episodes(data_directory: str, name: str, split_name: str = "train") -> Iterator[dict]:
    """Parses the shards back into {"steps": {...arrays with a leading step axis...}, "episode_metadata": {...}}."""
    with open(os.path.join(data_directory, "features.json")) as f:
        feat = json.load(f)
    h, w, _ = feat["steps"]["observation"]["overhead_camera/rgb"]["shape"]
    files = sorted(p for p in os.listdir(data_directory) if p.startswith(f"{name}-{split_name}.tfrecord-"))
    for p in files:
        for rec in read_records(os.path.join(data_directory, p)):
            e = decode_example(rec)
            T = len(e["steps/reward"])
            rgb = np.stack([np.frombuffer(b, np.uint8).reshape(h, w, 3) for b in e["steps/observation/overhead_camera/rgb"]])
            steps = {"observation": {"overhead_camera/rgb": rgb,
                                     "overhead_camera/depth": e["steps/observation/overhead_camera/depth"].reshape(T, h, w)},
                     "action": {"pose": e["steps/action/pose"].reshape(T, 7),
                                "pixel_coords": e["steps/action/pixel_coords"].reshape(T, 2),
                                "gripper_rot": e["steps/action/gripper_rot"]},
                     "reward": e["steps/reward"], "discount": e["steps/discount"],
                     "is_first": e["steps/is_first"].astype(bool), "is_last": e["steps/is_last"].astype(bool),
                     "is_terminal": e["steps/is_terminal"].astype(bool)}
            meta = {g: {k.split("/")[-1]: float(v[0]) for k, v in e.items() if k.startswith(f"episode_metadata/{g}/")}
                    for g in ("intrinsics", "extrinsics")}
            yield {"steps": steps, "episode_metadata": meta}


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
