"""Hydra-shaped configuration tree of the reference, without Hydra.

The reference composes ``config/rearrangement.yaml`` / ``config/lasa.yaml`` with nested defaults lists and
``group=option`` overrides at import time (tasks/rearrangement.py:31-49, tasks/lasa_draw.py:31-56,
transporter_network_data_generation.py:22-33).  hydra / omegaconf are not available on the target image, so the
tree is held here as plain data -- one entry of ``_TREE`` per config file of the reference, keyed by its path
below ``config/`` and citing it -- and ``compose()`` implements the part of the compose API the reference's
callers use:

  * defaults lists, recursively (an entry ``group: option`` of a config in group G selects ``G/group/option`` and
    packages it at ``G.group``); a config comes after its defaults (Hydra >= 1.1: no ``_self_`` = ``_self_`` last);
  * ``group[/sub]=option`` selection, ``a.b.c=value`` assignment of an existing key, ``+a.b=value`` addition;
  * ``${a.b}`` interpolation (resolved eagerly when the tree is returned).

It is re-entrant (no global state).  ``tests/test_config_golden.py`` pins it, leaf by
leaf, to the trees ``tools/make_config_golden.py`` composed from the reference's own YAML files for every call
site of the reference that the repo mirrors (tests/golden/config_tree.json).
"""
from __future__ import annotations

import copy
from typing import Any, Dict, List, Optional


class Cfg(dict):
    """dict with attribute access (the part of DictConfig the env uses)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


# ------------------------------------------------------------------ the tree, file by file
def _root(physics_dt, arena, madrona=None):
    """config/rearrangement.yaml:1-39 and config/lasa.yaml:1-42 (same keys; lasa adds `madrona`)."""
    d = dict(physics_dt=physics_dt, control_dt=0.005, gravity=[0.0, 0.0, -9.8], nconmax=1000, njmax=2000,
             offheight=640, offwidth=640, znear=0.0005, viewer=False)
    if madrona is not None:
        d["madrona"] = dict(use=madrona)
    d["wandb"] = dict(use=True, experiment_name="rearrangement_transporter", tags=["original transporter network"],
                      notes="Curently debugging.", track_gradients=False, resume=False,
                      resume_run=dict(id="xxxxx", load_epoch=1),
                      metrics=dict(pick_loss=dict(step_metric="epoch", summary="min")),
                      tables=dict(data_table=dict(columns=["placeholder"])))
    d["simulation_tuning_mode"] = False
    d["defaults"] = [{"arena": arena}, {"robots": "franka_robotiq_2f85"}, {"task": "rearrangement"},
                     {"dataset": "default"}]
    return d


def _props(lo, hi, min_size, max_size, shapes, colours):
    """config/arena/props/*.yaml"""
    return dict(min_objects=lo, max_objects=hi, min_object_size=min_size, max_object_size=max_size, sample_size=True,
                sample_colour=True, color_noise=0.1, shapes=shapes, colours=colours,
                textures=["plain", "stary", "dotted"])


def _camera(name, pos, quat):
    return dict(name=name, pos=pos, quat=quat, height=640, width=640, fovy=61)


_INITIALIZERS = dict(seed=1, workspace=dict(min_pose=[0.35, -0.4, 0.43], max_pose=[0.55, 0.4, 0.435]),
                     gripper=dict(min_pose=[0.4, -0.0, 0.7, 3.14, 0.0, 0.0], max_pose=[0.5, 0.0, 0.7, 3.14, 0.0, 0.0]))


def _target(name, loc, rgba):
    return dict(name=name, location=loc, size=[0.075, 0.15, 0.01], rgba=rgba)


_AC = "${robots.arm.actuator_config.default."


def _position_joint(ctrlrange=None, forcerange=None, kp=None, kv=None):
    """One joint of config/robots/arm/actuator_config/position.yaml: what is not given refers to `default`."""
    return dict(dyntype=_AC + "dyntype}", biastype=_AC + "biastype}",
                ctrlrange=ctrlrange if ctrlrange is not None else _AC + "ctrlrange}",
                forcerange=forcerange if forcerange is not None else _AC + "forcerange}",
                gainprm=str(kp) if kp is not None else _AC + "gainprm}",
                biasprm=f"0 -{kp} -{kv}" if kp is not None else _AC + "biasprm}")


def _sensors(kind, prefix):
    """One list entry of config/robots/arm/sensor_config/position.yaml."""
    d: Dict[str, Any] = dict(type=kind)
    for i in range(1, 8):
        d[f"{prefix}_joint{i}"] = dict(name=f"{prefix}_joint{i}", joint=f"joint{i}")
    d["joint_sensor_mapping"] = {f"{prefix}_joint{i}": f"joint{i}" for i in range(1, 8)}
    return d


_SIX = ["green", "blue", "red", "yellow", "cyan", "magenta"]

_TREE: Dict[str, Any] = {
    "rearrangement": _root(0.001, "rearrangement_table"),
    # config/lasa.yaml: physics_dt 0.01 (> control_dt); its arena has cameras and NO props group
    "lasa": _root(0.01, "lasa", madrona=False),
    # config/arena/rearrangement_table.yaml:1-3 -- `props: default` does not exist: every caller overrides it
    "arena/rearrangement_table": dict(defaults=[{"cameras": "transporter_data_collection"}, {"props": "default"}]),
    "arena/lasa": dict(defaults=[{"cameras": "lasa"}]),   # config/arena/lasa.yaml
    # config/arena/cameras/*.yaml
    "arena/cameras/transporter_data_collection": [dict(name="overhead_camera", pos=[0.7, 0.0, 1.3],
                                                       quat=[0.707, 0.0, 0.0, -0.707], height=480, width=640, fovy=61)],
    "arena/cameras/lasa": [_camera("main_camera", [0.363, -0.919, 1.663],
                                   [0.94215352, 0.3349722, -0.00401909, -0.01114498])],
    "arena/cameras/rearrangement": [_camera("front_camera", [2.5, 0.0, 1.4], [0.6133964, 0.3514872, 0.3512074, 0.6138851]),
                                    _camera("left_camera", [0.4, 1.6, 1.4], [0, 0, 0.5, 0.8660254])],
    # config/arena/props/{colour_splitter,single_block,apple}.yaml
    "arena/props/colour_splitter": _props(2, 5, 0.015, 0.016, ["cube"], _SIX),
    "arena/props/single_block": _props(1, 1, 0.02, 0.025, ["cube"],
                                       ["red", "green", "blue", "yellow", "magenta", "cyan", "grey"]),
    "arena/props/apple": _props(2, 3, 0.015, 0.016, ["apple"], _SIX),
    # config/robots/franka_robotiq_2f85.yaml
    "robots/franka_robotiq_2f85": dict(defaults=[{"arm": "franka_emika_panda"}, {"end_effector": "robotiq_2f85"}]),
    # config/robots/arm/franka_emika_panda.yaml
    "robots/arm/franka_emika_panda": dict(
        defaults=[{"actuator_config": "motor"}, {"sensor_config": "position"}, {"controller_config": "osc"}],
        default_configurations=dict(home=[0, -0.785, 0, -2.356, 0, 1.571, 0.785]),
        arm=dict(_target_="mujoco_robot_environments.models.arms.franka_emika.FER",
                 actuator_config="${robots.arm.actuator_config}", sensor_config="${robots.arm.sensor_config}",
                 controller_config="${robots.arm.controller_config}",
                 configuration_config="${robots.arm.default_configurations}")),
    # config/robots/arm/actuator_config/motor.yaml
    "robots/arm/actuator_config/motor": dict(
        type="motor", high_torque_motor=dict(ctrlrange="-87 87"), low_torque_motor=dict(ctrlrange="-12 12"),
        joint_actuator_mapping=dict(joint1="high_torque_motor", joint2="high_torque_motor", joint3="high_torque_motor",
                                    joint4="high_torque_motor", joint5="low_torque_motor", joint6="low_torque_motor",
                                    joint7="low_torque_motor")),
    # config/robots/arm/actuator_config/position.yaml (`general` actuators, kp / kv per joint: LasaDrawEnv's
    # deployment config, tasks/lasa_draw.py:45-56)
    "robots/arm/actuator_config/position": dict(
        type="general",
        default=dict(dyntype="none", biastype="affine", ctrlrange="-2.8973 2.8973", forcerange="-87 87", gainprm="4500",
                     biasprm="0 -4500 -450"),
        joint1=_position_joint(), joint2=_position_joint(ctrlrange="-1.7628 1.7628"),
        joint3=_position_joint(kp=3500, kv=350), joint4=_position_joint(ctrlrange="-3.0718 -0.0698", kp=3500, kv=350),
        joint5=_position_joint(forcerange="-12 12", kp=2000, kv=200),
        joint6=_position_joint(ctrlrange="-0.0175 3.7525", forcerange="-12 12", kp=2000, kv=200),
        joint7=_position_joint(forcerange="-12 12", kp=2000, kv=200),
        joint_actuator_mapping={f"joint{i}": f"joint{i}" for i in range(1, 8)}),
    # config/robots/arm/sensor_config/position.yaml (declared, never read: SURVEY.md a16)
    "robots/arm/sensor_config/position": [_sensors("jointpos", "pos"), _sensors("jointtorque", "torque")],
    # config/robots/arm/controller_config/osc.yaml
    "robots/arm/controller_config/osc": dict(
        controller_params=dict(
            name="osc", physics_dt="${physics_dt}", control_dt="${control_dt}",
            gains=dict(position=dict(kp=350.0, kd=20.0), orientation=dict(kp=500.0, kd=100.0),
                       nullspace=dict(kp=200.0, kd=30.0)),
            nullspace=dict(joint_config=[0, -0.785, 0, -2.356, 0, 1.571, 0.785]),
            convergence=dict(position_threshold=5e-3, orientation_threshold=68e-3)),
        controller=dict(_target_="mujoco_controllers.osc.OSC", _partial_=True,
                        controller_config="${robots.arm.controller_config.controller_params}")),
    # config/robots/end_effector/robotiq_2f85.yaml and its three groups
    "robots/end_effector/robotiq_2f85": dict(
        defaults=[{"actuator_config": "default"}, {"sensor_config": "default"}, {"controller_config": "min_max"}],
        end_effector=dict(_target_="mujoco_robot_environments.models.end_effectors.robotiq_2f85.Robotiq2F85",
                          actuator_config="${robots.end_effector.actuator_config}",
                          sensor_config="${robots.end_effector.sensor_config}",
                          controller_config="${robots.end_effector.controller_config}")),
    "robots/end_effector/actuator_config/default": dict(placeholder="placeholder"),
    "robots/end_effector/sensor_config/default": dict(placeholder="placeholder"),
    "robots/end_effector/controller_config/min_max": dict(
        controller=dict(_target_="mujoco_controllers.min_max.MinMax", min_val=0.0, max_val=255.0)),
    # config/task/{rearrangement,apple,rearrangement_w_targets}.yaml
    "task/rearrangement": dict(initializers=_INITIALIZERS),
    "task/apple": dict(initializers=_INITIALIZERS),
    "task/rearrangement_w_targets": dict(
        initializers=_INITIALIZERS,
        target_locations=dict(
            top_left=_target("top_left", [0.55, 0.4, 0.4], [1.0, 0.0, 0.0, 0.1]),
            bottom_left=_target("bottom_left", [0.35, 0.4, 0.4], [0.0, 1.0, 0.0, 0.1]),
            top_right=_target("top_right", [0.55, -0.4, 0.4], [0.0, 0.0, 1.0, 0.1]),
            bottom_right=_target("bottom_right", [0.35, -0.4, 0.4], [1.0, 1.0, 0.0, 0.1]),
            top_middle=_target("top_middle", [0.55, 0.0, 0.4], [1.0, 0.0, 1.0, 0.1]),
            bottom_middle=_target("bottom_middle", [0.35, 0.0, 0.4], [0.0, 1.0, 1.0, 0.1])),
        colour_target_map=dict(red="top_left", green="bottom_left", blue="top_right",
                               yellow="bottom_right", magenta="top_middle", cyan="bottom_middle")),
    # config/dataset/default.yaml
    "dataset/default": dict(num_episodes=1000, max_steps=10, max_episodes_per_file=10),
}


# ------------------------------------------------------------------ composition
class MissingConfig(KeyError):
    """Hydra's MissingConfigException: a defaults-list choice that names no config of the tree."""


def _merge(dst, src):
    if isinstance(dst, dict) and isinstance(src, dict):
        for k, v in src.items():
            dst[k] = _merge(dst[k], v) if k in dst else v
        return dst
    return src


def _place(tree: dict, package: List[str], value):
    if not package:
        return _merge(tree, value)
    node = tree
    for k in package[:-1]:
        node = node.setdefault(k, {})
    node[package[-1]] = _merge(node[package[-1]], value) if package[-1] in node else value
    return tree


def _expand(path: str, group: List[str], choices: Dict[str, str], tree: dict):
    if path not in _TREE:
        raise MissingConfig(f"no config '{path}' (the reference's tree has no such file; override the choice, e.g. "
                            f"'{'/'.join(group)}=<option>')")
    content = copy.deepcopy(_TREE[path])
    entries = content.pop("defaults", []) if isinstance(content, dict) else []
    if "_self_" not in entries:
        entries = list(entries) + ["_self_"]
    for ent in entries:
        if ent == "_self_":
            _place(tree, group, content)
            continue
        (g, opt), = ent.items()
        sub = group + [g]
        key = "/".join(sub)
        opt = choices.pop(key, opt)
        _expand(f"{key}/{opt}", sub, choices, tree)


def _is_group(key: str) -> bool:
    return any(p.startswith(key + "/") for p in _TREE)


def _parse_value(v: str):
    lv = v.strip()
    if lv.lower() in ("true", "false"):
        return lv.lower() == "true"
    if lv.lower() in ("null", "none", "~"):
        return None
    try:
        return int(lv)
    except ValueError:
        pass
    try:
        return float(lv)
    except ValueError:
        return lv


def _interpolate(node, root):
    if isinstance(node, dict):
        return {k: _interpolate(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_interpolate(v, root) for v in node]
    if isinstance(node, str) and node.startswith("${") and node.endswith("}"):
        cur = root
        for k in node[2:-1].split("."):
            cur = cur[k]
        return _interpolate(copy.deepcopy(cur), root)
    return node


def compose(config_name: str = "rearrangement", overrides: Optional[List[str]] = None) -> Cfg:
    if config_name not in _TREE or "/" in config_name:
        raise ValueError(f"unknown primary config '{config_name}' (mirrored: rearrangement, lasa)")
    choices: Dict[str, str] = {}
    assigns = []
    for ov in overrides or []:
        key, _, val = ov.partition("=")
        add = key.startswith("+")
        key = key.lstrip("+")
        if not add and _is_group(key):
            choices[key] = val
        else:
            assigns.append((key, _parse_value(val), add))
    tree: Dict[str, Any] = {}
    _expand(config_name, [], choices, tree)
    if choices:
        raise KeyError(f"could not override {sorted(choices)}: no such entry in a defaults list")
    for key, val, add in assigns:
        keys = key.split(".")
        node = tree
        for k in keys[:-1]:
            if add:
                node = node.setdefault(k, {})
            else:
                node = node[k]
        if not add and keys[-1] not in node:
            raise KeyError(f"could not override '{key}': key not in config (use +{key}= to add)")
        node[keys[-1]] = val
    return _wrap(_interpolate(tree, tree))


def default_config() -> Cfg:
    """tasks/rearrangement.py:34-40 DEFAULT_CONFIG (also tasks/base.py:33-39)."""
    return compose("rearrangement", ["arena/props=colour_splitter", "simulation_tuning_mode=False"])


def lasa_default_config() -> Cfg:
    """tasks/lasa_draw.py:31-41 generate_default_config."""
    return compose("lasa", ["simulation_tuning_mode=True"])


def lasa_deployment_config() -> Cfg:
    """tasks/lasa_draw.py:45-56 generate_deployment_config."""
    return compose("lasa", ["simulation_tuning_mode=False", "robots/arm/actuator_config=position"])


def push_default_config() -> Cfg:
    """tasks/push.py:31-41 generate_default_config."""
    return compose("rearrangement", ["arena/props=colour_splitter", "simulation_tuning_mode=False"])


def colour_separator_task_config() -> Cfg:
    """transporter_network_data_generation.py:26-33 COLOR_SEPERATOR_TASK_CONFIG."""
    return compose("rearrangement", ["+name=colour_splitter", "task=rearrangement_w_targets",
                                     "arena/props=colour_splitter"])


def controller_tuning_config() -> Cfg:
    """automated_controller_tuning/rearrangement_controller_tuning.py:134-138 TUNING_CONFIG."""
    return compose("rearrangement", ["arena/props=single_block", "physics_dt=0.001",
                                     "robots.arm.controller_config.controller_params.control_dt=0.005"])


# OSC gains found by the batched CMA-ES run of examples/controller_tuning.py (256 candidates x 16
# scenes, 5 generations, profiles/r01k_controller_tuning_256x16.log): fitness 2.6 against 523 for
# osc.yaml's gains on this repo's (unverified) arm model, whose joint torques osc.yaml's gains saturate.
TUNED_OSC_GAINS = {"position": (525.4, 215.2), "orientation": (766.6, 158.6), "nullspace": (22.6, 7.2)}


def apply_tuned_osc_gains(cfg: Cfg) -> Cfg:
    """Opt-in replacement of controller_params.gains by TUNED_OSC_GAINS (returns cfg)."""
    g = cfg.robots.arm.controller_config.controller_params.gains
    for k, (kp, kd) in TUNED_OSC_GAINS.items():
        g[k].kp, g[k].kd = kp, kd
    return cfg
