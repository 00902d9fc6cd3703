"""Hydra-shaped configuration tree of the reference, without Hydra.

The reference composes ``config/rearrangement.yaml`` with nested defaults lists and
``group=option`` overrides at import time (tasks/rearrangement.py:31-49,
transporter_network_data_generation.py:22-33).  hydra / omegaconf are not available on
the target image, so the same tree (same key names and values, cited per group) is held
here as plain data and ``compose()`` implements the subset of the compose API the
reference's callers use: defaults, ``group=option`` / ``group/sub=option`` selection,
``key=value`` and ``+key=value`` assignment, ``${a.b}`` interpolation.  It is re-entrant
(no global state).
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


# ------------------------------------------------------------------ option groups
# config/arena/props/*.yaml
_PROPS = {
    "colour_splitter": dict(min_objects=2, max_objects=5, min_object_size=0.015, max_object_size=0.016,
                            sample_size=True, sample_colour=True, color_noise=0.1, shapes=["cube"],
                            colours=["green", "blue", "red", "yellow", "cyan", "magenta"],
                            textures=["plain", "stary", "dotted"]),
    "single_block": dict(min_objects=1, max_objects=1, min_object_size=0.015, max_object_size=0.016,
                         sample_size=True, sample_colour=True, color_noise=0.1, shapes=["cube"],
                         colours=["red"], textures=["plain"]),
}
# config/arena/cameras/transporter_data_collection.yaml
_CAMERAS = {
    "transporter_data_collection": [dict(name="overhead_camera", pos=[0.7, 0.0, 1.3],
                                         quat=[0.707, 0.0, 0.0, -0.707], height=480, width=640, fovy=61)],
}
# config/arena/cameras/lasa.yaml
_CAMERAS["lasa"] = [
    dict(name="main_camera", pos=[0.363, -0.919, 1.663], quat=[0.94215352, 0.3349722, -0.00401909, -0.01114498],
         height=640, width=640, fovy=61),
    dict(name="front_camera", pos=[2.5, 0.0, 1.4], quat=[0.6133964, 0.3514872, 0.3512074, 0.6138851],
         height=640, width=640, fovy=61),
    dict(name="left_camera", pos=[0.4, 1.6, 1.4], quat=[0, 0, 0.5, 0.8660254], height=640, width=640, fovy=61),
]
_WORKSPACE = dict(seed=1, workspace=dict(min_pose=[0.35, -0.4, 0.43], max_pose=[0.55, 0.4, 0.435]),
                  gripper=dict(min_pose=[0.4, -0.0, 0.7, 3.14, 0.0, 0.0], max_pose=[0.5, 0.0, 0.7, 3.14, 0.0, 0.0]))


def _target(name, loc, rgba):
    return dict(name=name, location=loc, size=[0.075, 0.15, 0.01], rgba=rgba)


# config/task/*.yaml
_TASK = {
    "rearrangement": dict(initializers=_WORKSPACE),
    "rearrangement_w_targets": dict(
        initializers=_WORKSPACE,
        target_locations=dict(
            top_left=_target("top_left", [0.55, 0.4, 0.4], [1.0, 0.0, 0.0, 0.1]),
            bottom_left=_target("bottom_left", [0.35, 0.4, 0.4], [0.0, 1.0, 0.0, 0.1]),
            top_right=_target("top_right", [0.55, -0.4, 0.4], [0.0, 0.0, 1.0, 0.1]),
            bottom_right=_target("bottom_right", [0.35, -0.4, 0.4], [1.0, 1.0, 0.0, 0.1]),
            top_middle=_target("top_middle", [0.55, 0.0, 0.4], [1.0, 0.0, 1.0, 0.1]),
            bottom_middle=_target("bottom_middle", [0.35, 0.0, 0.4], [0.0, 1.0, 1.0, 0.1])),
        colour_target_map=dict(red="top_left", green="bottom_left", blue="top_right",
                               yellow="bottom_right", magenta="top_middle", cyan="bottom_middle")),
}
# config/robots/arm/*.yaml
_ARM = dict(
    actuator_config=dict(type="motor", high_torque_motor=dict(ctrlrange="-87 87"),
                         low_torque_motor=dict(ctrlrange="-12 12"),
                         joint_actuator_mapping=dict(joint1="high_torque_motor", joint2="high_torque_motor",
                                                     joint3="high_torque_motor", joint4="high_torque_motor",
                                                     joint5="low_torque_motor", joint6="low_torque_motor",
                                                     joint7="low_torque_motor")),
    controller_config=dict(
        controller_params=dict(
            name="osc", physics_dt="${physics_dt}", control_dt="${control_dt}",
            gains=dict(position=dict(kp=350.0, kd=20.0), orientation=dict(kp=500.0, kd=100.0),
                       nullspace=dict(kp=200.0, kd=30.0)),
            nullspace=dict(joint_config=[0, -0.785, 0, -2.356, 0, 1.571, 0.785]),
            convergence=dict(position_threshold=5e-3, orientation_threshold=68e-3)),
        controller=dict(_target_="mujoco_controllers.osc.OSC", _partial_=True,
                        controller_config="${robots.arm.controller_config.controller_params}")),
    default_configurations=dict(home=[0, -0.785, 0, -2.356, 0, 1.571, 0.785]),
    arm=dict(_target_="mujoco_robot_environments.models.arms.franka_emika.FER"),
)
# config/robots/arm/actuator_config/position.yaml (`general` actuators: kp, kv per joint; LasaDrawEnv's
# deployment config, tasks/lasa_draw.py:45-56)
def _position_joint(ctrlrange="-2.8973 2.8973", forcerange="-87 87", kp=4500, kv=450):
    return dict(dyntype="none", biastype="affine", ctrlrange=ctrlrange, forcerange=forcerange,
                gainprm=str(kp), biasprm=f"0 -{kp} -{kv}")


_ACTUATORS = {
    "motor": _ARM["actuator_config"],
    "position": dict(
        type="general", default=_position_joint(),
        joint1=_position_joint(), joint2=_position_joint(ctrlrange="-1.7628 1.7628"),
        joint3=_position_joint(kp=3500, kv=350), joint4=_position_joint(ctrlrange="-3.0718 -0.0698", kp=3500, kv=350),
        joint5=_position_joint(forcerange="-12 12", kp=2000, kv=200),
        joint6=_position_joint(ctrlrange="-0.0175 3.7525", forcerange="-12 12", kp=2000, kv=200),
        joint7=_position_joint(forcerange="-12 12", kp=2000, kv=200),
        joint_actuator_mapping={f"joint{i}": f"joint{i}" for i in range(1, 8)}),
}
# config/robots/end_effector/*.yaml
_EEF = dict(controller_config=dict(controller=dict(_target_="mujoco_controllers.min_max.MinMax",
                                                   min_val=0.0, max_val=255.0)),
            end_effector=dict(_target_="mujoco_robot_environments.models.end_effectors.robotiq_2f85.Robotiq2F85"))

# config/rearrangement.yaml (wandb block is unused by the env and omitted)
_ROOT = dict(physics_dt=0.001, control_dt=0.005, gravity=[0.0, 0.0, -9.8], nconmax=1000, njmax=2000,
             offheight=640, offwidth=640, znear=0.0005, viewer=False, simulation_tuning_mode=False,
             dataset=dict(num_episodes=1000, max_steps=10, max_episodes_per_file=10))
_DEFAULTS = {"arena/cameras": "transporter_data_collection", "arena/props": None, "task": "rearrangement",
             "robots/arm/actuator_config": "motor"}
# config/lasa.yaml: physics_dt 0.01 (> control_dt), arena: lasa (cameras: lasa, no props group)
_ROOT_LASA = dict(_ROOT, physics_dt=0.01)
_DEFAULTS_LASA = {"arena/cameras": "lasa", "arena/props": "single_block", "task": "rearrangement",
                  "robots/arm/actuator_config": "motor"}


def _set_path(tree: dict, path: str, value: Any, must_exist: bool):
    keys = path.split(".")
    node = tree
    for k in keys[:-1]:
        node = node.setdefault(k, {})
    if must_exist and keys[-1] not in node:
        raise KeyError(f"could not override '{path}': key not in config (use +{path}= to add)")
    node[keys[-1]] = value


def _parse_value(v: str):
    lv = v.strip()
    if lv.lower() in ("true", "false"):
        return lv.lower() == "true"
    if lv.lower() in ("null", "none"):
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
        for k, v in list(node.items()):
            node[k] = _interpolate(v, root)
        return node
    if isinstance(node, list):
        return [_interpolate(v, root) for v in node]
    if isinstance(node, str) and node.startswith("${") and node.endswith("}"):
        cur = root
        for k in node[2:-1].split("."):
            cur = cur[k]
        return _interpolate(copy.deepcopy(cur), root)
    return node


def compose(config_name: str = "rearrangement", overrides: Optional[List[str]] = None) -> Cfg:
    if config_name not in ("rearrangement", "lasa"):
        raise ValueError(f"unknown config '{config_name}' (mirrored: rearrangement, lasa)")
    groups = dict(_DEFAULTS if config_name == "rearrangement" else _DEFAULTS_LASA)
    assigns = []
    for ov in overrides or []:
        key, _, val = ov.partition("=")
        add = key.startswith("+")
        key = key.lstrip("+")
        if key in groups and not add:
            groups[key] = val
        else:
            assigns.append((key, _parse_value(val), add))
    if groups["arena/props"] is None:
        # config/arena/rearrangement_table.yaml:3 points at a non-existent `props: default`
        raise ValueError("arena/props has no default in the reference tree: pass 'arena/props=<option>'")
    tree: Dict[str, Any] = copy.deepcopy(_ROOT if config_name == "rearrangement" else _ROOT_LASA)
    tree["arena"] = dict(cameras=copy.deepcopy(_CAMERAS[groups["arena/cameras"]]),
                         props=copy.deepcopy(_PROPS[groups["arena/props"]]))
    tree["robots"] = dict(arm=copy.deepcopy(_ARM), end_effector=copy.deepcopy(_EEF))
    tree["robots"]["arm"]["actuator_config"] = copy.deepcopy(_ACTUATORS[groups["robots/arm/actuator_config"]])
    tree["task"] = copy.deepcopy(_TASK[groups["task"]])
    for key, val, add in assigns:
        _set_path(tree, key, val, must_exist=not add)
    return _wrap(_interpolate(tree, tree))


def default_config() -> Cfg:
    """tasks/rearrangement.py:34-40 DEFAULT_CONFIG."""
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
