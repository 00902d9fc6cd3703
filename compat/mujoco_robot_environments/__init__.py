"""Import-path shim: ``PYTHONPATH=compat`` makes the reference's module paths resolve to this
repo's MI355X implementation, so that a caller written against the reference --

    from mujoco_robot_environments.tasks.rearrangement import RearrangementEnv

(transporter_network_data_generation.py:19) -- runs unchanged on the batched HIP step.  The task
modules (rearrangement, base, push, lasa_draw), the robot arm and the config tree exist; everything else of
the reference (MJX prototypes, ROS export, HF upload) raises ImportError as it would without its
optional dependencies.  Not on sys.path unless asked for: it would shadow a real installation."""
import importlib
import sys

_MAP = {
    "tasks": "mujoco_robot_environments_amd.tasks",
    "tasks.rearrangement": "mujoco_robot_environments_amd.tasks.rearrangement",
    "tasks.base": "mujoco_robot_environments_amd.tasks.base",
    "tasks.push": "mujoco_robot_environments_amd.tasks.push",
    "tasks.lasa_draw": "mujoco_robot_environments_amd.tasks.lasa_draw",
    "models": "mujoco_robot_environments_amd.models",
    "models.robot_arm": "mujoco_robot_environments_amd.models.robot_arm",
    "config": "mujoco_robot_environments_amd.config",
}
for _name, _target in _MAP.items():
    sys.modules[f"{__name__}.{_name}"] = importlib.import_module(_target)
tasks = sys.modules[f"{__name__}.tasks"]
models = sys.modules[f"{__name__}.models"]
config = sys.modules[f"{__name__}.config"]
