"""Golden vectors of the reference's Hydra configuration tree -> tests/golden/config_tree.json.

Runs in the BUILD container only (it reads /root/reference, which does not travel): composes
``/root/reference/mujoco_robot_environments/config`` the way the reference's own call sites do and writes the
resolved trees as data.  ``tests/test_config_golden.py`` then requires ``mujoco_robot_environments_amd.config
.compose`` -- which holds the tree as Python data and never reads a YAML file -- to return the same tree, leaf by
leaf, for every call site the repo mirrors.

hydra-core / omegaconf are not installed here, so this file carries its own reading of the compose API, written
independently of the package's composer (it works on the YAML files; the package works on dicts):
  * YAML 1.1 through PyYAML with omegaconf's float resolver (``5e-3`` is a float there, a string in plain PyYAML);
  * defaults lists, recursively: an entry ``group: option`` of a config that lives in group G loads
    ``G/group/option.yaml`` into the package ``G.group`` (dots for slashes); a config without ``_self_`` in its
    defaults list comes AFTER its defaults (Hydra >= 1.1, which ``initialize(version_base=None)`` selects);
  * overrides: ``group[/sub]=option`` replaces a defaults-list choice when the key names a config group (a
    directory of the tree), otherwise ``a.b.c=value`` sets an existing key and ``+a.b=value`` adds one; values are
    parsed as YAML scalars;
  * ``${a.b.c}`` interpolations (absolute paths) are resolved in the dumped tree.

    python tools/make_config_golden.py
"""
import json
import os
import re
import sys

import yaml

REF = "/root/reference/mujoco_robot_environments/config"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "config_tree.json")

# (name, reference call site, config_name, overrides)
CALL_SITES = [
    ("rearrangement_default", "tasks/rearrangement.py:34-40 (also tasks/base.py:33-39, tasks/push.py:31-41)", "rearrangement",
     ["arena/props=colour_splitter", "simulation_tuning_mode=False"]),
    ("apple", "tasks/rearrangement.py:42-49", "rearrangement",
     ["arena/props=apple", "task=apple", "simulation_tuning_mode=False"]),
    ("colour_separator", "transporter_network_data_generation.py:26-33", "rearrangement",
     ["+name=colour_splitter", "task=rearrangement_w_targets", "arena/props=colour_splitter"]),
    ("controller_tuning", "automated_controller_tuning/rearrangement_controller_tuning.py:134-138", "rearrangement",
     ["arena/props=single_block", "physics_dt=0.001", "robots.arm.controller_config.controller_params.control_dt=0.005"]),
    ("lasa_default", "tasks/lasa_draw.py:31-41", "lasa", ["simulation_tuning_mode=True"]),
    ("lasa_deployment", "tasks/lasa_draw.py:45-56", "lasa",
     ["simulation_tuning_mode=False", "robots/arm/actuator_config=position"]),
]


class _Loader(yaml.SafeLoader):
    pass


_Loader.add_implicit_resolver(   # omegaconf.get_yaml_loader's float pattern
    "tag:yaml.org,2002:float",
    re.compile(r"""^(?:
     [-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
    |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
    |\.[0-9_]+(?:[eE][-+][0-9]+)?
    |[-+]?[0-9][0-9_]*(?::[0-5]?[0-9])+\.[0-9_]*
    |[-+]?\.(?:inf|Inf|INF)
    |\.(?:nan|NaN|NAN))$""", re.X),
    list("-+0123456789."))


def _load(rel):
    path = os.path.join(REF, rel + ".yaml")
    if not os.path.exists(path):
        raise FileNotFoundError(f"missing config {rel}.yaml (Hydra: MissingConfigException)")
    with open(path) as f:
        return yaml.load(f, Loader=_Loader)


def _merge(dst, src):
    """OmegaConf.merge for plain containers: dicts merge key by key, anything else is replaced."""
    if isinstance(dst, dict) and isinstance(src, dict):
        for k, v in src.items():
            dst[k] = _merge(dst[k], v) if k in dst else v
        return dst
    return src


def _place(tree, package, value):
    if not package:
        return _merge(tree, value)
    keys = package.split(".")
    node = tree
    for k in keys[:-1]:
        node = node.setdefault(k, {})
    node[keys[-1]] = _merge(node[keys[-1]], value) if keys[-1] in node else value
    return tree


def _expand(rel, group, choices, tree):
    """Load config `rel` (a path below the config root) that lives in config group `group`, its defaults first."""
    content = _load(rel)
    defaults = []
    if isinstance(content, dict) and "defaults" in content:
        defaults = content.pop("defaults") or []
    entries = list(defaults)
    if "_self_" not in entries:
        entries.append("_self_")
    for ent in entries:
        if ent == "_self_":
            _place(tree, group.replace("/", "."), content)
            continue
        (g, opt), = ent.items()
        sub = f"{group}/{g}" if group else g
        opt = choices.pop(sub, opt)
        _expand(f"{sub}/{opt}", sub, choices, tree)


def _scalar(text):
    return yaml.load(text, Loader=_Loader)


def _resolve(node, root):
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = re.fullmatch(r"\$\{([^}]+)\}", node)
        if m:
            cur = root
            for k in m.group(1).split("."):
                cur = cur[k]
            return _resolve(cur, root)
    return node


def compose(config_name, overrides):
    choices, assigns = {}, []
    for ov in overrides:
        key, _, val = ov.partition("=")
        add = key.startswith("+")
        key = key.lstrip("+")
        if not add and os.path.isdir(os.path.join(REF, key)):
            choices[key] = val
        else:
            assigns.append((key, _scalar(val), add))
    tree = {}
    _expand(config_name, "", choices, tree)
    if choices:
        raise KeyError(f"overrides matched no defaults-list entry: {sorted(choices)}")
    for key, val, add in assigns:
        node = tree
        keys = key.split(".")
        for k in keys[:-1]:
            node = node[k]
        if not add and keys[-1] not in node:
            raise KeyError(f"could not override '{key}' (not in the config; '+{key}=' would add it)")
        node[keys[-1]] = val
    return _resolve(tree, tree)


def main():
    if not os.path.isdir(REF):
        sys.exit(f"{REF} is not present: the golden file is generated in the build container only")
    out = {}
    for name, site, cfgname, ovs in CALL_SITES:
        out[name] = {"call_site": site, "config_name": cfgname, "overrides": ovs, "tree": compose(cfgname, ovs)}
    # the reference's own trap (config/arena/rearrangement_table.yaml:3): no `arena/props` choice -> no such file
    try:
        compose("rearrangement", [])
        out["_no_props_override"] = "composes"
    except FileNotFoundError:
        out["_no_props_override"] = "MissingConfigException"
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print(f"wrote {os.path.normpath(OUT)}: {', '.join(k for k in out if not k.startswith('_'))}")


if __name__ == "__main__":
    main()
