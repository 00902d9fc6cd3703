"""The Hydra tree, pinned to the reference: `config.compose` (Python data, no YAML, no Hydra) must return, leaf by
leaf, what `tools/make_config_golden.py` composed from /root/reference/mujoco_robot_environments/config/**/*.yaml
for every compose() call site of the reference that the repo mirrors (tests/golden/config_tree.json -- data
generated in the build container; the reference itself never travels)."""
import json
import os

import pytest

from mujoco_robot_environments_amd import config as C

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config_tree.json")))
SITES = [k for k in GOLD if not k.startswith("_")]


def _leaves(node, path=()):
    if isinstance(node, dict):
        if not node:
            yield path, {}
        for k, v in node.items():
            yield from _leaves(v, path + (k,))
    elif isinstance(node, list):
        yield path + ("#len",), len(node)
        for i, v in enumerate(node):
            yield from _leaves(v, path + (i,))
    else:
        yield path, node


@pytest.mark.parametrize("site", SITES)
def test_compose_equals_the_reference_tree_leaf_by_leaf(site):
    g = GOLD[site]
    got = dict(_leaves(C.compose(g["config_name"], g["overrides"])))
    want = dict(_leaves(g["tree"]))
    missing = sorted(set(want) - set(got), key=str)
    extra = sorted(set(got) - set(want), key=str)
    assert not missing, f"{site} ({g['call_site']}): keys of the reference tree absent here: {missing[:10]}"
    assert not extra, f"{site}: keys the reference tree does not have: {extra[:10]}"
    diff = [(k, got[k], want[k]) for k in want if got[k] != want[k] or type(got[k]) is not type(want[k])]
    assert not diff, f"{site}: {len(diff)} leaves differ, e.g. {diff[:5]}"
    assert len(want) > 150   # (the tree is not trivially small: ~300 leaves per call site)


def test_named_call_sites_are_the_golden_ones():
    """The helper functions the env modules use are exactly the reference's call sites."""
    pairs = {"rearrangement_default": C.default_config, "colour_separator": C.colour_separator_task_config,
             "controller_tuning": C.controller_tuning_config, "lasa_default": C.lasa_default_config,
             "lasa_deployment": C.lasa_deployment_config}
    for site, fn in pairs.items():
        assert dict(_leaves(fn())) == dict(_leaves(GOLD[site]["tree"])), site
    assert dict(_leaves(C.push_default_config())) == dict(_leaves(GOLD["rearrangement_default"]["tree"]))


def test_missing_props_choice_raises_like_hydra():
    """config/arena/rearrangement_table.yaml:3 selects `props: default`, a file the reference does not have: Hydra
    raises MissingConfigException unless the caller overrides arena/props (every call site does)."""
    assert GOLD["_no_props_override"] == "MissingConfigException"
    with pytest.raises(C.MissingConfig):
        C.compose("rearrangement", [])


def test_override_errors():
    with pytest.raises(KeyError):
        C.compose("rearrangement", ["arena/props=colour_splitter", "no_such_key=1"])
    with pytest.raises(C.MissingConfig):
        C.compose("rearrangement", ["arena/props=no_such_option"])
    c = C.compose("rearrangement", ["arena/props=colour_splitter", "+batch.num_envs=8"])
    assert c.batch.num_envs == 8
    # interpolation follows an override of its source (resolved after the assignments, like OmegaConf's lazy nodes)
    c = C.compose("rearrangement", ["arena/props=colour_splitter", "physics_dt=0.002"])
    assert c.robots.arm.controller_config.controller_params.physics_dt == 0.002
    assert c.robots.arm.arm.controller_config.controller.controller_config.physics_dt == 0.002
