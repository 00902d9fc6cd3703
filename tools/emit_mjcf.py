"""MJCF of this repo's scene spec (model/spec.py) -- the pinning hook of SURVEY.md section 8(c)/(d).

The reference's arithmetic is MuJoCo 3.2.7's (pyproject.toml:42), which is absent from this image and
cannot travel to the GPU box; the oracle is therefore "parity unpinned".  This module emits the
SAME model the oracle and the kernels consume (same body tree before fusing, inertials, hull boxes,
contact filters, equality / tendon / actuators, options) as an MJCF string, so that wherever
``import mujoco`` happens to succeed (bench.py's cpu_baseline leg and tests/test_mujoco_probe.py try
it at run time) genuine ``mj_step`` can be run on it and compared with the oracle.  Never a
dependency, never installed; nothing of the reference is read.

    python tools/emit_mjcf.py [--nprops 4] [--solver Newton] > scene.xml
    python tools/emit_mjcf.py --task push|lasa|base [--actuator position] > scene.xml   # the other tasks' own body trees
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import Optional, Sequence
from xml.sax.saxutils import quoteattr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mujoco_robot_environments_amd.model import spec as S  # noqa: E402

# contype / conaffinity bit masks reproducing compile.py's static pair filter: robot x robot and
# robot x ground never collide; props collide with everything; table and ground are static
_MASKS = {"robot": (1, 0), "pad": (1, 0), "prop": (2, 3), "table": (0, 3), "ground": (0, 2)}


def _r(t) -> float:
    """Model constants are defined as their nearest float32 (model/compile.py)."""
    import numpy as np
    return float(np.float32(t))


def _v(x) -> str:
    return " ".join(repr(_r(t)) for t in x)


def _geom(g: dict, size=None, masks=None) -> str:
    ct, ca = (masks or _MASKS)[g["group"]]
    a = [f"name={quoteattr(g['name'])}", f'contype="{ct}"', f'conaffinity="{ca}"',
         f'friction="{_v(g["friction"])}"', f'priority="{int(g["priority"])}"', f'condim="{int(g["condim"])}"',
         f'margin="{float(g["margin"])!r}"', f'gap="{float(g["gap"])!r}"', f'solref="{_v(g["solref"])}"',
         f'solimp="{_v(g["solimp"])}"']
    if g["type"] == "plane":
        a += ['type="plane"', 'size="2 2 2"']
    else:
        if g["type"] == "cylinder":   # MuJoCo: size = radius, half height
            a += ['type="cylinder"', f'size="{_v((g["size"][0], g["size"][2]))}"']
        else:
            a += ['type="box"', f'size="{_v(size if size is not None else g["size"])}"']
        a += [f'pos="{_v(g["pos"])}"', f'quat="{_v(g["quat"])}"']
        if g["mass"] is not None:
            a.append(f'mass="{float(g["mass"])!r}"')
    return "<geom " + " ".join(a) + "/>"


def _geom_groups(body: dict, acc=None) -> dict:
    """geom name -> group ("robot", "pad", "prop", "table", "ground") over a body tree of the scene spec."""
    acc = {} if acc is None else acc
    for g in body.get("geoms", []):
        acc[g["name"]] = g["group"]
    for c in body.get("children", []):
        _geom_groups(c, acc)
    return acc


def _joint(j: dict) -> str:
    if j["type"] == "free":
        return f"<freejoint name={quoteattr(j['name'])}/>"
    return (f"<joint name={quoteattr(j['name'])} type=\"hinge\" axis=\"{_v(j['axis'])}\" pos=\"{_v(j['pos'])}\" "
            f"range=\"{_v(j['range'])}\" limited=\"true\" armature=\"{float(j['armature'])!r}\" "
            f"damping=\"{float(j['damping'])!r}\" stiffness=\"{float(j['stiffness'])!r}\" "
            f"springref=\"{float(j['springref'])!r}\" solreflimit=\"{_v(j['solreflimit'])}\" "
            f"solimplimit=\"{_v(j['solimplimit'])}\"/>")


def _inertial(i: dict) -> str:
    if i["fullinertia"] is not None:
        return (f"<inertial mass=\"{float(i['mass'])!r}\" pos=\"{_v(i['pos'])}\" "
                f"fullinertia=\"{_v(i['fullinertia'])}\"/>")
    return (f"<inertial mass=\"{float(i['mass'])!r}\" pos=\"{_v(i['pos'])}\" quat=\"{_v(i['quat'])}\" "
            f"diaginertia=\"{_v(i['diaginertia'])}\"/>")


def _tool_inertial(g: dict) -> str:
    """A hull box that stands for a cylinder keeps the CYLINDER's mass and inertia (model/compile.py)."""
    m, (a, _, c) = float(g["mass"]), g["size"]
    ixx = m * (3 * a * a + 4 * c * c) / 12
    return (f"<inertial mass=\"{m!r}\" pos=\"{_v(g['pos'])}\" quat=\"{_v(g['quat'])}\" "
            f"diaginertia=\"{_v((ixx, ixx, m * a * a / 2))}\"/>")


def _body(node: dict, out: list, depth: int, prop_sizes, nprops: int, masks=None) -> None:
    name = node["name"]
    if name.startswith("prop_"):
        p = int(name.split("_")[1])
        if p >= nprops:
            return
    pad = "  " * depth
    if name == "world":
        for g in node["geoms"]:
            out.append(pad + _geom(g, masks=masks))
        for ch in node["children"]:
            _body(ch, out, depth, prop_sizes, nprops, masks)
        return
    out.append(f"{pad}<body name={quoteattr(name)} pos=\"{_v(node['pos'])}\" quat=\"{_v(node['quat'])}\">")
    if node["joint"] is not None:
        out.append(pad + "  " + _joint(node["joint"]))
    if node["inertial"] is not None:
        out.append(pad + "  " + _inertial(node["inertial"]))
    for g in node["geoms"]:
        if node["inertial"] is None and g.get("inertia_shape") == "cylinder":
            out.append(pad + "  " + _tool_inertial(g))
    for g in node["geoms"]:
        size = None
        if name.startswith("prop_") and prop_sizes is not None:
            size = prop_sizes[int(name.split("_")[1])]
        out.append(pad + "  " + _geom(g, size, masks))   # (a body with <inertial> ignores its geoms' masses)
    for s in node["sites"]:
        out.append(f"{pad}  <site name={quoteattr(s['name'])} pos=\"{_v(s['pos'])}\" quat=\"{_v(s['quat'])}\"/>")
    for ch in node["children"]:
        _body(ch, out, depth + 1, prop_sizes, nprops, masks)
    out.append(f"{pad}</body>")


def emit(scene: Optional[dict] = None, nprops: int = 4, prop_sizes: Optional[Sequence] = None,
         solver: Optional[str] = None) -> str:
    """MJCF string of ``scene`` (default: spec.default_scene()) with the first ``nprops`` cube slots and
    optional per-cube half sizes [4][3]."""
    scene = scene or S.default_scene()
    o = scene["option"]
    out = ['<mujoco model="rearrangement_amd_spec">',
           '  <compiler angle="radian" autolimits="true" inertiafromgeom="auto"/>',
           f'  <option timestep="{float(o["timestep"])!r}" gravity="{_v(o["gravity"])}" integrator="{o["integrator"]}" '
           f'cone="{o["cone"]}" impratio="{float(o["impratio"])!r}" solver="{solver or o["solver"]}" '
           f'iterations="{int(o["iterations"])}" tolerance="{float(o["tolerance"])!r}" '
           f'ls_iterations="{int(o.get("ls_iterations", 50))}" ls_tolerance="{float(o.get("ls_tolerance", 0.01))!r}"/>',
           '  <worldbody>']
    masks = dict(_MASKS)
    if scene.get("robot_ground_pairs", False):   # BaseEnv: the robot stands on the floor
        masks["ground"] = (0, 3)
    _body(scene["world"], out, 2, prop_sizes, nprops, masks)
    out.append('  </worldbody>')
    # robot self-collision: the compiled pair table keeps robot-robot pairs whose bodies are far enough apart along the
    # arm (model/compile.py: SELF_PAIR_MIN_LINKS) and the pads of opposite fingers.  The bit masks above switch
    # robot x robot off wholesale, so these pairs are listed explicitly, with every parameter spelled out
    from mujoco_robot_environments_amd.model import compile as MC
    A = MC.compile_scene(scene)
    gname = A["_names"]["geoms"]
    group = _geom_groups(scene["world"])
    pairs = []
    for k, (g1, g2) in enumerate(A["pair_geom"]):
        n1, n2 = gname[int(g1)], gname[int(g2)]
        if group.get(n1) in ("robot", "pad") and group.get(n2) in ("robot", "pad"):
            f = A["pair_friction"][k]
            pairs.append(f'    <pair geom1={quoteattr(n1)} geom2={quoteattr(n2)} condim="{int(A["pair_condim"][k])}" '
                         f'friction="{_v([f[0], f[0], f[1], f[2], f[2]])}" solref="{_v(A["pair_solref"][k])}" '
                         f'solimp="{_v(A["pair_solimp"][k])}" margin="{float(A["pair_margin"][k])!r}" gap="{float(A["pair_gap"][k])!r}"/>')
    if pairs:
        out.append('  <contact>')
        out += pairs
        out.append('  </contact>')
    out.append('  <equality>')
    for e in scene["equality"]:
        if e["type"] == "connect":
            out.append(f'    <connect body1={quoteattr(e["body1"])} body2={quoteattr(e["body2"])} anchor="{_v(e["anchor"])}" '
                       f'solref="{_v(e["solref"])}" solimp="{_v(e["solimp"])}"/>')
        else:
            out.append(f'    <joint joint1={quoteattr(e["joint1"])} joint2={quoteattr(e["joint2"])} polycoef="{_v(e["polycoef"])}" '
                       f'solref="{_v(e["solref"])}" solimp="{_v(e["solimp"])}"/>')
    out.append('  </equality>')
    t = scene["tendon"]
    if t["name"] != "null":   # (an arm-only scene fills the blob's tendon / eighth actuator slot with nulls)
        out.append('  <tendon>')
        out.append(f'    <fixed name={quoteattr(t["name"])}>')
        for j, c in zip(t["joints"], t["coef"]):
            out.append(f'      <joint joint={quoteattr(j)} coef="{float(c)!r}"/>')
        out.append('    </fixed>')
        out.append('  </tendon>')
    out.append('  <actuator>')
    for a in scene["actuators"]:
        if a["kind"] == "motor" and "gainprm" in a:   # position.yaml
            out.append(f'    <general name={quoteattr(a["name"])} joint={quoteattr(a["joint"])} gaintype="fixed" biastype="affine" '
                       f'gainprm="{float(a["gainprm"])!r} 0 0" biasprm="{_v(a["biasprm"])}" ctrlrange="{_v(a["ctrlrange"])}" '
                       f'ctrllimited="true" forcerange="{_v(a["forcerange"])}" forcelimited="true"/>')
        elif a.get("tendon") == "null":
            continue
        elif a["kind"] == "motor":
            out.append(f'    <motor name={quoteattr(a["name"])} joint={quoteattr(a["joint"])} gear="1" '
                       f'ctrlrange="{_v(a["ctrlrange"])}" ctrllimited="true"/>')
        else:
            out.append(f'    <general name={quoteattr(a["name"])} tendon={quoteattr(a["tendon"])} gaintype="fixed" biastype="affine" '
                       f'gainprm="{float(a["gainprm"])!r} 0 0" biasprm="{_v(a["biasprm"])}" ctrlrange="{_v(a["ctrlrange"])}" '
                       f'ctrllimited="true" forcerange="{_v(a["forcerange"])}" forcelimited="true"/>')
    out.append('  </actuator>')
    out.append('</mujoco>')
    return "\n".join(out) + "\n"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nprops", type=int, default=4)
    ap.add_argument("--solver", choices=["PGS", "Newton"], default=None)
    ap.add_argument("--task", choices=["rearrangement", "base", "push", "lasa"], default="rearrangement")
    ap.add_argument("--actuator", choices=["motor", "position"], default="motor")
    args = ap.parse_args()
    if args.task == "rearrangement":
        sys.stdout.write(emit(nprops=args.nprops, solver=args.solver))
    else:
        scene = S.other_task_scene(args.task, dict(actuator=args.actuator), embed=False)
        sizes = [[0.025] * 3] * 4 if args.task == "push" else None
        sys.stdout.write(emit(scene, nprops=1 if args.task == "push" else 0, prop_sizes=sizes, solver=args.solver))


if __name__ == "__main__":
    main()
