"""Scene specification for the RearrangementEnv hot path.

This is the *model data* the reference obtains at run time by loading MJCF
files that are NOT present under /root/reference:

* arena / table / props / camera: in-tree facts, cited per element below
  (reference paths relative to /root/reference/mujoco_robot_environments/).
* Franka Panda (``panda_nohand.xml``) and Robotiq 2F-85 (``2f85.xml``): come
  from mujoco_menagerie through a robot_descriptions wheel that is empty in
  the reference checkout (pyproject.toml:45, .gitmodules:7-9).  The numbers
  below are restated from the published menagerie models and are flagged
  ``unverified`` -- every parity claim in this repo is relative to THIS spec.
* collision *meshes* of the robot are not recoverable; they are replaced by
  conservative box hulls (``hull=True`` geoms).  Only the gripper pads (real
  box geoms in the menagerie model) are exact.

The spec is a plain tree of dicts; ``compile.py`` flattens it (fusing
joint-less bodies into their parents exactly as MuJoCo's ``fusestatic`` does)
into the arrays consumed by the HIP kernels and by the CPU oracle.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

SQ2 = math.sqrt(0.5)

# MuJoCo defaults (mjmodel.h / XML reference "default" column)
DEFAULT_SOLREF = (0.02, 1.0)
DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)
DEFAULT_FRICTION = (1.0, 0.005, 0.0001)


def body(name, pos=(0, 0, 0), quat=(1, 0, 0, 0), inertial=None, joint=None,
         geoms=None, sites=None, children=None):
    return dict(name=name, pos=tuple(pos), quat=tuple(quat), inertial=inertial,
                joint=joint, geoms=list(geoms or []), sites=list(sites or []),
                children=list(children or []))


def inertial(mass, pos, fullinertia=None, quat=None, diaginertia=None):
    return dict(mass=mass, pos=tuple(pos), fullinertia=fullinertia, quat=quat,
                diaginertia=diaginertia)


def hinge(name, axis, rng, armature, damping, pos=(0, 0, 0), stiffness=0.0,
          springref=0.0, solreflimit=DEFAULT_SOLREF, solimplimit=DEFAULT_SOLIMP):
    return dict(name=name, type="hinge", axis=tuple(axis), pos=tuple(pos),
                range=tuple(rng), armature=armature, damping=damping,
                stiffness=stiffness, springref=springref,
                solreflimit=tuple(solreflimit), solimplimit=tuple(solimplimit))


def free(name):
    return dict(name=name, type="free")


def box(name, size, pos=(0, 0, 0), quat=(1, 0, 0, 0), friction=DEFAULT_FRICTION,
        priority=0, condim=3, margin=0.0, gap=0.0, solref=DEFAULT_SOLREF,
        solimp=DEFAULT_SOLIMP, mass=None, hull=False, contype=1, conaffinity=1,
        group="robot"):
    return dict(name=name, type="box", size=tuple(size), pos=tuple(pos),
                quat=tuple(quat), friction=tuple(friction), priority=priority,
                condim=condim, margin=margin, gap=gap, solref=tuple(solref),
                solimp=tuple(solimp), mass=mass, hull=hull, contype=contype,
                conaffinity=conaffinity, group=group)


def plane(name, friction=DEFAULT_FRICTION):
    g = box(name, (0, 0, 0), friction=friction, group="ground")
    g["type"] = "plane"
    return g


# --------------------------------------------------------------------------
# Franka Emika Panda (menagerie franka_emika_panda/panda_nohand.xml) [3P,
# unverified].  Kinematic offsets: SURVEY.md App. A.3; the attachment frame
# is confirmed in-tree at models/arms/franka_emika.py:52-57; joint ranges are
# confirmed by config/robots/arm/actuator_config/position.yaml:5,19,33,47.
# Class defaults: armature 0.1, damping 1, axis 0 0 1.
# --------------------------------------------------------------------------
_PANDA_RANGES = [(-2.8973, 2.8973), (-1.7628, 1.7628), (-2.8973, 2.8973),
                 (-3.0718, -0.0698), (-2.8973, 2.8973), (-0.0175, 3.7525),
                 (-2.8973, 2.8973)]

_PANDA_LINKS = [
    # name, pos, quat, mass, com, fullinertia (xx yy zz xy xz yz)
    ("link1", (0, 0, 0.333), (1, 0, 0, 0), 4.970684,
     (0.003875, 0.002081, -0.04762),
     (0.70337, 0.70661, 0.0091170, -0.00013900, 0.0067720, 0.019169)),
    ("link2", (0, 0, 0), (SQ2, -SQ2, 0, 0), 0.646926,
     (-0.003141, -0.02872, 0.003495),
     (0.0079620, 0.02811, 0.025995, -0.003925, 0.010254, 0.000704)),
    ("link3", (0, -0.316, 0), (SQ2, SQ2, 0, 0), 3.228604,
     (0.027518, 0.039252, -0.066502),
     (0.037242, 0.036155, 0.01083, -0.004761, -0.011396, -0.012805)),
    ("link4", (0.0825, 0, 0), (SQ2, SQ2, 0, 0), 3.587895,
     (-0.05317, 0.104419, 0.027454),
     (0.025853, 0.019552, 0.028323, 0.007796, -0.001332, 0.008641)),
    ("link5", (-0.0825, 0.384, 0), (SQ2, -SQ2, 0, 0), 1.225946,
     (-0.011953, 0.041065, -0.038437),
     (0.035549, 0.029474, 0.008627, -0.002117, -0.004037, 0.000229)),
    ("link6", (0, 0, 0), (SQ2, SQ2, 0, 0), 1.666555,
     (0.060149, -0.014117, -0.010517),
     (0.001964, 0.004354, 0.005433, 0.000109, -0.001158, 0.000341)),
    ("link7", (0.088, 0, 0), (SQ2, SQ2, 0, 0), 0.735522,
     (0.010517, -0.004252, 0.061597),
     (0.012516, 0.010027, 0.004815, -0.000428, -0.001196, -0.000741)),
]

# box hulls standing in for the link collision meshes (deviation, see header)
_PANDA_HULLS = {
    # links 1..4 (the reference loads their menagerie meshes too, models/arms/franka_emika.py:7-11): the
    # elbow can reach the table and the cubes.  ``late``: these geoms get the ids AFTER the cubes so that
    # the ids of the geoms that existed before them (segmentation images, pair order) stay what they were.
    "link1": dict(size=(0.055, 0.065, 0.10), pos=(0.0, -0.02, -0.09), late=True),
    "link2": dict(size=(0.055, 0.10, 0.065), pos=(0.0, -0.07, 0.02), late=True),
    "link3": dict(size=(0.06, 0.065, 0.09), pos=(0.03, 0.02, -0.07), late=True),
    "link4": dict(size=(0.065, 0.085, 0.06), pos=(-0.045, 0.06, 0.0), late=True),
    "link5": dict(size=(0.055, 0.07, 0.11), pos=(0.0, 0.04, -0.12)),
    "link6": dict(size=(0.065, 0.055, 0.06), pos=(0.045, 0.0, 0.0)),
    "link7": dict(size=(0.05, 0.05, 0.04), pos=(0.0, 0.0, 0.07)),
}


def panda_spec(gripper: Optional[dict] = None, tool: Optional[dict] = None,
               late_links: Optional[Sequence[str]] = None, skip_links: Sequence[str] = ()) -> dict:
    """Body tree of the arm, rooted at link0 (welded to the robot base site).

    ``tool``: a geom added to the attachment body (tasks/push.py:156-163, tasks/lasa_draw.py:113-120);
    ``late_links``: the link hulls that take geom ids after the cubes (default: links 1..4, see
    _PANDA_HULLS); ``skip_links``: link hulls left out (pairs that can never touch in that scene).
    """
    attachment = body("attachment", pos=(0, 0, 0.107),
                      quat=(0.3826834, 0, 0, 0.9238795),
                      geoms=[tool] if tool is not None else [],
                      sites=[dict(name="attachment_site", pos=(0, 0, 0), quat=(1, 0, 0, 0))],
                      children=[gripper] if gripper is not None else [])
    child = attachment
    for i in range(6, -1, -1):
        name, pos, quat, mass, com, full = _PANDA_LINKS[i]
        geoms = []
        if name in _PANDA_HULLS and name not in skip_links:
            h = _PANDA_HULLS[name]
            g = box(f"{name}_hull", h["size"], pos=h["pos"], hull=True)
            g["late"] = bool(h.get("late", False)) if late_links is None else name in late_links
            geoms.append(g)
        child = body(name, pos=pos, quat=quat,
                     inertial=inertial(mass, com, fullinertia=full),
                     joint=hinge(f"joint{i + 1}", (0, 0, 1), _PANDA_RANGES[i],
                                 armature=0.1, damping=1.0),
                     geoms=geoms, children=[child])
    link0 = body("link0", inertial=inertial(
        0.629769, (-0.041018, -0.00014, 0.049974),
        fullinertia=(0.00315, 0.00388, 0.004285, 8.2904e-7, 0.00015, 8.2299e-6)),
        children=[child])
    return link0


# --------------------------------------------------------------------------
# Robotiq 2F-85 (menagerie robotiq_2f85/2f85.xml) [3P, unverified; the
# reference loads a private "_v4" fork, models/end_effectors/robotiq_2f85.py:7].
# option cone=elliptic impratio=10; class defaults restated per joint below.
# --------------------------------------------------------------------------
_LIM_SOLREF = (0.005, 1.0)
_LIM_SOLIMP = (0.95, 0.99, 0.001, 0.5, 2.0)
_PAD_SOLREF = (0.004, 1.0)
_PAD_SOLIMP = (0.95, 0.99, 0.001, 0.5, 2.0)


def _finger(side: str, mirrored: bool) -> List[dict]:
    q = (0, 0, 0, 1) if mirrored else (1, 0, 0, 0)
    sgn = -1.0 if mirrored else 1.0
    pad = body(f"{side}_pad", pos=(0, -0.0189, 0.01352),
               inertial=inertial(0.0035, (0, -0.0025, 0.0185),
                                 quat=(SQ2, 0, 0, SQ2),
                                 diaginertia=(4.73958e-07, 3.64583e-07, 1.23958e-07)),
               geoms=[
                   box(f"{side}_pad1", (0.011, 0.004, 0.009375), pos=(0, -0.0026, 0.028125),
                       friction=(0.7, 0.005, 0.0001), priority=1, solref=_PAD_SOLREF,
                       solimp=_PAD_SOLIMP, group="pad"),
                   box(f"{side}_pad2", (0.011, 0.004, 0.009375), pos=(0, -0.0026, 0.009375),
                       friction=(0.6, 0.005, 0.0001), priority=1, solref=_PAD_SOLREF,
                       solimp=_PAD_SOLIMP, group="pad"),
               ])
    follower = body(f"{side}_follower", pos=(0, 0.055, 0.0375),
                    inertial=inertial(0.0125222, (0, -0.011046, 0.0124786),
                                      quat=(1, 0.1664, 0, 0),
                                      diaginertia=(2.67415e-06, 2.4559e-06, 6.02031e-07)),
                    joint=hinge(f"{side}_follower_joint", (1, 0, 0), (-0.872664, 0.872664),
                                armature=0.001, damping=0.0, pos=(0, -0.018, 0.0065),
                                solreflimit=_LIM_SOLREF, solimplimit=_LIM_SOLIMP),
                    geoms=[box(f"{side}_follower_hull", (0.010, 0.005, 0.022),
                               pos=(0, 0.0005, 0.018), hull=True)],
                    children=[pad])
    spring_link = body(f"{side}_spring_link", pos=(0, sgn * 0.0132, 0.0609), quat=q,
                       inertial=inertial(0.0221642, (-8.65005e-09, 0.0181624, 0.0212658),
                                         quat=(0.663403, -0.244737, 0.244737, 0.663403),
                                         diaginertia=(8.96853e-06, 6.71733e-06, 2.63931e-06)),
                       joint=hinge(f"{side}_spring_link_joint", (1, 0, 0), (-0.29670597283, 0.8),
                                   armature=0.001, damping=0.00125, stiffness=0.05,
                                   springref=2.62),
                       children=[follower])
    coupler = body(f"{side}_coupler", pos=(0, 0.0315, -0.0041),
                   inertial=inertial(0.0140974, (0, 0.00301209, 0.0232175),
                                     quat=(0.705636, -0.0455904, 0.0455904, 0.705636),
                                     diaginertia=(4.16206e-06, 3.52216e-06, 8.88131e-07)),
                   joint=hinge(f"{side}_coupler_joint", (1, 0, 0), (-1.57, 0.0),
                               armature=0.001, damping=0.0,
                               solreflimit=_LIM_SOLREF, solimplimit=_LIM_SOLIMP))
    driver = body(f"{side}_driver", pos=(0, sgn * 0.0306011, 0.054904), quat=q,
                  inertial=inertial(0.00899563, (2.96931e-12, 0.0177547, 0.00107314),
                                    quat=(0.681301, 0.732003, 0, 0),
                                    diaginertia=(1.72352e-06, 1.60906e-06, 3.22006e-07)),
                  joint=hinge(f"{side}_driver_joint", (1, 0, 0), (0.0, 0.8),
                              armature=0.005, damping=0.1,
                              solreflimit=_LIM_SOLREF, solimplimit=_LIM_SOLIMP),
                  children=[coupler])
    return [driver, spring_link]


def _make_inert(node: dict) -> dict:
    """The same subtree with no mass, no geoms, no springs / dampers and wide joint ranges (armature kept)."""
    j = node["joint"]
    if j is not None:
        j = dict(j, stiffness=0.0, springref=0.0, damping=0.0, range=(-3.0, 3.0))
    return dict(node, inertial=None, geoms=[], joint=j, children=[_make_inert(c) for c in node["children"]])


def robotiq_spec(inert: bool = False) -> dict:
    """``inert``: the gripper's bodies and joints without mass, geoms or passive forces -- the
    placeholder that lets an arm-only scene (PushEnv, LasaDrawEnv) run on the kernels compiled for the
    arm + gripper topology: massless bodies add nothing to the arm rows of M or to its bias forces, the
    eight finger dofs keep M_ii = armature and see no force, so they stay at rest."""
    if inert:
        return _make_inert(robotiq_spec())
    base = body("base", pos=(0, 0, 0.0038), quat=(SQ2, 0, 0, -SQ2),
                inertial=inertial(0.777441, (0, -2.70394e-05, 0.0354675),
                                  quat=(1, -0.00152849, 0, 0),
                                  diaginertia=(0.000260285, 0.000225381, 0.000152708)),
                geoms=[box("gripper_base_hull", (0.032, 0.038, 0.045), pos=(0, 0, 0.04),
                           hull=True)],
                sites=[dict(name="pinch", pos=(0, 0, 0.145), quat=(1, 0, 0, 0))],
                children=_finger("right", False) + _finger("left", True))
    return body("base_mount", pos=(0, 0, 0.007), children=[base])


ROBOTIQ_EQUALITY = [
    # 2f85.xml <equality>: follower<->coupler closure of each 4-bar + driver sync
    dict(type="connect", body1="right_follower", body2="right_coupler", anchor=(0, 0, 0),
         solref=(0.005, 1.0), solimp=(0.95, 0.99, 0.001, 0.5, 2.0)),
    dict(type="connect", body1="left_follower", body2="left_coupler", anchor=(0, 0, 0),
         solref=(0.005, 1.0), solimp=(0.95, 0.99, 0.001, 0.5, 2.0)),
    dict(type="joint", joint1="right_driver_joint", joint2="left_driver_joint",
         polycoef=(0, 1, 0, 0, 0), solref=(0.005, 1.0),
         solimp=(0.95, 0.99, 0.001, 0.5, 2.0)),
]

ROBOTIQ_TENDON = dict(name="split", joints=("right_driver_joint", "left_driver_joint"),
                      coef=(0.5, 0.5))


def default_scene(cfg: Optional[Dict] = None, max_props: int = 4) -> dict:
    """Full scene = arena + table + robot + ``max_props`` cube slots.

    Mirrors RearrangementEnv.__init__ (tasks/rearrangement.py:73-210):
    ground plane friction 0.4 (models/arenas/empty_assets/arena.xml:14-15);
    table = box half-size (0.9,1.0,0.2) at (0.4,0,0.2), priority 10, margin/gap 0
    (tasks/rearrangement.py:86-101; the friction kw is not forwarded so the
    geom keeps (1,0.005,0.0001): environment/props.py:242-254,194-210);
    robot base at (0,0,0.4) (tasks/rearrangement.py:121-126); props are cubes
    with a free joint, mass 0.1, margin = gap = 0.15, priority 10
    (environment/props.py:238-240,194-210,258-304).
    Per-env prop count / half-size are *runtime* parameters of the batch.
    """
    cfg = cfg or {}
    motor = cfg.get("motor_ctrlrange",
                    [87.0, 87.0, 87.0, 87.0, 12.0, 12.0, 12.0])  # motor.yaml:2-13
    arm = panda_spec(gripper=robotiq_spec())
    robot_base = body("robot_base", pos=(0, 0, 0.4), children=[arm])
    table = body("table", pos=(0.4, 0.0, 0.2), geoms=[
        box("table", (0.9, 1.0, 0.2), priority=10, margin=0.0, gap=0.0, mass=10.0,
            group="table")])
    props = []
    for i in range(max_props):
        props.append(body(f"prop_{i}", joint=free(f"prop_{i}_free"), geoms=[
            box(f"prop_{i}", (0.0155, 0.0155, 0.0155), priority=10, margin=0.15, gap=0.15,
                mass=0.1, group="prop")]))
    world = body("world", geoms=[plane("ground", friction=(0.4, 0.005, 0.0001))],
                 children=[table, robot_base] + props)
    actuators = []
    for i in range(7):
        actuators.append(dict(name=f"actuator{i + 1}", kind="motor", joint=f"joint{i + 1}",
                              ctrlrange=(-motor[i], motor[i])))
    # models/end_effectors/robotiq_2f85.py:41-48 overwrites forcerange to +-1.5
    actuators.append(dict(name="fingers_actuator", kind="general_tendon", tendon="split",
                          gainprm=0.3137255, biasprm=(0.0, -100.0, -10.0),
                          ctrlrange=(0.0, 255.0), forcerange=(-1.5, 1.5)))
    return dict(
        world=world,
        equality=ROBOTIQ_EQUALITY,
        tendon=ROBOTIQ_TENDON,
        actuators=actuators,
        # robot self-collision (model/compile.py pair filter): opt-in.  The reference's CPU env keeps robot-robot
        # pairs (with the menagerie MESHES, which this repo replaces by box hulls); with them the arm of the bench's
        # full-range torque law folds onto itself and the closing pads press on each other, contact episodes that no
        # float32 arithmetic follows to 1e-4 (measured: 52 of 64 envs under the bar instead of 64) and that overflow
        # the large constraint capacities in 12 of 8192 grasps -- DESIGN.md section 9
        robot_self_pairs=bool(cfg.get("robot_self_pairs", False)),
        option=dict(
            timestep=float(cfg.get("physics_dt", 0.001)),       # config/rearrangement.yaml:2
            gravity=tuple(cfg.get("gravity", (0.0, 0.0, -9.8))),  # config/rearrangement.yaml:4
            integrator="implicitfast",   # inherited from panda_nohand.xml [3P]
            cone="elliptic", impratio=10.0,  # inherited from 2f85.xml [3P]
            # BASELINE.json's north_star prescribes PGS; the reference leaves MuJoCo's default, Newton
            # (tasks/rearrangement.py:77-80).  Both solvers are built; cfg["solver"] selects.
            solver=str(cfg.get("solver", "PGS")),
            iterations=100, tolerance=1e-8, ls_iterations=50, ls_tolerance=0.01,  # MuJoCo defaults
        ),
        arm_joints=[f"joint{i + 1}" for i in range(7)],
        eef_site="attachment_site",   # models/robot_arm.py:38
        tcp_site="pinch",             # models/robot_arm.py:57
        home=tuple(cfg.get("home", (0, -0.785, 0, -2.356, 0, 1.571, 0.785))),
        unverified=True,
    )


# --------------------------------------------------------------------------
# The reference's other tasks (tasks/base.py, tasks/push.py, tasks/lasa_draw.py) on the same kernels
# --------------------------------------------------------------------------
def tool_cylinder(shape: str = "cylinder") -> dict:
    """The cylinder added to the arm's attachment body for non-prehensile work (tasks/push.py:156-163,
    tasks/lasa_draw.py:113-120): radius 0.015, half height 0.05, centre 0.05 above the attachment site,
    default density 1000.  It collides as a cylinder (one contact per pair like MuJoCo's convex collider; narrow phase:
    oracle mro_cylbox / csrc/mre_collide.h cyl_box -- a closed-form separating-axis search).
    ``shape="hull"`` restores rounds 1-3's stand-in, the box hull of the cylinder.  ``size`` = (r, r, half height)
    either way (the camera draws the hull)."""
    r, hh = 0.015, 0.05
    g = box("tool_cylinder", (r, r, hh), pos=(0.0, 0.0, 0.05), hull=(shape == "hull"))
    if shape != "hull":
        g["type"] = "cylinder"
    g["mass"] = 1000.0 * math.pi * r * r * 2 * hh
    g["inertia_shape"] = "cylinder"
    return g


_POSITION_ACTUATORS = [
    # config/robots/arm/actuator_config/position.yaml: ctrlrange, forcerange, kp, kv per joint
    ((-2.8973, 2.8973), 87.0, 4500.0, 450.0), ((-1.7628, 1.7628), 87.0, 4500.0, 450.0),
    ((-2.8973, 2.8973), 87.0, 3500.0, 350.0), ((-3.0718, -0.0698), 87.0, 3500.0, 350.0),
    ((-2.8973, 2.8973), 12.0, 2000.0, 200.0), ((-0.0175, 3.7525), 12.0, 2000.0, 200.0),
    ((-2.8973, 2.8973), 12.0, 2000.0, 200.0),
]


def _arm_actuators(kind: str, motor, position_rows=None) -> List[dict]:
    acts = []
    position_rows = position_rows or _POSITION_ACTUATORS
    for i in range(7):
        if kind == "motor":      # config/robots/arm/actuator_config/motor.yaml
            acts.append(dict(name=f"actuator{i + 1}", kind="motor", joint=f"joint{i + 1}",
                             ctrlrange=(-motor[i], motor[i])))
        elif kind == "position":  # models/arms/franka_emika.py `general` branch of _add_actuators
            cr, fr, kp, kv = position_rows[i]
            acts.append(dict(name=f"actuator{i + 1}", kind="motor", joint=f"joint{i + 1}", ctrlrange=cr,
                             gainprm=kp, biasprm=(0.0, -kp, -kv), forcerange=(-fr, fr)))
        else:
            raise ValueError(f"actuator config {kind!r}: motor | position")
    return acts


def other_task_scene(task: str, cfg: Optional[Dict] = None, embed: bool = True, max_props: int = 4) -> dict:
    """Scenes of the reference's other environments.

    ``task``:
      * ``"base"`` -- BaseEnv (tasks/base.py:78-87): arm + gripper on the arena floor (robot base at the
        origin), no table, no props;
      * ``"push"`` -- PushEnv (tasks/push.py:83-221): nine table slabs (one 0.5 m wide under the robot, eight
        0.1 m wide in front of it; all boxes of half height 0.2 centred at z = 0.2), the arm WITHOUT a
        gripper but with the tool cylinder, one free block of half size 0.025 and mass 0.05.  Quirk
        preserved: ``Rectangle._build`` does not forward its ``friction`` argument to ``_make``
        (environment/props.py:226-260), so the friction gradient of tasks/push.py:112-125 never reaches the
        geoms -- every slab and the block keep MuJoCo's default (1, 0.005, 0.0001);
        ``cfg["forward_friction"] = True`` applies the gradient the author meant;
      * ``"lasa"`` -- LasaDrawEnv (tasks/lasa_draw.py:83-128): one table (half size 1, 1, 0.2 at
        (0.4, 0, 0.2)), the arm with the tool cylinder, no props; config/lasa.yaml: physics_dt 0.01;
        ``cfg["actuator"] = "position"`` is the deployment config (tasks/lasa_draw.py:45-56).

    ``embed``: True gives the scene in the topology the kernels are compiled for (an inert gripper below
    the attachment body when the task has none, four cube slots); False gives the task's own body tree
    (used by the oracle cross-check of the embedding and by tools/emit_mjcf.py).

    Without the 2F-85 nothing sets ``cone`` / ``impratio``: MuJoCo's defaults apply there, pyramidal / 1, and
    that is what push / lasa run here (``cfg["cone"] = "elliptic"`` selects the other cone; rounds 1-2 ran these
    tasks with it, tests/test_other_tasks.py measures what that cost).  The kernels and the oracle implement
    both (csrc/mre_solver.h: assemble_constraints).
    """
    cfg = cfg or {}
    motor = cfg.get("motor_ctrlrange", [87.0, 87.0, 87.0, 87.0, 12.0, 12.0, 12.0])
    has_gripper = task == "base"
    if has_gripper:
        gripper = robotiq_spec()
    else:
        gripper = robotiq_spec(inert=True) if embed else None
    tool = None if has_gripper else tool_cylinder(cfg.get("tool_shape", "cylinder"))
    statics, props, robot_z = [], [], 0.4
    if task == "base":
        robot_z = 0.0                                  # tasks/base.py:81-85
        arm = panda_spec(gripper=gripper)
    elif task == "push":
        frictions = [DEFAULT_FRICTION] * 9
        if cfg.get("forward_friction", False):
            frictions = [DEFAULT_FRICTION] + [(0.4 + 0.4 * k / 7.0, 0.005, 0.0001) for k in range(8)]
        centers = [-0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]   # tasks/push.py:98-101,128-149
        halfx = [0.25] + [0.05] * 8
        for k in range(9):
            statics.append(body(f"table_{k}", pos=(centers[k], 0.0, 0.2), geoms=[
                box(f"table_{k}", (halfx[k], 1.0, 0.2), priority=10, margin=0.0, gap=0.0, mass=10.0,
                    friction=frictions[k], group="table")]))
        # links 1..3 stay above z = 0.6: they reach neither the slabs nor the block
        arm = panda_spec(gripper=gripper, tool=tool, late_links=("link4", "link5", "link6"),
                         skip_links=("link1", "link2", "link3"))
    elif task == "lasa":
        statics.append(body("table", pos=(0.4, 0.0, 0.2), geoms=[
            box("table", (1.0, 1.0, 0.2), priority=10, margin=0.0, gap=0.0, mass=10.0, group="table")]))
        arm = panda_spec(gripper=gripper, tool=tool)
    else:
        raise ValueError(f"task {task!r}: base | push | lasa")
    nslots = max_props if embed else (1 if task == "push" else 0)
    for i in range(nslots):
        if task == "push":   # tasks/push.py:190-204 (friction not forwarded, see above)
            fr = (0.01, 0.005, 0.0001) if cfg.get("forward_friction", False) else DEFAULT_FRICTION
            g = box(f"prop_{i}", (0.025, 0.025, 0.025), priority=10, margin=0.0, gap=0.0, mass=0.05,
                    friction=fr, group="prop")
        else:
            g = box(f"prop_{i}", (0.0155, 0.0155, 0.0155), priority=10, margin=0.15, gap=0.15, mass=0.1,
                    group="prop")
        props.append(body(f"prop_{i}", joint=free(f"prop_{i}_free"), geoms=[g]))
    robot_base = body("robot_base", pos=(0, 0, robot_z), children=[arm])
    world = body("world", geoms=[plane("ground", friction=(0.4, 0.005, 0.0001))],
                 children=statics + [robot_base] + props)
    actuators = _arm_actuators(cfg.get("actuator", "motor"), motor, cfg.get("position_actuators"))
    if has_gripper:
        actuators.append(dict(name="fingers_actuator", kind="general_tendon", tendon="split",
                              gainprm=0.3137255, biasprm=(0.0, -100.0, -10.0),
                              ctrlrange=(0.0, 255.0), forcerange=(-1.5, 1.5)))
        equality, tendon = ROBOTIQ_EQUALITY, ROBOTIQ_TENDON
    elif embed:   # the inert gripper keeps its (force-free) loop closures; its actuator has zero gain
        actuators.append(dict(name="fingers_actuator", kind="general_tendon", tendon="split",
                              gainprm=0.0, biasprm=(0.0, 0.0, 0.0),
                              ctrlrange=(0.0, 255.0), forcerange=(-1.5, 1.5)))
        equality, tendon = ROBOTIQ_EQUALITY, ROBOTIQ_TENDON
    else:         # the oracle's blob reader has eight actuator slots: the last one is a null actuator
        actuators.append(dict(name="null_actuator", kind="general_tendon", tendon="null",
                              gainprm=0.0, biasprm=(0.0, 0.0, 0.0),
                              ctrlrange=(0.0, 255.0), forcerange=(-1.5, 1.5)))
        equality, tendon = [], dict(name="null", joints=("joint1", "joint1"), coef=(0.0, 0.0))
    default_dt = 0.01 if task == "lasa" else 0.001    # config/lasa.yaml:2, config/rearrangement.yaml:2
    return dict(
        world=world, equality=equality, tendon=tendon, actuators=actuators,
        option=dict(
            timestep=float(cfg.get("physics_dt", default_dt)),
            gravity=tuple(cfg.get("gravity", (0.0, 0.0, -9.8))),
            integrator="implicitfast", cone=str(cfg.get("cone", "elliptic" if has_gripper else "pyramidal")),
            impratio=10.0 if has_gripper else 1.0,   # base: inherited from 2f85.xml; push / lasa: MuJoCo's defaults
            solver=str(cfg.get("solver", "Newton")),   # MuJoCo's default: these tasks set no solver either
            iterations=100, tolerance=1e-8, ls_iterations=50, ls_tolerance=0.01,
        ),
        arm_joints=[f"joint{i + 1}" for i in range(7)],
        eef_site="attachment_site",
        tcp_site="pinch" if (has_gripper or embed) else "attachment_site",
        home=tuple(cfg.get("home", (0, -0.785, 0, -2.356, 0, 1.571, 0.785))),
        robot_ground_pairs=(task == "base"),
        layout=dict(prop_geom0=12, ngeom=20) if embed else None,
        task=task,
        unverified=True,
    )
