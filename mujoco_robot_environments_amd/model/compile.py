"""Model compiler: scene spec (spec.py) -> flat named arrays -> binary blob.

Does at build time what MuJoCo's XML compiler does for the reference's
``mjcf.Physics.from_mjcf_model`` (tasks/rearrangement.py:181):

* fuses joint-less bodies into their parent (``fusestatic`` semantics: the
  composite inertia is re-diagonalised, geoms/sites are re-expressed in the
  surviving body frame) so that every moving body carries exactly one joint;
* derives box inertias from geom mass, principal inertial frames from
  ``fullinertia``;
* builds the dof tree (``dof_parentid``, ``dof_Madr``) of the sparse
  mass matrix;
* evaluates the constants MuJoCo's ``mj_setConst`` derives at ``qpos0``
  (``dof_invweight0``, ``body_invweight0``, mean inertia) -- these feed the
  constraint regulariser R = (1-d)/d * diagApprox;
* enumerates collision pairs with MuJoCo's filters (contype/conaffinity,
  same body, parent-child, both-static) and mixes per-pair contact
  parameters (priority / max-friction / solmix, ``mj_contactParam``).

The output is a dict name -> np.ndarray (int32 or float64) plus a
self-describing binary blob (``to_blob``) that ``mre_create`` and the oracle
parse by entry name.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

from . import spec as _spec

BLOB_MAGIC = 0x4D524542  # 'MREB'
BLOB_VERSION = 4

JNT_NONE, JNT_HINGE, JNT_FREE = 0, 1, 2
GEOM_PLANE, GEOM_BOX, GEOM_CYLINDER = 0, 1, 2   # (cylinder: size = (r, r, half height), axis z)
EQ_CONNECT, EQ_JOINT = 0, 1
MINVAL = 1e-15


# ---------------------------------------------------------------- quaternions
def qnorm(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw])


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def q2m(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def m2q(m):
    """Rotation matrix -> unit quaternion (w,x,y,z), w >= 0 branch-stable."""
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s,
                      (m[1, 0] - m[0, 1]) / s])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s,
                      (m[0, 2] + m[2, 0]) / s])
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = np.array([(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s,
                      (m[1, 2] + m[2, 1]) / s])
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = np.array([(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s,
                      (m[1, 2] + m[2, 1]) / s, 0.25 * s])
    return qnorm(q)


def qrot(q, v):
    return q2m(q) @ np.asarray(v, dtype=np.float64)


def axisangle_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    s = np.sin(angle / 2)
    return np.array([np.cos(angle / 2), axis[0] * s, axis[1] * s, axis[2] * s])


def _full_to_mat(f):
    xx, yy, zz, xy, xz, yz = f
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]], dtype=np.float64)


# ------------------------------------------------------------------ flattening
class _Body:
    def __init__(self, name, parent, pos, quat):
        self.name = name
        self.parent = parent          # index into list
        self.pos = np.asarray(pos, dtype=np.float64)
        self.quat = qnorm(quat)
        self.joint = None
        self.masses = []              # list of (mass, com(3), I_com(3x3)) in body frame
        self.geoms = []
        self.sites = []
        self.names = [name]           # names of fused bodies (for equality lookup)
        self.frames = {name: (np.zeros(3), np.array([1.0, 0, 0, 0]))}


def _geom_inertia(g):
    if g["mass"] is None or g["type"] not in ("box", "cylinder"):
        return None
    m = float(g["mass"])
    a, b, c = g["size"]
    if g.get("inertia_shape") == "cylinder":   # radius a (= b), half height c, axis z
        I = np.diag([m * (3 * a * a + 4 * c * c) / 12, m * (3 * a * a + 4 * c * c) / 12, m * a * a / 2])
    else:
        I = np.diag([m / 3 * (b * b + c * c), m / 3 * (a * a + c * c), m / 3 * (a * a + b * b)])
    R = q2m(qnorm(g["quat"]))
    return m, np.asarray(g["pos"], dtype=np.float64), R @ I @ R.T


def _flatten(node, parent_idx, bodies, pos, quat):
    """DFS; joint-less non-world bodies are fused into ``parent_idx``.

    ``pos``/``quat`` = pose of ``node`` in the frame of body ``parent_idx``.
    """
    if node["joint"] is not None or parent_idx < 0:
        b = _Body(node["name"], parent_idx, pos, quat)
        b.joint = node["joint"]
        bodies.append(b)
        idx = len(bodies) - 1
        fpos, fquat = np.zeros(3), np.array([1.0, 0, 0, 0])
    else:
        idx = parent_idx
        b = bodies[idx]
        fpos, fquat = np.asarray(pos, dtype=np.float64), qnorm(quat)
        b.names.append(node["name"])
        b.frames[node["name"]] = (fpos, fquat)
    R = q2m(fquat)
    ine = node["inertial"]
    if ine is not None:
        if ine["fullinertia"] is not None:
            I = _full_to_mat(ine["fullinertia"])
        else:
            Ri = q2m(qnorm(ine["quat"]))
            I = Ri @ np.diag(ine["diaginertia"]) @ Ri.T
        b.masses.append((float(ine["mass"]), fpos + R @ np.asarray(ine["pos"], dtype=np.float64),
                         R @ I @ R.T))
    else:
        for g in node["geoms"]:
            gi = _geom_inertia(g)
            if gi is not None:
                m, c, I = gi
                b.masses.append((m, fpos + R @ c, R @ I @ R.T))
    for g in node["geoms"]:
        g2 = dict(g)
        g2["pos"] = fpos + R @ np.asarray(g["pos"], dtype=np.float64)
        g2["quat"] = qmul(fquat, qnorm(g["quat"]))
        b.geoms.append(g2)
    for s in node["sites"]:
        b.sites.append(dict(name=s["name"], pos=fpos + R @ np.asarray(s["pos"], dtype=np.float64),
                            quat=qmul(fquat, qnorm(s["quat"]))))
    for ch in node["children"]:
        cpos = fpos + R @ np.asarray(ch["pos"], dtype=np.float64)
        cquat = qmul(fquat, qnorm(ch["quat"]))
        _flatten(ch, idx, bodies, cpos, cquat)
    return idx


def _compose_inertia(masses):
    if not masses:
        return 0.0, np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3)
    M = sum(m for m, _, _ in masses)
    com = sum(m * c for m, c, _ in masses) / M
    I = np.zeros((3, 3))
    for m, c, Ic in masses:
        d = c - com
        I += Ic + m * (d @ d * np.eye(3) - np.outer(d, d))
    I = 0.5 * (I + I.T)
    w, V = np.linalg.eigh(I)
    order = np.argsort(-w)              # descending, like mju_eig3
    w, V = w[order], V[:, order]
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    if w[2] <= 0:
        raise ValueError(f"inertia not positive definite: {w}")
    if w[0] > (w[1] + w[2]) * (1 + 1e-9):
        # MuJoCo rejects A+B<C unless balanceinertia; recalled values may be off.
        raise ValueError(f"inertia violates triangle inequality: {w}")
    return M, com, m2q(V), w


# ----------------------------------------------------------------- kinematics
def forward_kinematics(mdl: Dict[str, np.ndarray], qpos: np.ndarray):
    """numpy restatement used only at compile time (qpos0 constants) and as an
    independent cross-check in tests. Returns (xpos, xquat) per body."""
    nb = int(mdl["nbody"][0])
    xpos = np.zeros((nb, 3))
    xquat = np.zeros((nb, 4))
    xquat[0] = [1, 0, 0, 0]
    for b in range(1, nb):
        p = int(mdl["body_parentid"][b])
        jt = int(mdl["body_jnttype"][b])
        qa = int(mdl["body_qposadr"][b])
        if jt == JNT_FREE:
            xpos[b] = qpos[qa:qa + 3]
            xquat[b] = qnorm(qpos[qa + 3:qa + 7])
            continue
        pos = xpos[p] + qrot(xquat[p], mdl["body_pos"][b])
        quat = qmul(xquat[p], mdl["body_quat"][b])
        if jt == JNT_HINGE:
            anchor = pos + qrot(quat, mdl["jnt_pos"][b])
            qloc = axisangle_quat(mdl["jnt_axis"][b], qpos[qa] - mdl["qpos0"][qa])
            quat = qnorm(qmul(quat, qloc))
            pos = anchor - qrot(quat, mdl["jnt_pos"][b])
        xpos[b], xquat[b] = pos, quat
    return xpos, xquat


def body_jacobian(mdl, xpos, xquat, b, point):
    """6 x nv Jacobian (translational rows 0:3, rotational 3:6) of ``point``
    rigidly attached to body ``b`` (world frame)."""
    nv = int(mdl["nv"][0])
    J = np.zeros((6, nv))
    while b > 0:
        jt = int(mdl["body_jnttype"][b])
        da = int(mdl["body_dofadr"][b])
        if jt == JNT_HINGE:
            axis = qrot(xquat[b], mdl["jnt_axis"][b])
            anchor = xpos[b] + qrot(xquat[b], mdl["jnt_pos"][b])
            J[0:3, da] = np.cross(axis, point - anchor)
            J[3:6, da] = axis
        elif jt == JNT_FREE:
            R = q2m(xquat[b])
            J[0:3, da:da + 3] = np.eye(3)
            for k in range(3):
                J[0:3, da + 3 + k] = np.cross(R[:, k], point - xpos[b])
                J[3:6, da + 3 + k] = R[:, k]
        b = int(mdl["body_parentid"][b])
    return J


def dense_mass_matrix(mdl, qpos):
    """M = sum_b J_b^T diag(m, I_b) J_b + armature  (independent of the CRB
    recursion used by the oracle / kernels)."""
    nb, nv = int(mdl["nbody"][0]), int(mdl["nv"][0])
    xpos, xquat = forward_kinematics(mdl, qpos)
    M = np.diag(mdl["dof_armature"].astype(np.float64))
    for b in range(1, nb):
        m = mdl["body_mass"][b]
        if m <= 0:
            continue
        com = xpos[b] + qrot(xquat[b], mdl["body_ipos"][b])
        Ri = q2m(qmul(xquat[b], mdl["body_iquat"][b]))
        Iw = Ri @ np.diag(mdl["body_inertia"][b]) @ Ri.T
        J = body_jacobian(mdl, xpos, xquat, b, com)
        M += m * J[0:3].T @ J[0:3] + J[3:6].T @ Iw @ J[3:6]
    return M


# --------------------------------------------------------------------- compile
def _mix_pair(g1, g2):
    """mj_contactParam: priority, else max condim / max friction / solmix."""
    if g1["priority"] != g2["priority"]:
        g = g1 if g1["priority"] > g2["priority"] else g2
        condim, fr = g["condim"], g["friction"]
        solref, solimp = g["solref"], g["solimp"]
    else:
        condim = max(g1["condim"], g2["condim"])
        fr = tuple(max(a, b) for a, b in zip(g1["friction"], g2["friction"]))
        mix = 0.5  # solmix = 1 on every geom of this scene
        solref = tuple(mix * a + (1 - mix) * b for a, b in zip(g1["solref"], g2["solref"]))
        solimp = tuple(mix * a + (1 - mix) * b for a, b in zip(g1["solimp"], g2["solimp"]))
    margin = max(g1["margin"], g2["margin"])
    gap = max(g1["gap"], g2["gap"])
    return condim, fr, solref, solimp, margin, gap


SELF_PAIR_MIN_LINKS = 4   # robot self-collision pairs: bodies at least this many arm links apart (see the pair filter)


def compile_scene(scene: dict = None) -> Dict[str, np.ndarray]:
    scene = scene or _spec.default_scene()
    bodies: List[_Body] = []
    _flatten(scene["world"], -1, bodies, (0, 0, 0), (1, 0, 0, 0))
    nb = len(bodies)
    name2body = {}
    for i, b in enumerate(bodies):
        for n in b.names:
            name2body[n] = i

    A: Dict[str, np.ndarray] = {}
    body_parent = np.array([max(b.parent, 0) for b in bodies], dtype=np.int32)
    jtype = np.zeros(nb, np.int32)
    dofadr = np.zeros(nb, np.int32)
    dofnum = np.zeros(nb, np.int32)
    qposadr = np.zeros(nb, np.int32)
    nv = nq = 0
    joint2body = {}
    for i, b in enumerate(bodies):
        dofadr[i], qposadr[i] = nv, nq
        if b.joint is None:
            assert i == 0, f"body {b.name} has no joint and was not fused"
            continue
        joint2body[b.joint["name"]] = i
        if b.joint["type"] == "hinge":
            jtype[i], dofnum[i] = JNT_HINGE, 1
            nv += 1
            nq += 1
        else:
            jtype[i], dofnum[i] = JNT_FREE, 6
            nv += 6
            nq += 7
    rootid = np.zeros(nb, np.int32)
    for i in range(1, nb):
        rootid[i] = i if body_parent[i] == 0 else rootid[body_parent[i]]

    body_pos = np.array([b.pos for b in bodies])
    body_quat = np.array([b.quat for b in bodies])
    body_mass = np.zeros(nb)
    body_ipos = np.zeros((nb, 3))
    body_iquat = np.tile([1.0, 0, 0, 0], (nb, 1))
    body_inertia = np.zeros((nb, 3))
    for i, b in enumerate(bodies):
        if i == 0:
            continue
        body_mass[i], body_ipos[i], body_iquat[i], body_inertia[i] = _compose_inertia(b.masses)

    jnt_pos = np.zeros((nb, 3))
    jnt_axis = np.tile([0.0, 0, 1.0], (nb, 1))
    jnt_range = np.zeros((nb, 2))
    jnt_limited = np.zeros(nb, np.int32)
    jnt_stiffness = np.zeros(nb)
    jnt_springref = np.zeros(nb)
    jnt_solref = np.tile(_spec.DEFAULT_SOLREF, (nb, 1)).astype(np.float64)
    jnt_solimp = np.tile(_spec.DEFAULT_SOLIMP, (nb, 1)).astype(np.float64)
    dof_bodyid = np.zeros(nv, np.int32)
    dof_parentid = np.full(nv, -1, np.int32)
    dof_armature = np.zeros(nv)
    dof_damping = np.zeros(nv)
    qpos0 = np.zeros(nq)
    prop_body = []
    for i, b in enumerate(bodies):
        if b.joint is None:
            continue
        j = b.joint
        da = dofadr[i]
        # parent dof = last dof of nearest ancestor body that has dofs
        p = body_parent[i]
        pd = -1
        while p > 0:
            if dofnum[p] > 0:
                pd = dofadr[p] + dofnum[p] - 1
                break
            p = body_parent[p]
        if j["type"] == "hinge":
            jnt_pos[i] = j["pos"]
            jnt_axis[i] = np.asarray(j["axis"], dtype=np.float64) / np.linalg.norm(j["axis"])
            jnt_range[i] = j["range"]
            jnt_limited[i] = 1           # compiler autolimits="true" in both robot XMLs
            jnt_stiffness[i] = j["stiffness"]
            jnt_springref[i] = j["springref"]
            jnt_solref[i] = j["solreflimit"]
            jnt_solimp[i] = j["solimplimit"]
            dof_bodyid[da] = i
            dof_parentid[da] = pd
            dof_armature[da] = j["armature"]
            dof_damping[da] = j["damping"]
        else:
            prop_body.append(i)
            for k in range(6):
                dof_bodyid[da + k] = i
                dof_parentid[da + k] = pd if k == 0 else da + k - 1
            qpos0[qposadr[i]:qposadr[i] + 7] = [0, 0, 0, 1, 0, 0, 0]
    dof_Madr = np.zeros(nv, np.int32)
    nM = 0
    for d in range(nv):
        dof_Madr[d] = nM
        k = d
        while k >= 0:
            nM += 1
            k = dof_parentid[k]

    # geoms / sites
    geoms, geom_body = [], []
    sites, site_body = [], []
    for late in (False, True):   # ``late`` geoms (spec.py: hulls of arm links 1..4) take the last ids
        for i, b in enumerate(bodies):
            for g in b.geoms:
                if bool(g.get("late", False)) == late:
                    geoms.append(g)
                    geom_body.append(i)
    for i, b in enumerate(bodies):
        for s in b.sites:
            sites.append(s)
            site_body.append(i)
    layout = scene.get("layout")
    if layout:
        # the kernels index the cube geoms from a fixed id and read a fixed geom count: disabled
        # placeholder boxes (no collision bits, 5 m under the floor) fill the ids a scene leaves free
        def placeholder(k):
            g = _spec.box(f"unused_{k}", (1e-3, 1e-3, 1e-3), pos=(0.0, 0.0, -5.0), contype=0, conaffinity=0,
                          group="none")
            g["late"] = False
            return g
        first_prop = min(i for i, g in enumerate(geoms) if g["group"] == "prop")
        n_world = next((i for i, b in enumerate(geom_body) if b != 0), len(geoms))   # world geoms come first
        pad = layout["prop_geom0"] - first_prop
        if pad < 0:
            raise ValueError(f"{first_prop} geoms precede the cubes, the kernels allow {layout['prop_geom0']}")
        for k in range(pad):
            geoms.insert(n_world, placeholder(k))
            geom_body.insert(n_world, 0)
        k = pad
        while len(geoms) < layout["ngeom"]:
            g = placeholder(k)
            g["late"] = True
            geoms.append(g)
            geom_body.append(0)
            k += 1
        if len(geoms) != layout["ngeom"]:
            raise ValueError(f"{len(geoms)} geoms, the kernels are compiled for {layout['ngeom']}")
    n_early = sum(1 for g in geoms if not g.get("late", False))
    # MuJoCo sorts nothing; keep spec order but put the plane first for pair typing
    ng = len(geoms)
    gtype = np.array([{"plane": GEOM_PLANE, "box": GEOM_BOX, "cylinder": GEOM_CYLINDER}[g["type"]] for g in geoms], np.int32)
    geom_size = np.array([g["size"] for g in geoms], dtype=np.float64)
    geom_pos = np.array([g["pos"] for g in geoms], dtype=np.float64)
    geom_quat = np.array([qnorm(g["quat"]) for g in geoms])
    geom_rbound = np.array([0.0 if t == GEOM_PLANE else (np.hypot(s[0], s[2]) if t == GEOM_CYLINDER else np.linalg.norm(s))
                            for t, s in zip(gtype, geom_size)])
    geom_propid = np.full(ng, -1, np.int32)
    body_propid = np.full(nb, -1, np.int32)
    for pi, bi in enumerate(prop_body):
        body_propid[bi] = pi
    for gi in range(ng):
        geom_propid[gi] = body_propid[geom_body[gi]]

    # arm link (1..7) a robot body hangs off (gripper and finger bodies: 7), 0 for the rest; finger side by the
    # branch below the arm's last link (bodies of one finger share the ancestor that hangs off link 7)
    arm_link_of_body = np.zeros(nb, np.int32)
    finger_side_of_body = np.zeros(nb, np.int32)
    arm_chain = [bi for bi in range(1, nb) if body_propid[bi] < 0 and int(jtype[bi]) == 1]   # hinge bodies, tree order
    n_arm = min(7, len(arm_chain))
    for k, bi in enumerate(arm_chain[:n_arm]):
        arm_link_of_body[bi] = k + 1
    for bi in arm_chain[n_arm:]:
        arm_link_of_body[bi] = 7
        r = bi
        while body_parent[r] != arm_chain[n_arm - 1] and body_parent[r] > 0:
            r = body_parent[r]
        finger_side_of_body[bi] = r
    # collision pairs with MuJoCo's static filters (engine_collision_driver.c:
    # contype/conaffinity, same body, filterparent, both bodies welded to world)
    pairs = []
    # pairs among the early geoms first, then the pairs that involve a late geom: the table keeps its
    # first entries (the kernels walk it 64 pairs per pass, contacts are listed in this order)
    order = [(a, c) for a in range(n_early) for c in range(a + 1, n_early)]
    order += [(a, c) for a in range(ng) for c in range(max(a + 1, n_early), ng)]
    for a, c in order:
        if True:
            ga, gc = geoms[a], geoms[c]
            ba, bc = geom_body[a], geom_body[c]
            if not ((ga["contype"] & gc["conaffinity"]) or (gc["contype"] & ga["conaffinity"])):
                continue
            if ba == bc:
                continue
            if ba == 0 and bc == 0:
                continue
            if (body_parent[ba] == bc and bc != 0) or (body_parent[bc] == ba and ba != 0):
                continue
            # robot self-collisions.  The reference's CPU env keeps them (MuJoCo's only static filter between two
            # robot geoms is parent - child; the author switches arm collisions off in the MJX port alone,
            # tasks/rearrangement_mjx.py:184-189).  The box hulls that stand in for the menagerie meshes are coarser
            # than the meshes, so hulls of NEARBY links overlap where the meshes clear each other (three links apart, the
            # finger hulls meet the hull of link 4 before joint 6 reaches its lower limit -- oracle KAT): a
            # robot-robot pair is kept when its two bodies are at least FOUR arm links apart (gripper and fingers
            # count as link 7), and the pads of opposite fingers collide with each other (a closing gripper)
            ra = ga["group"] in ("robot", "pad")
            rc = gc["group"] in ("robot", "pad")
            if ra and rc:
                if not scene.get("robot_self_pairs", False):
                    continue
                la, lc = arm_link_of_body[ba], arm_link_of_body[bc]
                pads_across = ga["group"] == "pad" and gc["group"] == "pad" and finger_side_of_body[ba] != finger_side_of_body[bc]
                if not (abs(la - lc) >= SELF_PAIR_MIN_LINKS or pads_across):
                    continue
            # robot vs ground: unreachable (ground is 0.4 m below the base)
            if not scene.get("robot_ground_pairs", False) and \
                    ((ra and gc["group"] == "ground") or (rc and ga["group"] == "ground")):
                continue
            g1, g2 = (a, c)
            if gtype[g1] > gtype[g2]:
                g1, g2 = g2, g1
            pairs.append((g1, g2))
    npair = len(pairs)
    pair_geom = np.array(pairs, dtype=np.int32).reshape(npair, 2)
    pair_condim = np.zeros(npair, np.int32)
    pair_friction = np.zeros((npair, 3))
    pair_solref = np.zeros((npair, 2))
    pair_solimp = np.zeros((npair, 5))
    pair_margin = np.zeros(npair)
    pair_gap = np.zeros(npair)
    pair_single = np.zeros(npair, np.int32)
    for k, (g1, g2) in enumerate(pairs):
        c, fr, sr, si, mg, gp = _mix_pair(geoms[g1], geoms[g2])
        pair_condim[k], pair_friction[k], pair_solref[k] = c, fr, sr
        pair_solimp[k], pair_margin[k], pair_gap[k] = si, mg, gp
        # hull geoms stand in for convex collision MESHES: MuJoCo's mesh narrow phase (mjc_Convex,
        # multiccd off) returns ONE contact per pair -> keep only the deepest point of the box test
        pair_single[k] = int(bool(geoms[g1]["hull"] or geoms[g2]["hull"]))
        if gtype[g2] == GEOM_CYLINDER:
            # the narrow phases cover cylinder - box (mro_cylbox / cyl_box): one contact, the cylinder as geom 2
            assert gtype[g1] == GEOM_BOX, "cylinder pairs: only cylinder - box is implemented"
            pair_single[k] = 1
        assert c == 3, "kernels implement condim 3 (all geoms of this scene)"

    A.update(
        nbody=np.array([nb], np.int32), nv=np.array([nv], np.int32), nq=np.array([nq], np.int32),
        nu=np.array([len(scene["actuators"])], np.int32), nM=np.array([nM], np.int32),
        ngeom=np.array([ng], np.int32), nsite=np.array([len(sites)], np.int32),
        npair=np.array([npair], np.int32), nprop=np.array([len(prop_body)], np.int32),
        body_parentid=body_parent, body_rootid=rootid, body_jnttype=jtype,
        body_dofadr=dofadr, body_dofnum=dofnum, body_qposadr=qposadr, body_propid=body_propid,
        body_pos=body_pos, body_quat=body_quat, body_ipos=body_ipos, body_iquat=body_iquat,
        body_mass=body_mass, body_inertia=body_inertia,
        jnt_pos=jnt_pos, jnt_axis=jnt_axis, jnt_range=jnt_range, jnt_limited=jnt_limited,
        jnt_stiffness=jnt_stiffness, jnt_springref=jnt_springref,
        jnt_solref=jnt_solref, jnt_solimp=jnt_solimp,
        dof_bodyid=dof_bodyid, dof_parentid=dof_parentid, dof_Madr=dof_Madr,
        dof_armature=dof_armature, dof_damping=dof_damping, qpos0=qpos0,
        geom_type=gtype, geom_bodyid=np.array(geom_body, np.int32), geom_size=geom_size,
        geom_pos=geom_pos, geom_quat=geom_quat, geom_rbound=geom_rbound,
        geom_propid=geom_propid,
        pair_geom=pair_geom, pair_condim=pair_condim, pair_friction=pair_friction,
        pair_solref=pair_solref, pair_solimp=pair_solimp, pair_margin=pair_margin,
        pair_gap=pair_gap, pair_single=pair_single,
        site_bodyid=np.array(site_body, np.int32),
        site_pos=np.array([s["pos"] for s in sites], dtype=np.float64).reshape(-1, 3),
        site_quat=np.array([s["quat"] for s in sites], dtype=np.float64).reshape(-1, 4),
        prop_bodyid=np.array(prop_body, np.int32),
    )
    site_names = [s["name"] for s in sites]
    A["eef_site"] = np.array([site_names.index(scene["eef_site"])], np.int32)
    A["tcp_site"] = np.array([site_names.index(scene["tcp_site"])], np.int32)
    arm_bodies = [joint2body[j] for j in scene["arm_joints"]]
    A["arm_dof"] = np.array([dofadr[b] for b in arm_bodies], np.int32)
    A["home_qpos"] = np.asarray(scene["home"], dtype=np.float64)

    # options
    o = scene["option"]
    A["opt_timestep"] = np.array([o["timestep"]])
    A["opt_gravity"] = np.asarray(o["gravity"], dtype=np.float64)
    A["opt_impratio"] = np.array([o["impratio"]])
    A["opt_cone"] = np.array([{"pyramidal": 0, "elliptic": 1}[o.get("cone", "elliptic")]], np.int32)  # mjtCone
    A["opt_tolerance"] = np.array([o["tolerance"]])
    A["opt_iterations"] = np.array([o["iterations"]], np.int32)
    A["opt_solver"] = np.array([{"PGS": 0, "Newton": 2}[o.get("solver", "PGS")]], np.int32)  # mjtSolver
    A["opt_ls_iterations"] = np.array([o.get("ls_iterations", 50)], np.int32)
    A["opt_ls_tolerance"] = np.array([o.get("ls_tolerance", 0.01)])

    # tendon + actuators
    t = scene["tendon"]
    A["ten_dof"] = np.array([dofadr[joint2body[j]] for j in t["joints"]], np.int32)
    A["ten_coef"] = np.asarray(t["coef"], dtype=np.float64)
    acts = scene["actuators"]
    A["act_ctrlrange"] = np.array([a["ctrlrange"] for a in acts], dtype=np.float64)
    A["act_dof"] = np.array([dofadr[joint2body[a["joint"]]] if a["kind"] == "motor" else -1
                             for a in acts], np.int32)
    ga = [a for a in acts if a["kind"] == "general_tendon"][0]
    A["grip_gainprm"] = np.array([ga["gainprm"]])
    A["grip_biasprm"] = np.asarray(ga["biasprm"], dtype=np.float64)
    A["grip_forcerange"] = np.asarray(ga["forcerange"], dtype=np.float64)
    # joint actuators as `general` actuators: motors are gain 1 / no bias / unlimited force
    A["act_gainprm"] = np.array([a.get("gainprm", 1.0) if a["kind"] == "motor" else 0.0 for a in acts])
    A["act_biasprm"] = np.array([a.get("biasprm", (0.0, 0.0, 0.0)) if a["kind"] == "motor" else (0.0, 0.0, 0.0)
                                 for a in acts], dtype=np.float64)
    A["act_forcerange"] = np.array([a.get("forcerange", (0.0, 0.0)) if a["kind"] == "motor" else (0.0, 0.0)
                                    for a in acts], dtype=np.float64)
    A["act_forcelimited"] = np.array([int(a["kind"] == "motor" and "forcerange" in a) for a in acts], np.int32)

    # ---- qpos0 constants (mj_setConst): invweight0, meaninertia
    xpos0, xquat0 = forward_kinematics(A, qpos0)
    M0 = dense_mass_matrix(A, qpos0)
    Minv = np.linalg.inv(M0)
    dof_invweight0 = np.zeros(nv)
    body_invweight0 = np.zeros((nb, 2))
    for i in range(1, nb):
        com = xpos0[i] + qrot(xquat0[i], body_ipos[i])
        J = body_jacobian(A, xpos0, xquat0, i, com)
        Ainv = J @ Minv @ J.T
        body_invweight0[i, 0] = np.trace(Ainv[0:3, 0:3]) / 3
        body_invweight0[i, 1] = np.trace(Ainv[3:6, 3:6]) / 3
        da = dofadr[i]
        if jtype[i] == JNT_HINGE:
            dof_invweight0[da] = Minv[da, da]
        elif jtype[i] == JNT_FREE:
            dof_invweight0[da:da + 3] = np.mean(np.diag(Minv)[da:da + 3])
            dof_invweight0[da + 3:da + 6] = np.mean(np.diag(Minv)[da + 3:da + 6])
    A["dof_invweight0"] = dof_invweight0
    A["body_invweight0"] = body_invweight0
    A["M0_diag"] = np.diag(M0).copy()

    # equality
    eqs = scene["equality"]
    neq = len(eqs)
    eq_type = np.zeros(neq, np.int32)
    eq_obj = np.zeros((neq, 2), np.int32)
    eq_data = np.zeros((neq, 8))
    eq_solref = np.zeros((neq, 2))
    eq_solimp = np.zeros((neq, 5))
    for k, e in enumerate(eqs):
        eq_solref[k], eq_solimp[k] = e["solref"], e["solimp"]
        if e["type"] == "connect":
            eq_type[k] = EQ_CONNECT
            b1, b2 = name2body[e["body1"]], name2body[e["body2"]]
            f1p, f1q = bodies[b1].frames[e["body1"]]
            f2p, f2q = bodies[b2].frames[e["body2"]]
            a1 = f1p + qrot(f1q, e["anchor"])          # anchor in fused-body-1 frame
            pw = xpos0[b1] + qrot(xquat0[b1], a1)      # world at qpos0
            a2 = q2m(xquat0[b2]).T @ (pw - xpos0[b2])  # same point in body-2 frame
            eq_obj[k] = (b1, b2)
            eq_data[k, 0:3], eq_data[k, 3:6] = a1, a2
        else:
            eq_type[k] = EQ_JOINT
            eq_obj[k] = (joint2body[e["joint1"]], joint2body[e["joint2"]])
            eq_data[k, 0:5] = e["polycoef"]
    A.update(neq=np.array([neq], np.int32), eq_type=eq_type, eq_obj=eq_obj, eq_data=eq_data,
             eq_solref=eq_solref, eq_solimp=eq_solimp)
    A["_names"] = dict(bodies=[b.name for b in bodies], geoms=[g["name"] for g in geoms],
                       sites=site_names)  # python-side only, not in blob
    # The model IS its fp32-representable form: every geometric / inertial / contact constant is
    # rounded to the nearest float32 here, so that the fp64 oracle and the fp32 kernels consume
    # bit-identical parameters.  (A link length of 0.0315 m held as 0.0315 by one side and as
    # float32(0.0315) = 0.031500000506... by the other differs by 5e-10 m: through the stiff `connect`
    # rows (K ~ 4e4 1/s^2) of the finger linkage that alone was a systematic 2e-3 rad/s^2 on the finger
    # accelerations.)  Solver / integrator options (opt_*) stay as specified.
    for k, v in A.items():
        if not k.startswith("_") and not k.startswith("opt_") and isinstance(v, np.ndarray) and v.dtype.kind == "f":
            A[k] = v.astype(np.float32).astype(np.float64)
    return A


def to_blob(A: Dict[str, np.ndarray]) -> bytes:
    """Self-describing little-endian blob: header, entry table, 8-byte aligned
    payloads.  Entry = name[32] | dtype(0=i32,1=f64) | count | byte offset."""
    entries = [(k, np.ascontiguousarray(v)) for k, v in A.items() if not k.startswith("_")]
    head = struct.pack("<III", BLOB_MAGIC, BLOB_VERSION, len(entries))
    table_size = len(entries) * (32 + 4 + 4 + 8)
    off = len(head) + 4 + table_size
    off = (off + 7) & ~7
    table, payload = b"", b""
    for name, arr in entries:
        if arr.dtype.kind in "iu":
            arr, code = arr.astype("<i4"), 0
        else:
            arr, code = arr.astype("<f8"), 1
        raw = arr.tobytes()
        table += struct.pack("<32sIIQ", name.encode(), code, arr.size, off + len(payload))
        payload += raw + b"\0" * ((-len(raw)) % 8)
    blob = head + struct.pack("<I", 0) + table
    blob += b"\0" * ((-len(blob)) % 8)
    assert len(blob) == off
    return blob + payload


def prop_constants(half_size, mass: float = 0.1):
    """Per-env cube constants (geom mass 0.1: environment/props.py:238).
    Returns (inertia diag(3), body_invweight0(2), dof_invweight0 (trans, rot))."""
    a, b, c = [float(x) for x in half_size]
    I = np.array([mass / 3 * (b * b + c * c), mass / 3 * (a * a + c * c),
                  mass / 3 * (a * a + b * b)])
    inv_rot = float(np.mean(1.0 / I))
    return I, np.array([1.0 / mass, inv_rot]), np.array([1.0 / mass, inv_rot])
