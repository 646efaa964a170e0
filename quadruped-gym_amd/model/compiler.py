"""MJCF-subset model compiler for the fixed 12-DoF quadruped.

Reads the reference robot description (``src/models/quadruped/scene.xml`` ->
``quadruped.xml`` + ``mesh/*.obj`` in antopio26/quadruped-gym) and derives the
constants the HIP rigid-body pipeline and the CPU oracle need:

* the kinematic tree (13 moving bodies: FRAME + 4 x (fema, shin, foot)),
* per-body mass / centre of mass / inertia tensor summed over the body's mesh
  geoms (each mesh's uniform-density inertia scaled to the geom's ``mass``),
* joint axes / ref / range / damping / armature (``quadruped.xml:9,24-37``),
* the position-servo parameters (``quadruped.xml:10-37,156-172``),
* a small set of ground-contact sample points per body, taken from the convex
  hull of the body's geoms (MuJoCo collides meshes through their hulls).

The reference delegates all of this to ``mujoco.MjModel.from_xml_path``
(``src/envs/quadruped.py:59``); MuJoCo is not available offline, so this is a
from-scratch restatement of the MJCF semantics that file relies on (SURVEY.md
Appendix A).  Only the MJCF subset the reference model uses is supported.

Output: a JSON document (``quadruped_model.json``) plus a generated C header
(``include/qg_model_data.h``) holding the same numbers as a ``qg_model``
initialiser.  The repository ships those derived constants; the XML/OBJ assets
themselves stay in the reference checkout.
"""
from __future__ import annotations

import json
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

NLEG = 4
NBODY = 13
NJNT = 12
MAXCP = 12          # storage width of the per-body contact-point table
CP_FRAME = 12       # FRAME: 3 orbits of its 4-fold symmetry (servo bottoms, servo tops, plate corners)
CP_LINK = 8         # fema / shin / foot

# Soft-constraint defaults of the engine being replaced (MuJoCo solref/solimp
# defaults): time constant 0.02 s, damping ratio 1,
# impedance d0 = 0.9.  Used only to pick the stiffness scale of our LCP-free
# penalty model so it is about as soft as the reference's contacts.
_SOLREF_TIMECONST = 0.02
_SOLIMP_D0 = 0.9


# --------------------------------------------------------------------------- #
# small rotation helpers
# --------------------------------------------------------------------------- #
def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def quat_to_mat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def axis_angle_quat(axis, angle):
    axis = np.asarray(axis, float)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([[math.cos(angle / 2)], math.sin(angle / 2) * axis])


def euler_to_quat(e, degrees):
    """MJCF default eulerseq="xyz": intrinsic rotations about x, then y, then z."""
    e = np.asarray(e, float)
    if degrees:
        e = np.deg2rad(e)
    q = np.array([1.0, 0, 0, 0])
    for k in range(3):
        ax = np.zeros(3)
        ax[k] = 1
        q = quat_mul(q, axis_angle_quat(ax, e[k]))
    return q


# --------------------------------------------------------------------------- #
# meshes
# --------------------------------------------------------------------------- #
def load_obj(path):
    verts, faces = [], []
    with open(path) as fh:
        for line in fh:
            if line.startswith("v "):
                verts.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) - 1 for tok in line.split()[1:]]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(verts, float), np.asarray(faces, int)


def solid_mass_properties(verts, faces):
    """Volume, centre of mass and unit-density inertia (about the COM) of the
    closed triangulated solid, by signed tetrahedra against the origin."""
    a = verts[faces[:, 0]]
    b = verts[faces[:, 1]]
    c = verts[faces[:, 2]]
    det = np.einsum("ij,ij->i", a, np.cross(b, c))  # 6 x signed tetra volume
    vol = det.sum() / 6.0
    com = (det[:, None] * (a + b + c)).sum(0) / (24.0 * vol)
    # second moments  int x_i x_j dV  of tetra (0,a,b,c): det/120 * (sum-of-pairs)
    s = a + b + c
    cov = (np.einsum("n,ni,nj->ij", det, a, a) + np.einsum("n,ni,nj->ij", det, b, b)
           + np.einsum("n,ni,nj->ij", det, c, c) + np.einsum("n,ni,nj->ij", det, s, s)) / 120.0
    cov_c = cov - vol * np.outer(com, com)
    inertia = np.trace(cov_c) * np.eye(3) - cov_c
    if vol < 0:  # inward-facing winding
        vol, inertia = -vol, -inertia
    return vol, com, inertia


def hull_faces(verts):
    from scipy.spatial import ConvexHull
    hull = ConvexHull(verts)
    faces = hull.simplices.copy()
    # orient outward using the facet normals scipy reports
    a = verts[faces[:, 0]]
    b = verts[faces[:, 1]]
    c = verts[faces[:, 2]]
    nrm = np.cross(b - a, c - a)
    flip = np.einsum("ij,ij->i", nrm, hull.equations[:, :3]) < 0
    faces[flip] = faces[flip][:, ::-1]
    return hull.vertices, faces


def mesh_properties(verts, faces, mode):
    """Shape properties of a mesh under one of MuJoCo's mesh-inertia modes.

    ``convex`` - uniform density over the convex hull (the engine's default
    since the 3.3 series; the reference pins no version);
    ``exact``  - uniform density over the closed surface itself.
    """
    if mode == "convex":
        _, hf = hull_faces(verts)
        return solid_mass_properties(verts, hf)
    if mode == "exact":
        return solid_mass_properties(verts, faces)
    raise ValueError(f"unknown mesh inertia mode {mode!r}")


# --------------------------------------------------------------------------- #
# MJCF parsing (subset)
# --------------------------------------------------------------------------- #
def _floats(s):
    return [float(x) for x in s.split()]


class _Defaults:
    """Nested <default class=...> tables with parent inheritance."""

    def __init__(self):
        self.tables = {}  # class -> {tag: attrs}
        self.parent = {}

    def load(self, node, parent=None):
        name = node.get("class", "main")
        self.parent[name] = parent
        tab = {}
        for child in node:
            if child.tag == "default":
                continue
            tab[child.tag] = dict(child.attrib)
        self.tables[name] = tab
        for child in node:
            if child.tag == "default":
                self.load(child, name)

    def resolve(self, cls, tag):
        chain = []
        while cls is not None:
            chain.append(cls)
            cls = self.parent.get(cls)
        out = {}
        for c in reversed(chain):
            out.update(self.tables.get(c, {}).get(tag, {}))
        return out


def _load_mjcf(path):
    """Parse an MJCF file, splicing <include file=...> in place."""
    tree = ET.parse(path)
    root = tree.getroot()
    base = os.path.dirname(os.path.abspath(path))
    roots = [(root, base)]
    for inc in root.findall("include"):
        sub_path = os.path.join(base, inc.get("file"))
        roots.extend(_load_mjcf(sub_path))
    return roots


def compile_mjcf(scene_path, mesh_inertia="convex", keep_hulls=False):
    """Compile the reference MJCF into the constant tables (a plain dict)."""
    if not os.path.exists(scene_path):
        raise FileNotFoundError(f"Model file not found: {scene_path}")
    roots = _load_mjcf(scene_path)

    degrees = True
    meshdir = "."
    integrator = "euler"
    timestep = 0.002  # MuJoCo default; the reference sets none (quadruped.xml:4)
    gravity = [0.0, 0.0, -9.81]
    defaults = _Defaults()
    defaults.parent["main"] = None
    defaults.tables["main"] = {}
    mesh_files = {}
    mesh_base = None
    floor_friction = 1.0
    robot_body = None
    actuators = []
    sensors = []

    for root, base in roots:
        comp = root.find("compiler")
        if comp is not None:
            degrees = comp.get("angle", "degree") == "degree"
            meshdir = comp.get("meshdir", ".")
            mesh_base = os.path.join(base, meshdir)
        opt = root.find("option")
        if opt is not None:
            integrator = opt.get("integrator", integrator)
            timestep = float(opt.get("timestep", timestep))
            if opt.get("gravity"):
                gravity = _floats(opt.get("gravity"))
        for d in root.findall("default"):
            defaults.load(d, None)
        for asset in root.findall("asset"):
            for m in asset.findall("mesh"):
                mesh_files[m.get("name")] = os.path.join(mesh_base or base, m.get("file"))
        wb = root.find("worldbody")
        if wb is not None:
            for g in wb.findall("geom"):
                if g.get("type") == "plane":
                    fr = g.get("friction")
                    floor_friction = _floats(fr)[0] if fr else 1.0
            for b in wb.findall("body"):
                robot_body = b
        act = root.find("actuator")
        if act is not None:
            actuators = list(act)
        sen = root.find("sensor")
        if sen is not None:
            sensors = list(sen)

    if integrator != "implicitfast":
        raise ValueError("only integrator=implicitfast (quadruped.xml:4) is supported")
    if robot_body is None:
        raise ValueError("no robot body under <worldbody>")

    mesh_cache = {}

    def mesh_data(name):
        if name not in mesh_cache:
            v, f = load_obj(mesh_files[name])
            vol, com, inertia = mesh_properties(v, f, mesh_inertia)
            hv, _ = hull_faces(v)
            mesh_cache[name] = dict(verts=v, vol=vol, com=com, inertia=inertia, hull=v[hv])
        return mesh_cache[name]

    def frame_of(node):
        pos = np.array(_floats(node.get("pos", "0 0 0")))
        if node.get("quat"):
            quat = np.array(_floats(node.get("quat")))
        elif node.get("euler"):
            quat = euler_to_quat(_floats(node.get("euler")), degrees)
        else:
            quat = np.array([1.0, 0, 0, 0])
        return pos, quat / np.linalg.norm(quat)

    bodies = []   # dicts in depth-first order
    joints = []   # hinge joints in depth-first order

    def walk(node, parent_idx, childclass):
        childclass = node.get("childclass", childclass)
        pos, quat = frame_of(node)
        body = dict(name=node.get("name"), parent=parent_idx, pos=pos, quat=quat, geoms=[], joint=None)
        idx = len(bodies)
        bodies.append(body)
        for j in node.findall("joint"):
            attrs = defaults.resolve(j.get("class", childclass), "joint")
            attrs.update(j.attrib)
            body["joint"] = attrs
        for g in node.findall("geom"):
            attrs = defaults.resolve(g.get("class", childclass), "geom")
            attrs.update(g.attrib)
            gpos, gquat = frame_of(g)
            body["geoms"].append(dict(name=g.get("name"), mesh=attrs["mesh"], mass=float(attrs["mass"]),
                                      friction=_floats(attrs.get("friction", "1"))[0],
                                      margin=float(attrs.get("margin", 0.0)), pos=gpos, quat=gquat))
        for child in node.findall("body"):
            walk(child, idx, childclass)

    walk(robot_body, -1, None)
    if len(bodies) != NBODY:
        raise ValueError(f"expected {NBODY} bodies, found {len(bodies)}")

    # --- body inertials ------------------------------------------------------
    out_bodies = []
    hull_clouds = []
    for b in bodies:
        mass = 0.0
        first = np.zeros(3)
        parts = []
        for g in b["geoms"]:
            md = mesh_data(g["mesh"])
            R = quat_to_mat(g["quat"])
            com = g["pos"] + R @ md["com"]
            inertia = R @ (md["inertia"] * (g["mass"] / md["vol"])) @ R.T
            parts.append((g["mass"], com, inertia))
            mass += g["mass"]
            first += g["mass"] * com
        ipos = first / mass
        inertia = np.zeros((3, 3))
        for m, com, I in parts:
            d = com - ipos
            inertia += I + m * (d @ d * np.eye(3) - np.outer(d, d))
        # symmetric bodies: remove round-off dust (far below the 1e-6 resolution of the OBJ vertices) so exact zeros stay zeros
        ipos = np.where(np.abs(ipos) < 1e-9, 0.0, ipos)
        inertia = np.where(np.abs(inertia) < 1e-6 * np.abs(np.diag(inertia)).max(), 0.0, inertia)
        # contact sample points: hull vertices of all the body's geoms, in the body frame
        cloud = []
        for g in b["geoms"]:
            md = mesh_data(g["mesh"])
            cloud.append(md["hull"] @ quat_to_mat(g["quat"]).T + g["pos"])
        cloud = np.vstack(cloud)
        hv, _ = hull_faces(cloud)
        hull_clouds.append(cloud[hv])
        if b["parent"] < 0:
            cps = select_contact_points(cloud[hv], CP_FRAME, fold=4)
            # exact 4-fold orbits (3 base points x 4 quarter turns, stored orbit-major): the mesh is symmetric
            # only to its 1e-6 vertex resolution; the one-leg-per-lane kernel rotates the base points itself
            orbits = max(1, len(cps) // 4)
            base_pts = [cps[4 * (o % orbits)] for o in range(CP_FRAME // 4)]     # a hull with few vertices repeats its orbits
            quarter = [np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1.0]]), np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]]),
                       np.array([[-1, 0, 0], [0, -1, 0], [0, 0, 1.0]]), np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]])]
            cps = np.array([R @ bp for bp in base_pts for R in quarter])
        else:
            cps = select_contact_points(cloud[hv], CP_LINK)
            if len(cps) < CP_LINK:                       # the kernels carry exactly CP_LINK points per link: repeat cyclically
                cps = np.array([cps[i % len(cps)] for i in range(CP_LINK)])
        out_bodies.append(dict(name=b["name"], parent=b["parent"], pos=b["pos"], quat=b["quat"], mass=mass,
                               ipos=ipos, inertia=inertia, contact_points=cps,
                               friction=max(g["friction"] for g in b["geoms"]),
                               margin=max(g["margin"] for g in b["geoms"])))

    # --- joints ----------------------------------------------------------------
    free = bodies[0]["joint"]
    if free is None or free.get("type") != "free":
        raise ValueError("root body must carry the free joint (quadruped.xml:63)")
    ang = (math.pi / 180.0) if degrees else 1.0
    out_joints = []
    for bi, b in enumerate(bodies[1:], start=1):
        j = b["joint"]
        if j is None or j.get("type", "hinge") != "hinge":
            raise ValueError(f"body {b['name']} must carry one hinge joint")
        rng = _floats(j["range"])
        out_joints.append(dict(name=j["name"], body=bi, axis=_floats(j.get("axis", "0 0 1")),
                               ref=float(j.get("ref", 0.0)) * ang, range=[rng[0] * ang, rng[1] * ang],
                               damping=float(j.get("damping", 0.0)), armature=float(j.get("armature", 0.0))))
    jname = {j["name"]: i for i, j in enumerate(out_joints)}

    # --- actuators ---------------------------------------------------------------
    out_act = []
    for a in actuators:
        if a.tag != "position":
            raise ValueError("only <position> actuators are supported")
        attrs = defaults.resolve(a.get("class", "main"), "position")
        attrs.update(a.attrib)
        out_act.append(dict(joint=jname[attrs["joint"]], kp=float(attrs["kp"]), kv=float(attrs.get("kv", 0.0)),
                            gear=float(attrs.get("gear", "1").split()[0]), timeconst=float(attrs.get("timeconst", 0.0)),
                            ctrlrange=_floats(attrs["ctrlrange"]), forcerange=_floats(attrs["forcerange"])))
    if [a["joint"] for a in out_act] != list(range(NJNT)):
        raise ValueError("actuator order must equal joint order (DOCS.md:349-363)")

    # --- sensors: only the layout is recorded (the pack itself is hard-wired) ----
    width = {"jointpos": 1}
    layout = []
    adr = 0
    for s in sensors:
        w = width.get(s.tag, 3)
        layout.append(dict(name=s.get("name"), type=s.tag, adr=adr, dim=w))
        adr += w

    model = dict(
        source=os.path.basename(scene_path), mesh_inertia=mesh_inertia, timestep=timestep, gravity=gravity,
        bodies=out_bodies, joints=out_joints, actuators=out_act, sensors=layout, nsensordata=adr,
        free_damping=float(free.get("damping", 0.0)), free_armature=float(free.get("armature", 0.0)),
        floor_friction=floor_friction,
    )
    _derive_soft_constraints(model)
    if keep_hulls:          # every vertex of each body's convex hull, body frame (tests: the geometry the sample points were picked from)
        model["hull_clouds"] = hull_clouds
    return model


def select_contact_points(hull_pts, k, fold=1):
    """Pick ``k`` hull vertices that best preserve the support function of the
    hull (greedy: repeatedly add the vertex with the largest support error over
    a fixed fan of directions).  With ``fold`` > 1 the choice is closed under
    the body's ``fold``-fold rotation symmetry about z (the FRAME carries its
    four hip servos at 90 degree steps, quadruped.xml:65-68).  Deterministic."""
    if len(hull_pts) <= k:
        return hull_pts.copy()
    n = 600
    i = np.arange(n) + 0.5
    phi = np.arccos(1 - 2 * i / n)
    th = math.pi * (1 + 5 ** 0.5) * i
    dirs = np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], 1)
    full = hull_pts @ dirs.T              # [nv, nd]
    target = full.max(0)

    def orbit(idx):
        out = [idx]
        for r in range(1, fold):
            a = 2 * math.pi * r / fold
            Rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
            out.append(int(np.argmin(np.linalg.norm(hull_pts - hull_pts[idx] @ Rz.T, axis=1))))
        return out

    centre = hull_pts.mean(0)
    chosen = orbit(int(np.argmax(np.linalg.norm(hull_pts - centre, axis=1))))
    while len(chosen) < k:
        err = target - full[chosen].max(0)
        cand = int(np.argmax(full[:, int(np.argmax(err))]))
        if cand in chosen:                # fall back: farthest-point sampling
            dist = np.min(np.linalg.norm(hull_pts[:, None] - hull_pts[chosen][None], axis=2), axis=1)
            cand = int(np.argmax(dist))
        for c in orbit(cand):
            if c not in chosen:
                chosen.append(c)
    return hull_pts[chosen[:k]]


def _derive_soft_constraints(model):
    """Penalty parameters of the LCP-free ground contact and joint limits.

    The reference's engine solves soft constraints whose reference dynamics are a
    critically damped spring of time constant 0.02 s scaled by an impedance
    d ~ 0.9 (Appendix A.5, A.9).  A penalty spring cannot reproduce that solver;
    its stiffness is chosen so that the static sag is comparable:
    k_total ~ m * d/(1-d) / (d*timeconst)^2.
    """
    total_mass = sum(b["mass"] for b in model["bodies"])
    kref = 1.0 / (_SOLIMP_D0 * _SOLREF_TIMECONST) ** 2
    stiff = _SOLIMP_D0 / (1.0 - _SOLIMP_D0)
    k_total = total_mass * kref * stiff                      # N/m carried by the whole robot
    model["contact"] = dict(
        stiffness=round(k_total / 12.0, 1),                  # per sample point (about 12 points share the weight)
        damping=400.0,                                       # N s/m per body in contact, treated implicitly
        margin=max(b["margin"] for b in model["bodies"]),    # quadruped.xml:8 (floor margin 0)
        friction=max(model["floor_friction"], max(b["friction"] for b in model["bodies"])),
        ramp=5.0e-4,                                         # m of summed penetration over which the damper ramps in
    )
    # joint limits: stiffness scaled by a typical joint-space inertia (armature + link)
    i_eff = 2.0e-3
    model["limit"] = dict(stiffness=round(i_eff * kref * stiff, 2), damping=round(2.0 * i_eff / (_SOLIMP_D0 * _SOLREF_TIMECONST), 4),
                          ramp=0.01)                         # rad over which the limit damper ramps in


# --------------------------------------------------------------------------- #
# emitters
# --------------------------------------------------------------------------- #
def qpos0(model):
    q = list(model["bodies"][0]["pos"]) + list(model["bodies"][0]["quat"])
    q += [j["ref"] for j in model["joints"]]
    return q


def to_jsonable(model):
    def conv(x):
        if isinstance(x, np.ndarray):
            return x.tolist()
        if isinstance(x, dict):
            return {k: conv(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [conv(v) for v in x]
        if isinstance(x, (np.floating,)):
            return float(x)
        if isinstance(x, (np.integer,)):
            return int(x)
        return x
    return conv(model)


def _c_arr(vals, fmt="%.17g"):
    return "{" + ", ".join(fmt % float(v) for v in vals) + "}"


def emit_header(model, path):
    """Write the constants as a C initialiser for ``qg_model`` (include/quadgym.h)."""
    B = model["bodies"]
    J = model["joints"]
    A = model["actuators"]
    L = []
    w = L.append
    w("/* GENERATED by quadruped-gym_amd/model/compiler.py -- do not edit.")
    w(" * Constants derived from the reference robot description")
    w(" * (src/models/quadruped/scene.xml, quadruped.xml and the OBJ meshes); mesh inertia mode: %s. */" % model["mesh_inertia"])
    w("#ifndef QG_MODEL_DATA_H")
    w("#define QG_MODEL_DATA_H")
    w("#define QG_MODEL_DEFAULT_INIT { \\")
    w("  /* timestep */ %.17g, \\" % model["timestep"])
    w("  /* gravity */ %s, \\" % _c_arr(model["gravity"]))
    w("  /* body_parent */ {%s}, \\" % ", ".join(str(b["parent"]) for b in B))
    w("  /* body_pos */ {%s}, \\" % ", ".join(_c_arr(b["pos"]) for b in B))
    w("  /* body_quat */ {%s}, \\" % ", ".join(_c_arr(b["quat"]) for b in B))
    w("  /* body_mass */ %s, \\" % _c_arr([b["mass"] for b in B]))
    w("  /* body_ipos */ {%s}, \\" % ", ".join(_c_arr(b["ipos"]) for b in B))

    def six(I):
        I = np.asarray(I)
        return [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]
    w("  /* body_inertia */ {%s}, \\" % ", ".join(_c_arr(six(b["inertia"])) for b in B))
    w("  /* jnt_axis */ {%s}, \\" % ", ".join(_c_arr(j["axis"]) for j in J))
    w("  /* jnt_ref */ %s, \\" % _c_arr([j["ref"] for j in J]))
    w("  /* jnt_range */ {%s}, \\" % ", ".join(_c_arr(j["range"]) for j in J))
    w("  /* jnt_damping */ %s, \\" % _c_arr([j["damping"] for j in J]))
    w("  /* jnt_armature */ %s, \\" % _c_arr([j["armature"] for j in J]))
    w("  /* free_damping */ %.17g, /* free_armature */ %.17g, \\" % (model["free_damping"], model["free_armature"]))
    w("  /* act_kp */ %s, \\" % _c_arr([a["kp"] for a in A]))
    w("  /* act_kv */ %s, \\" % _c_arr([a["kv"] for a in A]))
    w("  /* act_gear */ %s, \\" % _c_arr([a["gear"] for a in A]))
    w("  /* act_timeconst */ %s, \\" % _c_arr([a["timeconst"] for a in A]))
    w("  /* act_ctrlrange */ {%s}, \\" % ", ".join(_c_arr(a["ctrlrange"]) for a in A))
    w("  /* act_forcerange */ {%s}, \\" % ", ".join(_c_arr(a["forcerange"]) for a in A))
    w("  /* limit_stiffness */ %.17g, /* limit_damping */ %.17g, /* limit_ramp */ %.17g, \\"
      % (model["limit"]["stiffness"], model["limit"]["damping"], model["limit"]["ramp"]))
    w("  /* ncp */ {%s}, \\" % ", ".join(str(len(b["contact_points"])) for b in B))
    cps = []
    for b in B:
        pts = [list(p) for p in np.asarray(b["contact_points"])]
        while len(pts) < MAXCP:
            pts.append([0.0, 0.0, 0.0])
        cps.append("{" + ", ".join(_c_arr(p) for p in pts) + "}")
    w("  /* cp */ {%s}, \\" % ", \\\n    ".join(cps))
    c = model["contact"]
    w("  /* contact_stiffness */ %.17g, /* contact_damping */ %.17g, /* contact_margin */ %.17g, /* contact_friction */ %.17g, /* contact_ramp */ %.17g, \\"
      % (c["stiffness"], c["damping"], c["margin"], c["friction"], c["ramp"]))
    w("  /* qpos0 */ %s \\" % _c_arr(qpos0(model)))
    w("}")
    w("#endif")
    with open(path, "w") as fh:
        fh.write("\n".join(L) + "\n")


def main(argv=None):
    import argparse
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(os.path.dirname(here))
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--scene", default="/root/reference/src/models/quadruped/scene.xml")
    ap.add_argument("--mesh-inertia", default="convex", choices=["convex", "exact"])
    ap.add_argument("--json", default=os.path.join(here, "quadruped_model.json"))
    ap.add_argument("--header", default=os.path.join(repo, "include", "qg_model_data.h"))
    args = ap.parse_args(argv)
    model = compile_mjcf(args.scene, args.mesh_inertia)
    with open(args.json, "w") as fh:
        json.dump(to_jsonable(model), fh, indent=1)
    emit_header(model, args.header)
    tot = sum(b["mass"] for b in model["bodies"])
    print(f"compiled {args.scene}: {len(model['bodies'])} bodies, mass {tot:.4f} kg -> {args.json}, {args.header}")


if __name__ == "__main__":
    main()
