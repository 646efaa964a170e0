"""Turn a compiled model dictionary (``compiler.compile_mjcf`` / ``quadruped_model.json``) into
the ``qg_model`` struct of the C ABI."""
from __future__ import annotations

import json
import os

import numpy as np

from .._abi import MAXCP, NBODY, NJNT, QgModel
from . import compiler


def model_from_dict(d) -> QgModel:
    m = QgModel()
    m.timestep = float(d["timestep"])
    m.gravity[:] = [float(x) for x in d["gravity"]]
    B = d["bodies"]
    if len(B) != NBODY or len(d["joints"]) != NJNT or len(d["actuators"]) != NJNT:
        raise ValueError("the kernels serve the fixed 13-body / 12-hinge quadruped topology only")
    for b, body in enumerate(B):
        m.body_parent[b] = int(body["parent"])
        m.body_pos[b][:] = [float(x) for x in body["pos"]]
        m.body_quat[b][:] = [float(x) for x in body["quat"]]
        m.body_mass[b] = float(body["mass"])
        m.body_ipos[b][:] = [float(x) for x in body["ipos"]]
        I = np.asarray(body["inertia"], float)
        m.body_inertia[b][:] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]
        pts = np.asarray(body["contact_points"], float)
        if len(pts) > MAXCP:
            raise ValueError("too many contact points")
        m.ncp[b] = len(pts)
        for i, p in enumerate(pts):
            m.cp[b][i][:] = [float(x) for x in p]
    for j, jn in enumerate(d["joints"]):
        m.jnt_axis[j][:] = [float(x) for x in jn["axis"]]
        m.jnt_ref[j] = float(jn["ref"])
        m.jnt_range[j][:] = [float(x) for x in jn["range"]]
        m.jnt_damping[j] = float(jn["damping"])
        m.jnt_armature[j] = float(jn["armature"])
    m.free_damping = float(d["free_damping"])
    m.free_armature = float(d["free_armature"])
    for i, a in enumerate(d["actuators"]):
        m.act_kp[i], m.act_kv[i], m.act_gear[i] = float(a["kp"]), float(a["kv"]), float(a["gear"])
        m.act_timeconst[i] = float(a["timeconst"])
        m.act_ctrlrange[i][:] = [float(x) for x in a["ctrlrange"]]
        m.act_forcerange[i][:] = [float(x) for x in a["forcerange"]]
    m.limit_stiffness, m.limit_damping, m.limit_ramp = (float(d["limit"][k]) for k in ("stiffness", "damping", "ramp"))
    c = d["contact"]
    m.contact_stiffness, m.contact_damping = float(c["stiffness"]), float(c["damping"])
    m.contact_margin, m.contact_friction, m.contact_ramp = float(c["margin"]), float(c["friction"]), float(c["ramp"])
    m.qpos0[:] = [float(x) for x in compiler.qpos0(d)]
    return m


def load_model(model_path: str | None):
    """``model_path`` as ``QuadrupedEnv`` receives it (``src/envs/quadruped.py:41,55-59``).

    * ``None`` / ``"builtin"``: the constants compiled into the library from the reference model;
    * an MJCF file (the reference's ``scene.xml``): compiled on the spot by ``compiler.compile_mjcf``
      (raises ``FileNotFoundError`` for a missing path, as the reference does);
    * a ``.json`` file written by the compiler.
    Returns ``(QgModel, dict)`` -- the dict carries the sensor layout (names / addresses)."""
    here = os.path.dirname(os.path.abspath(__file__))
    if model_path is None or model_path == "builtin":
        with open(os.path.join(here, "quadruped_model.json")) as fh:
            d = json.load(fh)
        from .. import _abi
        return _abi.default_model(), d
    if not os.path.exists(model_path):
        raise FileNotFoundError(f"Model file not found: {model_path}")
    if model_path.endswith(".json"):
        with open(model_path) as fh:
            d = json.load(fh)
    else:
        d = compiler.to_jsonable(compiler.compile_mjcf(model_path))
    return model_from_dict(d), d
