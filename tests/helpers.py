"""Shared helpers for the test-suite (state sampling, model variants)."""
import numpy as np

NQ, NV, NU = 19, 18, 12


def random_state(rng, model, z=1.0, vel=1.0, joint_margin=0.05):
    """A random state well inside the joint ranges, base at height z."""
    qpos = np.zeros(NQ)
    qpos[0:2] = rng.uniform(-0.5, 0.5, 2)
    qpos[2] = z
    q = rng.normal(size=4)
    qpos[3:7] = q / np.linalg.norm(q)
    for j in range(12):
        lo, hi = model.jnt_range[j][0], model.jnt_range[j][1]
        qpos[7 + j] = rng.uniform(lo + joint_margin, hi - joint_margin)
    qvel = vel * rng.normal(size=NV)
    qvel[6:] *= 3.0
    return qpos, qvel


def make_conservative(model):
    """Strip every dissipative / driven term: what is left is a free-floating tree under gravity."""
    model.free_damping = 0.0
    for j in range(12):
        model.jnt_damping[j] = 0.0
        model.act_kp[j] = 0.0
        model.act_kv[j] = 0.0
        model.jnt_range[j][0] = -100.0
        model.jnt_range[j][1] = 100.0
    return model


def rotz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])
