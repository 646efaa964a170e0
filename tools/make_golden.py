"""Generate tests/golden/step_vectors.npz: seeded inputs (state + action, rounded to f32) and the
CPU oracle's outputs after one env-step, for the parity tests of the HIP path.

The reference itself cannot produce these vectors here (its engine, `mujoco`, is not installed
and not installable offline), so they come from the oracle (oracle/qg_oracle.c),
which is pinned by tests/test_oracle_physics.py.  Run:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def sample_states(model, task, n, seed):
    """States visited by random-action rollouts from reset (free flight, landing, crawling, joint
    limits), snapshot at random times, plus a few synthetic airborne tumbling states."""
    rng = np.random.default_rng(seed)
    qpos, qvel, act, nstep = [], [], [], []
    for i in range(n):
        e = O.reset(model, task, seed=seed, env_index=i, counter=0, flags=1)
        if i % 8 == 7:       # airborne, tumbling, joints anywhere in range
            q = rng.normal(size=4)
            e.qpos[2] = rng.uniform(0.3, 1.0)
            e.qpos[3:7] = list(q / np.linalg.norm(q))
            for j in range(12):
                e.qpos[7 + j] = rng.uniform(model.jnt_range[j][0], model.jnt_range[j][1])
            e.qvel[:] = list(rng.normal(size=18) * np.r_[np.ones(3), 3 * np.ones(3), 8 * np.ones(12)])
            e.act[:] = list(rng.uniform(-0.5, 0.5, 12))
            steps = 0
        else:
            steps = int(rng.integers(0, 400))
        hold = rng.uniform(-1, 1, 12)
        for s in range(steps):
            if s % 10 == 0:
                hold = rng.uniform(-1.2, 1.2, 12)
            O.step(model, task, e, hold)
        qpos.append(np.array(e.qpos[:])); qvel.append(np.array(e.qvel[:])); act.append(np.array(e.act[:])); nstep.append(e.nstep)
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    return f32(qpos), f32(qvel), f32(act), np.asarray(nstep, np.int32)


def contact_states(model, n, seed, bodies):
    """States in which one of `bodies` (indices into the model's body list: 0 = FRAME, 1/4/7/10 = the femurs) presses on the
    floor -- the paths the kernels' wave-uniform contact skips guard (the FRAME block and, on large grids, the femur block),
    which random-action rollouts practically never visit because the robot stands on its feet.  The base is tilted (every
    fourth state lies on its back), the hinges are anywhere in range, and the height is set so that the lowest sample point
    of the chosen bodies is 0..4 mm below the contact margin while no body is deeper than 6 mm; moderate velocities."""
    rng = np.random.default_rng(seed)
    cps = [np.array([[model.cp[b][i][d] for d in range(3)] for i in range(model.ncp[b])]) for b in range(13)]
    out_q, out_v, out_a = [], [], []
    while len(out_q) < n:
        k = len(out_q)
        roll, pitch, yaw = rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6), rng.uniform(-np.pi, np.pi)
        if k % 4 == 3:
            roll += np.pi                                  # on its back: the top of the FRAME / the femurs touch
        cr, sr, cp_, sp, cy, sy = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw / 2), np.sin(yaw / 2)
        quat = np.array([cr * cp_ * cy + sr * sp * sy, sr * cp_ * cy - cr * sp * sy, cr * sp * cy + sr * cp_ * sy, cr * cp_ * sy - sr * sp * cy])
        qpos = np.zeros(19)
        qpos[0:2] = rng.uniform(-1, 1, 2)
        qpos[2] = 1.0
        qpos[3:7] = quat
        for j in range(12):
            lo, hi = model.jnt_range[j][0], model.jnt_range[j][1]
            qpos[7 + j] = rng.uniform(lo, hi)
        xpos, xmat, _ = O.kinematics(model, qpos)
        zmin = np.array([(xpos[b][2] + cps[b] @ xmat[b][2]).min() for b in range(13)])      # lowest sample point of every body
        target = zmin[bodies].min()
        depth = rng.uniform(0.0, 0.004)
        shift = (model.contact_margin - depth) - target
        if (model.contact_margin - (zmin + shift)).max() > 0.006:      # some other body would be pressed in deeper: draw again
            continue
        qpos[2] += shift
        out_q.append(qpos)
        out_v.append(rng.normal(size=18) * np.r_[0.2 * np.ones(3), 1.0 * np.ones(3), 3.0 * np.ones(12)])
        out_a.append(rng.uniform(-0.5, 0.5, 12))
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    return f32(out_q), f32(out_v), f32(out_a), rng.integers(0, 2000, n).astype(np.int32)


def contact_census(model, qpos, qvel, act, nstep, actions):
    """How many of the states have the FRAME / a femur / a shin / a foot in contact at the start of the step (oracle diagnostics)."""
    frame = femur = shin = foot = 0
    for i in range(len(qpos)):
        e = O.make_env(qpos[i], qvel[i], act[i], nstep=int(nstep[i]))
        _, dg = O.substep(model, e, np.clip(actions[i], -1, 1), want_diag=True)
        W = np.array(dg.contact_W[:])
        frame += W[0] > 0; femur += (W[[1, 4, 7, 10]] > 0).any(); shin += (W[[2, 5, 8, 11]] > 0).any(); foot += (W[[3, 6, 9, 12]] > 0).any()
    return dict(frame=int(frame), femur=int(femur), shin=int(shin), foot=int(foot), states=len(qpos))


def run_case(model, task, qpos, qvel, act, nstep, actions):
    n = len(qpos)
    b = O.Batch(model, task, n)
    b.set_state(qpos.astype(np.float64), qvel.astype(np.float64), act.astype(np.float64), None, nstep)
    obs, rew, done, comps = b.step(actions.astype(np.float64))
    q1, v1, a1, c1, n1 = b.get_state()
    return dict(obs=obs, reward=rew, done=done, comps=comps, qpos1=q1, qvel1=v1, act1=a1, ctrl1=c1, nstep1=n1)


def main():
    model, task = O.default_model(), O.default_task()
    n = 96
    qpos, qvel, act, nstep = sample_states(model, task, n, seed=20251004)
    rng = np.random.default_rng(7)
    actions = rng.uniform(-1.3, 1.3, (n, 12)).astype(np.float32)      # some beyond the +-1 clip
    # round 2: 32 states with the FRAME on the floor and 32 with a femur on the floor (appended: the first 96 are unchanged)
    extra = [contact_states(model, 32, seed=411, bodies=[0]), contact_states(model, 32, seed=412, bodies=[1, 4, 7, 10])]
    qpos = np.concatenate([qpos] + [x[0] for x in extra]); qvel = np.concatenate([qvel] + [x[1] for x in extra])
    act = np.concatenate([act] + [x[2] for x in extra]); nstep = np.concatenate([nstep] + [x[3] for x in extra])
    actions = np.concatenate([actions, np.random.default_rng(8).uniform(-1.3, 1.3, (64, 12)).astype(np.float32)])
    n = len(qpos)
    print("contact census:", contact_census(model, qpos, qvel, act, nstep, actions))
    out = dict(qpos=qpos, qvel=qvel, act=act, nstep=nstep, actions=actions)
    # case A: BASELINE config 2 -- frame_skip 4, full 33-sensor obs, fall termination (threshold 0.05)
    task.use_fall = 1
    task.fall_height = 0.05
    for k, v in run_case(model, task, qpos, qvel, act, nstep, actions).items():
        out["A_" + k] = v
    # case B: BASELINE config 5 -- frame_skip 20, 21-value IMU + joint pack
    task.frame_skip = 20
    task.obs_mode = 1
    for k, v in run_case(model, task, qpos, qvel, act, nstep, actions).items():
        out["B_" + k] = v
    path = os.path.join(ROOT, "tests", "golden", "step_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", "done A:", int(out["A_done"].sum()), "min z:", float(qpos[:, 2].min()))


if __name__ == "__main__":
    main()
