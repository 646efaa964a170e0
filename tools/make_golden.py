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
