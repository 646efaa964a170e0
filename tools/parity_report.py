"""Quantifies HIP-vs-oracle agreement on a larger sample than the test suite holds: N seeded states visited by random-action
rollouts (tools/make_golden.sample_states), one env-step (frame_skip 4 and 20) from each through every work mapping and through
the f64 oracle, error percentiles per output.  The oracle is the checker (test infrastructure), the HIP path is what is
measured.  usage: python tools/parity_report.py [N] > profiles/<round>/parity_report.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401
from make_golden import sample_states
from oracle import oracle as O
from quadruped_gym_amd import _abi
from quadruped_gym_amd.sim import BatchedSim

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
model = O.default_model()
print(f"# HIP (f32) vs oracle (f64): one env-step from {n} seeded states; abs error percentiles [50 %, 99 %, max]; "
      f"accelerometer excluded from `obs` (reported separately)")
for fs in (4, 20):
    otask = O.default_task(); otask.frame_skip = fs; otask.use_fall = 1; otask.fall_height = 0.05
    task = _abi.default_task(); task.frame_skip = fs; task.use_fall = 1; task.fall_height = 0.05
    qpos, qvel, act, nstep = sample_states(model, otask, n, seed=2024 + fs)
    actions = np.random.default_rng(fs).uniform(-1, 1, (n, 12)).astype(np.float32)
    b = O.Batch(model, otask, n)
    b.set_state(qpos.astype(np.float64), qvel.astype(np.float64), act.astype(np.float64), None, nstep)
    obs_o, rew_o, done_o, _ = b.step(actions.astype(np.float64))
    q_o, v_o, a_o, _, _ = b.get_state()
    airborne = int((q_o[:, 2] > 0.25).sum())
    print(f"\n## frame_skip {fs}: {n} states ({airborne} airborne after the step, the rest in ground contact); |qvel| up to {np.abs(v_o).max():.1f}")
    for name, mp in (("lane", _abi.MAP_LANE), ("quad", _abi.MAP_QUAD), ("pair", _abi.MAP_PAIR), ("link", _abi.MAP_LINK)):
        sim = BatchedSim(n, task=task)
        sim.set_mapping(mp)
        sim.set_state(qpos, qvel, act, None, nstep)
        obs, rew, done, _ = sim.step(actions)
        q1, v1, a1, _, _ = sim.get_state()
        sim.close()
        mask = np.ones(33, bool); mask[12:15] = False

        def pct(x):
            x = np.abs(x).ravel()
            return f"[{np.quantile(x, 0.5):.1e}, {np.quantile(x, 0.99):.1e}, {x.max():.1e}]"
        sure = np.abs(q_o[:, 2] - 0.05) > 1e-4
        print(f"{name:5s} qpos {pct(q1 - q_o)}  qvel {pct(v1 - v_o)}  act {pct(a1 - a_o)}  obs {pct(obs[:, mask] - obs_o[:, mask])}  "
              f"accel {pct(obs[:, 12:15] - obs_o[:, 12:15])}  reward {pct(rew - rew_o)}  done mismatches {int((done[sure] != done_o[sure]).sum())}")
