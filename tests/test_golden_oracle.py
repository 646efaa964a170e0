"""The committed golden vectors (tests/golden/step_vectors.npz, written by tools/make_golden.py)
must be reproduced exactly by the CPU oracle -- this pins the oracle against silent changes and
keeps the fixture honest.  (The reference holds no golden vectors for this path)"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_vectors.npz")


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


def configure(task, case):
    task.use_fall = 1
    task.fall_height = 0.05
    if case == "B":
        task.frame_skip = 20
        task.obs_mode = 1
    return task


@pytest.mark.parametrize("case", ["A", "B"])
def test_oracle_reproduces_golden(oracle, model, task, gold, case):
    configure(task, case)
    n = len(gold["qpos"])
    b = oracle.Batch(model, task, n)
    b.set_state(gold["qpos"].astype(np.float64), gold["qvel"].astype(np.float64), gold["act"].astype(np.float64), None,
                gold["nstep"])
    obs, rew, done, comps = b.step(gold["actions"].astype(np.float64))
    q1, v1, a1, c1, n1 = b.get_state()
    assert np.array_equal(done, gold[case + "_done"])
    assert np.array_equal(n1, gold[case + "_nstep1"])
    for got, key in [(obs, "obs"), (rew, "reward"), (comps, "comps"), (q1, "qpos1"), (v1, "qvel1"), (a1, "act1"), (c1, "ctrl1")]:
        assert np.allclose(got, gold[case + "_" + key], rtol=1e-12, atol=1e-12), key


def test_golden_covers_the_interesting_regimes(gold, model):
    z = gold["qpos"][:, 2]
    assert (z < 0.12).sum() > 20 and (z > 0.3).sum() >= 10            # on the floor and airborne
    assert np.abs(gold["actions"]).max() > 1.0                        # exercises the +-1 clip (quadruped.py:160)
    lo = np.array([model.jnt_range[j][0] for j in range(12)]); hi = np.array([model.jnt_range[j][1] for j in range(12)])
    q = gold["qpos"][:, 7:]
    assert ((q < lo) | (q > hi)).any()                                # some joints beyond their soft limits
    assert gold["A_done"].any() and not gold["A_done"].all()
