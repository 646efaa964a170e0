"""The walking-task oracle (oracle/walking_oracle.py) against fixtures produced by the REFERENCE's own
code (math_utils.py, control_inputs.py imported by tools/make_walking_fixtures.py)."""
import os

import numpy as np
import pytest

from oracle import walking_oracle as W

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "walking_reference.npz")


@pytest.fixture(scope="module")
def ref():
    return dict(np.load(GOLD))


@pytest.mark.parametrize("fs", [4, 10, 20])
def test_estimator_matches_reference(ref, fs):
    dt = 0.002 * fs
    assert W.window_size(dt, 1.0) == int(ref[f"est_fs{fs}_window"]) == {4: 250, 10: 100, 20: 50}[fs]
    x = ref[f"est_fs{fs}_x"]
    est = W.FreqAmpEstimator(2, dt)                     # two envs fed the same signal
    for k in range(len(x)):
        f, a = est.update(np.stack([x[k], x[k]]))
        assert np.array_equal(f[0], ref[f"est_fs{fs}_f"][k]) and np.array_equal(f[1], f[0])
        assert np.array_equal(a[0], ref[f"est_fs{fs}_a"][k])


def test_unit_and_exp_dist_match_reference(ref):
    out = W.unit(ref["unit_in"])
    assert np.isnan(out[0]).all() and np.isnan(ref["unit_out"][0]).all()      # zero vector -> NaN, as in the reference
    assert np.allclose(out[1:], ref["unit_out"][1:], rtol=4e-16, atol=0)     # norm() of a row vs of a batch: 1 ulp
    assert np.array_equal(W.exp_dist(ref["exp_dist_in"]), ref["exp_dist_out"])


def test_controls_match_reference(ref):
    n = len(ref["ci_speed"])
    c = W.Controls(n)
    for i in range(n):
        c.set_orientation(i, ref["ci_theta"][i])
        c.set_velocity_speed_alpha(i, ref["ci_speed"][i], ref["ci_alpha"][i])
    assert np.array_equal(c.velocity, ref["ci_velocity"]) and np.array_equal(c.heading, ref["ci_heading"])
    assert np.array_equal(c.global_velocity, ref["ci_global_velocity"])


@pytest.mark.parametrize("name,opts", [("train", {"fixed_heading_angle": 0.0, "fixed_velocity_angle": 0.0, "fixed_speed": 0.3}),
                                       ("free", {}), ("speed", {"min_speed": 0.1, "max_speed": 0.4, "fixed_heading_angle": 0.5})])
def test_sampler_consumes_the_rng_like_the_reference(ref, name, opts):
    np.random.seed(77)
    c = W.Controls(1)
    for row in ref[f"sample_{name}"]:
        c.sample(0, opts)
        assert np.array_equal(np.r_[c.velocity[0], c.heading[0], c.global_velocity[0]], row)


def test_reward_stack_bookkeeping():
    """Quirks of walking_quad.py kept on purpose: previous_ctrl_cost is set once and never updated (:266-270),
    the derived term is zero on the first step after a reset (:388-393), the estimator sees the PREVIOUS ctrl."""
    n, dt = 3, 0.008
    o = W.WalkingOracle(n, dt, settling_time=0.02)
    for i in range(n):
        o.controls.sample(i, {"fixed_heading_angle": 0.0, "fixed_velocity_angle": 0.0, "fixed_speed": 0.3})
    rng = np.random.default_rng(0)
    sens = np.zeros((n, 33)); sens[:, 29] = 1.0; sens[:, 24] = 1.0; sens[:, 20] = 0.12; sens[:, 30] = 0.1
    a = rng.uniform(-1, 1, (n, 12))
    act = o.pre_step(np.array([0.0, 0.016, 0.03]), np.tile([0, 0, -0.5] * 4, (n, 1)), a)
    assert np.array_equal(act[0], [0, 0, -0.5] * 4) and np.array_equal(act[2], a[2])     # settling mask by data.time
    tot, comps, flip = o.post_step(sens, act)
    assert comps.shape == (n, 11) and np.allclose(tot, comps.sum(1)) and not flip.any()
    assert comps[:, 0] == pytest.approx(10.0) and (comps[:, 10] == 0).all()
    c0 = o.prev_ctrl_cost.copy()
    act2 = o.pre_step(np.full(n, 1.0), act, rng.uniform(-1, 1, (n, 12)))
    o.post_step(sens, act2)
    assert np.array_equal(o.prev_ctrl_cost, c0)
    o.reset(mask=[True, False, False])
    assert np.isnan(o.prev_derive[0]) and not np.isnan(o.prev_derive[1]) and (o.ideal[0] == 0).all()


def test_unit_of_a_zero_vector_is_nan_unless_the_option_says_zero():
    """math_utils.unit() (math_utils.py:7-8) divides by a zero norm: NaN in the direction term and in the total
    (walking_quad.py:197-205,422) -- kept by default.  ``unit_zero=True`` is the product's qg_walk_params.unit_zero: the direction
    term is 0 where either norm is exactly 0, every other component and every other env is untouched."""
    sens = np.zeros((3, 33)); sens[:, 29] = 1.0; sens[:, 24] = 1.0; sens[:, 20] = 0.12
    sens[0, 30:32] = [0.2, -0.1]          # env 0: moving body, zero command
    sens[2, 30:32] = [0.2, -0.1]          # env 2: moving body, command below; env 1: body at rest, command below
    out = {}
    for uz in (False, True):
        o = W.WalkingOracle(3, 0.008, unit_zero=uz)
        o.controls.velocity[1:, :2] = [0.3, 0.1]
        out[uz] = o.post_step(sens, np.tile([0, 0, -0.5] * 4, (3, 1)))
    (t0, c0, _), (t1, c1, _) = out[False], out[True]
    assert np.isnan(c0[:2, 2]).all() and np.isnan(t0[:2]).all() and np.isfinite(t0[2])
    assert (c1[:2, 2] == 0).all() and np.isfinite(t1).all()
    others = np.delete(np.arange(11), 2)
    assert np.array_equal(c0[:, others], c1[:, others]) and c0[2, 2] == c1[2, 2] and t0[2] == t1[2]
    assert np.allclose(t1, c1.sum(1))
