"""Generate tests/golden/walking_reference.npz by IMPORTING the reference's own pure-NumPy modules
(/root/reference/src/envs/math_utils.py and control_inputs.py -- the only reference code that imports in
this container) and driving them with seeded inputs.  The fixture holds inputs and the
reference's outputs only (data, no source).  It pins the restatements in oracle/walking_oracle.py and,
through them, the walking-reward kernels.  Run here only: the reference does not exist on the GPU box.

    python tools/make_walking_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/envs"
sys.path.insert(0, REF)
import control_inputs as ref_ci      # noqa: E402  (reference module)
import math_utils as ref_mu          # noqa: E402  (reference module)


def signals(rng, steps, dt):
    """12 channels covering the estimator's branches: sinusoids of several frequencies / amplitudes,
    a constant channel (zero derivative from the start), a plateau-then-sine channel (zero derivative
    after a sign is known), a clipped sine (repeated samples), noise, and a sign-flipping ramp."""
    t = np.arange(steps) * dt
    x = np.zeros((steps, 12))
    x[:, 0] = 0.5 * np.sin(2 * np.pi * 1.0 * t)
    x[:, 1] = 0.9 * np.sin(2 * np.pi * 2.5 * t + 0.3)
    x[:, 2] = -0.5                                              # constant
    x[:, 3] = np.where(t < 0.8, 0.2, 0.2 + 0.4 * np.sin(2 * np.pi * 3.0 * (t - 0.8)))
    x[:, 4] = np.clip(1.4 * np.sin(2 * np.pi * 1.5 * t), -1, 1)    # saturating: runs of equal samples
    x[:, 5] = rng.uniform(-1, 1, steps)                          # white noise
    x[:, 6] = 0.3 * np.sin(2 * np.pi * 0.5 * t) + 0.05 * np.sin(2 * np.pi * 9.0 * t)
    x[:, 7] = np.round(np.sin(2 * np.pi * 0.7 * t) * 4) / 4      # quantised
    x[:, 8] = 0.8 * np.sign(np.sin(2 * np.pi * 1.2 * t))         # square wave
    x[:, 9] = (t % 0.5) - 0.25                                   # sawtooth
    x[:, 10] = 0.0
    x[:, 11] = 0.6 * np.cos(2 * np.pi * 4.0 * t) * np.exp(-t / 3)
    return x.astype(np.float32).astype(np.float64)               # exactly representable in f32


def main():
    rng = np.random.default_rng(2025)
    out = {}
    # --- OnlineFrequencyAmplitudeEstimation (walking_quad.py:54-59: min_freq=1, ema_alpha=0.8, dt = h*frame_skip)
    for fs in (4, 10, 20):
        dt = 0.002 * fs
        est = ref_mu.OnlineFrequencyAmplitudeEstimation(n_channels=12, dt=dt, min_freq=1, ema_alpha=0.80)
        steps = 3 * est.window_size + 17
        x = signals(rng, steps, dt)
        f = np.zeros((steps, 12)); a = np.zeros((steps, 12))
        for k in range(steps):
            f[k], a[k] = est.update(x[k])
        out[f"est_fs{fs}_window"] = np.array(est.window_size)
        out[f"est_fs{fs}_x"] = x
        out[f"est_fs{fs}_f"] = f
        out[f"est_fs{fs}_a"] = a
    # --- exp_dist / unit
    v = rng.normal(size=(64, 2))
    v[0] = [0.0, 0.0]                                            # unit() of a zero vector -> NaN (math_utils.py:7-8)
    with np.errstate(all="ignore"):
        out["unit_in"] = v
        out["unit_out"] = np.array([ref_mu.unit(r) for r in v])
    e = np.linspace(-3, 2, 41)
    out["exp_dist_in"] = e
    out["exp_dist_out"] = ref_mu.exp_dist(e)
    # --- VelocityHeadingControls setters
    sp = rng.uniform(0, 1, 32); al = rng.uniform(-np.pi, np.pi, 32); th = rng.uniform(-np.pi, np.pi, 32)
    vel = np.zeros((32, 3)); head = np.zeros((32, 3)); gv = np.zeros((32, 3))
    for i in range(32):
        c = ref_ci.VelocityHeadingControls()
        c.set_orientation(th[i])
        c.set_velocity_speed_alpha(sp[i], al[i])
        vel[i], head[i], gv[i] = c.velocity, c.heading, c.global_velocity
    out.update(ci_speed=sp, ci_alpha=al, ci_theta=th, ci_velocity=vel, ci_heading=head, ci_global_velocity=gv)
    # --- sample(options): which RNG draws are consumed, in which order (global NumPy RNG)
    for name, opts in [("train", {"fixed_heading_angle": 0.0, "fixed_velocity_angle": 0.0, "fixed_speed": 0.3}),
                       ("free", {}), ("speed", {"min_speed": 0.1, "max_speed": 0.4, "fixed_heading_angle": 0.5})]:
        np.random.seed(77)
        c = ref_ci.VelocityHeadingControls()
        rows = []
        for _ in range(6):
            c.sample(options=opts)
            rows.append(np.r_[c.velocity, c.heading, c.global_velocity])
        out[f"sample_{name}"] = np.array(rows)
    path = os.path.join(ROOT, "tests", "golden", "walking_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; windows", [int(out[f'est_fs{fs}_window']) for fs in (4, 10, 20)])


if __name__ == "__main__":
    main()
