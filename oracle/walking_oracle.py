"""CPU restatement (NumPy, f64, batched over envs) of the reference's walking task layer:
``WalkingQuadrupedEnv.step/reset`` and its reward stack (``src/envs/walking_quad.py:96-428``), the
command inputs (``src/envs/control_inputs.py``) and ``exp_dist`` / ``unit`` /
``OnlineFrequencyAmplitudeEstimation`` (``src/envs/math_utils.py``).

TEST INFRASTRUCTURE ONLY -- the checker for the walking-reward kernels; the product never imports it.
Pinned against the reference's own code for everything that imports offline: the estimator, the command
setters / sampler and exp_dist / unit are checked against ``tests/golden/walking_reference.npz``, which
``tools/make_walking_fixtures.py`` produced by importing the reference modules.  The reward formulas
themselves live in ``walking_quad.py``, which cannot be imported here (it needs mujoco): for those the
parity is UNPINNED and this file follows the source text line by line (cited below).
"""
from __future__ import annotations

import numpy as np

REWARD_KEYS = ["alive_bonus", "control_cost", "progress_direction_reward_local", "progress_speed_cost_local",
               "heading_reward", "orientation_reward", "body_height_cost", "joint_posture_cost",
               "control_amplitude_cost", "control_frequency_cost", "diff_ideal_position_cost"]   # walking_quad.py:332-351

# sensordata addresses (quadruped.xml:174-217, DOCS.md:365-400)
S_POS, S_LINVEL, S_XAXIS, S_ZAXIS, S_VEL = 18, 21, 24, 27, 30


def default_params():
    """Constants hard-coded in walking_quad.py (line numbers in the comments)."""
    return dict(
        joint_centers=np.array([0.0, 0.0, -0.5] * 4),                   # :36-39
        ema_alpha=0.8, min_freq=1.0,                                     # :54-59
        control_cost_alpha=0.8,                                          # :254
        w=np.array([10.0, -2.0, 10.0, -50.0, 10.0, 10.0, -50.0, -1.0, -2.5, -8.0]),   # :362-373
        w_diff_ideal=-20.0,                                              # :383
        body_height=0.13,                                                # :369
        amp_target=np.array([1.5, 0.5, 0.0] * 4),                        # :282
        freq_target=np.array([1.0, 1.0, 0.0] * 4),                       # :275
        nu=12,
    )


def exp_dist(x):                       # math_utils.py:4-5
    return np.exp(x) - 1.0


def unit(x):                           # math_utils.py:7-8 (zero vector -> NaN, kept)
    with np.errstate(all="ignore"):
        return x / np.linalg.norm(x, axis=-1, keepdims=True)


def window_size(dt, min_freq):         # math_utils.py:26-28
    return int(np.ceil(2 / (min_freq * dt)))


class FreqAmpEstimator:
    """math_utils.py:11-158 for n envs x 12 channels at once."""

    def __init__(self, n, dt, min_freq=1.0, ema_alpha=0.8, channels=12):
        self.n, self.c, self.dt, self.alpha = n, channels, dt, ema_alpha
        self.W = window_size(dt, min_freq)
        self.cross = np.zeros((n, self.W, channels), dtype=np.int64)
        self.sig = np.zeros((n, self.W, channels))
        self.idx = np.zeros(n, dtype=np.int64)
        self.count = np.zeros((n, channels), dtype=np.int64)
        self.samples = np.zeros(n, dtype=np.int64)
        self.prev = np.zeros((n, channels))
        self.has_prev = np.zeros(n, dtype=bool)
        self.sign = np.zeros((n, channels))
        self.has_sign = np.zeros(n, dtype=bool)
        self.f = np.zeros((n, channels))
        self.a = np.zeros((n, channels))

    def update(self, x):
        x = np.asarray(x, dtype=np.float64)
        for e in range(self.n):
            self._update_one(e, x[e])
        return self.f.copy(), self.a.copy()

    def _update_one(self, e, x):
        if not self.has_prev[e]:                       # :66-72 first call: store and return zeros
            self.prev[e] = x
            self.has_prev[e] = True
            self.sig[e, self.idx[e]] = x
            self.samples[e] = 1
            self.idx[e] = (self.idx[e] + 1) % self.W
            return
        cur = np.sign(x - self.prev[e])                # :75-76
        if self.has_sign[e]:                            # :78-80 zero derivative keeps the previous sign
            z = cur == 0
            cur[z] = self.sign[e][z]
            crossing = (cur != self.sign[e]).astype(np.int64)     # :83-84
        else:
            crossing = np.zeros(self.c, dtype=np.int64)            # :86
        if self.samples[e] < self.W:                    # :89-90
            self.samples[e] += 1
        i = self.idx[e]
        self.count[e] -= self.cross[e, i]               # :94-96
        self.cross[e, i] = crossing
        self.count[e] += crossing
        self.sig[e, i] = x                              # :99
        self.idx[e] = (i + 1) % self.W                  # :102
        self.prev[e] = x                                # :105-106
        self.sign[e] = cur
        self.has_sign[e] = True
        dur = self.samples[e] * self.dt                 # :109
        f_cur = (self.count[e] / 2.0) / dur             # :113-114
        self.f[e] = self.alpha * self.f[e] + (1 - self.alpha) * f_cur          # :117
        win = self.sig[e, :self.samples[e]] if self.samples[e] < self.W else self.sig[e]   # :121-124
        amp = win.max(axis=0) - win.min(axis=0)         # :126
        self.a[e] = self.alpha * self.a[e] + (1 - self.alpha) * amp            # :129


class Controls:
    """control_inputs.py for n envs: velocity (local), heading (unit vector), global_velocity."""

    def __init__(self, n):
        self.velocity = np.zeros((n, 3))
        self.heading = np.zeros((n, 3))
        self.global_velocity = np.zeros((n, 3))

    def _update(self, i):                               # control_inputs.py:14-27
        v0, v1 = self.velocity[i, 0], self.velocity[i, 1]
        h0, h1 = self.heading[i, 0], self.heading[i, 1]
        self.global_velocity[i] = [h0 * v0 - h1 * v1, h1 * v0 + h0 * v1, 0.0]

    def set_orientation(self, i, theta):                # :45-51
        self.heading[i, 0], self.heading[i, 1] = np.cos(theta), np.sin(theta)
        self._update(i)

    def set_velocity_speed_alpha(self, i, speed, alpha):   # :37-43
        self.velocity[i, 0], self.velocity[i, 1] = speed * np.cos(alpha), speed * np.sin(alpha)
        self._update(i)

    def sample(self, i, options=None, uniform=None):    # :74-115; `uniform(lo, hi)` stands in for np.random.uniform
        uniform = uniform or np.random.uniform
        options = options or {}
        lo, hi = options.get("min_speed", 0.0), options.get("max_speed", 1.0)
        th = options.get("fixed_heading_angle")
        theta = th if th is not None else uniform(-np.pi, np.pi)
        self.set_orientation(i, theta)
        al = options.get("fixed_velocity_angle")
        alpha = al if al is not None else uniform(-np.pi, np.pi)
        sp = options.get("fixed_speed")
        speed = sp if sp is not None else uniform(lo, hi)
        self.set_velocity_speed_alpha(i, speed, alpha)


def sample_keyed(controls, i, options, seed, env_index, counter):
    """``Controls.sample`` (control_inputs.py:74-115) with the batch path's counter-based draws in place of the global NumPy
    RNG: env ``env_index`` in episode ``counter`` takes heading angle, velocity angle and speed from streams 13, 14, 15 of
    its key (a fixed quantity leaves its stream unused)."""
    from oracle import oracle as _o

    def u(stream):
        return _o.lib().qgo_uniform_stream(int(seed), int(env_index), int(counter), stream)
    options = options or {}
    lo, hi = options.get("min_speed", 0.0), options.get("max_speed", 1.0)
    th = options.get("fixed_heading_angle")
    theta = th if th is not None else -np.pi + 2 * np.pi * u(13)
    controls.set_orientation(i, theta)
    al = options.get("fixed_velocity_angle")
    alpha = al if al is not None else -np.pi + 2 * np.pi * u(14)
    sp = options.get("fixed_speed")
    speed = sp if sp is not None else lo + (hi - lo) * u(15)
    controls.set_velocity_speed_alpha(i, speed, alpha)


class WalkingOracle:
    """The task layer around the physics step, for n envs (walking_quad.py:96-148,352-428).

    Per env-step, in the reference's order:  pre_step() [ideal position integrates the commanded global
    velocity :93,133; the estimator takes data.ctrl = the PREVIOUS applied action :136; the action is
    replaced by the joint centres while data.time < settling_time :142-143]  ->  physics step  ->
    post_step() [input_control_reward on the step's sensordata and data.ctrl :352-421; flip termination
    :156-160]."""

    def __init__(self, n, dt, settling_time=0.0, params=None, unit_zero=False):
        # unit_zero: qg_walk_params.unit_zero of the product -- the direction term is 0 where either norm is exactly 0 instead of the
        # reference's NaN (math_utils.py:7-8); False = the reference
        self.unit_zero = bool(unit_zero)
        self.p = params or default_params()
        self.n, self.dt, self.settling = n, dt, settling_time
        self.controls = Controls(n)
        self.est = FreqAmpEstimator(n, dt, self.p["min_freq"], self.p["ema_alpha"])
        self.ideal = np.zeros((n, 3))
        self.prev_ctrl = np.tile(self.p["joint_centers"], (n, 1))
        self.prev_ctrl_cost = np.full(n, np.nan)        # None until the first call, then never updated (:266-270)
        self.prev_derive = np.full(n, np.nan)           # None after every reset (:109,388-390)
        self.f_est = np.zeros((n, 12))
        self.a_est = np.zeros((n, 12))

    def reset(self, mask=None):                          # walking_quad.py:96-126 (the estimator is NOT reset, :115)
        m = np.ones(self.n, bool) if mask is None else np.asarray(mask, bool)
        self.ideal[m] = 0.0
        self.prev_ctrl[m] = self.p["joint_centers"]
        self.prev_derive[m] = np.nan

    def pre_step(self, time, data_ctrl, action):
        self.ideal += self.controls.global_velocity * self.dt                 # :93,133
        self.f_est, self.a_est = self.est.update(data_ctrl)                   # :136
        act = np.array(action, dtype=np.float64, copy=True)
        settle = np.asarray(time) < self.settling                             # :142-143
        act[settle] = self.p["joint_centers"]
        return act

    def post_step(self, sens, ctrl):
        p = self.p
        sens = np.asarray(sens, np.float64)
        ctrl = np.asarray(ctrl, np.float64)
        pos, xax, zax, vel = sens[:, S_POS:S_POS + 3], sens[:, S_XAXIS:S_XAXIS + 3], sens[:, S_ZAXIS:S_ZAXIS + 3], sens[:, S_VEL:S_VEL + 3]
        cv = self.controls.velocity
        # control_cost :254-270
        diff = ctrl - self.prev_ctrl
        self.prev_ctrl = ctrl.copy()
        cost = np.sum(diff * diff, axis=1)
        first = np.isnan(self.prev_ctrl_cost)
        self.prev_ctrl_cost[first] = cost[first]
        control_cost = p["control_cost_alpha"] * self.prev_ctrl_cost + (1 - p["control_cost_alpha"]) * cost
        with np.errstate(all="ignore"):
            direction = np.sum(unit(vel[:, :2]) * unit(cv[:, :2]), axis=1)                      # :197-201
        if self.unit_zero:
            zero = (np.linalg.norm(vel[:, :2], axis=1) == 0) | (np.linalg.norm(cv[:, :2], axis=1) == 0)
            direction = np.where(zero, 0.0, direction)
        speed_cost = (np.linalg.norm(vel[:, :2], axis=1) - np.linalg.norm(cv[:, :2], axis=1)) ** 2   # :212-218
        heading = np.sum(xax[:, :2] * self.controls.heading[:, :2], axis=1)                     # :231-235
        orientation = zax[:, 2]                                                                  # :237-241
        height = np.abs(pos[:, 2] - p["body_height"])                                            # :243-247
        posture = np.linalg.norm((ctrl - p["joint_centers"]) / p["nu"], axis=1)                  # :249-253
        amp = np.linalg.norm((self.a_est - p["amp_target"].astype(np.float32)) / p["nu"], axis=1)     # :279-285
        freq = np.linalg.norm((self.f_est - p["freq_target"].astype(np.float32)) / p["nu"], axis=1)   # :272-277
        w = p["w"]
        values = np.stack([w[0] * np.ones(self.n), w[1] * control_cost, w[2] * direction, w[3] * speed_cost,
                           w[4] * exp_dist(heading), w[5] * exp_dist(orientation), w[6] * exp_dist(height),
                           w[7] * posture, w[8] * amp, w[9] * freq], axis=1)                     # :362-373
        ideal_cost = np.linalg.norm(pos[:, :2] - self.ideal[:, :2], axis=1)                      # :164-171
        derive = p["w_diff_ideal"] * ideal_cost                                                  # :383
        none = np.isnan(self.prev_derive)
        self.prev_derive[none] = derive[none]                                                    # :388-390
        derived = (derive - self.prev_derive) / self.dt                                          # :393
        self.prev_derive = derive.copy()                                                         # :396
        comps = np.concatenate([values, derived[:, None]], axis=1)
        total = comps.sum(axis=1)                                                                # :422
        flip = zax[:, 2] < 0                                                                     # :156-160
        return total, comps, flip
