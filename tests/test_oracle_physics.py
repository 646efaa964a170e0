"""Known-answer tests that pin the CPU oracle (oracle/qg_oracle.c).

The reference (antopio26/quadruped-gym) holds no tests or golden vectors for its
physics and its engine (`mujoco`, unpinned) is not available offline, so the
oracle is pinned by physics identities instead: the parity
of the oracle against MuJoCo itself is UNPINNED.
"""
import math
import os

import numpy as np
import pytest

from helpers import NQ, NV, NU, make_conservative, quat_mul, random_state, rotz


def test_struct_sizes_and_defaults(oracle, model, task):
    assert model.timestep == 0.002                       # engine default; quadruped.xml:4 sets none
    assert abs(sum(model.body_mass) - 1.110) < 1e-12      # robot = 1.110 kg
    q0 = np.array(model.qpos0[:])
    assert np.allclose(q0[:7], [0, 0, 0.13, 1, 0, 0, 0])  # quadruped.xml:62
    assert np.allclose(q0[7:10], np.deg2rad([-45, 37.5, 0]))   # ref of hip / knee / ankle, quadruped.xml:25,30,35
    assert np.allclose(task.default_ctrl[:], [0, 0, -0.5] * 4)  # quadruped.py:124
    assert list(model.ncp) == [12] + [8] * 12


def test_time_limit_matches_f64_accumulation(oracle):
    # `data.time >= max_time` with time accumulated in f64 (quadruped.py:149-151)
    for h, tmax in [(0.002, 10.0), (0.002, 20.0), (0.002, 1.0), (0.002, 0.3)]:
        t, n = 0.0, 0
        while not t >= tmax:
            t += h
            n += 1
        assert oracle.time_limit_substeps(h, tmax) == n
    assert oracle.time_limit_substeps(0.002, 10.0) == 5000
    assert oracle.time_limit_substeps(0.002, 20.0) == 10001   # accumulation lands just below 20.0 at 10000
    assert oracle.time_limit_substeps(0.002, 1e9) == 2 ** 31 - 1   # effectively no limit: answered without looping


def test_mass_matrix_spd_and_kinetic_energy(oracle, model):
    rng = np.random.default_rng(0)
    for _ in range(20):
        qpos, qvel = random_state(rng, model)
        M = oracle.mass_matrix(model, qpos)
        assert np.abs(M - M.T).max() == 0.0
        assert np.linalg.eigvalsh(M).min() > 1e-4
        T, _ = oracle.energy(model, qpos, qvel)           # body-by-body sum, independent of the CRBA code
        assert abs(0.5 * qvel @ M @ qvel - T) < 1e-12 * max(1.0, T)
        # total mass on the translational block (+ armature), no coupling between world axes
        assert np.allclose(M[:3, :3], (1.110 + model.free_armature) * np.eye(3), atol=1e-12)


def test_rne_is_consistent_with_crba(oracle, model):
    # inverse dynamics: RNE(q, v, a) - RNE(q, v, 0) == (M - armature) a
    rng = np.random.default_rng(1)
    for _ in range(20):
        qpos, qvel = random_state(rng, model)
        qacc = rng.normal(size=NV) * 10
        M = oracle.mass_matrix(model, qpos)
        arm = np.array([model.free_armature] * 6 + list(model.jnt_armature))
        lhs = oracle.rne(model, qpos, qvel, qacc) - oracle.rne(model, qpos, qvel, None)
        assert np.allclose(lhs, (M - np.diag(arm)) @ qacc, rtol=1e-10, atol=1e-10)


def test_bias_is_gravity_at_rest(oracle, model):
    # at zero velocity the bias force is minus the gradient of the potential energy
    rng = np.random.default_rng(2)
    qpos, _ = random_state(rng, model)
    c = oracle.rne(model, qpos, np.zeros(NV), None)
    assert np.allclose(c[:3], [0, 0, 1.110 * 9.81], atol=1e-12)   # weight on the translational DoFs
    eps = 1e-6
    for j in range(12):                                            # hinge DoFs: dV/dq by central differences
        qp, qm = qpos.copy(), qpos.copy()
        qp[7 + j] += eps
        qm[7 + j] -= eps
        dV = (oracle.energy(model, qp, np.zeros(NV))[1] - oracle.energy(model, qm, np.zeros(NV))[1]) / (2 * eps)
        assert abs(c[6 + j] - dV) < 1e-8


def _run_free(oracle, model, qpos, qvel, nsteps):
    e = oracle.make_env(qpos, qvel)
    for _ in range(nsteps):
        oracle.substep(model, e, np.zeros(NU))
    return np.array(e.qpos[:]), np.array(e.qvel[:])


def test_energy_conserved_without_dissipation(oracle, model):
    # no damping, no servos, no limits, far above the floor: E = T + V is an invariant of the
    # continuous dynamics; semi-implicit Euler drifts O(h), so the drift must shrink with h
    make_conservative(model)
    rng = np.random.default_rng(3)
    qpos, qvel = random_state(rng, model, z=50.0, vel=0.7)
    drifts = []
    for h, n in [(4e-4, 250), (1e-4, 1000)]:
        model.timestep = h
        T0, V0 = oracle.energy(model, qpos, qvel)
        q1, v1 = _run_free(oracle, model, qpos, qvel, n)
        T1, V1 = oracle.energy(model, q1, v1)
        drifts.append(abs((T1 + V1) - (T0 + V0)) / (abs(T0) + 1e-9))
    assert drifts[1] < 2e-3
    assert drifts[1] < 0.5 * drifts[0]


def test_momentum_conserved_in_free_flight(oracle, model):
    # zero gravity, no dissipation: linear and angular momentum are conserved.  Armature is a
    # reflected rotor inertia that does not belong to rigid-body momentum, so it is removed too.
    make_conservative(model)
    model.gravity[2] = 0.0
    model.free_armature = 0.0
    for j in range(12):
        model.jnt_armature[j] = 0.0
    model.timestep = 1e-4
    rng = np.random.default_rng(4)
    qpos, qvel = random_state(rng, model, z=50.0, vel=0.5)
    p0, L0 = oracle.momentum(model, qpos, qvel)
    q1, v1 = _run_free(oracle, model, qpos, qvel, 500)
    p1, L1 = oracle.momentum(model, q1, v1)
    assert np.allclose(p1, p0, atol=1e-9)
    assert np.allclose(L1, L0, atol=2e-4 * max(1.0, np.abs(L0).max()))


def test_exact_discrete_free_fall(oracle, model):
    # no damping, servos off: every body falls with g, nothing moves relative to the base and
    # semi-implicit Euler gives v_n = -g h n, z_n = z0 - g h^2 n (n + 1) / 2 exactly
    make_conservative(model)
    model.free_armature = 0.0
    q0 = np.array(model.qpos0[:])
    q0[2] = 30.0
    e = oracle.make_env(q0)
    h, g, n = model.timestep, 9.81, 400
    for _ in range(n):
        oracle.substep(model, e, np.zeros(NU))
    assert abs(e.qvel[2] + g * h * n) < 1e-9
    assert abs(e.qpos[2] - (30.0 - g * h * h * n * (n + 1) / 2)) < 1e-9
    assert np.allclose(np.array(e.qpos[7:]), q0[7:], atol=1e-9)
    assert np.allclose(np.array(e.qpos[3:7]), [1, 0, 0, 0], atol=1e-12)


def test_free_joint_armature_slows_free_fall(oracle, model):
    # the free joint inherits armature 0.001 through childclass (quadruped.xml:9,62-63): the base
    # translational DoFs carry m + 0.001 and the robot falls slightly slower than g (the legs, which
    # would fall with g, drag on their armatured hinges, so the value is not simply g m / (m + 0.001))
    make_conservative(model)
    q0 = np.array(model.qpos0[:])
    q0[2] = 30.0
    e = oracle.make_env(q0)
    _, dg = oracle.substep(model, e, np.zeros(NU), want_diag=True)
    assert -9.81 + 1e-3 < dg.qacc[2] < -9.81 * 1.110 / 1.111 + 0.02
    model.free_armature = 0.0
    e = oracle.make_env(q0)
    _, dg = oracle.substep(model, e, np.zeros(NU), want_diag=True)
    assert dg.qacc[2] == pytest.approx(-9.81, rel=1e-12)


def test_quaternion_stays_normalised_and_integrates_body_rates(oracle, model):
    # hinges frozen by a huge armature: the robot is one rigid body, and a torque-free spin about the
    # body z axis (a principal axis by 4-fold symmetry) keeps its rate; yaw advances by h*w per substep
    make_conservative(model)
    model.gravity[2] = 0.0
    for j in range(12):
        model.jnt_armature[j] = 1e9
    q0 = np.array(model.qpos0[:])
    q0[2] = 10.0
    v0 = np.zeros(NV)
    v0[5] = 2.0
    e = oracle.make_env(q0, v0)
    n = 300
    for _ in range(n):
        oracle.substep(model, e, np.zeros(NU))
    q = np.array(e.qpos[3:7])
    assert abs(np.linalg.norm(q) - 1) < 1e-14
    yaw = 2 * math.atan2(q[3], q[0])
    assert abs(yaw - 2.0 * model.timestep * n) < 1e-6
    assert abs(e.qvel[5] - 2.0) < 1e-6


def test_servo_filter_closed_form(oracle, model):
    # filterexact activation (quadruped.xml:14, timeconst 0.01): act_n = u (1 - exp(-h/tau)^n), with the
    # ctrl clamped to the servo's ctrlrange (hip 0.5, knee 0.91, ankle 1; quadruped.xml:26,31,36)
    q0 = np.array(model.qpos0[:])
    q0[2] = 10.0
    e = oracle.make_env(q0)
    ctrl = np.array([0.9, -1.0, 0.7] * 4)
    n = 7
    for _ in range(n):
        oracle.substep(model, e, ctrl)
    u = np.clip(ctrl, [-0.5, -0.91, -1.0] * 4, [0.5, 0.91, 1.0] * 4)
    expect = u * (1 - math.exp(-0.002 / 0.01) ** n)
    assert np.allclose(np.array(e.act[:]), expect, rtol=1e-12)
    assert abs(1 - math.exp(-0.002 / 0.01) - 0.18127) < 1e-5


def test_servo_force_uses_pre_update_activation_and_clamps(oracle, model):
    q0 = np.array(model.qpos0[:])
    q0[2] = 10.0
    act = np.linspace(-0.5, 0.5, 12)
    qvel = np.zeros(NV)
    qvel[6:] = np.linspace(-1, 1, 12)
    e = oracle.make_env(q0, qvel, act)
    _, dg = oracle.substep(model, e, np.zeros(NU), want_diag=True)
    g, kp, kv = 0.64, 100.0, 1.0
    raw = kp * act - kp * g * q0[7:] - kv * g * qvel[6:]
    force = np.clip(raw, -1.71, 1.71)
    assert np.allclose(np.array(dg.act_force[:]), force, atol=1e-12)
    assert np.allclose(np.array(dg.f_act[6:]), g * force, atol=1e-12)
    assert np.abs(np.array(dg.f_act[6:])).max() <= 0.64 * 1.71 + 1e-12   # <= 1.094 N m
    # implicit matrix: damping + kv g^2 on unclamped servos only
    A, M = np.array(dg.A[:]).reshape(NV, NV), np.array(dg.M[:]).reshape(NV, NV)
    add = np.diag(A - M) / model.timestep
    clamped = np.abs(raw) >= 1.71
    assert np.allclose(add[:6], 0.2)
    assert np.allclose(add[6:], 0.2 + np.where(clamped, 0.0, kv * g * g), atol=1e-9)


def test_fourfold_symmetry(oracle, model):
    # rotating the world by 90 degrees about z and relabelling the legs (leg k -> k+1) maps
    # trajectories onto trajectories (legs are 90-degree copies, quadruped.xml:71,89,107,125).
    # Off the floor the contact model plays no part; on the floor it must be equivariant too.
    rng = np.random.default_rng(5)
    for z in (5.0, 0.05):
        qpos, qvel = random_state(rng, model, z=z, vel=0.3)
        yaw = rng.uniform(0, 2 * np.pi)
        qpos[3:7] = [np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]   # near-upright so z = 0.05 is in contact
        act = rng.uniform(-0.3, 0.3, 12)
        ctrl = rng.uniform(-1, 1, 12)
        Rz = rotz(np.pi / 2)
        qz = np.array([np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])
        perm = np.r_[3:12, 0:3]                         # new leg j is old leg j+1: body frame turned by +90 deg
        qpos2, qvel2 = qpos.copy(), qvel.copy()
        qpos2[3:7] = quat_mul(qpos[3:7], qz)           # body frame turned by +90 deg about its own z ...
        qpos2[7:] = qpos[7:][perm]                      # ... and the legs relabelled: same physical robot
        qvel2[3:6] = Rz.T @ qvel[3:6]
        qvel2[6:] = qvel[6:][perm]
        e1 = oracle.make_env(qpos, qvel, act)
        e2 = oracle.make_env(qpos2, qvel2, act[perm])
        for _ in range(5):
            oracle.substep(model, e1, ctrl)
            oracle.substep(model, e2, ctrl[perm])
        assert np.allclose(np.array(e2.qpos[:3]), np.array(e1.qpos[:3]), atol=1e-9)
        assert np.allclose(np.array(e2.qvel[:3]), np.array(e1.qvel[:3]), atol=1e-8)
        assert np.allclose(np.array(e2.qpos[7:]), np.array(e1.qpos[7:])[perm], atol=1e-9)
        assert np.allclose(np.array(e2.qvel[6:]), np.array(e1.qvel[6:])[perm], atol=1e-7)
        assert np.allclose(np.array(e2.qvel[3:6]), Rz.T @ np.array(e1.qvel[3:6]), atol=1e-7)


def test_joint_limit_pushes_back(oracle, model):
    q0 = np.array(model.qpos0[:])
    q0[2] = 10.0
    q0[7] = model.jnt_range[0][0] - 0.05     # hip 1 below its lower limit
    q0[8] = model.jnt_range[1][1] + 0.05     # knee 1 above its upper limit
    e = oracle.make_env(q0)
    _, dg = oracle.substep(model, e, np.zeros(NU), want_diag=True)
    assert dg.f_limit[6] == pytest.approx(model.limit_stiffness * 0.05)
    assert dg.f_limit[7] == pytest.approx(-model.limit_stiffness * 0.05)
    assert all(dg.f_limit[d] == 0 for d in range(8, NV))


def test_contact_static_equilibrium_and_no_adhesion(oracle, model, task):
    # settle from the reset pose: the spring forces must carry the weight, the robot must come to rest
    e = oracle.reset(model, task)
    ctrl = np.array(task.default_ctrl[:])
    for _ in range(2500):
        _, dg = oracle.substep(model, e, ctrl, want_diag=True)
    assert np.abs(np.array(e.qvel[:])).max() < 1e-3
    Fz = sum(dg.contact_F[b][2] for b in range(13))
    assert abs(Fz - 1.110 * 9.81) < 1e-3
    assert all(dg.contact_F[b][2] >= 0 for b in range(13))          # no adhesion
    # penetration stays in the sub-millimetre range of the margin
    xpos, xmat, _ = oracle.kinematics(model, np.array(e.qpos[:]))
    zmin = min((xpos[b] + xmat[b] @ np.array(model.cp[b][i][:]))[2] for b in range(13) for i in range(model.ncp[b]))
    assert -1e-3 < zmin < model.contact_margin


def test_friction_is_coulomb_limited_and_stops_sliding(oracle, model, task):
    e = oracle.reset(model, task)
    ctrl = np.array(task.default_ctrl[:])
    for _ in range(1500):
        oracle.substep(model, e, ctrl)
    e.qvel[0] = 1.0                                         # shove the resting robot sideways
    _, dg = oracle.substep(model, e, ctrl, want_diag=True)
    for b in range(13):
        F = np.array(dg.contact_F[b][:])
        assert np.hypot(F[0], F[1]) <= model.contact_friction * F[2] + 1e-9
    for _ in range(1500):
        oracle.substep(model, e, ctrl)
    assert abs(e.qvel[0]) < 1e-3                            # friction has stopped it
    assert 0.0 < e.qpos[0] < 0.2                            # after sliding about v^2 / (2 mu g) = 5 cm


def test_random_actions_stay_bounded(oracle, model, task):
    # 10 simulated seconds of random actuation from reset: no blow-up, no sinking through the floor
    rng = np.random.default_rng(7)
    e = oracle.reset(model, task)
    zmin, vmax = 1.0, 0.0
    for _ in range(1250):
        a = rng.uniform(-1, 1, 12)
        obs, r, d, c = oracle.step(model, task, e, a)
        zmin = min(zmin, e.qpos[2])
        vmax = max(vmax, np.abs(np.array(e.qvel[:])).max())
    assert d and e.nstep == 5000                            # time limit fires on the 1250th env-step, as `terminated`
    assert zmin > 0.02 and vmax < 60.0
    assert np.isfinite(obs).all()


def test_step_contract(oracle, model, task):
    # quadruped.py:153-182: clip to +-1, frame_skip substeps, obs = lagged sensordata, rewards on the new state
    e = oracle.reset(model, task)
    a = np.array([2.0, -3.0, 0.5] * 4)
    obs, r, d, comps = oracle.step(model, task, e, a)
    assert e.nstep == 4 and not d
    assert np.allclose(np.array(e.ctrl[:]), np.clip(a, -1, 1))
    assert comps[1] == pytest.approx(-0.1 * np.sum(np.clip(a, -1, 1) ** 2))
    assert comps[0] == pytest.approx(e.qvel[0]) and comps[2] == 1.0 and r == pytest.approx(comps.sum())
    # lag: obs describes the state at the start of the 4th substep
    e2 = oracle.reset(model, task)
    for _ in range(3):
        oracle.substep(model, e2, np.clip(a, -1, 1))
    assert np.allclose(obs[:12], np.array(e2.qpos[7:]))
    assert np.allclose(obs[18:21], np.array(e2.qpos[:3])) and np.allclose(obs[21:24], np.array(e2.qvel[:3]))
    assert np.allclose(obs[15:18], np.array(e2.qvel[3:6]))
    # fall termination with the README's literal threshold fires at once (base starts at 0.13 < 0.2)
    task.use_fall = 1
    e3 = oracle.reset(model, task)
    assert oracle.step(model, task, e3, a)[2] is True
    task.fall_height = 0.05
    e4 = oracle.reset(model, task)
    assert oracle.step(model, task, e4, a)[2] is False


def test_sensor_pack_layout(oracle, model, task):
    # DOCS.md:365-400: axes and velocimeter against an independent rotation
    rng = np.random.default_rng(8)
    qpos, qvel = random_state(rng, model, z=3.0)
    e = oracle.make_env(qpos, qvel)
    sens, dg = oracle.substep(model, e, np.zeros(NU), want_sensors=True, want_diag=True)
    w, x, y, z = qpos[3:7]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    assert np.allclose(sens[24:27], R[:, 0]) and np.allclose(sens[27:30], R[:, 2])
    assert np.allclose(sens[30:33], R.T @ qvel[:3])
    assert np.allclose(sens[12:15], R.T @ (np.array(dg.qacc[:3]) + [0, 0, 9.81]))
    task.obs_mode = 1
    e = oracle.make_env(qpos, qvel)
    obs = oracle.step(model, task, e, np.zeros(NU))[0]
    assert obs.shape == (21,)


def test_random_yaw_reset(oracle, model, task):
    u = [oracle.uniform(123, i, 0) for i in range(2000)]
    assert 0.0 <= min(u) and max(u) < 1.0 and abs(np.mean(u) - 0.5) < 0.03
    assert all(float(np.float32(v)) == v for v in u)              # 24-bit: exact in f32
    e = oracle.reset(model, task, seed=123, env_index=5, counter=2, flags=1)
    a = 2 * np.pi * oracle.uniform(123, 5, 2)
    assert np.allclose(np.array(e.qpos[3:7]), [np.cos(a / 2), 0, 0, np.sin(a / 2)])   # walking_quad.py:73-75


def test_envelope_of_the_reference_notebook_run(oracle, model, task):
    """The one recorded output of the real engine in the reference: the joint-angle plot stored in `src/quadruped_model.ipynb`
    (cell 3) of a 10 s run from reset in which `data.ctrl` is redrawn from U(-1, 1) every 0.1 s (cell 2; NumPy's global RNG,
    unseeded, so only statistics are comparable).  Digitised from that plot (axes calibrated on its gridlines, 6.8 mrad and
    11.8 ms per pixel; the eight uniquely coloured curves, 65-80 % of each visible): the joint sensors start at
    (-0.785, 0.65, 0); hips stay within [-0.784, +0.805] (their +-45 deg range is also the servo's reach 0.5 / 0.64); knees peak
    at 1.43 (= 0.91 / 0.64, the ctrlrange clamp, reached without overshoot) and dip to -0.84 (55 mrad through the -45 deg limit);
    ankles swing to +1.46 / -1.40 of their +-1.5625 reach; the fastest sustained motion over 0.1 s is 5.4-5.8 rad/s (hips),
    6.5 (knees), 5.9-6.3 (ankles) -- the torque-limited slew of the servo (0.64 * 1.71 N m against 0.2 N m s/rad of joint damping
    and the load).  The same protocol through the oracle must give the same envelope and slew rates; this is a loose pin
    (one unseeded run, read off a picture), not bit-level parity."""
    lo = np.array([+9.0] * 3)
    hi = np.array([-9.0] * 3)
    rate = np.zeros(3)
    for seed in range(3):
        rng = np.random.default_rng(seed)
        e = oracle.reset(model, task)
        ctrl = np.array([0.0, 0.0, -0.5] * 4)
        count, time = 0, 0.0
        traj = np.empty((5000, 12))
        for s in range(5000):
            sens = oracle.substep(model, e, ctrl, want_sensors=True)
            traj[s] = np.array(sens[0][:12] if isinstance(sens, tuple) else sens[:12])
            time += 0.002
            if count < time * 10:                      # notebook cell 2: a new random command whenever count < time * 10
                ctrl = rng.uniform(-1, 1, 12)
                count += 1
        assert np.allclose(traj[0].reshape(4, 3), [[-0.7854, 0.6545, 0.0]] * 4, atol=2e-3)
        for j in range(3):
            q = traj[:, j::3]
            lo[j], hi[j] = min(lo[j], q.min()), max(hi[j], q.max())
            rate[j] = max(rate[j], (np.abs(q[50:] - q[:-50]) / 0.1).max())
    assert -0.83 < lo[0] < -0.76 and 0.76 < hi[0] < 0.83           # hips: plateaus at the +-45 deg limit
    assert 1.25 < hi[1] < 1.47 and -0.90 < lo[1] < -0.75           # knees: 1.42 reach, shallow dip through the lower limit
    assert 1.25 < hi[2] < 1.60 and -1.60 < lo[2] < -1.25           # ankles: most of the +-1.5625 reach, inside the +-90 deg range
    assert (rate > 5.0).all() and (rate < 7.8).all()               # torque-limited slew: 5.4-6.5 rad/s in the plot's visible parts


def _lowest_point_below_frame(oracle, model, hinge, points=None):
    """Distance from the FRAME origin down to the lowest point of the robot standing level with its hinges at `hinge`:
    over the model's contact sample points (vertices of the bodies' convex hulls), or over `points` = {body: [k, 3] array}."""
    qpos = np.array(model.qpos0[:], float)
    qpos[0:3] = 0.0
    qpos[3:7] = [1, 0, 0, 0]
    qpos[7:19] = hinge
    xpos, xmat, _ = oracle.kinematics(model, qpos)
    low = 0.0
    for b in range(13):
        pts = np.array([model.cp[b][i][:] for i in range(model.ncp[b])]) if points is None else points[b]
        low = min(low, float((xpos[b] + pts @ xmat[b].T)[:, 2].min()))
    return -low


def test_standing_height_at_the_joint_centre_command_follows_the_geometry(oracle, model, task):
    """SURVEY 8(c) names one loose standing-height anchor in the reference: `body_height_cost(0.13)` and the comment "0.12 is the
    default height" (src/envs/walking_quad.py:243-247,369).  It cannot pin anything: it is not the height the reference's own robot
    stands at.  At the joint-centre command [0, 0, -0.5] x 4 (walking_quad.py:36-39, = quadruped.py:124's default ctrl) the position
    servos hold the hinges at ctrl / gear = (0, 0, -0.78125) rad, and the XML + OBJ geometry then puts the lowest foot hull point
    0.1426 m below the FRAME origin -- 10 to 20 % above the reference's 0.13 / 0.12.  (At qpos0's folded pose it is 0.033 m: the
    0.13 m of quadruped.xml:62 is a drop height, not a stance.)  What CAN be pinned is that the oracle agrees with that geometry:
    it settles with the FRAME origin at the geometric height minus the sub-millimetre contact penetration and servo sag."""
    gear = np.array(model.act_gear[:])
    target = np.clip(np.array(task.default_ctrl[:]), np.array(model.act_ctrlrange[:])[:, 0], np.array(model.act_ctrlrange[:])[:, 1]) / gear
    assert np.allclose(target, [0, 0, -0.78125] * 4)
    h_geom = _lowest_point_below_frame(oracle, model, target)
    assert abs(h_geom - 0.1426) < 5e-4                              # the committed sample points hold the hull's lowest vertex
    assert abs(_lowest_point_below_frame(oracle, model, np.array(model.qpos0[7:19])) - 0.0329) < 5e-4    # the folded reset pose
    e = oracle.reset(model, task)
    ctrl = np.array(task.default_ctrl[:])
    for _ in range(4000):
        oracle.substep(model, e, ctrl)
    q = np.array(e.qpos[:])
    assert np.abs(np.array(e.qvel[:])).max() < 1e-6
    assert np.allclose(q[7:19], target, atol=4e-3)                  # the servos hold the hinges at ctrl / gear (sag < 3 mrad)
    h_pose = _lowest_point_below_frame(oracle, model, q[7:19])      # the geometry at the pose actually held
    assert 0.0 < h_pose - q[2] < 1e-3                               # the feet press < 1 mm into the contact margin
    assert abs(q[2] - 0.1423) < 3e-4 and abs(q[2] - h_geom) < 1e-3
    assert q[2] - 0.13 > 0.012 and q[2] - 0.12 > 0.022              # walking_quad.py:243-247,369: not this robot's stance height


REF_MODEL_DIR = "/root/reference/src/models/quadruped"


@pytest.mark.skipif(not os.path.isdir(REF_MODEL_DIR), reason="reference checkout not present on this host")
def test_standing_height_from_the_reference_meshes(oracle, model, task):
    """The same geometric height from the reference's own files: every vertex of the convex hulls of the OBJ meshes, placed by the
    XML's geom frames (model/compiler.py), instead of the eight sample points per body the model ships with."""
    from quadruped_gym_amd.model import compiler as MC
    full = MC.compile_mjcf(os.path.join(REF_MODEL_DIR, "scene.xml"), keep_hulls=True)
    clouds = full.get("hull_clouds")
    assert clouds is not None and len(clouds) == 13
    h_full = _lowest_point_below_frame(oracle, model, np.array([0, 0, -0.78125] * 4), {b: np.asarray(clouds[b]) for b in range(13)})
    assert abs(h_full - 0.1426) < 3e-4
    assert abs(h_full - _lowest_point_below_frame(oracle, model, np.array([0, 0, -0.78125] * 4))) < 2e-4
