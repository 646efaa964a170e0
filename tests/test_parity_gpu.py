"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden
vectors.  Floating point: the kernels compute in f32, the oracle in f64; tolerances are stated
per quantity below and hold for ONE env-step from identical f32 states (longer rollouts diverge
chaotically in any precision and are compared through invariants instead)."""
import os

import numpy as np
import pytest

from quadruped_gym_amd import _abi

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_vectors.npz")

# ---- stated tolerances: |gpu - oracle| <= atol + rtol * |oracle| after one env-step ----------
# The absolute terms are <= 5x the largest error MEASURED over 4096 seeded states per frame_skip and every work mapping
# (tools/parity_report.py -> profiles/r02/parity_report.txt; frame_skip 4: qpos 1.2e-6, qvel 4.3e-4, act 8.2e-8, obs 7.5e-5,
# accelerometer 5.3e-3 m/s^2, reward 3.5e-6), so a regression that costs an order of magnitude of accuracy fails.  frame_skip
# 20 compounds 20 substeps and a contact can switch inside the step (measured maxima qpos 1.5e-4, qvel 7.7e-2, obs 5.8e-4,
# accelerometer 0.85, reward 3.0e-4).
TOL = {
    "A": dict(qpos=(5e-6, 2e-6), qvel=(2e-3, 1e-4), act=(4e-7, 1e-6), obs=(3.5e-4, 1e-4), accel=(0.025, 1e-3), reward=(1.5e-5, 1e-5)),
    "B": dict(qpos=(4e-4, 1e-4), qvel=(5e-2, 2e-2), act=(6e-7, 1e-6), obs=(3e-3, 2e-3), accel=(2.0, 5e-2), reward=(1.5e-3, 1e-3)),
}
# frame_skip 20, accelerometer (round 4): the 2.0 m/s^2 above is set by the handful of states where a contact switches inside the
# step -- a body touches down or lifts off at a slightly different substep in f32 and the reading of the last substep jumps.  States
# whose contact census (the oracle's contact_W > 0 per body) does NOT change around the substep the sensors describe (the last one and
# two before it, one after: oracle.contact_switch_near_sensors) are held to this bound instead; the loose one only covers the rest.
ACCEL_B_STEADY = (0.01, 1e-3)        # measured: 4.2e-4 on the golden states (every mapping), 1.7e-3 on config 5's 4096-env sample


def close_accel_b(oracle, got, ref, model, frame_skip, state, actions, keep=None):
    qpos, qvel, act, nstep = state
    changed = oracle.contact_switch_near_sensors(model, frame_skip, qpos, qvel, act, nstep, actions)
    if keep is not None:
        got, ref, changed = got[keep], ref[keep], changed[keep]
    steady = ~changed
    assert steady.sum() >= 20, "enough states keep their contacts around the sensor substep"
    worst = close(got[steady], ref[steady], ACCEL_B_STEADY, "accelerometer (contact census steady)")
    if changed.any():
        close(got[changed], ref[changed], TOL["B"]["accel"], "accelerometer (a contact switches inside the step)")
    return worst, int(changed.sum())


def configure(task, case):
    task.use_fall = 1
    task.fall_height = 0.05
    if case == "B":
        task.frame_skip = 20
        task.obs_mode = 1
    return task


def close(got, ref, tol, what):
    atol, rtol = tol
    err = np.abs(np.asarray(got, np.float64) - ref)
    bound = atol + rtol * np.abs(ref)
    worst = np.unravel_index(np.argmax(err - bound), err.shape)
    assert (err <= bound).all(), f"{what}: max excess at {worst}: got {np.asarray(got)[worst]}, ref {ref[worst]}, err {err[worst]:.3e}"
    return float(err.max())


def accel_slice(case):
    return slice(12, 15)


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(GOLD))


MAPPINGS = {"lane": _abi.MAP_LANE, "quad": _abi.MAP_QUAD, "pair": _abi.MAP_PAIR, "link": _abi.MAP_LINK}


@pytest.mark.parametrize("mapping", ["lane", "quad", "pair", "link"])
@pytest.mark.parametrize("case", ["A", "B"])
def test_step_matches_golden_vectors(oracle, gold, case, mapping):
    """Every work mapping of the step kernel (one env per lane / one leg per lane / two legs per lane) against the fixture."""
    from quadruped_gym_amd.sim import BatchedSim
    task = configure(_abi.default_task(), case)
    n = len(gold["qpos"])
    sim = BatchedSim(n, task=task)
    sim.set_mapping(MAPPINGS[mapping])
    assert sim.mapping == MAPPINGS[mapping]
    sim.set_state(gold["qpos"], gold["qvel"], gold["act"], None, gold["nstep"])
    obs, rew, done, comps = sim.step(gold["actions"], want_components=True)
    q1, v1, a1, c1, n1 = sim.get_state()
    t = TOL[case]
    g = lambda k: gold[case + "_" + k]
    close(q1, g("qpos1"), t["qpos"], "qpos")
    close(v1, g("qvel1"), t["qvel"], "qvel")
    close(a1, g("act1"), t["act"], "act")
    assert np.array_equal(n1, g("nstep1"))
    assert np.array_equal(c1, g("ctrl1").astype(np.float32))       # data.ctrl = clip(action, -1, 1): exact
    sl = accel_slice(case)
    mask = np.ones(obs.shape[1], bool)
    mask[sl] = False
    close(obs[:, mask], g("obs")[:, mask], t["obs"], "obs")
    if case == "B":
        worst, nsw = close_accel_b(oracle, obs[:, sl], g("obs")[:, sl], oracle.default_model(), task.frame_skip,
                                   (gold["qpos"], gold["qvel"], gold["act"], gold["nstep"]), gold["actions"])
        print(f"frame_skip 20 accelerometer, {mapping}: worst error on steady-contact states {worst:.3e}, {nsw} states switch a contact")
    else:
        close(obs[:, sl], g("obs")[:, sl], t["accel"], "accelerometer")
    close(rew, g("reward"), t["reward"], "reward")
    close(comps, g("comps"), t["reward"], "reward components")
    # terminations are threshold tests on f32 vs f64 states: identical except within rounding of the threshold
    z1 = g("qpos1")[:, 2]
    sure = np.abs(z1 - 0.05) > 1e-4
    assert np.array_equal(done[sure], g("done")[sure])
    sim.close()


@pytest.mark.parametrize("mapping", ["lane", "quad", "pair", "link"])
def test_step_matches_oracle_on_fresh_states(oracle, mapping):
    """Seeded states the fixture does not hold, through the device-pointer entry points."""
    import torch
    from quadruped_gym_amd.sim import BatchedSim
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from make_golden import sample_states
    model, task = oracle.default_model(), configure(oracle.default_task(), "A")
    n = 200                                               # ragged: 3 full waves + 8 lanes
    qpos, qvel, act, nstep = sample_states(model, task, n, seed=99)
    rng = np.random.default_rng(5)
    actions = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
    b = oracle.Batch(model, task, n)
    b.set_state(qpos.astype(np.float64), qvel.astype(np.float64), act.astype(np.float64), None, nstep)
    obs_o, rew_o, done_o, comps_o = b.step(actions.astype(np.float64))
    q_o, v_o, a_o, _, n_o = b.get_state()

    sim = BatchedSim(n, task=configure(_abi.default_task(), "A"))
    sim.set_mapping(MAPPINGS[mapping])
    sim.set_state(qpos, qvel, act, None, nstep)
    dev = torch.device("cuda:0")
    a_d = torch.from_numpy(actions).to(dev)
    packed = torch.full((n, 35), float("nan"), device=dev)
    sim.step_device_packed(a_d, packed)
    torch.cuda.synchronize()
    p = packed.cpu().numpy()
    assert np.isfinite(p).all()
    t = TOL["A"]
    mask = np.ones(33, bool)
    mask[12:15] = False
    close(p[:, :33][:, mask], obs_o[:, mask], t["obs"], "obs")
    close(p[:, 12:15], obs_o[:, 12:15], t["accel"], "accelerometer")
    close(p[:, 33], rew_o, t["reward"], "reward")
    q1, v1, a1, _, n1 = sim.get_state()
    close(q1, q_o, t["qpos"], "qpos")
    close(v1, v_o, t["qvel"], "qvel")
    close(a1, a_o, t["act"], "act")
    assert np.array_equal(n1, n_o)
    sure = np.abs(q_o[:, 2] - 0.05) > 1e-4
    assert np.array_equal(p[sure, 34] > 0.5, done_o[sure])
    sim.close()


@pytest.mark.parametrize("mapping", ["lane", "quad", "pair", "link"])
def test_joint_limit_branches_match_oracle(oracle, mapping):
    """Every branch of the soft joint limits in one batch: hinges below the lower and above the upper limit (0.5 ... 9 degrees
    outside: inside and beyond the damper's ramp), moving further out, at rest, coming back slowly and LEAVING FAST (the spring
    would pull back: no torque, secant damping), plus hinges inside the range -- airborne robots, so that nothing but the limits,
    the servos and the rigid-body terms act.  The one-link-per-lane kernel evaluates both limits in one signed expression
    (qg_kernel_link.hip), the other kernels side by side: all of them against the oracle's two-sided form."""
    import torch
    from quadruped_gym_amd.sim import BatchedSim
    from helpers import random_state
    model, task = oracle.default_model(), configure(oracle.default_task(), "A")
    rng = np.random.default_rng(2026)
    n = 192
    qpos = np.zeros((n, 19)); qvel = np.zeros((n, 18))
    outside = np.deg2rad([0.5, 2.0, 5.0, 9.0])
    speeds = np.array([-12.0, -3.0, -0.3, 0.0, 0.3, 3.0, 12.0])
    beyond = 0
    for e in range(n):
        q, v = random_state(rng, model, z=1.0, vel=0.3)
        for j in range(12):
            lo, hi = model.jnt_range[j][0], model.jnt_range[j][1]
            kind = (e + 5 * j) % 3                        # 0: inside the range, 1: below the lower limit, 2: above the upper one
            if kind:
                d = outside[(e // 3 + j) % 4]
                q[7 + j] = lo - d if kind == 1 else hi + d
                v[6 + j] = speeds[(e + 3 * j) % 7]
                beyond += 1
        qpos[e], qvel[e] = q, v
    assert beyond > n * 6
    qpos, qvel = qpos.astype(np.float32), qvel.astype(np.float32)
    act = rng.uniform(-0.5, 0.5, (n, 12)).astype(np.float32)
    nstep = np.zeros(n, np.int32)
    actions = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
    b = oracle.Batch(model, task, n)
    b.set_state(qpos.astype(np.float64), qvel.astype(np.float64), act.astype(np.float64), None, nstep)
    obs_o, rew_o, done_o, _ = b.step(actions.astype(np.float64))
    q_o, v_o, a_o, _, n_o = b.get_state()
    sim = BatchedSim(n, task=configure(_abi.default_task(), "A"))
    sim.set_mapping(MAPPINGS[mapping])
    sim.set_state(qpos, qvel, act, None, nstep)
    dev = torch.device("cuda:0")
    packed = torch.full((n, 35), float("nan"), device=dev)
    sim.step_device_packed(torch.from_numpy(actions).to(dev), packed)
    torch.cuda.synchronize()
    pk = packed.cpu().numpy()
    assert np.isfinite(pk).all()
    t = TOL["A"]
    q1, v1, a1, _, n1 = sim.get_state()
    close(q1, q_o, t["qpos"], "qpos")
    close(v1, v_o, t["qvel"], "qvel")             # the hinge velocities carry the limit torques of four substeps
    close(a1, a_o, t["act"], "act")
    mask = np.ones(33, bool); mask[12:15] = False
    close(pk[:, :33][:, mask], obs_o[:, mask], t["obs"], "obs")
    close(pk[:, 33], rew_o, t["reward"], "reward")
    assert np.array_equal(n1, n_o)
    sim.close()


@pytest.mark.parametrize("mapping", ["lane", "quad", "link"])
def test_generic_variant_matches_oracle_on_a_modified_robot(oracle, mapping):
    """Any model other than the compiled-in default runs the generic kernel variant (tables read from
    device memory instead of literals).  Heavier feet, a different servo gain and a shifted hip mount
    must track the oracle given the same modified model."""
    from quadruped_gym_amd.sim import BatchedSim
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from make_golden import sample_states

    def tweak(m):
        for k in range(4):
            m.body_mass[3 + 3 * k] *= 1.3
            m.act_kp[1 + 3 * k] = 80.0
        m.body_pos[4][0] += 0.004           # leg 2's hip mount: legs no longer identical
        m.contact_friction = 0.7
        return m
    model, task = tweak(oracle.default_model()), configure(oracle.default_task(), "A")
    n = 96
    qpos, qvel, act, nstep = sample_states(model, task, n, seed=123)
    actions = np.random.default_rng(6).uniform(-1, 1, (n, 12)).astype(np.float32)
    b = oracle.Batch(model, task, n)
    b.set_state(qpos.astype(np.float64), qvel.astype(np.float64), act.astype(np.float64), None, nstep)
    obs_o, rew_o, done_o, _ = b.step(actions.astype(np.float64))
    q_o, v_o, a_o, _, _ = b.get_state()
    sim = BatchedSim(n, model=tweak(_abi.default_model()), task=configure(_abi.default_task(), "A"))
    sim.set_mapping(MAPPINGS[mapping])
    assert not sim.baked and sim.mapping == MAPPINGS[mapping]
    ref = BatchedSim(4)
    assert ref.baked                        # the default robot takes the literal-constant variant
    ref.close()
    sim.set_state(qpos, qvel, act, None, nstep)
    obs, rew, done, _ = sim.step(actions)
    q1, v1, a1, _, _ = sim.get_state()
    t = TOL["A"]
    close(q1, q_o, t["qpos"], "qpos")
    close(v1, v_o, t["qvel"], "qvel")
    close(a1, a_o, t["act"], "act")
    mask = np.ones(33, bool)
    mask[12:15] = False
    close(obs[:, mask], obs_o[:, mask], t["obs"], "obs")
    close(rew, rew_o, t["reward"], "reward")
    sim.close()


def test_reset_contract_and_first_observation():
    """quadruped.py:115-139: qpos0, zero velocity/activation, ctrl = [0, 0, -0.5]*4, time 0."""
    from quadruped_gym_amd.sim import BatchedSim
    sim = BatchedSim(130)
    m = sim.model
    qpos, qvel, act, ctrl, nstep = sim.get_state()
    assert np.allclose(qpos, np.array(m.qpos0[:], np.float32)[None])
    assert not qvel.any() and not act.any() and not nstep.any()
    assert np.array_equal(ctrl, np.tile(np.array([0, 0, -0.5] * 4, np.float32), (130, 1)))
    # masked reset leaves the other envs alone
    a = np.random.default_rng(0).uniform(-1, 1, (130, 12)).astype(np.float32)
    sim.step(a)
    before = sim.get_state()
    mask = np.zeros(130, np.uint8)
    mask[[0, 64, 129]] = 1
    sim.reset(mask=mask)
    after = sim.get_state()
    assert np.allclose(after[0][mask == 1], np.array(m.qpos0[:], np.float32)[None])
    assert np.array_equal(after[0][mask == 0], before[0][mask == 0])
    assert np.array_equal(after[4], np.where(mask == 1, 0, 4))
    sim.close()


def test_random_yaw_reset_matches_oracle_stream(oracle):
    from quadruped_gym_amd.sim import BatchedSim
    sim = BatchedSim(100, env_index_base=1000)
    sim.reset(seed=42, flags=_abi.RESET_RANDOM_YAW)
    qpos = sim.get_state()[0]
    a = np.array([2 * np.pi * oracle.uniform(42, 1000 + i, 0) for i in range(100)])
    assert np.allclose(qpos[:, 3], np.cos(a / 2), atol=2e-7) and np.allclose(qpos[:, 6], np.sin(a / 2), atol=2e-7)
    assert not qpos[:, 4:6].any()
    sim.close()


def test_joint_jitter_reset_matches_oracle_stream(oracle):
    """QG_RESET_JOINT_JITTER (the reference's open 'randomize starting pose, joints' item, TODO.md:8): every hinge starts at
    qpos0 + jitter * U(-1, 1) from its own stream of the (seed, global env index, episode) key, kept inside its range."""
    from quadruped_gym_amd.sim import BatchedSim
    n, base = 100, 5000
    task = _abi.default_task()
    task.reset_joint_jitter = 0.25
    sim = BatchedSim(n, task=task, env_index_base=base)
    flags = _abi.RESET_RANDOM_YAW | _abi.RESET_JOINT_JITTER
    for episode in range(2):                      # the explicit reset advances the per-env episode counter
        sim.reset(seed=77, flags=flags)
        qpos = sim.get_state()[0]
        ot = oracle.default_task()
        ot.reset_joint_jitter = 0.25
        want = np.array([np.array(oracle.reset(oracle.default_model(), ot, seed=77, env_index=base + i, counter=episode,
                                               flags=flags).qpos[:]) for i in range(n)])
        assert np.allclose(qpos, want, atol=3e-7)
        lo = np.array([r[0] for r in sim.model.jnt_range]); hi = np.array([r[1] for r in sim.model.jnt_range])
        assert (qpos[:, 7:] >= lo - 1e-6).all() and (qpos[:, 7:] <= hi + 1e-6).all()
        assert np.ptp(qpos[:, 9]) > 0.3          # the draw really spreads (shin hinge, range wider than the jitter)
    sim.reset(seed=77, flags=0)
    assert np.allclose(sim.get_state()[0], np.array(sim.model.qpos0[:], np.float32)[None])
    sim.close()


@pytest.mark.parametrize("mapping", ["lane", "quad", "pair", "link"])
def test_auto_reset_draws_match_oracle_stream(oracle, mapping):
    """The in-kernel auto-reset of each mapping applies the same yaw + hinge-jitter draws as the oracle's reset with the env's
    episode counter."""
    from quadruped_gym_amd.sim import BatchedSim
    n, base = 70, 300
    task = _abi.default_task()
    task.max_time = 0.016          # 8 substeps: every second env-step ends the episode
    task.auto_reset = 1
    task.reset_joint_jitter = 0.2
    flags = _abi.RESET_RANDOM_YAW | _abi.RESET_JOINT_JITTER
    task.reset_flags = flags
    sim = BatchedSim(n, task=task, env_index_base=base)
    sim.set_mapping(MAPPINGS[mapping])
    sim.reset(seed=5, flags=flags)                # episode 0 draws
    ot = oracle.default_task()
    ot.reset_joint_jitter = 0.2
    a = np.zeros((n, 12), np.float32)
    for episode in (1, 2):
        d = None
        for _ in range(2):
            d = sim.step(a)[2]
        assert d.all()
        qpos, qvel, act, _, nstep = sim.get_state()
        want = np.array([np.array(oracle.reset(oracle.default_model(), ot, seed=5, env_index=base + i, counter=episode,
                                               flags=flags).qpos[:]) for i in range(n)])
        assert np.allclose(qpos, want, atol=3e-7), np.abs(qpos - want).max()
        assert not qvel.any() and not act.any() and not nstep.any()
    sim.close()


@pytest.mark.parametrize("mapping", ["lane", "quad", "pair", "link"])
def test_time_limit_terminates_and_auto_reset(mapping):
    """`time >= max_time` is reported as terminated on the exact substep the f64-accumulated clock
    crosses it (quadruped.py:149-151); with auto_reset the env restarts inside the same launch."""
    from quadruped_gym_amd.sim import BatchedSim
    task = _abi.default_task()
    task.max_time = 0.05           # 25 substeps -> 7th env-step at frame_skip 4 (28 >= 25... first >= is step 7)
    task.auto_reset = 1
    sim = BatchedSim(70, task=task)
    sim.set_mapping(MAPPINGS[mapping])
    lim = sim.limit_substeps
    a = np.zeros((70, 12), np.float32)
    k_done = None
    for k in range(1, 12):
        obs, rew, done, _ = sim.step(a)
        if done.all():
            k_done = k
            break
        assert not done.any()
    assert k_done == -(-lim // 4)
    qpos, qvel, act, ctrl, nstep = sim.get_state()
    assert not nstep.any() and not qvel.any() and np.allclose(qpos, np.array(sim.model.qpos0[:], np.float32)[None])
    assert np.abs(obs).sum() > 0                      # the terminal observation is returned, not the reset one
    sim.close()


@pytest.mark.parametrize("mapping", ["lane", "quad", "pair", "link"])
def test_full_size_invariants(mapping):
    """BASELINE config 2 size (4096 envs): properties that need no oracle."""
    import torch
    from quadruped_gym_amd.sim import BatchedSim
    n = 4096
    task = _abi.default_task()
    sim = BatchedSim(n, task=task)
    sim.set_mapping(MAPPINGS[mapping])
    rng = np.random.default_rng(3)
    dev = torch.device("cuda:0")
    packed = torch.empty((n, 35), device=dev)
    acts = [torch.from_numpy(rng.uniform(-1, 1, (n, 12)).astype(np.float32)).to(dev) for _ in range(8)]
    for k in range(120):
        sim.step_device_packed(acts[k % 8], packed)
    torch.cuda.synchronize()
    qpos, qvel, act, _, nstep = sim.get_state()
    assert np.isfinite(qpos).all() and np.isfinite(qvel).all()
    assert np.allclose(np.linalg.norm(qpos[:, 3:7], axis=1), 1.0, atol=1e-5)      # unit quaternions
    assert (nstep == 480).all()
    assert qpos[:, 2].min() > 0.0 and np.abs(qvel).max() < 100.0                     # nobody fell through the floor
    assert (np.abs(act) <= np.array([0.5, 0.91, 1.0] * 4) + 1e-6).all()              # activations inside the ctrlrange
    # identical envs in different lanes / waves produce identical bits; different actions differ
    sim2 = BatchedSim(n, task=task)
    sim2.set_mapping(MAPPINGS[mapping])
    same = torch.from_numpy(np.tile(rng.uniform(-1, 1, (1, 12)).astype(np.float32), (n, 1))).to(dev)
    for k in range(30):
        sim2.step_device_packed(same, packed)
    torch.cuda.synchronize()
    q2 = sim2.get_state()[0]
    assert (q2 == q2[0:1]).all()
    p = packed.cpu().numpy()
    assert (p == p[0:1]).all()
    sim.close(); sim2.close()


@pytest.mark.parametrize("mapping", ["lane", "quad", "pair", "link"])
def test_diverged_envs_are_reported_done_and_reset(mapping):
    """A state with NaN / Inf (here injected through set_state) must not linger: the env is reported done and,
    with auto_reset, restarts -- the counterpart of the engine's own bad-state reset."""
    from quadruped_gym_amd.sim import BatchedSim
    task = _abi.default_task()
    task.auto_reset = 1
    sim = BatchedSim(40, task=task)
    sim.set_mapping(MAPPINGS[mapping])
    qpos, qvel, act, ctrl, nstep = sim.get_state()
    qvel[5, 7] = np.nan
    qpos[21, 2] = np.inf
    qvel[33, 0] = 1e30
    sim.set_state(qpos, qvel, act)
    obs, rew, done, _ = sim.step(np.zeros((40, 12), np.float32))
    assert done[[5, 21, 33]].all() and done.sum() == 3
    q1, v1, a1, _, n1 = sim.get_state()
    assert np.isfinite(q1).all() and np.isfinite(v1).all() and (n1[[5, 21, 33]] == 0).all()
    sim.close()


def test_auto_mapping_policy():
    """AUTO = the measured optimum per batch size (profiles/r01/pair_sweep.txt, profiles/r02/map_sweep.txt); a modified robot never
    takes PAIR; up to 4096 envs it runs the one-link-per-lane kernel with its tables in LDS."""
    from quadruped_gym_amd.sim import BatchedSim
    for n, want in ((64, _abi.MAP_LINK), (4096, _abi.MAP_LINK), (4097, _abi.MAP_QUAD), (16384, _abi.MAP_QUAD), (16385, _abi.MAP_PAIR),
                    (32768, _abi.MAP_PAIR), (32769, _abi.MAP_QUAD), (57343, _abi.MAP_QUAD), (57344, _abi.MAP_PAIR)):
        sim = BatchedSim(n)
        assert sim.baked and sim.mapping == want, (n, sim.mapping)
        sim.close()
    t = _abi.default_task()
    t.sensor_lag = 0                       # un-lagged sensors: the one-link-per-lane kernel does not serve them
    sim = BatchedSim(2048, task=t)
    assert sim.mapping == _abi.MAP_QUAD
    sim.set_mapping(_abi.MAP_LINK)
    assert sim.mapping == _abi.MAP_QUAD    # an explicit request falls back as well
    sim.close()
    m = _abi.default_model()
    m.contact_friction = 0.7
    sim = BatchedSim(2048, model=m)
    assert not sim.baked and sim.mapping == _abi.MAP_LINK
    sim.close()
    sim = BatchedSim(4097, model=m)
    assert not sim.baked and sim.mapping == _abi.MAP_QUAD
    sim.close()
    sim = BatchedSim(20000, model=m)
    assert not sim.baked and sim.mapping == _abi.MAP_QUAD
    with pytest.raises(RuntimeError, match="compiled-in robot"):
        sim.set_mapping(_abi.MAP_PAIR)
    assert sim.mapping == _abi.MAP_QUAD
    sim.close()


def test_mappings_agree_with_each_other():
    """The mappings run the same arithmetic per leg; only the order of the four-leg sums differs,
    so a 50-step rollout from reset stays within rounding-level drift of one another."""
    from quadruped_gym_amd.sim import BatchedSim
    n = 256
    sims = [BatchedSim(n), BatchedSim(n), BatchedSim(n), BatchedSim(n)]
    sims[0].set_mapping(_abi.MAP_LANE)
    sims[1].set_mapping(_abi.MAP_QUAD)
    sims[2].set_mapping(_abi.MAP_PAIR)
    sims[3].set_mapping(_abi.MAP_LINK)
    rng = np.random.default_rng(21)
    for k in range(50):
        a = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
        o = [s.step(a) for s in sims]
        if k == 0:
            assert np.allclose(o[0][0], o[1][0], atol=1e-4, rtol=1e-4)
            assert np.allclose(o[0][0], o[2][0], atol=1e-4, rtol=1e-4)
            assert np.allclose(o[0][0], o[3][0], atol=1e-4, rtol=1e-4)
    q = [s.get_state()[0] for s in sims]
    assert np.allclose(q[0], q[1], atol=5e-3) and np.allclose(q[0], q[2], atol=5e-3) and np.allclose(q[0], q[3], atol=5e-3)
    for s in sims:
        s.close()


@pytest.mark.parametrize("mapping", ["lane", "quad", "pair", "link"])
def test_sharding_does_not_change_results(mapping):
    """Two handles of 128 envs with env_index_base 0 / 128 reproduce one handle of 256 bit for bit
    (per-env random streams are keyed by the global env index)."""
    from quadruped_gym_amd.sim import BatchedSim
    task = _abi.default_task()
    task.auto_reset = 1
    task.max_time = 0.04
    task.reset_flags = _abi.RESET_RANDOM_YAW | _abi.RESET_JOINT_JITTER
    rng = np.random.default_rng(11)
    whole = BatchedSim(256, task=task)
    parts = [BatchedSim(128, task=task, env_index_base=0), BatchedSim(128, task=task, env_index_base=128)]
    for s in [whole] + parts:
        s.set_mapping(MAPPINGS[mapping])
        s.reset(seed=9, flags=_abi.RESET_RANDOM_YAW | _abi.RESET_JOINT_JITTER)
    for k in range(12):
        a = rng.uniform(-1, 1, (256, 12)).astype(np.float32)
        ow = whole.step(a)
        op = [parts[0].step(a[:128]), parts[1].step(a[128:])]
        assert np.array_equal(ow[0], np.concatenate([op[0][0], op[1][0]]))
        assert np.array_equal(ow[2], np.concatenate([op[0][2], op[1][2]]))
    sw = whole.get_state()
    sp = [p.get_state() for p in parts]
    assert np.array_equal(sw[0], np.concatenate([sp[0][0], sp[1][0]]))
    for s in [whole] + parts:
        s.close()


@pytest.mark.parametrize("mapping", ["lane", "quad", "pair", "link"])
def test_non_finite_actions_stay_contained(mapping):
    """NaN / Inf in one env's action must not leak: the other envs step bit-identically to a clean run, +-Inf acts like the
    clip bound (quadruped.py:160), and the state of every env stays finite (a NaN command either clips or trips the
    divergence guard, which reports done and resets the env)."""
    from quadruped_gym_amd.sim import BatchedSim
    n = 64
    task = _abi.default_task()
    task.auto_reset = 1
    rng = np.random.default_rng(8)
    clean, dirty = BatchedSim(n, task=task), BatchedSim(n, task=task)
    for s in (clean, dirty):
        s.set_mapping(MAPPINGS[mapping])
    bad = np.array([5, 17, 40])
    for k in range(6):
        a = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
        b = a.copy()
        b[5, 3] = np.nan
        b[17, :] = np.inf
        b[40, 7] = -np.inf
        a[17, :] = 1.0                                    # what the clip makes of +Inf
        a[40, 7] = -1.0
        oc, od = clean.step(a), dirty.step(b)
        keep = np.setdiff1d(np.arange(n), [5])
        assert np.array_equal(oc[0][keep], od[0][keep]) and np.array_equal(oc[1][keep], od[1][keep])
        qd, vd = dirty.get_state()[:2]
        assert np.isfinite(qd).all() and np.isfinite(vd).all()
    for s in (clean, dirty):
        s.close()


@pytest.mark.parametrize("mapping", ["lane", "quad", "pair", "link"])
@pytest.mark.parametrize("n", [1, 15, 33])
def test_tiny_and_ragged_batches(oracle, mapping, n):
    """A single env, less than one quad-wave's 16 envs, and one env past the pair-wave's 32: the tail lanes shadow the last
    env and must neither write nor disturb it."""
    from quadruped_gym_amd.sim import BatchedSim
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from make_golden import sample_states
    model, task = oracle.default_model(), configure(oracle.default_task(), "A")
    qpos, qvel, act, nstep = sample_states(model, task, n, seed=300 + n)
    actions = np.random.default_rng(n).uniform(-1, 1, (n, 12)).astype(np.float32)
    b = oracle.Batch(model, task, n)
    b.set_state(qpos.astype(np.float64), qvel.astype(np.float64), act.astype(np.float64), None, nstep)
    obs_o, rew_o, done_o, _ = b.step(actions.astype(np.float64))
    q_o, v_o = b.get_state()[:2]
    sim = BatchedSim(n, task=configure(_abi.default_task(), "A"))
    sim.set_mapping(MAPPINGS[mapping])
    sim.set_state(qpos, qvel, act, None, nstep)
    obs, rew, done, _ = sim.step(actions)
    q1, v1 = sim.get_state()[:2]
    t = TOL["A"]
    mask = np.ones(33, bool)
    mask[12:15] = False
    close(obs[:, mask], obs_o[:, mask], t["obs"], "obs")
    close(rew, rew_o, t["reward"], "reward")
    close(q1, q_o, t["qpos"], "qpos")
    close(v1, v_o, t["qvel"], "qvel")
    sim.close()


def test_reset_is_ordered_after_steps_on_a_side_stream():
    """include/quadgym.h ordering contract: qg_reset waits for device-pointer steps still in flight on ANY stream of the
    device before it rewrites the state (it runs on the library's own non-blocking stream).  Many steps queued on a side
    stream, then reset(): every env must stand at qpos0 with a zero clock, nothing of the steps may land afterwards."""
    import torch
    from quadruped_gym_amd.sim import BatchedSim
    n = 32768
    sim = BatchedSim(n)
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    acts = torch.rand((n, 12), device=dev) * 2 - 1
    packed = torch.empty((n, 35), device=dev)
    torch.cuda.synchronize()
    for rounds in range(3):
        for _ in range(200):                      # ~6 ms of queued kernels: the host gets far ahead of the device
            sim.step_device_packed(acts, packed, stream=side)
        sim.reset()
        qpos, qvel, act, ctrl, nstep = sim.get_state()
        assert not nstep.any() and not qvel.any() and not act.any()
        assert np.array_equal(qpos, np.tile(np.array(sim.model.qpos0[:], np.float32), (n, 1)))
    sim.close()


def test_host_step_is_ordered_after_steps_on_a_side_stream():
    """The host-pointer step only pays a device-wide wait when a device-pointer step has been enqueued on a caller's stream since the
    last one -- and then it must: 150 steps queued on a side stream followed at once by a host step give the state of 151 steps in
    order (a twin batch stepped through the host entry point only is the reference; same kernel, same inputs: bit-identical)."""
    import torch
    from quadruped_gym_amd.sim import BatchedSim
    n = 32768
    a, b = BatchedSim(n), BatchedSim(n)
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    acts_h = np.random.default_rng(3).uniform(-1, 1, (n, 12)).astype(np.float32)
    acts = torch.from_numpy(acts_h).to(dev)
    packed = torch.empty((n, 35), device=dev)
    torch.cuda.synchronize()
    for _ in range(150):                          # ~4 ms of queued kernels: the host gets far ahead of the device
        a.step_device_packed(acts, packed, stream=side)
    obs_a, rew_a, done_a, _ = a.step(acts_h)      # must wait for the 150
    for _ in range(150):
        b.step(acts_h)
    obs_b, rew_b, done_b, _ = b.step(acts_h)
    assert np.array_equal(obs_a, obs_b) and np.array_equal(rew_a, rew_b) and np.array_equal(done_a, done_b)
    for x, y in zip(a.get_state(), b.get_state()):
        assert np.array_equal(x, y)
    a.close(); b.close()


def test_masked_reset_keeps_the_batch_seed(oracle):
    """The seed keys the reset streams of every env (auto-resets included); a masked reset must not re-key the batch: it
    draws from the seed of the last whole-batch reset, whatever seed argument it is given."""
    from quadruped_gym_amd.sim import BatchedSim
    n, base = 64, 40
    sim = BatchedSim(n, env_index_base=base)
    sim.reset(seed=1234, flags=_abi.RESET_RANDOM_YAW)             # episode counter 0 -> 1
    mask = np.zeros(n, np.uint8)
    mask[[3, 17]] = 1
    sim.reset(mask=mask, seed=999, flags=_abi.RESET_RANDOM_YAW)   # counter 1 for the two, seed stays 1234
    qpos = sim.get_state()[0]
    for i in range(n):
        a = 2 * np.pi * oracle.uniform(1234, base + i, 1 if mask[i] else 0)
        assert np.allclose(qpos[i, [3, 6]], [np.cos(a / 2), np.sin(a / 2)], atol=2e-7), i
    sim.close()


def _one_step_against_oracle(oracle, n, seed, mapping, sensor_lag=1, frame_skip=4):
    """n seeded states (tools/make_golden.sample_states), one env-step through `mapping` and through the oracle; returns nothing,
    asserts the stated tolerances on state, observation, reward."""
    from quadruped_gym_amd.sim import BatchedSim
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from make_golden import sample_states
    model, otask = oracle.default_model(), configure(oracle.default_task(), "A")
    otask.sensor_lag, otask.frame_skip = sensor_lag, frame_skip
    qpos, qvel, act, nstep = sample_states(model, otask, n, seed=seed)
    actions = np.random.default_rng(seed).uniform(-1, 1, (n, 12)).astype(np.float32)
    b = oracle.Batch(model, otask, n)
    b.set_state(qpos.astype(np.float64), qvel.astype(np.float64), act.astype(np.float64), None, nstep)
    obs_o, rew_o, done_o, _ = b.step(actions.astype(np.float64))
    q_o, v_o, a_o, _, _ = b.get_state()
    task = configure(_abi.default_task(), "A")
    task.sensor_lag, task.frame_skip = sensor_lag, frame_skip
    sim = BatchedSim(n, task=task)
    sim.set_mapping(mapping)
    sim.set_state(qpos, qvel, act, None, nstep)
    obs, rew, done, _ = sim.step(actions)
    q1, v1, a1, _, _ = sim.get_state()
    sim.close()
    t = TOL["A"]
    mask = np.ones(33, bool)
    mask[12:15] = False
    close(obs[:, mask], obs_o[:, mask], t["obs"], "obs")
    close(obs[:, 12:15], obs_o[:, 12:15], t["accel"], "accelerometer")
    close(rew, rew_o, t["reward"], "reward")
    close(q1, q_o, t["qpos"], "qpos")
    close(v1, v_o, t["qvel"], "qvel")
    close(a1, a_o, t["act"], "act")
    return obs, obs_o


@pytest.mark.parametrize("mapping", ["lane", "quad", "pair", "link"])
def test_unlagged_sensors_match_oracle(oracle, mapping):
    """task.sensor_lag = 0 (an option beyond the reference, whose observation always lags by one substep): the sensors describe
    the state the step ends in -- one extra forward pass whose state changes are discarded.  The state must advance exactly as
    with lagged sensors, the observation must be the oracle's un-lagged one and differ from the lagged one."""
    obs0, ref0 = _one_step_against_oracle(oracle, 150, 77, MAPPINGS[mapping], sensor_lag=0)
    obs1, ref1 = _one_step_against_oracle(oracle, 150, 77, MAPPINGS[mapping], sensor_lag=1)
    assert np.abs(ref0[:, 21:24] - ref1[:, 21:24]).max() > 1e-3         # world linear velocity: one substep apart
    assert np.abs(obs0[:, 21:24] - obs1[:, 21:24]).max() > 1e-3


def test_host_step_is_ordered_after_a_graph_replay():
    """Replays of a hipGraph enqueue device-pointer steps the library never sees.  Capture 4 steps on a side stream, do a host step
    (which takes the device-wide wait and clears the in-flight flag), then replay the graph 40 times (~2 ms of queued kernels at this
    size) and call the host step at once: it must come AFTER the replays (the handle remembers that it has been captured).  A twin
    stepped in that order through the host entry point only is the reference -- same kernel, same inputs: bit-identical."""
    import torch
    from quadruped_gym_amd.sim import BatchedSim
    n, G, R = 32768, 4, 40
    a, b = BatchedSim(n), BatchedSim(n)
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    acts_h = np.random.default_rng(5).uniform(-1, 1, (n, 12)).astype(np.float32)
    acts = torch.from_numpy(acts_h).to(dev)
    packed = torch.empty((n, 35), device=dev)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(G):
            a.step_device_packed(acts, packed, stream=torch.cuda.current_stream(dev))
    a.step(acts_h)                                  # host step: synchronises, the per-launch in-flight flag is clear again
    b.step(acts_h)
    with torch.cuda.stream(side):
        for _ in range(R):
            graph.replay()
    obs_a, rew_a, done_a, _ = a.step(acts_h)        # must wait for the 160 replayed steps
    for _ in range(G * R):
        b.step(acts_h)
    obs_b, rew_b, done_b, _ = b.step(acts_h)
    assert np.array_equal(obs_a, obs_b) and np.array_equal(rew_a, rew_b) and np.array_equal(done_a, done_b)
    for x, y in zip(a.get_state(), b.get_state()):
        assert np.array_equal(x, y)
    a.close(); b.close()


def test_link_step_replays_from_a_hipgraph_like_eager():
    """The plain env-step of the headline mapping (one link per lane, 4096 envs, auto-reset with random yaw) captured into a hipGraph
    of 8 steps and replayed 20 times leaves the same packed rows and the same state as 160 eager steps of a twin: nothing of the
    step lives on the host between launches (episode counters and reset streams are device state)."""
    import torch
    from quadruped_gym_amd.sim import BatchedSim
    n, G, R = 4096, 8, 20
    task = _abi.default_task()
    task.auto_reset = 1
    task.max_time = 0.2                             # 25 env-steps per episode: several auto-resets inside the replays
    task.reset_flags = _abi.RESET_RANDOM_YAW
    a, b = BatchedSim(n, task=task), BatchedSim(n, task=task)
    assert a.mapping == _abi.MAP_LINK
    a.reset(seed=3, flags=task.reset_flags); b.reset(seed=3, flags=task.reset_flags)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(2)
    acts = [torch.rand((n, 12), generator=gen, device=dev) * 2 - 1 for _ in range(G)]
    pa = [torch.empty((n, 35), device=dev) for _ in range(G)]
    pb = [torch.empty((n, 35), device=dev) for _ in range(G)]
    side = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for g in range(G):
            a.step_device_packed(acts[g], pa[g], stream=torch.cuda.current_stream(dev))
    finished = 0
    for rep in range(R):
        graph.replay()
        for g in range(G):
            b.step_device_packed(acts[g], pb[g])
        torch.cuda.synchronize()
        for g in range(G):
            assert torch.equal(pa[g], pb[g]), (rep, g)
            finished += int(pa[g][:, 34].sum())
    assert finished >= 5 * n                        # every env restarted several times inside the graph
    for x, y in zip(a.get_state(), b.get_state()):
        assert np.array_equal(x, y)
    ea, eb = a.get_reset_streams(), b.get_reset_streams()
    assert np.array_equal(ea[0], eb[0]) and ea[1] == eb[1] and ea[0].min() >= 5
    a.close(); b.close()


def test_config5_at_full_size_invariants_and_oracle_sample(oracle):
    """BASELINE config 5 at its size: 4096 envs x frame_skip 20 with the 21-value IMU + joint pack, AUTO = one link per lane.  60
    env-steps of random actions with the invariants of a healthy batch, then one more step of the whole batch with a strided
    256-env sample (every lane position of a wave) against the oracle from the same f32 states, within the frame_skip-20 bounds."""
    from quadruped_gym_amd.sim import BatchedSim
    n, fs = 4096, 20
    task = _abi.default_task()
    task.frame_skip = fs
    task.obs_mode = 1
    task.auto_reset = 1
    task.use_fall = 1
    task.fall_height = 0.05
    task.reset_flags = _abi.RESET_RANDOM_YAW
    sim = BatchedSim(n, task=task)
    assert sim.mapping == _abi.MAP_LINK and sim.obs_dim == 21
    sim.reset(seed=11, flags=task.reset_flags)
    rng = np.random.default_rng(8)
    for k in range(60):
        a = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
        obs, rew, done, _ = sim.step(a)
        assert obs.shape == (n, 21) and np.isfinite(obs).all() and np.isfinite(rew).all()
    qpos, qvel, act, ctrl, nstep = sim.get_state()
    assert np.isfinite(qpos).all() and np.isfinite(qvel).all()
    assert np.allclose(np.linalg.norm(qpos[:, 3:7], axis=1), 1.0, atol=1e-5)
    assert (np.abs(act) <= 1.0 + 1e-6).all() and (qpos[:, 2] > 0.02).all() and (qpos[:, 2] < 0.3).all()
    assert (nstep % fs == 0).all() and nstep.max() <= 60 * fs
    # one more step: the whole batch on the GPU, a strided sample through the oracle from the same f32 states
    idx = np.arange(0, n, 16) + (np.arange(n // 16) % 16)          # 256 envs, every position within a wave's four envs and beyond
    a = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
    otask = oracle.default_task()
    otask.frame_skip = fs
    otask.obs_mode = 1
    otask.use_fall = 1
    otask.fall_height = 0.05
    batch = oracle.Batch(oracle.default_model(), otask, len(idx))
    batch.reset()
    batch.set_state(qpos[idx].astype(np.float64), qvel[idx].astype(np.float64), act[idx].astype(np.float64), None, nstep[idx])
    obs_o, rew_o, done_o, _ = batch.step(a[idx].astype(np.float64))
    obs, rew, done, _ = sim.step(a)
    keep = ~(done_o.astype(bool) | np.asarray(done)[idx].astype(bool))          # envs that finished were auto-reset on the GPU
    assert keep.sum() > 200
    T = TOL["B"]
    jp = np.r_[0:12, 15:21]                                         # joint positions, gyro, velocimeter
    assert np.allclose(obs[idx][keep][:, jp], obs_o[keep][:, jp], atol=T["obs"][0], rtol=T["obs"][1])
    worst, nsw = close_accel_b(oracle, obs[idx][:, 12:15], obs_o[:, 12:15], oracle.default_model(), fs,
                               (qpos[idx], qvel[idx], act[idx], nstep[idx]), a[idx], keep=keep)
    print(f"config 5 accelerometer: worst error on steady-contact states {worst:.3e}, {nsw} of {int(keep.sum())} states switch a contact")
    assert np.allclose(rew[idx][keep], rew_o[keep], atol=T["reward"][0], rtol=T["reward"][1])
    q1 = sim.get_state()[0]
    assert np.allclose(q1[idx][keep], batch.get_state()[0][keep], atol=T["qpos"][0], rtol=T["qpos"][1])
    sim.close()


def test_standing_height_at_the_joint_centre_command_4096_envs(oracle):
    """GPU mirror of tests/test_oracle_physics.py::test_standing_height_at_the_joint_centre_command_follows_the_geometry: 4096 robots
    (AUTO = one link per lane) held at the joint-centre command [0, 0, -0.5] x 4 for 4000 substeps settle where the oracle does --
    FRAME origin 0.1423 m above the floor, the height the reference's XML + OBJ geometry gives for that pose (not the 0.12 / 0.13 of
    walking_quad.py:243-247,369) -- hinges at ctrl / gear, every env on the same bits."""
    from quadruped_gym_amd.sim import BatchedSim
    n = 4096
    task = _abi.default_task()
    task.frame_skip = 20
    task.use_time_limit = 0
    sim = BatchedSim(n, task=task)
    assert sim.mapping == _abi.MAP_LINK
    sim.reset()
    a = np.tile(np.array([0, 0, -0.5] * 4, np.float32), (n, 1))
    for _ in range(200):
        obs, rew, done, _ = sim.step(a)
        assert not done.any()
    qpos, qvel = sim.get_state()[:2]
    model, otask = oracle.default_model(), oracle.default_task()
    e = oracle.reset(model, otask)
    for _ in range(4000):
        oracle.substep(model, e, np.array(otask.default_ctrl[:]))
    z_o = e.qpos[2]
    assert abs(z_o - 0.1423) < 3e-4
    assert np.abs(qpos[:, 2] - z_o).max() < 2e-4 and np.abs(qvel).max() < 1e-4
    assert np.allclose(qpos[:, 7:19], np.array(e.qpos[7:19]), atol=2e-4)
    assert np.allclose(qpos[:, 7:19], [0, 0, -0.78125] * 4, atol=4e-3)           # ctrl / gear
    assert (qpos == qpos[0]).all()                                                # identical envs, identical bits
    assert (qpos[:, 2] - 0.13).min() > 0.012
    sim.close()


@pytest.mark.parametrize("trial", range(10))
def test_random_task_and_model_variants_match_oracle(oracle, trial):
    """Ten seeded variants the other tests do not visit: a perturbed robot (masses, inertias, servo gains, damping, contact and limit
    parameters -- the kernel variants that read the model tables from LDS / scalar loads), frame_skip 1 ... 8, either observation pack,
    random reward weights and fall height, every mapping that serves the variant.  One env-step from seeded rollout states against the
    oracle run on the SAME perturbed numbers, within the frame_skip-4 bounds scaled by frame_skip / 4 (the error compounds per substep)."""
    import sys
    from quadruped_gym_amd.sim import BatchedSim
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from make_golden import sample_states
    rng = np.random.default_rng(1000 + trial)
    fs = int(rng.choice([1, 2, 3, 5, 8]))
    obs_mode = int(rng.integers(0, 2))
    perturb = trial % 3 != 0                              # every third trial keeps the compiled-in robot (literal-constant kernels)

    def shape(model, task):
        if perturb:
            for b in range(13):
                model.body_mass[b] *= float(r_model[0][b])
                for i in range(6):
                    model.body_inertia[b][i] *= float(r_model[0][b])
            for j in range(12):
                model.act_kp[j] *= float(r_model[1][j]); model.act_kv[j] *= float(r_model[2][j]); model.jnt_damping[j] *= float(r_model[3][j])
            model.contact_stiffness *= float(r_model[4][0]); model.contact_damping *= float(r_model[4][1])
            model.limit_stiffness *= float(r_model[4][2]); model.free_damping *= float(r_model[4][3])
        task.frame_skip, task.obs_mode = fs, obs_mode
        task.use_fall, task.fall_height = 1, float(r_task[0])
        task.w_forward, task.w_ctrl, task.alive_bonus = float(r_task[1]), float(r_task[2]), float(r_task[3])
        return model, task
    # trials 1, 4, 7: the legs stay quarter-turn copies of one another (the same factor for the four copies of a link / hinge)
    per_link = rng.uniform(0.8, 1.25, 4)                  # FRAME, fema, shin, foot
    body_f = np.array([per_link[0]] + [per_link[1 + (b % 3)] for b in range(12)])
    r_model = [body_f, np.tile(rng.uniform(0.8, 1.2, 3), 4), np.tile(rng.uniform(0.8, 1.2, 3), 4), np.tile(rng.uniform(0.7, 1.3, 3), 4),
               rng.uniform(0.8, 1.2, 4)]
    if trial % 3 == 2:                                    # ... and a robot whose four legs DIFFER from one another (a payload on one side,
        r_model[0] = rng.uniform(0.8, 1.25, 13)           # an ageing servo): nothing in the table-driven kernels assumes the symmetry
        r_model[1], r_model[2], r_model[3] = rng.uniform(0.8, 1.2, 12), rng.uniform(0.8, 1.2, 12), rng.uniform(0.7, 1.3, 12)
    r_task = [rng.uniform(0.03, 0.08), rng.uniform(-2, 2), rng.uniform(-0.5, 0.0), rng.uniform(0, 2)]
    model, otask = shape(oracle.default_model(), oracle.default_task())
    n = 96
    qpos, qvel, act, nstep = sample_states(model, otask, n, seed=500 + trial)
    actions = rng.uniform(-1.3, 1.3, (n, 12)).astype(np.float32)
    b = oracle.Batch(model, otask, n)
    b.set_state(qpos.astype(np.float64), qvel.astype(np.float64), act.astype(np.float64), None, nstep)
    obs_o, rew_o, done_o, comps_o = b.step(actions.astype(np.float64))
    q_o, v_o, a_o, c_o, n_o = b.get_state()
    gmodel, gtask = shape(_abi.default_model(), _abi.default_task())
    scale = max(1.0, fs / 4.0)
    t = {k: (v[0] * scale, v[1] * scale) for k, v in TOL["A"].items()}
    od = 21 if obs_mode else 33
    mask = np.ones(od, bool)
    mask[12:15] = False
    for name in (["lane", "quad", "link"] + ([] if perturb else ["pair"])):
        sim = BatchedSim(n, model=gmodel, task=gtask)
        assert sim.baked == (not perturb)
        sim.set_mapping(MAPPINGS[name])
        sim.set_state(qpos, qvel, act, None, nstep)
        obs, rew, done, comps = sim.step(actions, want_components=True)
        q1, v1, a1, c1, n1 = sim.get_state()
        sim.close()
        what = f"trial {trial} ({'perturbed' if perturb else 'built-in'} robot, frame_skip {fs}, obs {od}) {name}: "
        close(q1, q_o, t["qpos"], what + "qpos")
        close(v1, v_o, t["qvel"], what + "qvel")
        close(a1, a_o, t["act"], what + "act")
        assert np.array_equal(n1, n_o) and np.array_equal(c1, c_o.astype(np.float32))
        close(obs[:, mask], obs_o[:, mask], t["obs"], what + "obs")
        close(obs[:, 12:15], obs_o[:, 12:15], t["accel"], what + "accelerometer")
        close(rew, rew_o, t["reward"], what + "reward")
        close(comps, comps_o, t["reward"], what + "reward components")
        sure = np.abs(q_o[:, 2] - gtask.fall_height) > 1e-4
        assert np.array_equal(np.asarray(done)[sure], done_o[sure])


@pytest.mark.parametrize("mapping,n", [("link", 700), ("quad", 700), ("pair", 700), ("lane", 700)])
def test_reward_is_the_f32_sum_of_its_components_in_every_kernel(mapping, n):
    """``total_reward += fn()`` over the reward dict (quadruped.py:170-175) in f32: every step kernel adds the rounded products
    ``forward + control_cost`` and then ``alive_bonus`` with no multiply-add contraction (reward_total, round 4), so the reward a step
    reports IS the sum of the components it reports, to the bit, whatever the mapping and the weights."""
    from quadruped_gym_amd.sim import BatchedSim
    task = configure(_abi.default_task(), "A")
    task.w_forward, task.w_ctrl, task.alive_bonus = 0.7, -0.37, 0.3
    sim = BatchedSim(n, task=task)
    sim.set_mapping(MAPPINGS[mapping])
    sim.reset(seed=4)
    rng = np.random.default_rng(12)
    for _ in range(30):
        obs, rew, done, comps = sim.step(rng.uniform(-1, 1, (n, 12)).astype(np.float32), want_components=True)
        want = (comps[:, 0] + comps[:, 1]).astype(np.float32) + comps[:, 2]
        assert np.array_equal(rew, want.astype(np.float32))
    sim.close()
