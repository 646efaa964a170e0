"""The device walking task layer (qg_walk_*) against the walking oracle, which is itself pinned to the
reference's estimator / command code by tests/test_walking_oracle.py.  The oracle is fed the GPU's own
observations, so the comparison isolates the task layer from the (chaotic) physics rollout."""
import numpy as np
import pytest

from oracle import walking_oracle as W
from quadruped_gym_amd import _abi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mapping", ["auto", "quad", "pair", "lane"])
@pytest.mark.parametrize("frame_skip", [4, 20])
def test_walking_rewards_match_oracle(frame_skip, mapping):
    """`auto` (= one link per lane at this size), `quad` (one leg per lane) and `pair` (two legs per lane) run the whole walking
    env-step as ONE launch (task layer fused into the step kernel); `lane` keeps the three launches estimator -> physics -> reward.
    All go through the same device functions."""
    from quadruped_gym_amd.envs.walking import REWARD_KEYS, WalkingQuadrupedVecEnv
    n = 70
    settle = 0.05
    env = WalkingQuadrupedVecEnv(n, settling_time=settle, frame_skip=frame_skip, max_time=1000.0)
    env._sim.set_mapping({"auto": _abi.MAP_AUTO, "quad": _abi.MAP_QUAD, "pair": _abi.MAP_PAIR, "lane": _abi.MAP_LANE}[mapping])
    if mapping == "auto":
        assert env._sim.mapping == _abi.MAP_LINK
    assert REWARD_KEYS == W.REWARD_KEYS
    dt = 0.002 * frame_skip
    o = W.WalkingOracle(n, dt, settling_time=settle)
    rng = np.random.default_rng(3)
    sp, al, th = rng.uniform(0.1, 0.5, n), rng.uniform(-np.pi, np.pi, n), rng.uniform(-np.pi, np.pi, n)
    for i in range(n):
        o.controls.set_orientation(i, th[i])
        o.controls.set_velocity_speed_alpha(i, sp[i], al[i])
    env.set_commands(o.controls.velocity[:, :2], o.controls.heading[:, :2])
    env.reset()
    o.reset()
    steps = 2 * o.est.W + 30                       # let the estimator's ring buffer wrap
    t = 0.0
    data_ctrl = np.tile([0, 0, -0.5] * 4, (n, 1)).astype(np.float64)       # quadruped.py:124
    ph = rng.uniform(0, 2 * np.pi, (n, 12)); fr = rng.uniform(0.5, 4.0, (n, 12)); am = rng.uniform(0.1, 1.2, (n, 12))
    saw_nan = False
    for k in range(steps):
        a = (am * np.sin(2 * np.pi * fr * k * dt + ph) + 0.05 * rng.normal(size=(n, 12))).astype(np.float32)
        if k % 7 == 0:
            a[:, 3] = a[:, 3]                       # keep a channel exactly repeating its previous value sometimes
        obs, rew, dones, infos = env.step(a)
        assert not dones.any()
        act = o.pre_step(np.full(n, t), data_ctrl, a.astype(np.float64))
        ctrl = np.clip(act, -1, 1)
        tot, comps, flip = o.post_step(obs.astype(np.float64), ctrl)
        got = env.last_components.astype(np.float64)
        # f32 kernel vs f64 oracle on identical inputs; the derived term divides a difference of f32 sensors by dt
        # (unit() of an exactly zero local velocity -- the vertical drop of the first steps -- is NaN in the reference too)
        assert np.allclose(got[:, :10], comps[:, :10], rtol=2e-4, atol=2e-4, equal_nan=True), (k, np.nanmax(np.abs(got[:, :10] - comps[:, :10])))
        assert np.allclose(got[:, 10], comps[:, 10], rtol=1e-3, atol=2e-2 / dt * 1e-3 + 1e-3), k
        assert np.allclose(rew, got.sum(1), rtol=1e-5, atol=1e-4, equal_nan=True)
        saw_nan = saw_nan or bool(np.isnan(got).any())
        for i in (0, n - 1):
            assert infos[i][REWARD_KEYS[3]] == pytest.approx(got[i, 3], nan_ok=True)
        data_ctrl = ctrl
        for _ in range(frame_skip):
            t += 0.002
    f, amp, ideal = env.estimates()
    assert np.allclose(f, o.f_est, rtol=1e-4, atol=1e-4) and np.allclose(amp, o.a_est, rtol=1e-4, atol=1e-5)
    assert np.allclose(ideal, o.ideal[:, :2], rtol=1e-4, atol=1e-5)
    assert (f > 0.2).any() and (amp > 0.2).any()
    env.close()


def test_settling_mask_flip_and_auto_reset():
    from quadruped_gym_amd.envs.walking import WalkingQuadrupedVecEnv
    n = 40
    env = WalkingQuadrupedVecEnv(n, settling_time=0.1, max_time=0.3, random_init=True, random_controls=True,
                                 reset_options={"fixed_heading_angle": 0.0, "fixed_velocity_angle": 0.0, "fixed_speed": 0.3})
    env.reset()
    assert np.allclose(env.velocity, [0.3, 0.0]) and np.allclose(env.heading, [1.0, 0.0])    # train_quadruped.py:40-46
    a = np.ones((n, 12), np.float32)
    env.step(a)
    ctrl = env._sim.get_state()[3]
    assert np.allclose(ctrl, [0, 0, -0.5] * 4)             # data.time < settling_time: joint centres applied
    k_done = None
    for k in range(2, 60):
        obs, rew, dones, infos = env.step(a)
        if dones.all():
            k_done = k
            break
    assert k_done == 38                                    # 0.3 s -> 151 substeps (f64 clock) -> 38 env-steps of 4
    assert "terminal_observation" in infos[0] and not obs.any()
    ctrl = env._sim.get_state()[3]
    assert np.allclose(ctrl, [0, 0, -0.5] * 4)             # auto-reset restored the default ctrl
    # flip termination: put one robot on its back
    env2 = WalkingQuadrupedVecEnv(8, max_time=100.0)
    env2.reset()
    qpos = env2._sim.get_state()[0]
    qpos[3, 2] = 0.3
    qpos[3, 3:7] = [0, 1, 0, 0]                             # rolled by 180 degrees
    env2._sim.set_state(qpos=qpos)
    obs, rew, dones, infos = env2.step(np.zeros((8, 12), np.float32))
    assert dones[3] and not dones[[0, 1, 2, 4, 5, 6, 7]].any()
    env.close(); env2.close()


def test_device_command_sampler_matches_oracle_stream():
    """random_controls on the device: at reset and at every auto-reset each env draws a new command with the
    semantics of VelocityHeadingControls.sample(options) (control_inputs.py:74-115) from its own (seed, global env index,
    episode) stream; the oracle's Controls.sample fed with the same stream must give the same velocity / heading."""
    from quadruped_gym_amd.envs.walking import WalkingQuadrupedVecEnv
    from oracle.walking_oracle import Controls, sample_keyed
    n, base, seed = 48, 700, 31
    for options in ({"min_speed": 0.2, "max_speed": 0.8}, {"fixed_heading_angle": 0.5, "max_speed": 2.0},
                    {"fixed_velocity_angle": -1.0, "fixed_speed": 0.3}, None):
        env = WalkingQuadrupedVecEnv(n, max_time=0.016, random_init=True, random_controls=True, reset_options=options,
                                     env_index_base=base, seed=seed, device_commands=True)
        env.reset()                                           # episode 0
        a = np.zeros((n, 12), np.float32)
        for episode in range(3):
            want = Controls(n)
            for i in range(n):
                sample_keyed(want, i, options, seed, base + i, episode)
            v, h = env.commands()
            assert np.allclose(v, want.velocity[:, :2], atol=2e-6) and np.allclose(h, want.heading[:, :2], atol=2e-6)
            d = None
            for _ in range(2):                                # 8 substeps: the second env-step ends the episode
                d = env.step(a)[2]
            assert d.all()
        env.close()
    # without random_controls nothing is drawn
    env = WalkingQuadrupedVecEnv(4, device_commands=True)
    env.reset()
    v, h = env.commands()
    assert not v.any() and not h.any()
    env.close()


def test_single_env_facade_has_reference_signature():
    import inspect
    from quadruped_gym_amd.envs.walking import REWARD_KEYS, WalkingQuadrupedEnv
    sig = list(inspect.signature(WalkingQuadrupedEnv.__init__).parameters)
    assert sig[:5] == ["self", "settling_time", "random_controls", "random_init", "reset_options"]   # walking_quad.py:11
    env = WalkingQuadrupedEnv(settling_time=0.0, random_controls=True, reset_options={"fixed_heading_angle": 0.0,
                              "fixed_velocity_angle": 0.0, "fixed_speed": 0.3}, model_path="builtin", max_time=20.0, frame_skip=10)
    obs, info = env.reset()
    assert obs.shape == (33,) and info == {}
    rng = np.random.default_rng(0)
    for _ in range(5):      # the symmetric drop of the first steps has exactly zero local xy velocity: unit() -> NaN, as math_utils.unit
        obs, r, term, trunc, info = env.step(rng.uniform(-1, 1, 12).astype(np.float32))
    assert set(info) == set(REWARD_KEYS) and trunc is False and isinstance(r, float)
    assert np.isfinite(r) and r == pytest.approx(sum(info.values()), rel=1e-5)
    env.close()


@pytest.mark.parametrize("nan_direction", [True, False])
def test_zero_norm_direction_is_nan_as_in_the_reference(nan_direction):
    """math_utils.unit() of a zero vector is NaN (math_utils.py:7-8) and that NaN reaches progress_direction_reward_local and
    the total (walking_quad.py:197-205,422).  The device code is compiled with -ffinite-math-only, so the kernel produces that
    NaN explicitly (quiet-NaN bit pattern through an integer select): a zero command must read NaN in exactly those two
    places, every other component stays finite, and envs with a non-zero command and a moving body stay finite.
    ``nan_direction=False`` (qg_walk_params.unit_zero = 1) is the way out for training: the direction term is 0 there, every
    other number of the step is unchanged, and the oracle with the same option agrees."""
    from oracle import walking_oracle as WO
    from quadruped_gym_amd.envs.walking import WalkingQuadrupedVecEnv
    n = 8
    # (settling_time: the first six env-steps apply the joint centres -- four-fold symmetric actuation, as at the start of every
    # training episode, train_quadruped.py:40-46)
    env = WalkingQuadrupedVecEnv(n, max_time=100.0, settling_time=0.05, nan_direction=nan_direction)
    ref = WalkingQuadrupedVecEnv(n, max_time=100.0, settling_time=0.05)   # the reference's behaviour, same actions
    vel = np.tile(np.array([[0.3, 0.1]], np.float32), (n, 1))
    vel[2] = 0.0                                            # zero commanded velocity: unit(command) is NaN
    head = np.tile(np.array([[1.0, 0.0]], np.float32), (n, 1))
    for e in (env, ref):
        e.set_commands(vel, head)
        e.reset()
    rng = np.random.default_rng(1)
    first_rew = None
    for k in range(40):                                     # asymmetric actions: the body picks up a local xy velocity
        a = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
        obs, rew, dones, infos = env.step(a)
        obs_r, rew_r, _, _ = ref.step(a)
        if k == 0:
            first_rew = rew.copy()
        assert np.array_equal(obs, obs_r)                   # the option touches the reward only
    comps, comps_r = env.last_components, ref.last_components
    others = np.delete(np.arange(11), 2)
    keep = np.delete(np.arange(n), 2)
    assert np.isnan(comps_r[2, 2]) and np.isnan(rew_r[2])
    assert np.isfinite(comps[keep]).all() and np.isfinite(rew[keep]).all()
    assert np.array_equal(comps[:, others], comps_r[:, others]) and np.array_equal(rew[keep], rew_r[keep])
    if nan_direction:
        assert np.isnan(comps[2, 2]) and np.isnan(rew[2])
        assert np.isfinite(comps[2, others]).all()
        # the first step of an episode: the robot drops straight down from its symmetric start pose under symmetric actuation, the four
        # legs' reactions cancel EXACTLY in f32 and the local xy velocity is 0.0 -- every env's first reward is NaN (what
        # INTEGRATION.md section 4 warns a training run about; the engine's f64 state, summed in another order, would hold 1e-19 there)
        assert np.isnan(first_rew).all()
    else:
        assert comps[2, 2] == 0.0 and np.isfinite(rew[2]) and np.isfinite(first_rew).all()
        # the total without the direction term, summed in the reference's order
        tot = np.zeros((), np.float32)
        for j in range(11):
            tot = np.float32(tot + comps[2, j])
        assert rew[2] == tot
        # and the oracle's option does the same thing to the same numbers
        wo = WO.WalkingOracle(1, env.dt, unit_zero=True)
        wo.controls.velocity[0, :2] = 0.0
        sens = np.zeros((1, 33)); sens[0, 30:32] = [0.2, -0.1]
        total, oc, _ = wo.post_step(sens, np.zeros((1, 12)))
        assert oc[0, 2] == 0.0 and np.isfinite(total[0])
        wn = WO.WalkingOracle(1, env.dt)
        wn.controls.velocity[0, :2] = 0.0
        tn, on, _ = wn.post_step(sens, np.zeros((1, 12)))
        assert np.isnan(on[0, 2]) and np.isnan(tn[0])
    env.close(); ref.close()


def test_walk_destroy_restores_the_sim():
    """qg_walk_create switches the bound sim's flip termination and data.ctrl tracking on; qg_walk_destroy must hand the sim
    back as it found it (include/quadgym.h): an upside-down robot terminates while the layer is bound and not afterwards."""
    import ctypes as C
    from quadruped_gym_amd._abi import check, load_library
    from quadruped_gym_amd.sim import BatchedSim
    lib = load_library()
    sim = BatchedSim(4)
    sim.set_track_ctrl(False)

    def flipped_done():
        qpos = sim.get_state()[0]
        qpos[:, 2] = 0.3
        qpos[1, 3:7] = [0, 1, 0, 0]                         # env 1 rolled by 180 degrees
        sim.set_state(qpos=qpos, qvel=np.zeros((4, 18), np.float32))
        return sim.step(np.zeros((4, 12), np.float32))[2]
    assert not flipped_done().any()                         # base task: no flip termination
    w = C.c_void_p()
    check(lib.qg_walk_create(sim._h, None, C.byref(w)), "qg_walk_create")
    d = flipped_done()
    assert d[1] and d.sum() == 1
    a = np.full((4, 12), 0.25, np.float32)
    sim.step(a)
    assert np.array_equal(sim.get_state()[3], a)            # data.ctrl tracked while the layer is bound
    check(lib.qg_walk_destroy(w), "qg_walk_destroy")
    assert not flipped_done().any()
    before = sim.get_state()[3]
    sim.step(np.full((4, 12), -0.5, np.float32))
    assert np.array_equal(sim.get_state()[3], before)       # tracking is off again, as the caller had set it
    sim.close()


@pytest.mark.parametrize("fused_mapping", ["auto", "quad"])
def test_fused_and_three_launch_walking_agree_on_a_modified_robot(fused_mapping):
    """Any model other than the compiled-in one runs the table-driven kernel variants; the walking env-step is then the fused launch
    of the generic one-link-per-lane kernel (AUTO up to 4096 envs) or one-leg-per-lane kernel (QUAD), or estimator -> generic
    one-env-per-lane kernel -> reward (LANE).  Same states,
    same actions: rewards, components, estimates and dones must agree (the two differ only in the summation order of the four
    legs / twelve channels), across the settling phase, an estimator window wrap at frame_skip 20 and auto-resets."""
    from quadruped_gym_amd.envs.walking import WalkingQuadrupedVecEnv
    from quadruped_gym_amd.model.loader import load_model
    n = 40

    def make(mapping):
        env = WalkingQuadrupedVecEnv(n, settling_time=0.1, frame_skip=20, max_time=1.5, random_init=True, seed=4)
        return env
    # a heavier, weaker robot: patch the model the envs load
    import quadruped_gym_amd.envs.walking as W
    orig = W.load_model

    def tweaked(path):
        m, layout = orig(path)
        for k in range(4):
            m.body_mass[3 + 3 * k] *= 1.25
            m.act_kp[1 + 3 * k] = 85.0
        m.contact_friction = 0.8
        return m, layout
    W.load_model = tweaked
    try:
        fused, three = make("auto"), make("lane")
    finally:
        W.load_model = orig
    three._sim.set_mapping(_abi.MAP_LANE)
    if fused_mapping == "quad":
        fused._sim.set_mapping(_abi.MAP_QUAD)
    want = _abi.MAP_LINK if fused_mapping == "auto" else _abi.MAP_QUAD
    assert not fused._sim.baked and fused._sim.mapping == want and three._sim.mapping == _abi.MAP_LANE
    cmd_v = np.tile(np.array([[0.25, 0.05]], np.float32), (n, 1)); cmd_h = np.tile(np.array([[0.8, 0.6]], np.float32), (n, 1))
    for e in (fused, three):
        e.set_commands(cmd_v, cmd_h)
        e.reset()
    rng = np.random.default_rng(12)
    finished = 0
    for k in range(90):                                    # 1.5 s / 0.04 s: every env restarts at step 38 and 76; window = 50 samples
        a = (0.6 * np.sin(0.4 * k + np.arange(12)) + 0.1 * rng.normal(size=(n, 12))).astype(np.float32)
        three._sim.set_state(*fused._sim.get_state())      # same physics state into both: the comparison is one env-step deep
        o1, r1, d1, i1 = fused.step(a)
        o2, r2, d2, i2 = three.step(a)
        assert np.array_equal(d1, d2), k
        assert np.allclose(o1, o2, atol=2e-3, rtol=2e-3), (k, np.abs(o1 - o2).max())      # two summation orders, 20 substeps
        c1, c2 = fused.last_components, three.last_components
        # unit() of an EXACTLY zero local velocity is NaN (the symmetric drop of the first steps); whether a 1e-9 survives depends
        # on the summation order, so the direction term may be NaN in one form and finite in the other: compare it where both are
        ok = np.isfinite(c1) & np.isfinite(c2)
        others = np.delete(np.arange(11), 2)
        assert np.isfinite(c1[:, others]).all() and np.isfinite(c2[:, others]).all()
        assert np.allclose(c1[ok], c2[ok], atol=5e-2, rtol=2e-2), (k, np.abs(c1 - c2)[ok].max())
        finished += int(d1.sum())
    assert finished >= 2 * n
    f1, a1, id1 = fused.estimates(); f2, a2, id2 = three.estimates()
    assert np.allclose(f1, f2, atol=1e-5) and np.allclose(a1, a2, atol=1e-5) and np.allclose(id1, id2, atol=1e-5)
    fused.close(); three.close()


def test_estimator_is_bit_identical_across_mappings_at_scale():
    """The estimator is fed with data.ctrl (the clipped actions), not with anything the physics computes, so its outputs must not
    depend on the work mapping AT ALL: the six-channels-per-lane form with 12-byte accesses inside the two-legs-per-lane kernel
    (AUTO at 20 000 envs), the three-channel form of the one-leg-per-lane kernel and the one-channel stand-alone kernel (LANE: three
    launches) give bit-identical frequency / amplitude estimates over 270 steps -- past the wrap of the 250-sample window, i.e.
    through block entries with and without old samples behind the write index, the partial last block and the other-blocks cache."""
    from quadruped_gym_amd.envs.walking import WalkingQuadrupedVecEnv
    n = 20000
    envs = {}
    for name, mapping in (("pair", None), ("quad", _abi.MAP_QUAD), ("lane", _abi.MAP_LANE)):
        e = WalkingQuadrupedVecEnv(n, frame_skip=4, max_time=100.0, seed=1)
        if mapping is not None:
            e._sim.set_mapping(mapping)
        e.reset()
        envs[name] = e
    assert envs["pair"]._sim.mapping == _abi.MAP_PAIR
    import torch
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(21)
    obs = torch.empty((n, 33), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, device=dev, dtype=torch.uint8)
    t = torch.arange(12, device=dev, dtype=torch.float32)
    for k in range(270):
        # smooth signals of different frequency per channel plus noise, beyond the clip now and then
        a = 0.9 * torch.sin(0.05 * k * (1.0 + t) + t)[None, :] + 0.3 * (torch.rand((n, 12), generator=gen, device=dev) - 0.5)
        a = a.contiguous()
        for e in envs.values():
            e.step_tensor(a, obs, rew, done)
        if k in (5, 17, 249, 250, 251, 269):
            torch.cuda.synchronize()
            ref = envs["lane"].estimates()
            for name in ("pair", "quad"):
                got = envs[name].estimates()
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), (k, name)
    f_est, a_est, _ = envs["lane"].estimates()
    assert np.isfinite(f_est).all() and (a_est > 0.5).all() and (f_est > 0).any()
    for e in envs.values():
        e.close()


def test_walking_at_config3_size_matches_oracle_on_a_sample():
    """The walking env-step at BASELINE config 3's size (32 768 envs: the two-legs-per-lane kernel with the task layer fused in, 1 024
    waves in four-wave workgroups) against the NumPy oracle on a strided sample of 256 envs, 290 steps (past the wrap of the
    250-sample estimator window); the oracle is fed the sampled envs' own observations, as in the small-batch test above."""
    from quadruped_gym_amd.envs.walking import WalkingQuadrupedVecEnv
    n, m, fs = 32768, 256, 4
    pick = np.arange(m) * (n // m) + 7
    env = WalkingQuadrupedVecEnv(n, frame_skip=fs, max_time=1000.0)
    assert env._sim.mapping == _abi.MAP_PAIR
    dt = 0.002 * fs
    o = W.WalkingOracle(m, dt, settling_time=0.0)
    rng = np.random.default_rng(8)
    sp, al, th = rng.uniform(0.1, 0.5, n), rng.uniform(-np.pi, np.pi, n), rng.uniform(-np.pi, np.pi, n)
    vel = np.stack([sp * np.cos(al), sp * np.sin(al)], 1).astype(np.float32)
    head = np.stack([np.cos(th), np.sin(th)], 1).astype(np.float32)
    for j, i in enumerate(pick):
        o.controls.set_orientation(j, th[i])
        o.controls.set_velocity_speed_alpha(j, sp[i], al[i])
    env.set_commands(vel, head)
    env.reset()
    o.reset()
    t = 0.0
    data_ctrl = np.tile([0, 0, -0.5] * 4, (m, 1)).astype(np.float64)
    ph = rng.uniform(0, 2 * np.pi, (n, 12)).astype(np.float32); fr = rng.uniform(0.5, 4.0, (n, 12)).astype(np.float32)
    am = rng.uniform(0.1, 1.2, (n, 12)).astype(np.float32)
    worst = 0.0
    for k in range(290):
        a = (am * np.sin(2 * np.pi * fr * (k * dt) + ph)).astype(np.float32)
        obs, rew, dones, infos = env.step(a)
        assert not dones[pick].any()
        act = o.pre_step(np.full(m, t), data_ctrl, a[pick].astype(np.float64))
        ctrl = np.clip(act, -1, 1)
        tot, comps, flip = o.post_step(obs[pick].astype(np.float64), ctrl)
        got = env.last_components[pick].astype(np.float64)
        ok = np.isfinite(comps[:, :10]) & np.isfinite(got[:, :10])
        assert np.allclose(got[:, :10][ok], comps[:, :10][ok], rtol=2e-4, atol=2e-4), (k, np.abs(got[:, :10] - comps[:, :10])[ok].max())
        assert np.allclose(got[:, 10], comps[:, 10], rtol=1e-3, atol=2e-2 / dt * 1e-3 + 1e-3), k
        worst = max(worst, float(np.abs(got[:, :10] - comps[:, :10])[ok].max()))
        data_ctrl = ctrl
        for _ in range(fs):
            t += 0.002
    f, amp, ideal = env.estimates()
    assert np.allclose(f[pick], o.f_est, rtol=1e-4, atol=1e-4) and np.allclose(amp[pick], o.a_est, rtol=1e-4, atol=1e-5)
    assert np.isfinite(env._sim.get_state()[0]).all()
    print(f"largest component difference on the sample: {worst:.2e}")
    env.close()


@pytest.mark.parametrize("frame_skip,mapping", [(2, "auto"), (3, "quad"), (1, "pair")])
def test_estimator_windows_beyond_256_samples_match_oracle(frame_skip, mapping):
    """math_utils.py:26-28 puts no bound on the estimator window ceil(2 / (min_freq * timestep * frame_skip)): at the shipped 2 ms
    timestep frame_skip 1 / 2 / 3 give 1000 / 500 / 334 samples (63 / 32 / 21 blocks of 16: more than the 16 block summaries the
    unrolled rebuild reads).  The estimates over more than one wrap of the ring against the oracle (pinned to the reference's
    estimator), one fused mapping each."""
    from quadruped_gym_amd.envs.walking import WalkingQuadrupedVecEnv
    n = 24
    env = WalkingQuadrupedVecEnv(n, frame_skip=frame_skip, max_time=1000.0)
    env._sim.set_mapping({"auto": _abi.MAP_AUTO, "quad": _abi.MAP_QUAD, "pair": _abi.MAP_PAIR}[mapping])
    dt = 0.002 * frame_skip
    o = W.WalkingOracle(n, dt, settling_time=0.0)
    assert o.est.W == {1: 1000, 2: 500, 3: 334}[frame_skip]
    env.reset(); o.reset()
    rng = np.random.default_rng(4)
    ph = rng.uniform(0, 2 * np.pi, (n, 12)); fr = rng.uniform(0.5, 4.0, (n, 12)); am = rng.uniform(0.1, 1.2, (n, 12))
    data_ctrl = np.tile([0, 0, -0.5] * 4, (n, 1)).astype(np.float64)
    steps = o.est.W + o.est.W // 4 + 37
    checks = 0
    for k in range(steps):
        # a slow amplitude envelope makes the window's extrema move from block to block as the ring wraps
        env_amp = 1.0 + 0.5 * np.sin(2 * np.pi * 0.37 * k * dt)
        a = (env_amp * am * np.sin(2 * np.pi * fr * k * dt + ph)).astype(np.float32)
        dones = env.step(a)[2]
        o.pre_step(np.zeros(n), data_ctrl, a.astype(np.float64))
        data_ctrl = np.clip(a.astype(np.float64), -1, 1)
        data_ctrl[dones] = [0, 0, -0.5] * 4               # an env that flipped over was auto-reset: data.ctrl back at its default
        if k % 61 == 0 or k >= steps - 40:
            f, amp, _ = env.estimates()
            assert np.allclose(f, o.f_est, rtol=1e-4, atol=1e-4), (k, np.abs(f - o.f_est).max())
            assert np.allclose(amp, o.a_est, rtol=1e-4, atol=1e-5), (k, np.abs(amp - o.a_est).max())
            checks += 1
    assert checks > 40 and (o.a_est > 0.2).any()
    env.close()


def test_second_walking_layer_on_one_simulator_is_refused():
    """One task layer per simulator: a second qg_walk_create would save the flags the first one has already switched as the sim's own,
    and destroying the first would then switch the flip termination and data.ctrl tracking off under the second."""
    import ctypes as C
    from quadruped_gym_amd.envs.walking import WalkingQuadrupedVecEnv
    env = WalkingQuadrupedVecEnv(8)
    lib = _abi.load_library()
    h = C.c_void_p()
    assert lib.qg_walk_create(env._sim._h, None, C.byref(h)) == -1 and not h.value
    assert b"already bound" in lib.qg_last_error()
    env.reset()
    env.step(np.zeros((8, 12), np.float32))                # the bound layer is unharmed
    env.close()


@pytest.mark.parametrize("po,n,mapping", [(False, 300, "auto"), (True, 300, "auto"), (True, 5000, "auto"), (False, 20000, "auto")])
def test_task_state_snapshot_restores_bit_identical_rollouts(po, n, mapping):
    """Checkpoint / resume of everything the env keeps per robot: step 300, snapshot, step 50 more (recorded), restore
    -- into the SAME env and into a freshly built twin -- and step the same 50 again: observations, rewards, dones, components and the
    final state are identical to the bit.  Auto-resets with random yaw and re-drawn commands happen inside both windows, so the reset
    streams, the estimator (ring, block summaries, counters) and the frame ring must all have come back."""
    from quadruped_gym_amd.envs.walking import POWalkingQuadrupedVecEnv, WalkingQuadrupedVecEnv
    kw = dict(settling_time=0.05, frame_skip=4, max_time=0.4, random_init=True, random_controls=True, device_commands=True, seed=13,
              reset_options={"min_speed": 0.1, "max_speed": 0.4})
    make = (lambda: POWalkingQuadrupedVecEnv(n, obs_window=5, **kw)) if po else (lambda: WalkingQuadrupedVecEnv(n, **kw))
    env = make()
    env.reset()
    rng = np.random.default_rng(6)
    acts = [rng.uniform(-1, 1, (n, 12)).astype(np.float32) for _ in range(350)]
    for k in range(300):
        env.step(acts[k])
    snap = env.snapshot()

    def run50(e):
        out = []
        for k in range(300, 350):
            obs, rew, dones, infos = e.step(acts[k])
            out.append((obs.copy(), rew.copy(), dones.copy(), e.last_components.copy()))
        return out, e._sim.get_state(), e.commands(), e.estimates()
    first, st1, cmd1, est1 = run50(env)
    assert sum(int(d.sum()) for _, _, d, _ in first) > 0            # auto-resets inside the window
    cmd1 = tuple(x.copy() for x in cmd1)
    twin = make()
    twin.reset()
    for e in (env, twin):
        e.restore(snap)
        again, st2, cmd2, est2 = run50(e)
        for (o1, r1, d1, c1), (o2, r2, d2, c2) in zip(first, again):
            assert np.array_equal(o1, o2) and np.array_equal(d1, d2)
            assert np.array_equal(r1, r2, equal_nan=True) and np.array_equal(c1, c2, equal_nan=True)
        for x, y in zip(st1, st2):
            assert np.array_equal(x, y)
        assert all(np.array_equal(x, y) for x, y in zip(cmd1, cmd2)) and all(np.array_equal(x, y) for x, y in zip(est1, est2))
    # a blob of another shape is refused, not misread
    other = (POWalkingQuadrupedVecEnv(n + 1, obs_window=5, **kw) if po else WalkingQuadrupedVecEnv(n + 1, **kw))
    with pytest.raises((ValueError, _abi.QuadGymError)):
        other.restore(snap)
    env.close(); twin.close(); other.close()


@pytest.mark.parametrize("po,n,steps", [(False, 300, 130), (True, 300, 130), (False, 4096, 40), (True, 4090, 40), (True, 37, 60),
                                        (False, 5001, 130), (False, 16384, 35), (True, 5001, 60), (True, 16384, 32)])
def test_helper_waves_equal_the_one_role_kernel(po, n, steps, monkeypatch):
    """qg_step_kernel_link<WALK, .., HELP> (up to 4096 envs) and qg_step_kernel_quad<2, .., HELP> (4097 .. 16 384 envs: 5001 and
    16 384 here, with and without the observation pack): four helper waves per workgroup run the estimator update (and, in the link kernel, the
    observation pack's history copy, frame and rows) beside the physics waves.  The two forms (helpers: the default; QG_LINK_HELPERS=0
    when the simulator is created: the one-role kernel) are different instantiations of the physics, whose contraction choices differ in an ulp
    here and there, and this robot's contacts amplify an ulp by 1e4 per env-step -- so the comparison is per STEP from IDENTICAL state:
    before every step the one-role env is restored from the helper env's snapshot.  What does not pass through the physics must then
    agree to the BIT: the estimator's estimates (they read data.ctrl), the commands, the dones, every frame of the 260-value stack but
    the newest and the newest frame's data.ctrl / command columns; what does, within one step's amplified rounding (2e-2 absolute,
    typically 1e-8; 2 m/s² on the accelerometer, whose reading jumps when a contact switches inside the step).  Auto-resets with re-drawn commands, a ragged last workgroup (4090, 37 envs) and a wrap of the estimator's window included."""
    from quadruped_gym_amd.envs.walking import POWalkingQuadrupedVecEnv, WalkingQuadrupedVecEnv
    kw = dict(settling_time=0.05, frame_skip=10, max_time=0.6, random_init=True, random_controls=True, device_commands=True, seed=5,
              reset_options={"min_speed": 0.1, "max_speed": 0.4})
    make = (lambda: POWalkingQuadrupedVecEnv(n, obs_window=10, **kw)) if po else (lambda: WalkingQuadrupedVecEnv(n, **kw))
    envs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("QG_LINK_HELPERS", flag)
        e = make()
        e.reset()
        envs.append(e)
    rng = np.random.default_rng(9)
    resets = 0
    worst = 0.0
    for k in range(steps):                                           # the estimator's window at frame_skip 10 is 100 samples
        a = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
        envs[1].restore(envs[0].snapshot())
        (o0, r0, d0, _), (o1, r1, d1, _) = [e.step(a) for e in envs]
        assert np.array_equal(d0, d1), k
        fin = d0.astype(bool)
        if po:
            o0f, o1f = o0.reshape(n, 10, 26), o1.reshape(n, 10, 26)
            live = ~fin                                                                # (a finished env shows its reset stack: computed)
            assert np.array_equal(o0f[live, :9], o1f[live, :9]), k                    # history: copied, not computed
            assert np.array_equal(o0f[live, 9, 11:], o1f[live, 9, 11:]), k            # data.ctrl and the command of the newest frame
        live = ~fin
        acc = slice(9 * 26 + 3, 9 * 26 + 6) if po else slice(12, 15)                   # the accelerometer: contact switches inside the step
        dd = np.abs(o0[live] - o1[live])
        assert float(dd[:, acc].max(initial=0.0)) < 2.0, (k, float(dd[:, acc].max()))
        dd[:, acc] = 0.0
        assert float(dd.max(initial=0.0)) < 2e-2, (k, float(dd.max()))
        # reset observation (PO: the orientation filter's Euler angles of the new heading, computed by each instantiation with its own
        # contraction choices: measured 1.5e-6 (round 3 build) .. 3.0e-6 (round 4, packed base solve) over these cases)
        assert np.allclose(o0[fin], o1[fin], rtol=0, atol=(2e-5 if po else 0.0)), k
        assert np.allclose(r0, r1, rtol=0, atol=5e-2, equal_nan=True), k
        assert np.allclose(envs[0].last_components, envs[1].last_components, rtol=0, atol=5e-2, equal_nan=True), k
        assert all(np.array_equal(x, y) for x, y in zip(envs[0].commands(), envs[1].commands())), k
        assert all(np.array_equal(x, y) for x, y in zip(envs[0].estimates(), envs[1].estimates())), k
        worst = max(worst, float(dd.max(initial=0.0)))
        resets += int(d0.sum())
    assert resets > 0
    for e in envs:
        e.close()


@pytest.mark.parametrize("trial,mapping", [(0, "auto"), (1, "quad"), (2, "pair"), (3, "auto")])
def test_walking_rewards_with_other_task_parameters_match_oracle(trial, mapping):
    """The walking task layer with every constant of walking_quad.py moved (reward weights :362-383, control-cost and estimator
    smoothing :54-59,:254, body height :369, joint centres :36-39, amplitude / frequency targets :272-285, estimator window through
    min_freq) and frame_skip 10 (the training setting, train_quadruped.py:18) or 5: the kernels take all of them as data
    (qg_walk_params), and the oracle given the same numbers must agree on all eleven components over a window wrap."""
    from quadruped_gym_amd.envs.walking import WalkingQuadrupedVecEnv, default_walk_params
    rng = np.random.default_rng(40 + trial)
    fs = 10 if trial % 2 == 0 else 5
    n, dt, settle = 50, 0.002 * fs, 0.04
    p = default_walk_params()
    po = W.default_params()
    po["w"] = np.array(po["w"]) * rng.uniform(0.5, 1.5, 10)
    po["w_diff_ideal"] = float(-20.0 * rng.uniform(0.5, 1.5))
    po["control_cost_alpha"] = float(rng.uniform(0.5, 0.95)); po["ema_alpha"] = float(rng.uniform(0.6, 0.95))
    po["min_freq"] = float(rng.choice([1.0, 2.0, 0.8])); po["body_height"] = float(rng.uniform(0.1, 0.15))
    po["joint_centers"] = np.tile(rng.uniform(-0.4, 0.3, 3), 4)
    po["amp_target"] = np.tile(rng.uniform(0.0, 1.5, 3), 4); po["freq_target"] = np.tile(rng.uniform(0.0, 2.0, 3), 4)
    for i in range(10):
        p.w[i] = float(po["w"][i])
    p.w_diff_ideal, p.control_cost_alpha, p.ema_alpha = po["w_diff_ideal"], po["control_cost_alpha"], po["ema_alpha"]
    p.min_freq, p.body_height = po["min_freq"], po["body_height"]
    for j in range(12):
        p.joint_centers[j], p.amp_target[j], p.freq_target[j] = float(po["joint_centers"][j]), float(po["amp_target"][j]), float(po["freq_target"][j])
    env = WalkingQuadrupedVecEnv(n, settling_time=settle, frame_skip=fs, max_time=1000.0, walk_params=p)
    env._sim.set_mapping({"auto": _abi.MAP_AUTO, "quad": _abi.MAP_QUAD, "pair": _abi.MAP_PAIR}[mapping])
    o = W.WalkingOracle(n, dt, settling_time=settle, params=po)
    sp, al, th = rng.uniform(0.1, 0.5, n), rng.uniform(-np.pi, np.pi, n), rng.uniform(-np.pi, np.pi, n)
    for i in range(n):
        o.controls.set_orientation(i, th[i])
        o.controls.set_velocity_speed_alpha(i, sp[i], al[i])
    env.set_commands(o.controls.velocity[:, :2], o.controls.heading[:, :2])
    env.reset(); o.reset()
    data_ctrl = np.tile(po["joint_centers"], (n, 1)) * 0 + np.tile([0, 0, -0.5] * 4, (n, 1))        # quadruped.py:124, whatever the centres
    ph = rng.uniform(0, 2 * np.pi, (n, 12)); fr = rng.uniform(0.5, 4.0, (n, 12)); am = rng.uniform(0.1, 1.2, (n, 12))
    t = 0.0
    for k in range(o.est.W + 25):
        a = (am * np.sin(2 * np.pi * fr * k * dt + ph) + 0.05 * rng.normal(size=(n, 12))).astype(np.float32)
        obs, rew, dones, infos = env.step(a)
        assert not dones.any()
        act = o.pre_step(np.full(n, t), data_ctrl, a.astype(np.float64))
        ctrl = np.clip(act, -1, 1)
        tot, comps, flip = o.post_step(obs.astype(np.float64), ctrl)
        got = env.last_components.astype(np.float64)
        assert np.allclose(got[:, :10], comps[:, :10], rtol=3e-4, atol=3e-4, equal_nan=True), (k, np.nanmax(np.abs(got[:, :10] - comps[:, :10])))
        assert np.allclose(got[:, 10], comps[:, 10], rtol=1e-3, atol=3e-2 / dt * 1e-3 + 1e-3), k
        assert np.allclose(rew, got.sum(1), rtol=1e-5, atol=2e-4, equal_nan=True)
        data_ctrl = ctrl
        t += 0.002 * fs
    f, amp, ideal = env.estimates()
    assert np.allclose(f, o.f_est, rtol=1e-4, atol=1e-4) and np.allclose(amp, o.a_est, rtol=1e-4, atol=1e-5)
    env.close()
