"""Many env-steps per launch (include/quadgym.h: qg_step_device_seq and the RESIDENT form qg_resident_*): both must leave exactly the
bits of the per-launch one-link-per-lane kernel -- they run the same substep code on state that stays in registers between env-steps
(/root/reference keeps it in MjData across steps the same way, src/envs/quadruped.py:163-165) -- and the resident kernel must never
be able to hang: it leaves by itself when nobody rings, and every entry point that needs the state in memory retires it first."""
import os
import time

import numpy as np
import pytest

from quadruped_gym_amd import _abi

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_vectors.npz")


def _task(frame_skip=4, imu=False, max_time=0.2, yaw=True):
    t = _abi.default_task()
    t.frame_skip = frame_skip
    t.auto_reset = 1
    t.max_time = max_time                             # short episodes: several auto-resets inside every test
    t.use_fall = 1
    t.fall_height = 0.05
    t.reset_flags = _abi.RESET_RANDOM_YAW if yaw else 0
    if imu:
        t.obs_mode = 1
    return t


def _twin(n, task, seed=5):
    from quadruped_gym_amd.sim import BatchedSim
    a, b = BatchedSim(n, task=task), BatchedSim(n, task=task)
    want = _abi.MAP_LINK if n <= 4096 else _abi.MAP_QUAD if (n <= 16384 or 32768 < n < 57344) else _abi.MAP_PAIR
    assert a.mapping == want
    a.reset(seed=seed, flags=task.reset_flags); b.reset(seed=seed, flags=task.reset_flags)
    return a, b


def _sync():
    # the caller's stream only: a device-wide wait (torch.cuda.synchronize) also waits for the resident kernel, i.e. until it has
    # left for lack of rings
    import torch
    torch.cuda.current_stream().synchronize()


def _same_state(a, b):
    for x, y in zip(a.get_state(), b.get_state()):
        assert np.array_equal(x, y)
    ea, eb = a.get_reset_streams(), b.get_reset_streams()
    assert np.array_equal(ea[0], eb[0]) and ea[1] == eb[1]
    return ea[0]


@pytest.mark.parametrize("n,fs,imu", [(4096, 4, False), (1000, 4, False), (3, 20, True), (4096, 20, True),
                                      # the two-legs-per-lane mapping's own one-launch form (BASELINE configs 3 / 4's sizes; one- and
                                      # four-wave workgroups, a ragged last wave, the 21-value pack)
                                      (32768, 4, False), (20001, 4, True), (16400, 8, False), (60000, 4, False),
                                      # ... and the one-leg-per-lane mapping's (one wave per SIMD: 4 097 .. 16 384 envs; two: 32 769 .. 57 343)
                                      (8192, 4, False), (12001, 20, True), (40000, 4, False)])
def test_sequence_launch_is_bit_identical_to_per_step_launches(n, fs, imu):
    """ONE launch of K env-steps against K launches of the per-launch kernel: same packed rows, same final state, same episode
    counters -- through auto-resets with random yaw (25 or 5 env-steps per episode), a ragged last workgroup, both observation packs."""
    import torch
    K, rounds = (24, 4) if n <= 4096 else (12, 3)
    task = _task(fs, imu)
    a, b = _twin(n, task)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    finished = 0
    for rnd in range(rounds):
        acts = torch.rand((K, n, 12), generator=gen, device=dev) * 3 - 1.5          # beyond the clip range on purpose
        pa = torch.empty((K, n, a.obs_dim + 2), device=dev)
        pb = torch.empty((K, n, a.obs_dim + 2), device=dev)
        a.step_device_seq(acts, pa)
        for k in range(K):
            b.step_device_packed(acts[k], pb[k])
        torch.cuda.synchronize()
        assert torch.equal(pa, pb), rnd
        finished += int(pa[:, :, -1].sum())
    episodes = 2 if n <= 4096 else 1                  # (the large batches run 36 env-steps: one time limit of 25, or of 12 - 13 at frame_skip 8 / 20)
    assert finished >= episodes * n
    ep = _same_state(a, b)
    assert ep.min() >= episodes
    a.close(); b.close()


def test_sequence_launch_matches_golden_vectors():
    """The golden states (tests/golden/step_vectors.npz: rollout states plus FRAME / femur contacts) through a sequence launch of ONE
    env-step: the same bits as the per-launch kernel, which tests/test_parity_gpu.py checks against the fixture and the oracle."""
    import torch
    from quadruped_gym_amd.sim import BatchedSim
    gold = dict(np.load(GOLD))
    n = len(gold["qpos"])
    task = _abi.default_task()
    task.use_fall = 1
    task.fall_height = 0.05
    a, b = BatchedSim(n, task=task), BatchedSim(n, task=task)
    for s in (a, b):
        s.set_mapping(_abi.MAP_LINK)
        s.set_state(gold["qpos"], gold["qvel"], gold["act"], None, gold["nstep"])
    dev = torch.device("cuda:0")
    acts = torch.from_numpy(gold["actions"].astype(np.float32)).to(dev)
    pa = torch.empty((1, n, 35), device=dev); pb = torch.empty((n, 35), device=dev)
    a.step_device_seq(acts[None].contiguous(), pa)
    b.step_device_packed(acts, pb)
    torch.cuda.synchronize()
    assert torch.equal(pa[0], pb)
    _same_state(a, b)
    # and against the fixture itself, within the stated one-step tolerances of tests/test_parity_gpu.py
    q1 = a.get_state()[0]
    assert np.abs(q1 - gold["A_qpos1"]).max() <= 5e-6 + 2e-6 * np.abs(gold["A_qpos1"]).max()
    a.close(); b.close()


@pytest.mark.parametrize("n,slots", [(4096, 1), (4096, 4), (1000, 2), (3, 2), (1, 1)])
def test_resident_closed_loop_is_bit_identical_to_per_step_launches(n, slots):
    """The resident kernel rung ONE env-step at a time (a policy in the loop: fresh actions written into the slot before every ring,
    the rows read back after it) against the per-launch kernel: every step's rows, the final state, the episode counters -- 120
    env-steps through auto-resets with random yaw."""
    import torch
    steps = 120
    task = _task()
    a, b = _twin(n, task)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(2)
    mail_a = torch.zeros((slots, n, 12), device=dev)
    mail_p = torch.zeros((slots, n, 35), device=dev)
    pb = torch.empty((n, 35), device=dev)
    a.resident_start(mail_a, mail_p)
    finished = 0
    for i in range(steps):
        act = torch.rand((n, 12), generator=gen, device=dev) * 3 - 1.5
        s = i % slots
        mail_a[s].copy_(act)                          # the "policy": a kernel on the caller's stream writes the slot ...
        a.resident_step(1)                            # ... the ring makes the step runnable and holds the stream until its rows are out
        got = mail_p[s].clone()
        b.step_device_packed(act, pb)
        _sync()
        assert torch.equal(got, pb), i
        finished += int(pb[:, 34].sum())
    assert finished >= 3 * n
    st = a.resident_status()
    assert st["rung"] == steps and st["not_executed"] == 0
    _same_state(a, b)                                 # get_state retires the kernel: the registers' state is in memory
    a.resident_stop(); a.close(); b.close()


def test_resident_run_ahead_rings_many_steps_at_once():
    """Rings of 16 env-steps over a 16-slot mailbox (the kernel runs ahead through slots filled beforehand), then per-launch steps
    on the same handle (which retire the kernel), then rings again: one continuous trajectory, bit-identical to a per-launch twin."""
    import torch
    n, slots = 4096, 16
    task = _task()
    a, b = _twin(n, task)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    mail_a = torch.zeros((slots, n, 12), device=dev)
    mail_p = torch.zeros((slots, n, 35), device=dev)
    pb = torch.empty((slots, n, 35), device=dev)
    one = torch.empty((n, 35), device=dev)
    a.resident_start(mail_a, mail_p)
    for rnd in range(6):
        mail_a.copy_(torch.rand((slots, n, 12), generator=gen, device=dev) * 2 - 1)
        a.resident_step(slots)
        for k in range(slots):
            b.step_device_packed(mail_a[k], pb[k])
        _sync()
        assert torch.equal(mail_p, pb), rnd
        if rnd == 2:                                  # a per-launch step in between: the resident kernel hands the state back first
            act = torch.rand((n, 12), generator=gen, device=dev)
            a.step_device_packed(act, one)
            b.step_device_packed(act, pb[0])
            _sync()
            assert torch.equal(one, pb[0])
            assert not a.resident_status()["running"]
            # the mailbox goes on where the resident sequence stood (96 env-steps rung so far: slot 0 again)
    _same_state(a, b)
    a.resident_stop(); a.close(); b.close()


def test_resident_kernel_leaves_by_itself_and_the_handle_stays_usable():
    """Nobody rings: within the idle time-out (here 1 ms) the kernel has stored its state and left -- seen from the host without any
    synchronisation -- and the handle is usable at once: state read-back, a ring (which launches it again), a reset."""
    import torch
    n = 4096
    task = _task()
    a, b = _twin(n, task)
    dev = torch.device("cuda:0")
    mail_a = torch.rand((1, n, 12), device=dev)
    mail_p = torch.zeros((1, n, 35), device=dev)
    pb = torch.empty((n, 35), device=dev)
    a.resident_start(mail_a, mail_p, idle_timeout_us=1000)
    assert a.resident_status()["running"]
    t0 = time.perf_counter()
    while a.resident_status()["running"] and time.perf_counter() - t0 < 1.0:
        time.sleep(0.0005)
    gone_after = time.perf_counter() - t0
    assert not a.resident_status()["running"] and gone_after < 0.010, gone_after          # within 10 ms, no doorbell ever rung
    t0 = time.perf_counter()
    _same_state(a, b)                                 # usable at once (nothing to wait for: the kernel is gone)
    assert time.perf_counter() - t0 < 0.010
    a.resident_step(1)                                # launches the kernel again, then rings
    b.step_device_packed(mail_a[0], pb)
    _sync()
    assert torch.equal(mail_p[0], pb)
    assert a.resident_status()["not_executed"] == 0
    a.reset(seed=9, flags=task.reset_flags); b.reset(seed=9, flags=task.reset_flags)       # retires it again
    a.resident_step(1)
    b.step_device_packed(mail_a[0], pb)
    _sync()
    assert torch.equal(mail_p[0], pb)
    _same_state(a, b)
    a.resident_stop(); a.close(); b.close()


def test_ring_that_meets_a_retired_kernel_runs_nothing_and_says_so():
    """A ring replayed from a hipGraph after the kernel has left (the host check of qg_resident_step_device is not part of a replay):
    the env-steps are NOT executed, the state stays that of the last executed step, and the next resident call reports it."""
    import torch
    n = 512
    task = _task()
    a, b = _twin(n, task)
    dev = torch.device("cuda:0")
    mail_a = torch.rand((1, n, 12), device=dev)
    mail_p = torch.zeros((1, n, 35), device=dev)
    a.resident_start(mail_a, mail_p, idle_timeout_us=500)
    side = torch.cuda.Stream(dev)
    _sync()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        a.resident_step(1, stream=torch.cuda.current_stream(dev))
    graph.replay(); _sync()          # (capture does not execute: this is env-step 0)
    pb = torch.empty((n, 35), device=dev)
    b.step_device_packed(mail_a[0], pb)
    _sync()
    assert torch.equal(mail_p[0], pb)
    time.sleep(0.02)                                  # the kernel leaves (0.5 ms idle) ...
    assert not a.resident_status()["running"]
    before = mail_p.clone()
    graph.replay(); _sync()          # ... and the replayed ring meets a retired door
    assert a.resident_status()["not_executed"] == 1
    assert torch.equal(mail_p, before)
    assert a.resident_status()["rung"] == 1          # (rings of a replayed graph do not go through the API: one counted so far)
    with pytest.raises(_abi.QuadGymError, match="NOT executed"):
        a.resident_step(1)
    _same_state(a, b)                                 # one env-step has run, not two
    a.resident_ensure()                               # what a caller does before replaying captured rings
    graph.replay(); _sync()
    b.step_device_packed(mail_a[0], pb); _sync()
    assert torch.equal(mail_p[0], pb)
    _same_state(a, b)
    a.resident_stop(); a.close(); b.close()


def test_resident_rings_replay_from_a_hipgraph():
    """A graph of 8 rings (one env-step each, the slot's actions rewritten by a captured copy in front of every ring) replayed 15
    times -- 120 env-steps without a host call per step -- against eager per-launch steps of a twin."""
    import torch
    n, G, R = 4096, 8, 15
    task = _task()
    a, b = _twin(n, task)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(4)
    acts = torch.rand((G, n, 12), generator=gen, device=dev) * 2 - 1
    mail_a = torch.zeros((1, n, 12), device=dev)
    mail_p = torch.zeros((1, n, 35), device=dev)
    rows = torch.zeros((G, n, 35), device=dev)
    pb = torch.empty((G, n, 35), device=dev)
    a.resident_start(mail_a, mail_p, idle_timeout_us=20000)
    side = torch.cuda.Stream(dev)
    _sync()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for g in range(G):
            mail_a[0].copy_(acts[g])
            a.resident_step(1, stream=torch.cuda.current_stream(dev))
            rows[g].copy_(mail_p[0])
    for rep in range(R):
        a.resident_ensure()
        graph.replay()
        for g in range(G):
            b.step_device_packed(acts[g], pb[g])
        _sync()
        assert torch.equal(rows, pb), rep
    assert a.resident_status()["not_executed"] == 0
    ep = _same_state(a, b)
    assert ep.min() >= 4
    a.resident_stop(); a.close(); b.close()


def test_resident_and_sequence_forms_refuse_what_they_do_not_cover():
    import torch
    from quadruped_gym_amd.sim import BatchedSim
    dev = torch.device("cuda:0")
    big, twin = BatchedSim(8192), BatchedSim(8192)   # AUTO = one leg per lane: the sequence call is then K per-step launches
    acts = torch.rand((3, 8192, 12), device=dev) * 2 - 1
    pa, pb = torch.zeros((3, 8192, 35), device=dev), torch.zeros((3, 8192, 35), device=dev)
    big.step_device_seq(acts, pa)
    for k in range(3):
        twin.step_device_packed(acts[k], pb[k])
    torch.cuda.synchronize()
    assert torch.equal(pa, pb)
    with pytest.raises(_abi.QuadGymError, match="one-link-per-lane"):
        big.resident_start(torch.zeros((1, 8192, 12), device=dev), torch.zeros((1, 8192, 35), device=dev))
    big.close(); twin.close()
    t = _abi.default_task()
    t.auto_reset = 1
    t.reset_flags = _abi.RESET_JOINT_JITTER
    jit = BatchedSim(64, task=t)
    with pytest.raises(_abi.QuadGymError, match="jitter"):
        jit.resident_start(torch.zeros((1, 64, 12), device=dev), torch.zeros((1, 64, 35), device=dev))
    jit.close()
    sim = BatchedSim(64)
    ma, mp = torch.zeros((2, 64, 12), device=dev), torch.zeros((2, 64, 35), device=dev)
    sim.resident_start(ma, mp)
    with pytest.raises(_abi.QuadGymError, match="slots"):
        sim.resident_step(3)
    with pytest.raises(_abi.QuadGymError, match="resident"):
        sim.set_mapping(_abi.MAP_QUAD)
    sim.resident_stop()
    sim.set_mapping(_abi.MAP_QUAD)
    sim.close()


def test_vec_env_sequence_steps_equal_single_steps():
    """``QuadrupedVecEnv.step_sequence_tensor``: K env-steps in one call (one launch up to 4096 envs, K launches above) leave the rows of
    K ``step_tensor`` calls, bit for bit, auto-resets included."""
    import torch
    from quadruped_gym_amd.envs.vec_env import QuadrupedVecEnv
    dev = torch.device("cuda:0")
    for n in (512, 6000):
        kw = dict(reward_fns={"forward": 1.0, "control_cost": -0.1, "alive_bonus": 1.0}, termination_fns={"fall": 0.05}, max_time=0.1)
        a, b = QuadrupedVecEnv(n, **kw), QuadrupedVecEnv(n, **kw)
        a.reset(); b.reset()
        acts = torch.rand((20, n, 12), device=dev) * 2 - 1
        rows = a.step_sequence_tensor(acts)
        ref = torch.stack([b.step_tensor(acts[k]).clone() for k in range(20)])
        torch.cuda.synchronize()
        assert torch.equal(rows, ref) and int(rows[:, :, -1].sum()) >= n
        a.close(); b.close()


def test_sequence_and_resident_forms_with_other_model_numbers():
    """Any other robot (here: one mass and one servo gain changed) runs the variants that stage the model tables in LDS
    (``qg_step_kernel_link_multi<BAKED = false, ..>``): the sequence launch and closed-loop rings against per-step launches of the same
    tables, bit for bit, through auto-resets."""
    _other_model_numbers(1500, True)
    _other_model_numbers(6000, False)


def _other_model_numbers(n, with_ring):
    import torch
    from quadruped_gym_amd.sim import BatchedSim
    K = 16
    model = _abi.default_model()
    model.body_mass[3] *= 1.25
    model.act_kp[1] = 85.0
    task = _task()
    sims = [BatchedSim(n, model=model, task=task) for _ in range(3 if with_ring else 2)]
    seq, ref = sims[0], sims[-1]
    ring = sims[1] if with_ring else None
    for s in sims:
        assert not s.baked and s.mapping == (_abi.MAP_LINK if n <= 4096 else _abi.MAP_QUAD)
        s.reset(seed=2, flags=task.reset_flags)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(6)
    mail_a = torch.zeros((1, n, 12), device=dev); mail_p = torch.zeros((1, n, 35), device=dev)
    if with_ring:
        ring.resident_start(mail_a, mail_p)
    finished = 0
    for rnd in range(4):
        acts = torch.rand((K, n, 12), generator=gen, device=dev) * 2 - 1
        ps = torch.empty((K, n, 35), device=dev); pr = torch.empty((K, n, 35), device=dev); pg = torch.empty((K, n, 35), device=dev)
        seq.step_device_seq(acts, ps)
        for k in range(K):
            ref.step_device_packed(acts[k], pr[k])
            if with_ring:
                mail_a[0].copy_(acts[k])
                ring.resident_step(1)
                pg[k].copy_(mail_p[0])
        _sync()
        assert torch.equal(ps, pr) and (not with_ring or torch.equal(pg, pr)), rnd
        finished += int(pr[:, :, -1].sum())
    assert finished >= 2 * n
    _same_state(seq, ref)
    if with_ring:
        assert ring.resident_status()["not_executed"] == 0
        _same_state(ring, ref)
        ring.resident_stop()
    for s in sims:
        s.close()
