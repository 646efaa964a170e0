"""Soak of the resident / sequence forms of the step against a per-launch twin: random interleavings of closed-loop rings, run-ahead
rings, idle gaps longer than the time-out (the kernel leaves and is launched again), state read-backs, masked resets, per-launch steps
on the same handle, sequence launches and task changes.  After EVERY operation the rows both handles produced must be equal bit for
bit; the full state and the episode counters are compared every few hundred operations.  No ring may be lost, nothing may hang.
usage (GPU box): python tools/soak_resident_gpu.py [n_envs] [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_gym_amd import _abi  # noqa: E402
from quadruped_gym_amd.sim import BatchedSim  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
SLOTS = 8


def task(max_time=0.3, fs=4):
    t = _abi.default_task()
    t.frame_skip, t.auto_reset, t.max_time, t.use_fall, t.fall_height = fs, 1, max_time, 1, 0.05
    t.reset_flags = _abi.RESET_RANDOM_YAW
    return t


a, b = BatchedSim(n, task=task()), BatchedSim(n, task=task())
a.reset(seed=1, flags=1); b.reset(seed=1, flags=1)
mail_a = torch.zeros((SLOTS, n, 12), device=dev)
mail_p = torch.zeros((SLOTS, n, 35), device=dev)
pb = torch.zeros((SLOTS, n, 35), device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(seed)
sync = lambda: torch.cuda.current_stream().synchronize()
a.resident_start(mail_a, mail_p, idle_timeout_us=500)
rung = 0                        # env-steps rung since resident_start: the next one uses slot rung % SLOTS
ops = {k: 0 for k in ("ring1", "ringm", "idle", "state", "reset", "launch", "seq", "task")}
steps = relaunch = 0
t_end = time.perf_counter() + seconds
t_report = time.perf_counter() + 30.0
it = 0
while time.perf_counter() < t_end:
    it += 1
    op = rng.choice(["ring1", "ring1", "ring1", "ringm", "ringm", "idle", "state", "reset", "launch", "seq", "task"])
    ops[op] += 1
    if op in ("ring1", "ringm"):
        m = 1 if op == "ring1" else int(rng.integers(2, SLOTS + 1))
        acts = torch.rand((m, n, 12), generator=gen, device=dev) * 3 - 1.5
        sl = [(rung + i) % SLOTS for i in range(m)]
        for i, s in enumerate(sl):
            mail_a[s].copy_(acts[i])
        if not a.resident_status()["running"]:
            relaunch += 1
        a.resident_step(m)
        for i in range(m):
            b.step_device_packed(acts[i], pb[i])
        sync()
        for i, s in enumerate(sl):
            assert torch.equal(mail_p[s], pb[i]), (it, op, i, a.resident_status())
        rung += m; steps += m
    elif op == "idle":
        time.sleep(float(rng.uniform(0.0002, 0.003)))            # around and beyond the 0.5 ms idle time-out
    elif op == "state":
        for x, y in zip(a.get_state(), b.get_state()):
            assert np.array_equal(x, y), (it, op)
    elif op == "reset":
        mask = (rng.random(n) < 0.1).astype(np.uint8)
        a.reset(mask=mask, flags=1); b.reset(mask=mask, flags=1)
    elif op == "launch":
        act = torch.rand((n, 12), generator=gen, device=dev) * 2 - 1
        a.step_device_packed(act, mail_p[0]); b.step_device_packed(act, pb[0])
        sync()
        assert torch.equal(mail_p[0], pb[0]), (it, op)
        steps += 1
    elif op == "seq":
        m = int(rng.integers(1, SLOTS + 1))
        acts = torch.rand((m, n, 12), generator=gen, device=dev) * 2 - 1
        out = torch.empty((m, n, 35), device=dev)
        a.step_device_seq(acts, out)
        for i in range(m):
            b.step_device_packed(acts[i], pb[i])
        sync()
        assert torch.equal(out, pb[:m]), (it, op)
        steps += m
    elif op == "task":
        t = task(max_time=float(rng.choice([0.2, 0.3, 0.5])), fs=int(rng.choice([2, 4, 4, 6])))
        a.set_task(t); b.set_task(t)
    if time.perf_counter() > t_report:
        t_report += 30.0
        print(f"  ... {it} operations, {steps} env-steps, {relaunch} launches after leaving, rings not executed {a.resident_status()['not_executed']}", flush=True)
    if it % 300 == 0:
        for x, y in zip(a.get_state(), b.get_state()):
            assert np.array_equal(x, y), (it, "periodic state")
        ea, eb = a.get_reset_streams(), b.get_reset_streams()
        assert np.array_equal(ea[0], eb[0])
st = a.resident_status()
assert st["not_executed"] == 0 and st["rung"] == rung, st
for x, y in zip(a.get_state(), b.get_state()):
    assert np.array_equal(x, y)
print(f"resident soak: {n} envs, {seconds:.0f} s, {it} operations {ops}, {steps} env-steps, kernel launched again {relaunch} times after leaving; "
      f"every row, state and episode counter equal to the per-launch twin; rings not executed: {st['not_executed']}")
a.resident_stop(); a.close(); b.close()
