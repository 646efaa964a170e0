"""Randomised API sequences against the C ABI (through BatchedSim): random batch sizes, mappings and task flags, then a random walk
over step / step_device_packed on side streams / masked reset / set_state / get_state / set_task / set_mapping.  Checked after every
operation that reads the state back: everything finite, the per-env substep counters equal a host-side model of them, quaternions of
unit length, an auto-reset env stands at its start pose.  usage (GPU box): python tools/fuzz_api_gpu.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_gym_amd import _abi  # noqa: E402
from quadruped_gym_amd.sim import BatchedSim  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
MAPS = [_abi.MAP_AUTO, _abi.MAP_LANE, _abi.MAP_QUAD, _abi.MAP_PAIR, _abi.MAP_LINK]
t_end = time.time() + budget
handles = ops = 0
while time.time() < t_end:
    n = int(rng.choice([1, 3, 17, 64, 130, 1000, 4096, 4097, 20000]))
    task = _abi.default_task()
    task.frame_skip = int(rng.choice([1, 2, 4, 7, 20]))
    task.use_fall = int(rng.integers(0, 2)); task.fall_height = 0.05
    task.auto_reset = 1
    task.obs_mode = int(rng.integers(0, 2))
    task.sensor_lag = int(rng.integers(0, 4) > 0)
    task.reset_flags = int(rng.integers(0, 4))
    task.reset_joint_jitter = 0.1
    task.max_time = float(rng.choice([0.05, 0.3, 10.0]))
    sim = BatchedSim(n, task=task, env_index_base=int(rng.integers(0, 1 << 20)))
    handles += 1
    fs = task.frame_skip
    sim.set_mapping(int(rng.choice(MAPS)))
    sim.reset(seed=int(rng.integers(0, 1 << 30)), flags=task.reset_flags)
    model_nstep = np.zeros(n, np.int64)
    limit = None
    side = torch.cuda.Stream(dev)
    od = sim.obs_dim
    for _ in range(int(rng.integers(5, 40))):
        op = rng.choice(["step", "dev", "mask_reset", "state", "mapping", "check"], p=[0.35, 0.25, 0.1, 0.1, 0.05, 0.15])
        ops += 1
        if op == "step":
            a = rng.uniform(-1.3, 1.3, (n, 12)).astype(np.float32)
            obs, rew, done, _ = sim.step(a)
            assert obs.shape == (n, od) and np.isfinite(obs).all() and np.isfinite(rew).all()
            model_nstep = np.where(done, 0, model_nstep + fs)
        elif op == "dev":
            k = int(rng.integers(1, 6))
            acts = torch.from_numpy(rng.uniform(-1, 1, (n, 12)).astype(np.float32)).to(dev)
            packed = torch.empty((n, od + 2), device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            for _k in range(k):
                sim.step_device_packed(acts, packed, stream=side)
                side.synchronize()
                d = packed[:, od + 1].cpu().numpy() > 0.5
                model_nstep = np.where(d, 0, model_nstep + fs)
        elif op == "mask_reset":
            mask = (rng.random(n) < 0.3).astype(np.uint8)
            sim.reset(mask=mask, flags=task.reset_flags)
            model_nstep = np.where(mask > 0, 0, model_nstep)
        elif op == "state":
            qpos, qvel, act, ctrl, nstep = sim.get_state()
            sim.set_state(qpos, qvel, act, ctrl, nstep)
        elif op == "mapping":
            try:
                sim.set_mapping(int(rng.choice(MAPS)))
            except _abi.QuadGymError:
                pass
        else:
            qpos, qvel, act, ctrl, nstep = sim.get_state()
            assert np.isfinite(qpos).all() and np.isfinite(qvel).all() and np.isfinite(act).all() and np.isfinite(ctrl).all()
            assert np.array_equal(nstep.astype(np.int64), model_nstep), (n, fs, np.flatnonzero(nstep != model_nstep)[:5])
            qn = np.linalg.norm(qpos[:, 3:7], axis=1)
            assert np.allclose(qn, 1.0, atol=1e-4), float(np.abs(qn - 1).max())
            fresh = model_nstep == 0
            if fresh.any() and not (task.reset_flags & 2):
                assert np.allclose(qpos[fresh, 7:], np.array(sim.model.qpos0[7:19], np.float32)[None], atol=1e-6)
    sim.close()
print(f"fuzz ok: {handles} handles, {ops} operations in {budget:.0f} s")
