"""PCIe-inclusive rate of the host-pointer entry points (qg_step / qg_walk_step / qg_po_step): NumPy actions in, NumPy obs /
reward / done out, one stream synchronise per step.  Not the bench's `value` (that one keeps everything resident in HBM)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (loads the HIP runtime the library links against)
from quadruped_gym_amd import _abi
from quadruped_gym_amd.sim import BatchedSim
from quadruped_gym_amd.envs.vec_env import QuadrupedVecEnv
from quadruped_gym_amd.envs.walking import WalkingQuadrupedVecEnv, POWalkingQuadrupedVecEnv

steps = 300
for n in (4096, 32768):
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-1, 1, (n, 12)).astype(np.float32) for _ in range(8)]
    task = _abi.default_task(); task.auto_reset = 1; task.use_fall = 1; task.fall_height = 0.05
    sim = BatchedSim(n, task=task); sim.reset()
    for k in range(20): sim.step(acts[k & 7])
    t0 = time.perf_counter()
    for k in range(steps): sim.step(acts[k & 7])
    dt = time.perf_counter() - t0
    print(f"qg_step (host buffers)            n={n:6d}: {dt / steps * 1e6:8.1f} us/step  {n * steps / dt / 1e6:7.1f} M env-steps/s")
    sim.close()
    for name, make in (("QuadrupedVecEnv.step (+infos)", lambda: QuadrupedVecEnv(n, reward_fns={"forward": 1.0, "control_cost": -0.1, "alive_bonus": 1.0}, termination_fns={"fall": 0.05})),
                       ("WalkingQuadrupedVecEnv.step", lambda: WalkingQuadrupedVecEnv(n)),
                       ("POWalkingQuadrupedVecEnv.step w10", lambda: POWalkingQuadrupedVecEnv(n, obs_window=10))):
        env = make(); env.reset()
        for k in range(5): env.step(acts[k & 7])
        m = 40 if n > 10000 else 100
        t0 = time.perf_counter()
        for k in range(m): env.step(acts[k & 7])
        dt = time.perf_counter() - t0
        print(f"{name:33s} n={n:6d}: {dt / m * 1e6:8.1f} us/step  {n * m / dt / 1e6:7.1f} M env-steps/s")
        env.close()
