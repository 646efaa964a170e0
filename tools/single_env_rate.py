"""Per-step cost of the single-robot façades (the reference's own calling convention: one env, Python callables in the dicts).
usage (GPU box): python tools/single_env_rate.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_gym_amd.envs.quadruped import QuadrupedEnv  # noqa: E402
from quadruped_gym_amd.envs.walking import POWalkingQuadrupedEnv, WalkingQuadrupedEnv  # noqa: E402


def rate(name, env, K=300):
    env.reset()
    a = env.action_space.sample()
    for _ in range(20):
        env.step(a)
    t0 = time.perf_counter()
    for _ in range(K):
        out = env.step(a)
        if out[2]:
            env.reset()
    dt = (time.perf_counter() - t0) / K
    print(f"{name:58s} {dt * 1e6:8.1f} us per step = {1 / dt / 1e3:6.1f} k steps/s")
    env.close()


e = QuadrupedEnv("builtin", frame_skip=4)
rate("QuadrupedEnv, default dicts", e)
e = QuadrupedEnv("builtin", frame_skip=4)
e.reward_fns["forward"] = lambda: e.data.qvel[0]
e.reward_fns["control_cost"] = lambda: -0.1 * float(np.sum(np.square(e.data.ctrl)))
e.termination_fns["fall"] = lambda: e.data.qpos[2] < 0.05
rate("QuadrupedEnv, README lambdas (host-evaluated)", e)
rate("WalkingQuadrupedEnv", WalkingQuadrupedEnv(model_path="builtin", frame_skip=4))
rate("POWalkingQuadrupedEnv(obs_window=10, frame_skip=10)", POWalkingQuadrupedEnv(obs_window=10, model_path="builtin", frame_skip=10))
