"""Step rate of the VecEnv protocol (host arrays in and out, infos included), alone and followed by the walk SB3's rollout collection
does over `infos` every step (`for info in infos: info.get("episode")`).  usage (GPU box): python tools/vecenv_rate.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_gym_amd.envs.vec_env import QuadrupedVecEnv  # noqa: E402
from quadruped_gym_amd.envs.walking import POWalkingQuadrupedVecEnv, WalkingQuadrupedVecEnv  # noqa: E402


def sb3_walk(infos):
    got = 0
    for info in infos:
        if info.get("episode") is not None:
            got += 1
    return got


for name, mk in (("QuadrupedVecEnv", lambda n, m: QuadrupedVecEnv(n, infos_mode=m)),
                 ("WalkingQuadrupedVecEnv", lambda n, m: WalkingQuadrupedVecEnv(n, infos_mode=m)),
                 ("POWalkingQuadrupedVecEnv(w=10, fs=10)", lambda n, m: POWalkingQuadrupedVecEnv(n, obs_window=10, frame_skip=10, infos_mode=m))):
    for n in (64, 4096):
        for mode in ("lazy", "finished"):
            env = mk(n, mode)
            env.reset()
            a = np.random.default_rng(0).uniform(-1, 1, (n, 12)).astype(np.float32)
            for _ in range(5):
                env.step(a)
            K = 50
            t0 = time.perf_counter()
            for _ in range(K):
                env.step(a)
            dt = (time.perf_counter() - t0) / K
            t0 = time.perf_counter()
            for _ in range(K):
                sb3_walk(env.step(a)[3])
            dw = (time.perf_counter() - t0) / K
            print(f"{name:40s} n={n:5d} infos_mode={mode:8s}: step {dt * 1e6:8.1f} us = {n / dt / 1e6:6.2f} M env-steps/s;  "
                  f"step + SB3's walk over infos {dw * 1e6:8.1f} us = {n / dw / 1e6:6.2f} M")
            env.close()
