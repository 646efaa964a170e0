import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from quadruped_gym_amd.envs.walking import POWalkingQuadrupedVecEnv, WalkingQuadrupedVecEnv
from quadruped_gym_amd.envs.vec_env import QuadrupedVecEnv
import cProfile, pstats
for name, mk in (("QuadrupedVecEnv", lambda n: QuadrupedVecEnv(n)), ("WalkingQuadrupedVecEnv", lambda n: WalkingQuadrupedVecEnv(n)),
                 ("POWalkingQuadrupedVecEnv(w=10, fs=10)", lambda n: POWalkingQuadrupedVecEnv(n, obs_window=10, frame_skip=10))):
    for n in (64, 4096):
        env = mk(n)
        env.reset()
        a = np.random.default_rng(0).uniform(-1, 1, (n, 12)).astype(np.float32)
        for _ in range(5):
            env.step(a)
        K = 50
        t0 = time.perf_counter()
        for _ in range(K):
            env.step(a)
        dt = (time.perf_counter() - t0) / K
        print(f"{name:40s} n={n:5d}: {dt * 1e6:9.1f} us per step(actions) = {n / dt / 1e6:8.2f} M env-steps/s")
        if n == 4096 and name.startswith("PO"):
            pr = cProfile.Profile(); pr.enable()
            for _ in range(20): env.step(a)
            pr.disable(); pstats.Stats(pr).sort_stats("cumtime").print_stats(8)
        env.close()
