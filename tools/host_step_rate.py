import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
from quadruped_gym_amd.sim import BatchedSim
from quadruped_gym_amd import _abi
for n in (1, 64, 4096):
    sim = BatchedSim(n); sim.reset(seed=0, flags=0)
    a = np.zeros((n, 12), np.float32)
    for _ in range(20): sim.step(a)
    K = 300
    t0 = time.perf_counter()
    for _ in range(K): sim.step(a)
    t_py = (time.perf_counter() - t0) / K
    lib = sim._lib
    obs = np.empty((n, 33), np.float32); rew = np.empty(n, np.float32); done = np.empty(n, np.uint8)
    t0 = time.perf_counter()
    for _ in range(K): lib.qg_step(sim._h, a.ctypes.data, obs.ctypes.data, rew.ctypes.data, done.ctypes.data, None)
    t_c = (time.perf_counter() - t0) / K
    print(f"n={n:5d}: BatchedSim.step {t_py*1e6:7.1f} us   raw qg_step {t_c*1e6:7.1f} us")
    sim.close()
