"""Where the time of ONE launch of the one-link-per-lane kernel goes, from inside the kernel: the first wave of the grid stamps the
100 MHz clock at its phase marks (QG_MARK in qg_kernel_link.hip; development build only).
usage (GPU box):  make -C quadruped-gym_amd/csrc clean && make -C quadruped-gym_amd/csrc CXXFLAGS+=-DQG_PHASE_TIMES  (see tools/phase_times.sh)
                  python tools/phase_times.py [plain|walking|po] [n_envs] [frame_skip]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_gym_amd import _abi  # noqa: E402
from quadruped_gym_amd._abi import check  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
fs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lib = _abi.load_library()
dev = torch.device("cuda:0")
acts = [torch.rand((n, 12), device=dev) * 2 - 1 for _ in range(8)]
NAMES = ["entry", "state loaded", "physics done", "obs written", "channel sums", "reward", "reset block", "frame built", "rows written",
         "stores issued"]
if mode == "plain":
    from quadruped_gym_amd.sim import BatchedSim
    t = _abi.default_task(); t.frame_skip = fs
    sim = BatchedSim(n, task=t)
    packed = torch.empty((n, 35), device=dev)
    step = lambda k: sim.step_device_packed(acts[k % 8], packed)
elif mode == "walking":
    from quadruped_gym_amd.envs.walking import WalkingQuadrupedVecEnv
    env = WalkingQuadrupedVecEnv(n, frame_skip=fs, max_time=20.0)
    env.reset()
    obs = torch.empty((n, 33), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, device=dev, dtype=torch.uint8)
    comps = torch.empty((n, 11), device=dev)
    step = lambda k: env.step_tensor(acts[k % 8], obs, rew, done, comps)
else:
    from quadruped_gym_amd.envs.walking import POWalkingQuadrupedVecEnv
    env = POWalkingQuadrupedVecEnv(n, obs_window=10, frame_skip=fs, max_time=20.0, random_controls=True, device_commands=True)
    env.reset()
    obs = torch.empty((n, env.obs_dim), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, device=dev, dtype=torch.uint8)
    step = lambda k: env.step_tensor(acts[k % 8], obs, rew, done)
for k in range(300):
    step(k)
torch.cuda.synchronize()
acc = np.zeros(16)
reps = 50
for rpt in range(reps):
    step(rpt)
    out = (C.c_uint64 * 16)()
    check(lib.qg_debug_phase_times(out), "qg_debug_phase_times")
    v = np.array(list(out), dtype=np.float64)
    acc += (v - v[0]) * 10.0          # ns since the wave's entry
acc /= reps
print(f"{mode}, {n} envs, frame_skip {fs}: ns since the first wave's entry (mean of {reps} launches)")
prev = 0.0
for i, nm in enumerate(NAMES):
    if acc[i] <= 0 and i > 0:
        continue
    print(f"  {nm:14s} {acc[i]:9.0f} ns   (+{acc[i] - prev:7.0f})")
    prev = acc[i]
