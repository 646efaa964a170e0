"""Soak of the fused partially observable walking step: N envs x K env-steps of random actions with auto-resets, device commands and
random start poses; every observation stack must stay finite, rewards finite except where the reference itself yields NaN (unit() of
an exactly zero velocity) -- and nowhere at all with the fourth argument 0 (nan_direction=False, qg_walk_params.unit_zero = 1).
usage (GPU box): python tools/soak_po_gpu.py [n_envs] [steps] [frame_skip] [nan_direction: 1 | 0]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_gym_amd.envs.walking import POWalkingQuadrupedVecEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
fs = int(sys.argv[3]) if len(sys.argv) > 3 else 10
nan_direction = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
env = POWalkingQuadrupedVecEnv(n, obs_window=10, frame_skip=fs, max_time=4.0, random_controls=True, random_init=True, device_commands=True,
                               reset_options={"min_speed": 0.0, "max_speed": 0.5}, seed=7, nan_direction=nan_direction)
env.reset()
dev = torch.device("cuda:0")
obs = torch.empty((n, env.obs_dim), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, device=dev, dtype=torch.uint8)
gen = torch.Generator(device=dev); gen.manual_seed(3)
bad_obs = 0; nan_rew = 0; dones = 0; lo = float("inf"); hi = -float("inf")
for k in range(steps):
    a = torch.rand((n, 12), generator=gen, device=dev) * 2.4 - 1.2          # beyond the +-1 clip now and then
    env.step_tensor(a, obs, rew, done)
    if k % 50 == 0 or k == steps - 1:
        bad_obs += int((~torch.isfinite(obs)).sum())
        nan_rew += int(torch.isnan(rew).sum())
        dones += int(done.sum())
        f = rew[torch.isfinite(rew)]
        if f.numel():
            lo = min(lo, float(f.min())); hi = max(hi, float(f.max()))
print(f"PO soak (nan_direction={nan_direction}): {n} envs x {steps} env-steps (frame_skip {fs}): non-finite observation values {bad_obs}, NaN rewards {nan_rew} "
      f"(sampled every 50th step), finished episodes in the sampled steps {dones}, finite rewards in [{lo:.1f}, {hi:.1f}]")
assert bad_obs == 0
assert nan_direction or nan_rew == 0
env.close()
