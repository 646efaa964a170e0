"""Soak: N envs x K random-action env-steps with only the time limit enabled; every `done` must fall on the time-limit
grid -- anything else is the divergence guard (non-finite / huge state).  Also tracks height / speed extremes."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quadruped_gym_amd import _abi
from quadruped_gym_amd.sim import BatchedSim

n, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 5000
task = _abi.default_task(); task.auto_reset = 1; task.reset_flags = _abi.RESET_RANDOM_YAW
sim = BatchedSim(n, task=task); sim.set_track_ctrl(False); sim.reset(seed=1, flags=task.reset_flags)
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(7)
packed = torch.empty((n, 35), device=dev)
bad = 0; zmin = 1e9; zmax = -1e9; vmax = 0.0
hold = torch.rand((n, 12), generator=gen, device=dev) * 2 - 1
for k in range(1, steps + 1):
    if k % 5 == 0:      # new random targets at 25 Hz, beyond the clip range sometimes
        hold = torch.rand((n, 12), generator=gen, device=dev) * 2.6 - 1.3
    sim.step_device_packed(hold, packed)
    if k % 50 == 0 or k % 1250 == 0:
        p = packed.cpu().numpy()
        d = p[:, 34] > 0.5
        if k % 1250 == 0:
            assert d.all(), (k, int(d.sum()))
        else:
            bad += int(d.sum())
        assert np.isfinite(p).all(), k
        zmin = min(zmin, p[:, 20].min()); zmax = max(zmax, p[:, 20].max()); vmax = max(vmax, np.abs(p[:, 21:24]).max())
print(f"soak ok: {n} envs x {steps} steps, off-grid dones (divergence) = {bad}, base z in [{zmin:.3f}, {zmax:.3f}], max |v| = {vmax:.2f} m/s")
assert bad == 0
