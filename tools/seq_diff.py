"""Bit-for-bit comparison of one sequence launch (qg_step_device_seq) against per-step launches: where and by how much they differ
(GPU box: python tools/seq_diff.py)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from quadruped_gym_amd import _abi
from quadruped_gym_amd.sim import BatchedSim
n, K = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 8
t = _abi.default_task(); t.auto_reset = 1; t.max_time = 0.2; t.use_fall = 1; t.fall_height = 0.05; t.reset_flags = 1
import os
t.w_forward = float(os.environ.get("QG_W_FORWARD", "1.0")); t.w_ctrl = float(os.environ.get("QG_W_CTRL", "-0.1")); t.alive_bonus = float(os.environ.get("QG_ALIVE", "1.0"))
a, b = BatchedSim(n, task=t), BatchedSim(n, task=t)
a.reset(seed=5, flags=1); b.reset(seed=5, flags=1)
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(1)
acts = torch.rand((K, n, 12), generator=gen, device=dev) * 3 - 1.5
pa = torch.empty((K, n, 35), device=dev); pb = torch.empty((K, n, 35), device=dev)
a.step_device_seq(acts, pa)
for k in range(K): b.step_device_packed(acts[k], pb[k])
torch.cuda.synchronize()
A, B = pa.cpu().numpy(), pb.cpu().numpy()
for k in range(K):
    d = np.abs(A[k] - B[k])
    bad = np.argwhere(d > 0)
    print("step", k, "differing entries", len(bad), "max", d.max(), "cols", sorted(set(bad[:, 1].tolist()))[:40], "envs", len(set(bad[:, 0].tolist())))
sa, sb = a.get_state(), b.get_state()
for name, x, y in zip(("qpos", "qvel", "act", "ctrl", "nstep"), sa, sb):
    d = np.abs(x.astype(np.float64) - y)
    print(name, "max diff", d.max(), "cols", sorted(set(np.argwhere(d > 0)[:, -1].tolist())) if d.ndim > 1 else int((d > 0).sum()))
