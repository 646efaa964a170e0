"""(GPU box: python tools/accel_census.py)  frame_skip-20 accelerometer error of the link kernel against the oracle, by contact census (GPU box)."""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from oracle import oracle as O
from quadruped_gym_amd import _abi
from quadruped_gym_amd.sim import BatchedSim
from make_golden import sample_states
model = O.default_model()
task = O.default_task(); task.frame_skip = 20; task.obs_mode = 1; task.use_fall = 1; task.fall_height = 0.05
n = 1024
qpos, qvel, act, nstep = sample_states(model, task, n, seed=123)
rng = np.random.default_rng(9)
actions = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
b = O.Batch(model, task, n)
b.set_state(qpos.astype(np.float64), qvel.astype(np.float64), act.astype(np.float64), None, nstep)
obs_o, _, done_o, _ = b.step(actions.astype(np.float64))
gt = _abi.default_task(); gt.frame_skip = 20; gt.obs_mode = 1; gt.use_fall = 1; gt.fall_height = 0.05
sim = BatchedSim(n, task=gt); sim.set_mapping(_abi.MAP_LINK)
sim.set_state(qpos, qvel, act, None, nstep)
obs, _, done, _ = sim.step(actions)
err = np.abs(obs[:, 12:15] - obs_o[:, 12:15]).max(axis=1)
c = O.contact_census(model, 20, qpos, qvel, act, nstep, actions, extra=1)
for before in (0, 1, 2, 4, 8, 19):
    lo = max(0, 19 - before)
    win = c[:, lo:]
    sw = (win != win[:, :1]).any(axis=(1, 2))
    st = ~sw
    print(f"window last {before}+1+1 substeps: steady {st.sum():4d} max err {err[st].max():.4f}  99% {np.quantile(err[st], 0.99):.4f} | switching {sw.sum():4d} max err {err[sw].max() if sw.any() else 0:.4f}")
order = np.argsort(-err)[:10]
print("largest errors:", [(int(i), float(err[i])) for i in order])
