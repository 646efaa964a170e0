"""``QuadrupedVecEnv`` -- batched counterpart of the reference's ``SubprocVecEnv([...])`` of
``QuadrupedEnv`` instances (``src/train_quadruped.py:49-50``), with the Stable-Baselines3 VecEnv
calling convention: ``reset() -> obs[N, D]``, ``step(actions) -> (obs, rewards, dones, infos)``,
auto-reset of finished envs with ``infos[i]["terminal_observation"]``.

All N robots advance in one kernel launch.  The README reward / termination set
(``README.md:64-90``) runs on the device as named built-ins:

    reward_fns      = {"forward": 1.0, "control_cost": -0.1, "alive_bonus": 1.0}   # name -> weight
    termination_fns = {"fall": 0.2}                                                  # name -> threshold

Arbitrary Python callables cannot run on the GPU: they stay available through the single-robot
``QuadrupedEnv`` (host evaluation, as in the reference); the batched env rejects them up front.
"""
from __future__ import annotations

import numpy as np

from .. import _abi
from ..model.loader import load_model
from ..sim import BatchedSim
from .quadruped import ModelView
from .spaces import Box

_REWARD_BUILTINS = ("forward", "control_cost", "alive_bonus")


class BatchedData:
    """Batched host view of the state: ``qpos [N,19]``, ``qvel [N,18]``, ``act``, ``ctrl``, ``time [N]``, ``sensordata``."""

    def __init__(self, n, obs_dim):
        self.qpos = np.zeros((n, 19), np.float32)
        self.qvel = np.zeros((n, 18), np.float32)
        self.act = np.zeros((n, 12), np.float32)
        self.ctrl = np.zeros((n, 12), np.float32)
        self.time = np.zeros(n)
        self.sensordata = np.zeros((n, obs_dim), np.float32)


class QuadrupedVecEnv:
    def __init__(self, num_envs: int, model_path: str | None = "builtin", max_time: float = 10.0, frame_skip: int = 4,
                 reward_fns: dict | None = None, termination_fns: dict | None = None, use_default_termination: bool = True,
                 obs_mode: int = _abi.OBS_FULL, random_init: bool = False, device: int = 0, env_index_base: int = 0,
                 seed: int = 0):
        qg_model, layout = load_model(model_path)
        self.model = ModelView(qg_model, layout)
        self.num_envs = int(num_envs)
        self.frame_skip = int(frame_skip)
        self.max_time = float(max_time)
        task = _abi.default_task()
        task.frame_skip = self.frame_skip
        task.max_time = self.max_time
        task.use_time_limit = 1 if use_default_termination else 0
        task.obs_mode = obs_mode
        task.auto_reset = 1
        task.reset_flags = _abi.RESET_RANDOM_YAW if random_init else 0
        self._host_rewards, self._host_terms = {}, {}
        rf = {"forward": 0.0, "control_cost": 0.0, "alive_bonus": 0.0}
        for name, val in (reward_fns or {}).items():
            if callable(val):
                self._host_rewards[name] = val
            elif name in _REWARD_BUILTINS:
                rf[name] = float(val)
            else:
                raise ValueError(f"unknown built-in reward {name!r}; pass a callable fn(vec_env) -> array[N] instead")
        task.w_forward, task.w_ctrl, task.alive_bonus = rf["forward"], rf["control_cost"], rf["alive_bonus"]
        task.use_fall = 0
        for name, val in (termination_fns or {}).items():
            if callable(val):
                self._host_terms[name] = val
            elif name == "fall":
                task.use_fall, task.fall_height = 1, float(val)
            else:
                raise ValueError(f"unknown built-in termination {name!r}; pass a callable fn(vec_env) -> bool array[N]")
        if self._host_rewards or self._host_terms:
            raise NotImplementedError("Python reward/termination callables run through QuadrupedEnv (one robot, host "
                                      "evaluation as in the reference); QuadrupedVecEnv runs the named built-ins "
                                      f"{_REWARD_BUILTINS} / 'fall' on the device")
        self.reward_keys = [k for k in _REWARD_BUILTINS if k in (reward_fns or {})]
        self._sim = BatchedSim(self.num_envs, device=device, model=qg_model, task=task, env_index_base=env_index_base)
        self._reset_flags = task.reset_flags
        self._seed = int(seed)
        self.obs_dim = self._sim.obs_dim
        self.action_space = Box(low=-1.0, high=1.0, shape=(12,), dtype=np.float32)
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.obs_dim,), dtype=np.float32)
        self.data = BatchedData(self.num_envs, self.obs_dim)
        self._actions = None
        self.render_mode = None

    # -- SB3 VecEnv protocol ------------------------------------------------------------------------
    def reset(self):
        self._sim.reset(seed=self._seed, flags=self._reset_flags)
        self.data.time[:] = 0.0
        return np.zeros((self.num_envs, self.obs_dim), np.float32)      # the reference's first obs is all zeros

    def step_async(self, actions):
        self._actions = np.asarray(actions, dtype=np.float32)

    def step_wait(self):
        obs, rew, done, comps = self._sim.step(self._actions, want_components=True)
        names = _REWARD_BUILTINS
        infos = []
        for row in comps.tolist():                          # one C-level conversion; per-element float() costs 4x as much
            rc = dict(zip(names, row))
            info = dict(rc)
            info["reward_components"] = rc
            infos.append(info)
        for i in np.nonzero(done)[0]:
            infos[i]["terminal_observation"] = obs[i].copy()
            infos[i]["TimeLimit.truncated"] = False         # the reference reports the time limit as `terminated`
        obs = obs.copy()
        obs[done] = 0.0                                    # envs that finished were reset: their next obs is the reset obs
        return obs, rew, done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if getattr(self, "_sim", None) is not None:
            self._sim.close()
            self._sim = None

    def seed(self, seed=None):
        self._seed = 0 if seed is None else int(seed)
        return [self._seed + i for i in range(self.num_envs)]

    def get_attr(self, attr_name, indices=None):
        idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
        return [getattr(self, attr_name) for _ in idx]

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
        return [getattr(self, method_name)(*args, **kwargs) for _ in idx]

    def env_is_wrapped(self, wrapper_class, indices=None):
        idx = range(self.num_envs) if indices is None else ([indices] if isinstance(indices, int) else indices)
        return [False for _ in idx]

    # -- zero-copy path for policies that live on the GPU --------------------------------------------------
    def step_tensor(self, actions, packed=None, stream=None):
        """``actions``: float32 CUDA tensor ``[N, 12]``; returns the packed ``[N, obs_dim + 2]`` tensor
        (obs, reward, done) written by the kernel -- nothing touches the host."""
        import torch
        if packed is None:
            packed = torch.empty((self.num_envs, self.obs_dim + 2), device=actions.device, dtype=torch.float32)
        self._sim.step_device_packed(actions, packed, stream=stream)
        return packed

    def sync_data(self):
        """Refresh the batched host view ``self.data`` from the device."""
        qpos, qvel, act, ctrl, nstep = self._sim.get_state()
        self.data.qpos, self.data.qvel, self.data.act, self.data.ctrl = qpos, qvel, act, ctrl
        self.data.time = nstep.astype(np.float64) * self.model.opt.timestep
        return self.data
