"""``QuadrupedVecEnv`` -- batched counterpart of the reference's ``SubprocVecEnv([...])`` of
``QuadrupedEnv`` instances (``src/train_quadruped.py:49-50``), with the Stable-Baselines3 VecEnv
calling convention: ``reset() -> obs[N, D]``, ``step(actions) -> (obs, rewards, dones, infos)``,
auto-reset of finished envs with ``infos[i]["terminal_observation"]``.

All N robots advance in one kernel launch.  ``reward_fns`` / ``termination_fns`` are the reference's
mutable dicts (``src/envs/quadruped.py:97-100,170-178``; README.md:64-90) and may hold

* **named built-ins** evaluated on the device: ``{"forward": w, "control_cost": w, "alive_bonus": w}``
  (name -> weight) and ``{"fall": z_threshold}``; the default time-limit termination
  (``quadruped.py:99-100,149-151``) sits in ``termination_fns["default"]`` exactly as in the reference
  and is removed by deleting that key;
* **zero-argument Python callables**, exactly what the reference's dicts hold
  (``env.reward_fns["x"] = lambda: f(env)``).  They cannot run on the GPU, so they are evaluated on the
  host after the kernel, over ``env.data`` -- the view the reference's lambdas read -- and are added to
  the device reward / OR-ed into ``done``; envs they finish are reset with a masked reset.  Two modes:

  - ``callable_mode="per_env"`` (default, exact): every callable is called once per env with
    ``env.data`` showing that env alone (``data.qpos`` is ``float64[19]`` ...), i.e. precisely the
    reference's semantics for ANY lambda, at N Python calls per callable and step;
  - ``callable_mode="batched"``: every callable is called ONCE per step; ``env.data.qpos`` is a
    field-major ``[19, N]`` array whose last axis is the env axis and whose reductions keep that axis
    (``EnvAxisArray``), so reference-style code -- ``env.data.qvel[0]``,
    ``-0.1 * np.sum(np.square(env.data.ctrl))``, ``env.data.qpos[2] < 0.2`` -- yields one value per env
    unchanged.

Both dicts may be edited at any time; the device task follows at the next step (``qg_set_task``).  The
zero-copy ``step_tensor`` path runs the device built-ins only.
"""
from __future__ import annotations

import numpy as np

from .. import _abi
from ..model.loader import load_model
from ..sim import BatchedSim
from .infos import LazyInfos, finished_only_infos
from .quadruped import ModelView
from .spaces import Box

try:  # pragma: no cover - depends on the host (stable_baselines3 is absent from the build image)
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase
    HAVE_SB3 = True
except Exception:
    _VecEnvBase = object
    HAVE_SB3 = False

_REWARD_BUILTINS = ("forward", "control_cost", "alive_bonus")
_REDUCTIONS = ("sum", "mean", "prod", "max", "min", "any", "all", "std", "var")


class EnvAxisArray(np.ndarray):
    """An array whose LAST axis runs over the envs of the batch.  Reductions called without ``axis`` reduce every
    other axis and keep the env axis, so an expression written for one robot's ``data`` -- ``np.sum(np.square(ctrl))``,
    ``np.linalg.norm(qvel[0:3])``, ``np.dot(a, b)`` -- evaluates to one value per env on the batched view."""

    def __array_finalize__(self, obj):
        pass

    def _other_axes(self):
        return tuple(range(self.ndim - 1))

    def _reduce(self, name, axis, kwargs):
        base = self.view(np.ndarray)
        if axis is None:
            axis = self._other_axes()
            if not axis:                              # already one value per env
                return base.copy() if name not in ("any", "all") else base.astype(bool)
        return getattr(base, name)(axis=axis, **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        name = getattr(func, "__name__", "")
        plain = [a.view(np.ndarray) if isinstance(a, EnvAxisArray) else a for a in args]
        if name in ("sum", "mean", "prod", "amax", "amin", "max", "min", "any", "all", "std", "var") and kwargs.get("axis") is None \
                and len(args) == 1:
            kw = {k: v for k, v in kwargs.items() if k != "axis"}
            return args[0]._reduce({"amax": "max", "amin": "min"}.get(name, name), None, kw)
        if name == "norm" and kwargs.get("axis") is None and len(args) == 1 and kwargs.get("ord") in (None, 2):
            x = plain[0]
            return np.sqrt((x * x).sum(axis=tuple(range(x.ndim - 1)))) if x.ndim > 1 else np.abs(x)
        if name in ("dot", "inner", "vdot") and len(args) == 2 and all(np.ndim(a) == 2 for a in plain):
            return np.einsum("in,in->n", plain[0], plain[1])
        out = func(*plain, **{k: (v.view(np.ndarray) if isinstance(v, EnvAxisArray) else v) for k, v in kwargs.items()})
        return out.view(EnvAxisArray) if isinstance(out, np.ndarray) and out.ndim >= 1 and out.shape[-1] == self.shape[-1] else out


def _make_reduction(name):
    def method(self, axis=None, dtype=None, out=None, keepdims=False, **kw):
        extra = dict(kw)
        if dtype is not None:
            extra["dtype"] = dtype
        if keepdims:
            extra["keepdims"] = keepdims
        return self._reduce(name, axis, extra)
    method.__name__ = name
    return method


for _name in _REDUCTIONS:
    setattr(EnvAxisArray, _name, _make_reduction(_name))


class BatchedData:
    """Env-major host snapshot of the state (``sync_data()``): ``qpos [N,19]``, ``qvel [N,18]``, ``act``, ``ctrl``,
    ``time [N]``, ``sensordata [N,D]``."""

    def __init__(self, n, obs_dim):
        self.qpos = np.zeros((n, 19), np.float32)
        self.qvel = np.zeros((n, 18), np.float32)
        self.act = np.zeros((n, 12), np.float32)
        self.ctrl = np.zeros((n, 12), np.float32)
        self.time = np.zeros(n)
        self.sensordata = np.zeros((n, obs_dim), np.float32)


class CallableDataView:
    """What reward / termination callables read as ``env.data``: the fields of the reference's ``MjData`` the env exposes
    (``qpos, qvel, act, ctrl, time, sensordata``; ``quadruped.py:121-124,141-143``), float64 as there.  Field-major storage
    ``[width, N]``; with a cursor set (per-env evaluation) every attribute shows that env alone, without one the whole
    batch as an ``EnvAxisArray``."""

    _FIELDS = ("qpos", "qvel", "act", "ctrl", "sensordata")

    def __init__(self, n, obs_dim):
        self._store = {"qpos": np.zeros((19, n)), "qvel": np.zeros((18, n)), "act": np.zeros((12, n)), "ctrl": np.zeros((12, n)),
                       "sensordata": np.zeros((obs_dim, n))}
        self._time = np.zeros(n)
        self._cursor = None

    def __getattr__(self, name):
        if name in CallableDataView._FIELDS:
            a = self._store[name]
            return np.ascontiguousarray(a[:, self._cursor]) if self._cursor is not None else a.view(EnvAxisArray)
        if name == "time":
            return float(self._time[self._cursor]) if self._cursor is not None else self._time.view(EnvAxisArray)
        raise AttributeError(name)

    def load(self, qpos, qvel, act, ctrl, nstep, sensordata, timestep):
        s = self._store
        s["qpos"][:] = qpos.T; s["qvel"][:] = qvel.T; s["act"][:] = act.T; s["ctrl"][:] = ctrl.T
        s["sensordata"][:] = sensordata.T
        self._time[:] = nstep.astype(np.float64) * timestep


class QuadrupedVecEnv(_VecEnvBase):
    def __init__(self, num_envs: int, model_path: str | None = "builtin", max_time: float = 10.0, frame_skip: int = 4,
                 reward_fns: dict | None = None, termination_fns: dict | None = None, use_default_termination: bool = True,
                 obs_mode: int = _abi.OBS_FULL, random_init: bool = False, device: int = 0, env_index_base: int = 0,
                 seed: int = 0, callable_mode: str = "per_env", infos_mode: str = "lazy"):
        if callable_mode not in ("per_env", "batched"):
            raise ValueError("callable_mode must be 'per_env' or 'batched'")
        if infos_mode not in ("lazy", "finished"):
            raise ValueError("infos_mode must be 'lazy' or 'finished'")
        self.infos_mode = infos_mode
        qg_model, layout = load_model(model_path)
        self.model = ModelView(qg_model, layout)
        self.num_envs = int(num_envs)
        self.frame_skip = int(frame_skip)
        self.max_time = float(max_time)
        self.callable_mode = callable_mode
        # the reference's dicts (quadruped.py:97-100): mutable, looked at again at every step
        self.reward_fns = dict(reward_fns) if reward_fns is not None else {"default": self._default_reward}
        self.termination_fns = dict(termination_fns) if termination_fns is not None else {}
        if use_default_termination:
            self.termination_fns["default"] = self._default_termination
        self._reset_flags = _abi.RESET_RANDOM_YAW if random_init else 0
        self._obs_mode = obs_mode
        task, self._host_rewards, self._host_terms = self._plan()
        self._sim = BatchedSim(self.num_envs, device=device, model=qg_model, task=task, env_index_base=env_index_base)
        self._task_key = self._key(task)
        self._seed = int(seed)
        self.obs_dim = self._sim.obs_dim
        self.action_space = Box(low=-1.0, high=1.0, shape=(12,), dtype=np.float32)
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.obs_dim,), dtype=np.float32)
        if HAVE_SB3:  # pragma: no cover
            _VecEnvBase.__init__(self, self.num_envs, self.observation_space, self.action_space)
        self.data = CallableDataView(self.num_envs, self.obs_dim)
        self._snapshot = BatchedData(self.num_envs, self.obs_dim)
        self._actions = None
        self.render_mode = None

    # -- the reference's defaults (quadruped.py:145-151): markers here, evaluated on the device ---------------------------
    def _default_reward(self):
        return 0.0

    def _default_termination(self):
        """``data.time >= max_time`` -- run on the device as the integer test on the f64-accumulated clock."""
        return self.data.time >= self.max_time

    @property
    def reward_keys(self):
        return [k for k in self.reward_fns if k != "default" or self.reward_fns[k] != self._default_reward]

    # -- dict -> (device task, host callables) -------------------------------------------------------------------------
    def _plan(self):
        task = _abi.default_task()
        task.frame_skip = self.frame_skip
        task.max_time = self.max_time
        task.obs_mode = self._obs_mode
        task.reset_flags = self._reset_flags
        task.use_time_limit = 0
        task.use_fall = 0
        rf = {"forward": 0.0, "control_cost": 0.0, "alive_bonus": 0.0}
        host_r, host_t = {}, {}
        for name, val in self.reward_fns.items():
            if callable(val):
                if val != self._default_reward:          # the default reward is the constant 0 (quadruped.py:145-147)
                    host_r[name] = val
            elif name in _REWARD_BUILTINS:
                rf[name] = float(val)
            else:
                raise ValueError(f"reward {name!r}: pass a zero-argument callable, or a weight for one of {_REWARD_BUILTINS}")
        task.w_forward, task.w_ctrl, task.alive_bonus = rf["forward"], rf["control_cost"], rf["alive_bonus"]
        for name, val in self.termination_fns.items():
            if callable(val):
                if val == self._default_termination:
                    task.use_time_limit = 1
                else:
                    host_t[name] = val
            elif name == "fall":
                task.use_fall, task.fall_height = 1, float(val)
            else:
                raise ValueError(f"termination {name!r}: pass a zero-argument callable, or a height threshold for 'fall'")
        # envs finished by a host callable can only be reset from the host, and host rewards must see the terminal state:
        # with callables present the kernel leaves finished envs alone and step_wait resets them
        task.auto_reset = 0 if (host_r or host_t) else 1
        return task, host_r, host_t

    @staticmethod
    def _key(task):
        return bytes(task)

    def _dict_key(self):
        return (tuple(self.reward_fns.items()), tuple(self.termination_fns.items()), self.max_time, self.frame_skip)

    def _sync_task(self):
        """Follow edits of the two dicts (README.md:74-89 assigns them after construction).  The common case -- nothing changed
        since the last step -- costs two tuple comparisons."""
        key = self._dict_key()
        if key == getattr(self, "_dicts_seen", None):
            return
        task, self._host_rewards, self._host_terms = self._plan()
        if self._key(task) != self._task_key:
            self._sim.set_task(task)
            self._task_key = self._key(task)
        self._dicts_seen = key

    # -- SB3 VecEnv protocol ------------------------------------------------------------------------
    def reset(self):
        self._sync_task()
        self._sim.reset(seed=self._seed, flags=self._reset_flags)
        return np.zeros((self.num_envs, self.obs_dim), np.float32)      # the reference's first obs is all zeros

    def step_async(self, actions):
        self._actions = np.asarray(actions, dtype=np.float32)

    def _eval_callables(self, obs, state=None):
        """Host evaluation of the Python callables over ``self.data``; returns (components {name: [N]}, done [N])."""
        n = self.num_envs
        qpos, qvel, act, ctrl, nstep = state if state is not None else self._sim.get_state()
        self.data.load(qpos, qvel, act, ctrl, nstep, obs, self.model.opt.timestep)
        comps = {name: np.zeros(n) for name in self._host_rewards}
        done = np.zeros(n, bool)
        if self.callable_mode == "batched":
            self.data._cursor = None
            for name, fn in self._host_rewards.items():
                comps[name][:] = np.broadcast_to(np.asarray(fn(), dtype=np.float64), (n,))
            for fn in self._host_terms.values():
                done |= np.broadcast_to(np.asarray(fn(), dtype=bool), (n,))
        else:
            try:
                for i in range(n):
                    self.data._cursor = i
                    for name, fn in self._host_rewards.items():
                        comps[name][i] = fn()
                    done[i] = any(fn() for fn in self._host_terms.values())       # quadruped.py:178
            finally:
                self.data._cursor = None
        return comps, done

    def step_wait(self):
        self._sync_task()
        host = bool(self._host_rewards or self._host_terms)
        state = None
        if host:                                            # step + state snapshot in one call and one synchronisation
            (obs, rew, done, comps), state = self._sim.step_mirror(self._actions, want_components=True)
        else:
            obs, rew, done, comps = self._sim.step(self._actions, want_components=True)
        names = _REWARD_BUILTINS
        host_comps = {}
        if host:
            host_comps, host_done = self._eval_callables(obs, state)
            rew = rew.astype(np.float64)
            for v in host_comps.values():
                rew = rew + v
            rew = rew.astype(np.float32)
            done = done | host_done
        active = [k for k in names if k in self.reward_fns and not callable(self.reward_fns[k])]
        cols = [names.index(k) for k in active]
        defaults = [k for k, v in self.reward_fns.items() if callable(v) and v == self._default_reward]
        finished = np.nonzero(done)[0]
        # the reference reports the time limit as `terminated`
        extra = {int(i): {"terminal_observation": obs[i].copy(), "TimeLimit.truncated": False} for i in finished}
        picked = comps[:, cols] if cols else None

        def make(i):                                        # infos[i], built the first time it is touched (envs/infos.py)
            rc = dict(zip(active, picked[i].tolist())) if cols else {}
            for name in defaults:
                rc[name] = 0.0                              # quadruped.py:145-147
            for name, v in host_comps.items():
                rc[name] = float(v[i])
            info = dict(rc)
            info["reward_components"] = rc
            e = extra.get(i)
            if e:
                info.update(e)
            return info
        infos = (finished_only_infos(self.num_envs, {i: make(i) for i in extra}) if self.infos_mode == "finished"
                 else LazyInfos(self.num_envs, make))
        if host and finished.size:
            self._sim.reset(mask=done.astype(np.uint8), flags=self._reset_flags)    # draws from the batch's own streams
        if finished.size:
            obs = obs.copy()
            obs[done] = 0.0                                 # envs that finished were reset: their next obs is the reset obs
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
        (obs, reward, done) written by the kernel -- nothing touches the host.  Device built-ins only."""
        import torch
        self._sync_task()
        if self._host_rewards or self._host_terms:
            raise RuntimeError("step_tensor runs the device built-ins only; Python reward / termination callables need step()")
        if packed is None:
            packed = torch.empty((self.num_envs, self.obs_dim + 2), device=actions.device, dtype=torch.float32)
        self._sim.step_device_packed(actions, packed, stream=stream)
        return packed

    def step_sequence_tensor(self, actions, packed=None, stream=None):
        """``K`` env-steps on actions known ahead: ``actions`` float32 CUDA ``[K, N, 12]`` -> the packed rows of every step
        ``[K, N, obs_dim + 2]``.  ONE kernel launch with the state in registers between the steps where the handle allows it (up to
        4096 envs: 9.2 instead of 11.8 us per env-step), ``K`` launches otherwise -- the rows are the same bits either way
        (``qg_step_device_seq``).  Action repeat, evaluating a planned sequence, open-loop rollouts; device built-ins only."""
        import torch
        self._sync_task()
        if self._host_rewards or self._host_terms:
            raise RuntimeError("step_sequence_tensor runs the device built-ins only; Python reward / termination callables need step()")
        if packed is None:
            packed = torch.empty((int(actions.shape[0]), self.num_envs, self.obs_dim + 2), device=actions.device, dtype=torch.float32)
        self._sim.step_device_seq(actions, packed, stream=stream)
        return packed

    def sync_data(self):
        """Env-major host snapshot of the device state (``qpos [N,19]`` ...)."""
        d = self._snapshot
        d.qpos, d.qvel, d.act, d.ctrl, nstep = self._sim.get_state()
        d.time = nstep.astype(np.float64) * self.model.opt.timestep
        return d
