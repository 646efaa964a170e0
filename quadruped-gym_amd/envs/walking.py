"""Walking task envs on the device task layer (``qg_walk_*`` in ``include/quadgym.h``): the batched
counterpart of the reference's ``WalkingQuadrupedEnv`` (``src/envs/walking_quad.py``) -- velocity / heading
command, settling-time action mask, control-signal frequency / amplitude estimator, the 11-term
``input_control_reward`` and the flip + time-limit terminations -- for N robots per kernel launch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _abi
from .._abi import NWALKREWARD, QgWalkParams, check
from ..model.loader import load_model
from ..sim import BatchedSim
from .quadruped import ModelView
from .infos import LazyInfos, finished_only_infos
from .vec_env import HAVE_SB3, _VecEnvBase
from .spaces import Box

REWARD_KEYS = ["alive_bonus", "control_cost", "progress_direction_reward_local", "progress_speed_cost_local",
               "heading_reward", "orientation_reward", "body_height_cost", "joint_posture_cost",
               "control_amplitude_cost", "control_frequency_cost", "diff_ideal_position_cost"]   # walking_quad.py:332-351


def default_walk_params() -> QgWalkParams:
    p = QgWalkParams()
    check(_abi.load_library().qg_walk_default_params(C.byref(p)), "qg_walk_default_params")
    return p


def sample_command(options=None, uniform=None):
    """``VelocityHeadingControls.sample`` (``src/envs/control_inputs.py:74-115``): returns
    ``(velocity_xy, heading_xy)``; draws from ``np.random.uniform`` in the reference's order."""
    uniform = uniform or np.random.uniform
    options = options or {}
    lo, hi = options.get("min_speed", 0.0), options.get("max_speed", 1.0)
    th = options.get("fixed_heading_angle")
    theta = th if th is not None else uniform(-np.pi, np.pi)
    al = options.get("fixed_velocity_angle")
    alpha = al if al is not None else uniform(-np.pi, np.pi)
    sp = options.get("fixed_speed")
    speed = sp if sp is not None else uniform(lo, hi)
    return (speed * np.cos(alpha), speed * np.sin(alpha)), (np.cos(theta), np.sin(theta))


def _component_infos(n, comps, extra, mode="lazy"):
    """``infos`` of a walking step: per env the reward-component dict the reference returns as ``info`` (walking_quad.py:419), plus
    the SB3 entries of the envs in ``extra``; built on first access (see envs/infos.py), or -- ``mode="finished"`` -- only for the
    envs that finished."""
    def make(i):
        d = dict(zip(REWARD_KEYS, comps[i].tolist()))
        e = extra.get(i)
        if e:
            d.update(e)
        return d
    if mode == "finished":
        return finished_only_infos(n, {i: make(i) for i in extra})
    return LazyInfos(n, make)


class WalkingQuadrupedVecEnv(_VecEnvBase):      # SB3's VecEnv where that package is importable (what `PPO(..., env)` checks for), else object
    """N walking robots; SB3 VecEnv calling convention (replaces ``SubprocVecEnv([make_env]*N)`` at
    ``src/train_quadruped.py:50``).  ``infos[i]`` is the reward-component dict the reference returns as
    ``info`` (``walking_quad.py:146-148,419``), which ``RewardCallback`` reads (``train_quadruped.py:86-97``)."""

    reward_keys = REWARD_KEYS

    def __init__(self, num_envs, settling_time=0, random_controls=False, random_init=False, reset_options=None,
                 model_path="builtin", max_time=10.0, frame_skip=4, device=0, env_index_base=0, seed=0, walk_params=None,
                 device_commands=False, auto_reset=True, use_default_termination=True, infos_mode="lazy", nan_direction=True):
        if infos_mode not in ("lazy", "finished"):
            raise ValueError("infos_mode must be 'lazy' (every env's component dict, built when touched) or 'finished' (content "
                             "for the envs that finished only; `last_components` holds every env's components as one array)")
        self.infos_mode = infos_mode
        qg_model, layout = load_model(model_path)
        self.model = ModelView(qg_model, layout)
        self.num_envs = int(num_envs)
        self.frame_skip, self.max_time = int(frame_skip), float(max_time)
        self.random_controls, self.random_init, self.reset_options = random_controls, random_init, reset_options
        task = _abi.default_task()
        task.frame_skip = self.frame_skip
        task.max_time = self.max_time
        task.use_time_limit = 1 if use_default_termination else 0   # quadruped.py:52,99-100
        task.use_fall = 0
        task.use_flip = 1                                   # walking_quad.py:156-166
        # VecEnv semantics: finished envs restart inside the step.  The single-robot facades switch it off: there the caller
        # resets, as with the reference's env (a finished robot left alone keeps reporting `terminated`)
        self.auto_reset = bool(auto_reset)
        task.auto_reset = 1 if self.auto_reset else 0
        task.reset_flags = _abi.RESET_RANDOM_YAW if random_init else 0     # walking_quad.py:68-75,118-119
        self._sim = BatchedSim(self.num_envs, device=device, model=qg_model, task=task, env_index_base=env_index_base)
        self._lib = _abi.load_library()
        self.params = walk_params if walk_params is not None else default_walk_params()
        self.params.settling_time = float(settling_time)
        # nan_direction=True is the reference: unit() of an exactly zero local velocity (or command) is NaN and so are the direction
        # term and the reward of that step (math_utils.py:7-8, walking_quad.py:197-205).  In this f32 pipeline the local xy velocity is
        # EXACTLY zero on the first step of every episode, so a training run wants False: the term is 0 there.
        self.nan_direction = bool(nan_direction)
        if not self.nan_direction:
            self.params.unit_zero = 1
        h = C.c_void_p()
        check(self._lib.qg_walk_create(self._sim._h, C.byref(self.params), C.byref(h)), "qg_walk_create")
        self._w = h
        self._seed = int(seed)
        self._flags = task.reset_flags
        # random_controls (walking_quad.py:121-122).  device_commands=False: commands are redrawn on the host with
        # np.random.uniform in the reference's order (VelocityHeadingControls.sample); True: on the GPU, inside reset and
        # the step's auto-reset, from the env's own counter-based stream -- rollouts through step_tensor never touch the host.
        self.device_commands = bool(device_commands) and bool(random_controls)
        if self.device_commands:
            sampler = _abi.QgCommandSampler.from_options(reset_options)
            check(self._lib.qg_walk_set_command_sampler(self._w, C.byref(sampler)), "qg_walk_set_command_sampler")
        self.action_space = Box(low=-1.0, high=1.0, shape=(12,), dtype=np.float32)
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(33,), dtype=np.float32)
        if HAVE_SB3:  # pragma: no cover - stable_baselines3 is absent from the build image
            _VecEnvBase.__init__(self, self.num_envs, self.observation_space, self.action_space)
        self.velocity = np.zeros((self.num_envs, 2), np.float32)
        self.heading = np.zeros((self.num_envs, 2), np.float32)
        self.dt = qg_model.timestep * self.frame_skip
        self.render_mode = None

    # -- commands ---------------------------------------------------------------------------------
    def set_commands(self, velocity_xy, heading_xy):
        self.velocity[:] = np.asarray(velocity_xy, np.float32).reshape(self.num_envs, 2)
        self.heading[:] = np.asarray(heading_xy, np.float32).reshape(self.num_envs, 2)
        check(self._lib.qg_walk_set_commands(self._w, self.velocity.ctypes.data, self.heading.ctypes.data), "qg_walk_set_commands")

    def commands(self):
        """``(velocity_xy, heading_xy)`` as they stand on the device, ``[num_envs, 2]`` each."""
        check(self._lib.qg_walk_get_commands(self._w, self.velocity.ctypes.data, self.heading.ctypes.data), "qg_walk_get_commands")
        return self.velocity, self.heading

    # -- checkpoint / resume (the reference resumes the policy only, train_quadruped.py:114-141) -------------------
    def snapshot(self):
        """Everything this env keeps per robot -- physics state, reset streams, the walking layer's task state (estimator included)
        and, for the partially observable env, the filter estimate and the frame ring -- as a ``dict`` of NumPy arrays.
        ``restore(snapshot)`` on an env of the same shape continues bit for bit."""
        snap = {"sim": self._sim.snapshot()}
        blob = np.empty(self._blob_bytes(self._lib.qg_walk_state_bytes, self._w, "qg_walk_state_bytes"), np.uint8)
        check(self._lib.qg_walk_get_state(self._w, blob.ctypes.data), "qg_walk_get_state")
        snap["walk"] = blob
        if getattr(self, "_po", None):
            pblob = np.empty(self._blob_bytes(self._lib.qg_po_state_bytes, self._po, "qg_po_state_bytes"), np.uint8)
            check(self._lib.qg_po_get_state(self._po, pblob.ctypes.data), "qg_po_get_state")
            snap["po"] = pblob
        snap["velocity"], snap["heading"] = self.velocity.copy(), self.heading.copy()
        return snap

    def _blob_bytes(self, fn, handle, what):
        nbytes = int(fn(handle))
        if nbytes <= 0:                   # the C side reports errors as small negative numbers
            check(nbytes if nbytes < 0 else -1, what)
        return nbytes

    def restore(self, snap):
        """All-or-nothing: every part of the snapshot is checked against this env BEFORE anything is written.  Continues bit for
        bit with ``device_commands=True`` (or fixed commands); with host-side command sampling the commands an auto-reset re-draws
        come from NumPy's global generator, which a snapshot does not hold."""
        po = getattr(self, "_po", None)
        blob = np.ascontiguousarray(snap["walk"], dtype=np.uint8)
        if blob.size != self._blob_bytes(self._lib.qg_walk_state_bytes, self._w, "qg_walk_state_bytes"):
            raise ValueError("the snapshot was taken from an env of another shape (num_envs / estimator window)")
        pblob = None
        if po:
            if "po" not in snap:
                raise ValueError("the snapshot holds no observation-pack state (it was taken from an env without one)")
            pblob = np.ascontiguousarray(snap["po"], dtype=np.uint8)
            if pblob.size != self._blob_bytes(self._lib.qg_po_state_bytes, po, "qg_po_state_bytes"):
                raise ValueError("the snapshot was taken from an env of another shape (num_envs / obs_window)")
        sim = snap["sim"]
        if np.asarray(sim["qpos"]).shape != (self.num_envs, 19) or np.asarray(sim["episode"]).shape != (self.num_envs,):
            raise ValueError("the snapshot was taken from an env of another size")
        if np.asarray(snap["velocity"]).shape != self.velocity.shape or np.asarray(snap["heading"]).shape != self.heading.shape:
            raise ValueError("the snapshot's commands do not fit this env")
        self._sim.restore(sim)
        check(self._lib.qg_walk_set_state(self._w, blob.ctypes.data), "qg_walk_set_state")
        if po:
            check(self._lib.qg_po_set_state(po, pblob.ctypes.data), "qg_po_set_state")
        self.velocity[:], self.heading[:] = snap["velocity"], snap["heading"]

    def _resample(self, idx):
        if self.device_commands:          # already redrawn on the device by the reset / auto-reset itself
            return
        for i in idx:
            v, hd = sample_command(self.reset_options)
            self.velocity[i], self.heading[i] = v, hd
        check(self._lib.qg_walk_set_commands(self._w, self.velocity.ctypes.data, self.heading.ctypes.data), "qg_walk_set_commands")

    # -- VecEnv protocol ------------------------------------------------------------------------------
    def reset(self):
        check(self._lib.qg_walk_reset(self._w, None, self._seed, self._flags), "qg_walk_reset")
        if self.random_controls:                              # walking_quad.py:121-122
            self._resample(range(self.num_envs))
        return np.zeros((self.num_envs, 33), np.float32)

    def step_async(self, actions):
        self._actions = np.ascontiguousarray(actions, dtype=np.float32)

    def step_wait(self):
        n = self.num_envs
        a = self._actions
        if a.shape != (n, 12):
            raise ValueError(f"actions must have shape ({n}, 12)")
        obs = np.empty((n, 33), np.float32)
        rew = np.empty(n, np.float32)
        done = np.empty(n, np.uint8)
        comps = np.empty((n, NWALKREWARD), np.float32)
        check(self._lib.qg_walk_step(self._w, a.ctypes.data, obs.ctypes.data, rew.ctypes.data, done.ctypes.data, comps.ctypes.data),
              "qg_walk_step")
        dones = done.astype(bool)
        extra = {}
        if self.auto_reset:
            for i in np.nonzero(dones)[0]:
                extra[int(i)] = {"terminal_observation": obs[i].copy(), "TimeLimit.truncated": False}
        infos = _component_infos(n, comps, extra, self.infos_mode)
        if self.auto_reset and dones.any():
            obs = obs.copy()
            obs[dones] = 0.0
            if self.random_controls:
                self._resample(np.nonzero(dones)[0])
        self.last_components = comps
        return obs, rew, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def estimates(self):
        f = np.empty((self.num_envs, 12), np.float32)
        a = np.empty((self.num_envs, 12), np.float32)
        ideal = np.empty((self.num_envs, 2), np.float32)
        check(self._lib.qg_walk_get_estimates(self._w, f.ctypes.data, a.ctypes.data, ideal.ctypes.data), "qg_walk_get_estimates")
        return f, a, ideal

    def step_tensor(self, actions, obs, reward, done, components=None, stream=None):
        """Zero-copy step on CUDA tensors (float32 ``[N,12]``, ``[N,33]``, ``[N]``, uint8 ``[N]``, ``[N,11]``)."""
        check(self._lib.qg_walk_step_device(self._w, actions.data_ptr(), obs.data_ptr(), reward.data_ptr(), done.data_ptr(),
                                            components.data_ptr() if components is not None else None,
                                            self._sim._stream_ptr(stream)), "qg_walk_step_device")

    def close(self):
        if getattr(self, "_w", None):
            self._lib.qg_walk_destroy(self._w)
            self._w = None
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


def _facade_kwargs(kwargs, allowed):
    """Keyword arguments the reference forwards to ``QuadrupedEnv.__init__`` (``walking_quad.py:11-12``).  The rendering ones
    are accepted when they ask for nothing; ``reward_fns`` / ``termination_fns`` would REPLACE the task's reward and
    terminations in the reference (``quadruped.py:97-100``) -- the device task layer cannot honour that, so it refuses
    instead of silently running the built-in walking reward."""
    display = {"render_mode", "width", "height", "render_fps", "save_video", "video_path"}
    extra = set(kwargs) - allowed - display - {"reward_fns", "termination_fns"}
    if extra:
        raise TypeError(f"unexpected keyword arguments {sorted(extra)}")
    if kwargs.get("render_mode") is not None or kwargs.get("save_video"):
        raise NotImplementedError("rendering / video recording is not part of the HIP path")
    for key in ("reward_fns", "termination_fns"):
        if kwargs.get(key) is not None:
            raise NotImplementedError(f"{key}: the walking task layer evaluates the reference's input_control_reward and flip / "
                                      "time-limit terminations on the device; custom callables run through QuadrupedEnv / "
                                      "QuadrupedVecEnv")
    args = {k: v for k, v in kwargs.items() if k in allowed}
    args.setdefault("model_path", "./models/quadruped/scene.xml")      # quadruped.py:41
    return args


class WalkingQuadrupedEnv:
    """One walking robot with the reference's constructor (``walking_quad.py:11``): ``settling_time``,
    ``random_controls``, ``random_init``, ``reset_options`` plus the base env's keyword arguments;
    ``step`` returns ``(obs, reward, terminated, False, info)`` with ``info`` = the component dict."""

    reward_keys = REWARD_KEYS

    def __init__(self, settling_time=0, random_controls=False, random_init=False, reset_options=None, **kwargs):
        args = _facade_kwargs(kwargs, {"model_path", "max_time", "frame_skip", "device", "use_default_termination"})
        self._vec = WalkingQuadrupedVecEnv(1, settling_time, random_controls, random_init, reset_options, auto_reset=False, **args)
        self.action_space, self.observation_space = self._vec.action_space, self._vec.observation_space
        self.model = self._vec.model
        self.info = {}

    def reset(self, seed=None, options=None):
        if options is not None:
            self._vec.reset_options = options                  # walking_quad.py:100-101
        obs = self._vec.reset()
        self.info = {}
        return obs[0].astype(np.float64), self.info

    def step(self, action):
        obs, rew, dones, infos = self._vec.step(np.asarray(action, np.float32)[None])
        self.info = dict(infos[0])                             # walking_quad.py:146-148: info = the component dict
        return obs[0].astype(np.float64), float(rew[0]), bool(dones[0]), False, self.info

    def set_command(self, velocity_xy, heading_xy):
        self._vec.set_commands([velocity_xy], [heading_xy])

    def close(self):
        self._vec.close()


class POWalkingQuadrupedVecEnv(WalkingQuadrupedVecEnv):
    """Batched ``POWalkingQuadrupedEnv`` (``src/envs/po_walking_quad.py``): the observation is the stack of the last
    ``obs_window`` 26-value frames [gyro, accel, Madgwick-IMU Euler angles, body_vel xy, data.ctrl, command vx vy theta]
    (``:48-56``); rewards and terminations are the walking task's."""

    FRAME = 26

    def __init__(self, num_envs, obs_window=1, **kwargs):
        super().__init__(num_envs, **kwargs)
        self.obs_window = int(obs_window)
        h = C.c_void_p()
        check(self._lib.qg_po_create(self._w, self.obs_window, C.byref(h)), "qg_po_create")
        self._po = h
        self.obs_dim = int(self._lib.qg_po_obs_dim(self._po))
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.obs_dim,), dtype=np.float32)   # po_walking_quad.py:27
        if HAVE_SB3:  # pragma: no cover
            _VecEnvBase.__init__(self, self.num_envs, self.observation_space, self.action_space)

    def reset(self):
        obs = np.empty((self.num_envs, self.obs_dim), np.float32)
        check(self._lib.qg_po_reset(self._po, None, self._seed, self._flags, obs.ctypes.data), "qg_po_reset")
        if self.random_controls:
            self._resample(range(self.num_envs))
        return obs

    def step_wait(self):
        n = self.num_envs
        a = self._actions
        if a.shape != (n, 12):
            raise ValueError(f"actions must have shape ({n}, 12)")
        obs = np.empty((n, self.obs_dim), np.float32)
        if getattr(self, "_term_buf", None) is None:       # scratch for the library's terminal stacks (rows are copied out below)
            self._term_buf = np.empty((n, self.obs_dim), np.float32)
        term = self._term_buf
        rew = np.empty(n, np.float32)
        done = np.empty(n, np.uint8)
        comps = np.empty((n, NWALKREWARD), np.float32)
        check(self._lib.qg_po_step(self._po, a.ctypes.data, obs.ctypes.data, rew.ctypes.data, done.ctypes.data, comps.ctypes.data,
                                   term.ctypes.data), "qg_po_step")
        dones = done.astype(bool)
        extra = {}
        if self.auto_reset:
            for i in np.nonzero(dones)[0]:
                extra[int(i)] = {"terminal_observation": term[i].copy(), "TimeLimit.truncated": False}
        infos = _component_infos(n, comps, extra, self.infos_mode)
        if self.auto_reset and dones.any() and self.random_controls:
            self._resample(np.nonzero(dones)[0])
        self.last_components = comps
        return obs, rew, dones, infos

    def step_tensor(self, actions, obs, reward, done, components=None, terminal_obs=None, stream=None):
        """Zero-copy step on CUDA tensors: ``actions`` float32 ``[N,12]`` -> ``obs`` float32 ``[N, 26 * obs_window]`` (the
        stacked frames; rows of envs that finished already hold the reset stack), ``reward`` ``[N]``, ``done`` uint8
        ``[N]``, optionally ``components`` ``[N,11]`` and ``terminal_obs`` (the stack each finished env ended with)."""
        check(self._lib.qg_po_step_device(self._po, actions.data_ptr(), obs.data_ptr(), reward.data_ptr(), done.data_ptr(),
                                          components.data_ptr() if components is not None else None,
                                          terminal_obs.data_ptr() if terminal_obs is not None else None,
                                          self._sim._stream_ptr(stream)), "qg_po_step_device")

    def close(self):
        if getattr(self, "_po", None):
            self._lib.qg_po_destroy(self._po)
            self._po = None
        super().close()


class POWalkingQuadrupedEnv:
    """One robot with the reference's constructor ``POWalkingQuadrupedEnv(obs_window=1, **kwargs)``
    (``po_walking_quad.py:10``; ``train_quadruped.py:16-22`` builds it with ``obs_window=10``)."""

    reward_keys = REWARD_KEYS

    def __init__(self, obs_window=1, **kwargs):
        args = _facade_kwargs(kwargs, {"settling_time", "random_controls", "random_init", "reset_options", "model_path", "max_time",
                                       "frame_skip", "device", "use_default_termination"})
        self._vec = POWalkingQuadrupedVecEnv(1, obs_window=obs_window, auto_reset=False, **args)
        self.obs_window = obs_window
        self.action_space, self.observation_space = self._vec.action_space, self._vec.observation_space
        self.model = self._vec.model
        self.info = {}

    def reset(self, seed=None, options=None):
        if options is not None:
            self._vec.reset_options = options
        obs = self._vec.reset()
        self.info = {}
        return obs[0].astype(np.float64), self.info

    def step(self, action):
        obs, rew, dones, infos = self._vec.step(np.asarray(action, np.float32)[None])
        self.info = dict(infos[0])
        return obs[0].astype(np.float64), float(rew[0]), bool(dones[0]), False, self.info

    def close(self):
        self._vec.close()
