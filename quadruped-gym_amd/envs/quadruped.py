"""``QuadrupedEnv`` -- the reference's single-robot Gymnasium environment
(``src/envs/quadruped.py:9-182`` of antopio26/quadruped-gym) on the HIP pipeline.

Same constructor arguments, same ``reset`` / ``step`` contract, same public attributes
(``model``, ``data``, ``reward_fns``, ``termination_fns``, ``action_space``, ``observation_space``),
so user code written against the reference -- README-style reward/termination lambdas that read
``env.data.qpos`` etc. -- runs unchanged.  The physics of ``mujoco.mj_step`` (``quadruped.py:165``) is
replaced by one launch of the step kernel for a batch of one env; reward and termination callables
stay ordinary Python evaluated on the host mirror of the state, exactly as in the reference.
Rendering and video (``quadruped.py:184-316``) are outside the accelerated path and not provided.
"""
from __future__ import annotations

import numpy as np

from .. import _abi
from ..model.loader import load_model
from ..sim import BatchedSim
from .spaces import Box, EnvBase


class _Opt:
    def __init__(self, timestep):
        self.timestep = timestep


class ModelView:
    """The slice of ``mujoco.MjModel`` the reference reads: ``nu``, ``nsensordata``, ``opt.timestep``,
    ``sensor_adr`` and sensor name lookup (``quadruped.py:90,93``; ``walking_quad.py:19,56``)."""

    def __init__(self, qg_model, layout):
        self.nq, self.nv, self.nu, self.na = 19, 18, 12, 12
        self.nsensordata = int(layout["nsensordata"])
        self.nsensor = len(layout["sensors"])
        self.opt = _Opt(qg_model.timestep)
        self.sensor_names = [s["name"] for s in layout["sensors"]]
        self.sensor_adr = np.array([s["adr"] for s in layout["sensors"]], dtype=np.int32)
        self.sensor_dim = np.array([s["dim"] for s in layout["sensors"]], dtype=np.int32)
        self.qpos0 = np.array(qg_model.qpos0[:])
        self.qg = qg_model

    def sensor_id(self, name: str) -> int:
        """``mj_name2id(model, mjtObj.mjOBJ_SENSOR, name)`` (``walking_quad.py:19``)."""
        return self.sensor_names.index(name)


class DataView:
    """Host mirror of ``mujoco.MjData``: ``qpos, qvel, act, ctrl, time, sensordata`` as float64 arrays."""

    def __init__(self):
        self.qpos = np.zeros(19)
        self.qvel = np.zeros(18)
        self.act = np.zeros(12)
        self.ctrl = np.zeros(12)
        self.sensordata = np.zeros(33)
        self.time = 0.0


class QuadrupedEnv(EnvBase):
    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 30}

    def __init__(self,
                 model_path: str = "./models/quadruped/scene.xml",
                 max_time: float = 10.0,
                 frame_skip: int = 4,
                 render_mode: str = None,
                 width: int = 720,
                 height: int = 480,
                 render_fps: int = 30,
                 reward_fns: dict = None,
                 termination_fns: dict = None,
                 save_video: bool = False,
                 video_path: str = "videos/simulation.mp4",
                 use_default_termination: bool = True,
                 device: int = 0):
        super().__init__()
        self.model_path = model_path
        qg_model, layout = load_model(model_path)          # FileNotFoundError for a bad path (quadruped.py:55-56)
        self.model = ModelView(qg_model, layout)
        self.data = DataView()
        self.max_time = max_time
        self.frame_skip = frame_skip
        if render_mode is not None or save_video:
            raise NotImplementedError("rendering / video recording (quadruped.py:184-316) is not part of the HIP path")
        self.render_mode = render_mode
        self.width, self.height, self.render_fps = width, height, render_fps
        self.metadata = dict(self.metadata, render_fps=render_fps)

        task = _abi.default_task()
        task.frame_skip = int(frame_skip)
        task.use_time_limit = 0         # terminations are the Python callables below, as in the reference
        task.use_fall = 0
        task.auto_reset = 0
        self._sim = BatchedSim(1, device=device, model=qg_model, task=task)

        self.action_space = Box(low=-1.0, high=1.0, shape=(self.model.nu,), dtype=np.float32)          # quadruped.py:90
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.model.nsensordata,), dtype=np.float32)

        self.reward_fns = reward_fns if reward_fns is not None else {"default": self._default_reward}
        self.termination_fns = termination_fns if termination_fns is not None else {}
        if use_default_termination:
            self.termination_fns["default"] = self._default_termination
        self.save_video, self.video_path = save_video, video_path
        self._synced = None
        self.seed()

    # -- reference API --------------------------------------------------------------------------
    def seed(self, seed=None):
        np.random.seed(seed)              # quadruped.py:111-113 (global NumPy RNG)
        return [seed]

    def reset(self, seed=None, options=None):
        self._sim.reset()
        self._pull()
        self.data.time = 0.0
        self.data.ctrl[:] = np.array([0, 0, -0.5] * 4)     # quadruped.py:124
        self.data.sensordata[:] = 0.0                       # no mj_forward after mj_resetData: first obs is zeros
        return self._get_obs(), {}

    def _get_obs(self):
        return self.data.sensordata.copy()

    def _default_reward(self):
        return 0.0

    def _default_termination(self):
        return self.data.time >= self.max_time

    def step(self, action):
        # contract of quadruped.py:153-182 -- clip to the action space, frame_skip physics substeps in one launch,
        # lagged sensordata as the observation, reward / termination callables evaluated on the host mirror
        lo, hi = self.action_space.low, self.action_space.high
        applied = np.minimum(np.maximum(np.asarray(action, dtype=np.float64), lo), hi)
        self._push_if_edited()
        (sensed, _, _, _), state = self._sim.step_mirror(applied.astype(np.float32)[None])    # one launch, one synchronisation
        self._pull(state)
        h = self.model.opt.timestep
        for _ in range(self.frame_skip):                    # the engine's f64 clock: one addition per substep
            self.data.time += h
        self.data.ctrl[:] = applied
        self.data.sensordata[:] = sensed[0]

        components = {name: fn() for name, fn in self.reward_fns.items()}
        total = 0.0
        for value in components.values():
            total += value
        finished = False
        for fn in self.termination_fns.values():            # `any(...)`: stops at the first condition that fires
            if fn():
                finished = True
                break
        return self._get_obs(), total, finished, False, {"time": self.data.time, "reward_components": components}

    def render(self):
        return None

    def close(self):
        if getattr(self, "_sim", None) is not None:
            self._sim.close()
            self._sim = None

    # -- host mirror <-> device state -----------------------------------------------------------------
    def _pull(self, state=None):
        qpos, qvel, act, ctrl, nstep = state if state is not None else self._sim.get_state()
        self.data.qpos[:] = qpos[0]
        self.data.qvel[:] = qvel[0]
        self.data.act[:] = act[0]
        self._synced = (self.data.qpos.copy(), self.data.qvel.copy(), self.data.act.copy())

    def _push_if_edited(self):
        """User code may write ``env.data.qpos[...]`` between steps (``walking_quad.py:68-75`` does); push it."""
        s = self._synced
        if s is None or not (np.array_equal(s[0], self.data.qpos) and np.array_equal(s[1], self.data.qvel)
                             and np.array_equal(s[2], self.data.act)):
            self._sim.set_state(self.data.qpos[None], self.data.qvel[None], self.data.act[None])
