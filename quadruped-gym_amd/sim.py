"""``BatchedSim`` -- thin Python handle over the C ABI (``include/quadgym.h``).

PyTorch is used only as plumbing: it owns the device buffers that are handed to
``qg_step_device*`` as raw pointers and provides the HIP stream.  The physics
runs in ``libquadgym.so`` (hand-written gfx950 kernels); there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi
from ._abi import NQ, NV, NU, NREWARD, QgModel, QgTask, check


class BatchedSim:
    """n independent quadrupeds on one GPU.

    Replaces, for a whole batch, what ``QuadrupedEnv.__init__/reset/step`` do for
    one robot through ``mujoco`` (``src/envs/quadruped.py:59-60,115-139,153-182``).
    """

    def __init__(self, n_envs: int, device: int = 0, model: QgModel | None = None, task: QgTask | None = None,
                 env_index_base: int = 0):
        self._lib = _abi.load_library()
        self.model = model if model is not None else _abi.default_model()
        self.task = task if task is not None else _abi.default_task()
        self.n = int(n_envs)
        self.device = int(device)
        self.env_index_base = int(env_index_base)
        h = C.c_void_p()
        check(self._lib.qg_create(self.n, self.device, C.byref(self.model), C.byref(self.task), self.env_index_base,
                                  C.byref(h)), "qg_create")
        self._h = h
        # the step time is a staircase in the batch size: just past a stair's top a batch is SLOWER in absolute
        # terms than the top itself -- say so once, where it is worst (4097 .. ~6500 envs on an MI355X run below 4096 envs' env-steps/s)
        top = _abi.recommended_batch(1, self.device)          # the first stair: one wave of the one-link-per-lane kernel per SIMD
        if top < self.n <= int(1.6 * top):
            import warnings
            warnings.warn(f"{self.n} envs cost {_abi.recommended_batch(self.n, self.device)} envs' step time and run at fewer env-steps/s than "
                          f"{top} envs do; use {top}, two handles of <= {top} on two streams, or >= {int(1.6 * top)} "
                          f"(quadruped_gym_amd._abi.recommended_batch, INTEGRATION.md section 5)", stacklevel=2)
        self.obs_dim = self._lib.qg_obs_dim(self._h)
        self.baked = bool(self._lib.qg_uses_baked_model(self._h))
        self.limit_substeps = int(self._lib.qg_time_limit_substeps(self.model.timestep, self.task.max_time))

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.qg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reset / step through host (NumPy) buffers ------------------------------------------------
    def reset(self, mask=None, seed: int = 0, flags: int = 0):
        mp = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            assert mask.shape == (self.n,)
            mp = mask.ctypes.data
        check(self._lib.qg_reset(self._h, mp, int(seed), int(flags)), "qg_reset")

    def step(self, actions, want_components: bool = False):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        if a.shape != (self.n, NU):
            raise ValueError(f"actions must have shape ({self.n}, {NU}), got {a.shape}")
        obs = np.empty((self.n, self.obs_dim), np.float32)
        rew = np.empty(self.n, np.float32)
        done = np.empty(self.n, np.uint8)
        comps = np.empty((self.n, NREWARD), np.float32) if want_components else None
        check(self._lib.qg_step(self._h, a.ctypes.data, obs.ctypes.data, rew.ctypes.data, done.ctypes.data,
                                comps.ctypes.data if want_components else None), "qg_step")
        return obs, rew, done.astype(bool), comps

    def step_mirror(self, actions, want_components: bool = False):
        """``step`` and ``get_state`` in one call and one synchronisation: ``(obs, rew, done, comps), (qpos, qvel, act, ctrl, nstep)``."""
        a = np.ascontiguousarray(actions, dtype=np.float32)
        if a.shape != (self.n, NU):
            raise ValueError(f"actions must have shape ({self.n}, {NU}), got {a.shape}")
        obs = np.empty((self.n, self.obs_dim), np.float32)
        rew = np.empty(self.n, np.float32)
        done = np.empty(self.n, np.uint8)
        comps = np.empty((self.n, NREWARD), np.float32) if want_components else None
        qpos = np.empty((self.n, NQ), np.float32)
        qvel = np.empty((self.n, NV), np.float32)
        act = np.empty((self.n, NU), np.float32)
        ctrl = np.empty((self.n, NU), np.float32)
        nstep = np.empty(self.n, np.int32)
        check(self._lib.qg_step_mirror(self._h, a.ctypes.data, obs.ctypes.data, rew.ctypes.data, done.ctypes.data,
                                       comps.ctypes.data if want_components else None, qpos.ctypes.data, qvel.ctypes.data, act.ctypes.data,
                                       ctrl.ctypes.data, nstep.ctypes.data), "qg_step_mirror")
        return (obs, rew, done.astype(bool), comps), (qpos, qvel, act, ctrl, nstep)

    # -- zero-copy forms on torch (ROCm) tensors ---------------------------------------------------
    def _stream_ptr(self, stream):
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        return C.c_void_p(stream.cuda_stream)

    def _check_tensor(self, t, shape, dtype):
        if not t.is_cuda or t.device.index != self.device:
            raise ValueError(f"tensor must live on cuda:{self.device}")
        if tuple(t.shape) != tuple(shape) or t.dtype != dtype or not t.is_contiguous():
            raise ValueError(f"expected contiguous {dtype} tensor of shape {tuple(shape)}, got {t.dtype} {tuple(t.shape)}")

    def step_device(self, actions, obs, reward, done, comps=None, stream=None):
        import torch
        self._check_tensor(actions, (self.n, NU), torch.float32)
        self._check_tensor(obs, (self.n, self.obs_dim), torch.float32)
        self._check_tensor(reward, (self.n,), torch.float32)
        self._check_tensor(done, (self.n,), torch.uint8)
        if comps is not None:
            self._check_tensor(comps, (self.n, NREWARD), torch.float32)
        check(self._lib.qg_step_device(self._h, actions.data_ptr(), obs.data_ptr(), reward.data_ptr(), done.data_ptr(),
                                       comps.data_ptr() if comps is not None else None, self._stream_ptr(stream)),
              "qg_step_device")

    def step_device_packed(self, actions, packed, stream=None):
        """obs, reward and done (0/1) in one ``[n, obs_dim + 2]`` f32 buffer: the unit of the per-step RCCL gather."""
        import torch
        self._check_tensor(actions, (self.n, NU), torch.float32)
        self._check_tensor(packed, (self.n, self.obs_dim + 2), torch.float32)
        check(self._lib.qg_step_device_packed(self._h, actions.data_ptr(), packed.data_ptr(), self._stream_ptr(stream)),
              "qg_step_device_packed")

    def bind_step_packed(self, actions_list, packed_list, stream=None):
        """Pre-validate a set of action / packed tensors and return ``step(i, j)`` that launches one env-step on
        ``actions_list[i] -> packed_list[j]`` with no per-call checks (a throughput loop's hot path)."""
        import torch
        for a in actions_list:
            self._check_tensor(a, (self.n, NU), torch.float32)
        for p in packed_list:
            self._check_tensor(p, (self.n, self.obs_dim + 2), torch.float32)
        a_ptr = [C.c_void_p(a.data_ptr()) for a in actions_list]
        p_ptr = [C.c_void_p(p.data_ptr()) for p in packed_list]
        st = self._stream_ptr(stream)
        fn, h = self._lib.qg_step_device_packed, self._h

        def step(i, j):
            if fn(h, a_ptr[i], p_ptr[j], st) != 0:
                check(-1, "qg_step_device_packed")
        return step

    # -- many env-steps per launch (include/quadgym.h: qg_step_device_seq, qg_resident_*) ---------------------------------------
    def step_device_seq(self, actions, packed, stream=None):
        """ONE launch runs ``K`` env-steps: ``actions[K, n, 12] -> packed[K, n, obs_dim + 2]`` with the state in registers in
        between (open-loop sequences).  Bit-identical to ``K`` calls of ``step_device_packed``."""
        import torch
        k = int(actions.shape[0])
        self._check_tensor(actions, (k, self.n, NU), torch.float32)
        self._check_tensor(packed, (k, self.n, self.obs_dim + 2), torch.float32)
        check(self._lib.qg_step_device_seq(self._h, actions.data_ptr(), packed.data_ptr(), k, self._stream_ptr(stream)),
              "qg_step_device_seq")

    def resident_start(self, actions, packed, idle_timeout_us: int = 0):
        """Launch the RESIDENT step kernel on the mailbox ``actions[slots, n, 12]`` / ``packed[slots, n, obs_dim + 2]`` (torch tensors
        the caller keeps alive until ``resident_stop``).  Env-step ``i`` of the resident sequence uses slot ``i % slots``."""
        import torch
        slots = int(actions.shape[0])
        self._check_tensor(actions, (slots, self.n, NU), torch.float32)
        self._check_tensor(packed, (slots, self.n, self.obs_dim + 2), torch.float32)
        torch.cuda.synchronize(self.device)        # whatever filled the slots has landed before the kernel may read them
        check(self._lib.qg_resident_start(self._h, slots, int(idle_timeout_us), actions.data_ptr(), packed.data_ptr()),
              "qg_resident_start")
        self._resident_keep = (actions, packed)

    def resident_step(self, count: int = 1, stream=None):
        """Ring ``count`` env-steps on ``stream`` (default: torch's current stream): what follows on that stream finds their rows
        in the output slots."""
        check(self._lib.qg_resident_step_device(self._h, int(count), self._stream_ptr(stream)), "qg_resident_step_device")

    def bind_resident_step(self, count: int = 1, stream=None):
        st = self._stream_ptr(stream)
        fn, h, c = self._lib.qg_resident_step_device, self._h, int(count)

        def ring():
            if fn(h, c, st) != 0:
                check(-1, "qg_resident_step_device")
        return ring

    def resident_ensure(self):
        check(self._lib.qg_resident_ensure(self._h), "qg_resident_ensure")

    def resident_status(self):
        """``dict(rung, running, completed_at_exit, not_executed)`` -- no synchronisation."""
        rung, run, comp, lost = C.c_int64(0), C.c_int32(0), C.c_int64(0), C.c_int64(0)
        check(self._lib.qg_resident_status(self._h, C.byref(rung), C.byref(run), C.byref(comp), C.byref(lost)), "qg_resident_status")
        return {"rung": int(rung.value), "running": bool(run.value), "completed_at_exit": int(comp.value), "not_executed": int(lost.value)}

    def resident_stop(self):
        check(self._lib.qg_resident_stop(self._h), "qg_resident_stop")
        self._resident_keep = None

    def time_step_kernel(self, actions, packed, iters: int) -> float:
        """Mean milliseconds per launch of the step kernel over ``iters`` back-to-back launches,
        measured with HIP events on the stream the kernel is launched on."""
        import torch
        self._check_tensor(actions, (self.n, NU), torch.float32)
        self._check_tensor(packed, (self.n, self.obs_dim + 2), torch.float32)
        ms = C.c_float()
        check(self._lib.qg_time_step_kernel(self._h, actions.data_ptr(), packed.data_ptr(), int(iters), C.byref(ms)),
              "qg_time_step_kernel")
        return float(ms.value)

    def set_mapping(self, mapping: int):
        """``_abi.MAP_AUTO`` / ``MAP_LANE`` (one env per lane) / ``MAP_QUAD`` (one leg per lane) / ``MAP_PAIR`` (two legs per
        lane, packed f32; built-in robot only) / ``MAP_LINK`` (one link per lane; lagged sensors only)."""
        check(self._lib.qg_set_mapping(self._h, int(mapping)), "qg_set_mapping")

    @property
    def mapping(self) -> int:
        return int(self._lib.qg_get_mapping(self._h))

    def set_task(self, task: QgTask):
        """Replace the task constants of the live handle (``qg_set_task``): reward weights, terminations, auto-reset,
        frame_skip -- what assigning ``env.reward_fns`` / ``env.termination_fns`` after construction does in the reference."""
        check(self._lib.qg_set_task(self._h, C.byref(task)), "qg_set_task")
        self.task = task
        self.limit_substeps = int(self._lib.qg_time_limit_substeps(self.model.timestep, self.task.max_time))

    def set_track_ctrl(self, on: bool):
        check(self._lib.qg_set_track_ctrl(self._h, 1 if on else 0), "qg_set_track_ctrl")

    # -- state snapshot / restore ---------------------------------------------------------------
    def get_state(self):
        qpos = np.empty((self.n, NQ), np.float32)
        qvel = np.empty((self.n, NV), np.float32)
        act = np.empty((self.n, NU), np.float32)
        ctrl = np.empty((self.n, NU), np.float32)
        nstep = np.empty(self.n, np.int32)
        check(self._lib.qg_get_state(self._h, qpos.ctypes.data, qvel.ctypes.data, act.ctypes.data, ctrl.ctypes.data,
                                     nstep.ctypes.data), "qg_get_state")
        return qpos, qvel, act, ctrl, nstep

    def get_reset_streams(self):
        """``(episode[n] int32, seed)``: the per-env episode counters and the batch seed that key every random draw of a (re)set."""
        ep = np.empty(self.n, np.int32)
        seed = C.c_uint64(0)
        check(self._lib.qg_get_reset_streams(self._h, ep.ctypes.data, C.byref(seed)), "qg_get_reset_streams")
        return ep, int(seed.value)

    def set_reset_streams(self, episode, seed):
        ep = np.ascontiguousarray(episode, dtype=np.int32)
        if ep.shape != (self.n,):
            raise ValueError(f"expected shape ({self.n},), got {ep.shape}")
        check(self._lib.qg_set_reset_streams(self._h, ep.ctypes.data, int(seed)), "qg_set_reset_streams")

    def snapshot(self):
        """Everything the simulator keeps per env: ``dict`` of NumPy arrays (physics state + reset streams)."""
        qpos, qvel, act, ctrl, nstep = self.get_state()
        ep, seed = self.get_reset_streams()
        return {"qpos": qpos, "qvel": qvel, "act": act, "ctrl": ctrl, "nstep": nstep, "episode": ep, "seed": seed}

    def restore(self, snap):
        self.set_state(snap["qpos"], snap["qvel"], snap["act"], snap["ctrl"], snap["nstep"])
        self.set_reset_streams(snap["episode"], snap["seed"])

    def set_state(self, qpos=None, qvel=None, act=None, ctrl=None, nstep=None):
        def f32(x, w):
            if x is None:
                return None, None
            a = np.ascontiguousarray(x, dtype=np.float32)
            if a.shape != (self.n, w):
                raise ValueError(f"expected shape ({self.n}, {w}), got {a.shape}")
            return a, a.ctypes.data
        qp, qpp = f32(qpos, NQ)
        qv, qvp = f32(qvel, NV)
        ac, acp = f32(act, NU)
        ct, ctp = f32(ctrl, NU)
        ns, nsp = None, None
        if nstep is not None:
            ns = np.ascontiguousarray(nstep, dtype=np.int32)
            nsp = ns.ctypes.data
        check(self._lib.qg_set_state(self._h, qpp, qvp, acp, ctp, nsp), "qg_set_state")
