"""ctypes mirrors of the C structs in ``include/quadgym.h`` and the loader of
``libquadgym.so`` (the HIP pipeline behind the C ABI).

There is no CPU fallback: if the shared library is missing, or no HIP device is
usable, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

NBODY, NJNT, NQ, NV, NU, NSENSOR, MAXCP, NREWARD = 13, 12, 19, 18, 12, 33, 12, 3
OBS_FULL, OBS_IMU = 0, 1
RESET_RANDOM_YAW = 1
RESET_JOINT_JITTER = 2
CMD_FIXED_HEADING, CMD_FIXED_VELOCITY_ANGLE, CMD_FIXED_SPEED = 1, 2, 4
MAP_AUTO, MAP_LANE, MAP_QUAD, MAP_PAIR, MAP_LINK = 0, 1, 2, 3, 4
OBS_DIM = {OBS_FULL: 33, OBS_IMU: 21}


class QgModel(C.Structure):
    _fields_ = [
        ("timestep", C.c_double),
        ("gravity", C.c_double * 3),
        ("body_parent", C.c_int32 * NBODY),
        ("body_pos", (C.c_double * 3) * NBODY),
        ("body_quat", (C.c_double * 4) * NBODY),
        ("body_mass", C.c_double * NBODY),
        ("body_ipos", (C.c_double * 3) * NBODY),
        ("body_inertia", (C.c_double * 6) * NBODY),
        ("jnt_axis", (C.c_double * 3) * NJNT),
        ("jnt_ref", C.c_double * NJNT),
        ("jnt_range", (C.c_double * 2) * NJNT),
        ("jnt_damping", C.c_double * NJNT),
        ("jnt_armature", C.c_double * NJNT),
        ("free_damping", C.c_double),
        ("free_armature", C.c_double),
        ("act_kp", C.c_double * NU),
        ("act_kv", C.c_double * NU),
        ("act_gear", C.c_double * NU),
        ("act_timeconst", C.c_double * NU),
        ("act_ctrlrange", (C.c_double * 2) * NU),
        ("act_forcerange", (C.c_double * 2) * NU),
        ("limit_stiffness", C.c_double),
        ("limit_damping", C.c_double),
        ("limit_ramp", C.c_double),
        ("ncp", C.c_int32 * NBODY),
        ("cp", ((C.c_double * 3) * MAXCP) * NBODY),
        ("contact_stiffness", C.c_double),
        ("contact_damping", C.c_double),
        ("contact_margin", C.c_double),
        ("contact_friction", C.c_double),
        ("contact_ramp", C.c_double),
        ("qpos0", C.c_double * NQ),
    ]


class QgTask(C.Structure):
    _fields_ = [
        ("frame_skip", C.c_int32),
        ("max_time", C.c_double),
        ("use_time_limit", C.c_int32),
        ("use_fall", C.c_int32),
        ("fall_height", C.c_double),
        ("use_flip", C.c_int32),
        ("w_forward", C.c_double),
        ("w_ctrl", C.c_double),
        ("alive_bonus", C.c_double),
        ("obs_mode", C.c_int32),
        ("sensor_lag", C.c_int32),
        ("auto_reset", C.c_int32),
        ("reset_flags", C.c_uint32),
        ("default_ctrl", C.c_double * NU),
        ("reset_joint_jitter", C.c_double),
    ]


class QgCommandSampler(C.Structure):
    """``qg_command_sampler``: the options of ``VelocityHeadingControls.sample`` (``control_inputs.py:74-115``)."""
    _fields_ = [
        ("fixed", C.c_uint32),
        ("min_speed", C.c_double),
        ("max_speed", C.c_double),
        ("fixed_heading_angle", C.c_double),
        ("fixed_velocity_angle", C.c_double),
        ("fixed_speed", C.c_double),
    ]

    @classmethod
    def from_options(cls, options=None):
        o = options or {}
        s = cls()
        s.min_speed, s.max_speed = float(o.get("min_speed", 0.0)), float(o.get("max_speed", 1.0))
        for bit, key, field in ((CMD_FIXED_HEADING, "fixed_heading_angle", "fixed_heading_angle"),
                                (CMD_FIXED_VELOCITY_ANGLE, "fixed_velocity_angle", "fixed_velocity_angle"),
                                (CMD_FIXED_SPEED, "fixed_speed", "fixed_speed")):
            if o.get(key) is not None:
                s.fixed |= bit
                setattr(s, field, float(o[key]))
        return s


class QgWalkParams(C.Structure):
    _fields_ = [
        ("settling_time", C.c_double),
        ("joint_centers", C.c_double * NU),
        ("ema_alpha", C.c_double),
        ("min_freq", C.c_double),
        ("control_cost_alpha", C.c_double),
        ("w", C.c_double * 10),
        ("w_diff_ideal", C.c_double),
        ("body_height", C.c_double),
        ("amp_target", C.c_double * NU),
        ("freq_target", C.c_double * NU),
        ("unit_zero", C.c_int32),
    ]


NWALKREWARD = 11


def package_dir() -> str:
    return os.path.dirname(os.path.abspath(__file__))


def library_path() -> str:
    # QUADGYM_LIB selects another build of the same library (A/B timing of kernel variants on one box)
    return os.environ.get("QUADGYM_LIB") or os.path.join(package_dir(), "csrc", "libquadgym.so")


_lib = None


def load_library():
    """Load ``libquadgym.so`` and declare the prototypes of every exported entry
    point (``include/quadgym.h``).  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C quadruped-gym_amd/csrc`). "
            "There is no CPU fallback.")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 (same SONAME as the system's).  Loaded after
    # ours it becomes a SECOND runtime and torch then finds no GPU; loaded first, libquadgym.so's dependency resolves
    # to that same copy.  So when torch is installed, load it before the library.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path)
    vp, u8p, fp, i32p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int32)
    lib.qg_version.restype = C.c_char_p
    lib.qg_version.argtypes = []
    lib.qg_build_id.restype = C.c_char_p
    lib.qg_build_id.argtypes = []
    lib.qg_last_error.restype = C.c_char_p
    lib.qg_last_error.argtypes = []
    lib.qg_device_pci_bus_id.argtypes = [C.c_int32, C.c_char_p, C.c_int32]
    lib.qg_recommended_batch.restype = C.c_int32
    lib.qg_recommended_batch.argtypes = [C.c_int32, C.c_int32]
    lib.qg_default_model.argtypes = [C.POINTER(QgModel)]
    lib.qg_default_task.argtypes = [C.POINTER(QgTask)]
    lib.qg_time_limit_substeps.restype = C.c_int64
    lib.qg_time_limit_substeps.argtypes = [C.c_double, C.c_double]
    lib.qg_create.argtypes = [C.c_int32, C.c_int32, C.POINTER(QgModel), C.POINTER(QgTask), C.c_uint64, C.POINTER(vp)]
    lib.qg_destroy.argtypes = [vp]
    lib.qg_num_envs.argtypes = [vp]
    lib.qg_obs_dim.argtypes = [vp]
    lib.qg_reset.argtypes = [vp, vp, C.c_uint64, C.c_uint32]
    lib.qg_step.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.qg_step_device.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.qg_step_device_packed.argtypes = [vp, vp, vp, vp]
    lib.qg_get_state.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.qg_step_mirror.argtypes = [vp] * 11
    lib.qg_set_state.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.qg_time_step_kernel.argtypes = [vp, vp, vp, C.c_int32, C.POINTER(C.c_float)]
    lib.qg_set_track_ctrl.argtypes = [vp, C.c_int32]
    lib.qg_debug_phase_times.argtypes = [vp]
    lib.qg_set_task.argtypes = [vp, C.POINTER(QgTask)]
    lib.qg_get_task.argtypes = [vp, C.POINTER(QgTask)]
    lib.qg_uses_baked_model.argtypes = [vp]
    lib.qg_set_mapping.argtypes = [vp, C.c_int32]
    lib.qg_get_mapping.argtypes = [vp]
    lib.qg_comm_unique_id.argtypes = [vp]
    lib.qg_comm_create.argtypes = [vp, C.c_int32, C.c_int32, vp, C.POINTER(vp)]
    lib.qg_comm_destroy.argtypes = [vp]
    lib.qg_comm_rollout.argtypes = [vp, vp, C.c_int32, vp, vp, C.c_int32, C.c_int32]
    lib.qg_comm_synchronize.argtypes = [vp]
    lib.qg_walk_default_params.argtypes = [C.POINTER(QgWalkParams)]
    lib.qg_walk_create.argtypes = [vp, C.POINTER(QgWalkParams), C.POINTER(vp)]
    lib.qg_walk_destroy.argtypes = [vp]
    lib.qg_walk_set_commands.argtypes = [vp, vp, vp]
    lib.qg_walk_reset.argtypes = [vp, vp, C.c_uint64, C.c_uint32]
    lib.qg_walk_step.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.qg_walk_step_device.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.qg_walk_get_estimates.argtypes = [vp, vp, vp, vp]
    lib.qg_walk_set_command_sampler.argtypes = [vp, vp]
    lib.qg_walk_get_commands.argtypes = [vp, vp, vp]
    lib.qg_po_create.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    lib.qg_po_destroy.argtypes = [vp]
    lib.qg_po_obs_dim.argtypes = [vp]
    lib.qg_po_reset.argtypes = [vp, vp, C.c_uint64, C.c_uint32, vp]
    lib.qg_po_step.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.qg_walk_state_bytes.restype = C.c_int64
    lib.qg_walk_state_bytes.argtypes = [vp]
    lib.qg_walk_get_state.argtypes = [vp, vp]
    lib.qg_walk_set_state.argtypes = [vp, vp]
    lib.qg_po_state_bytes.restype = C.c_int64
    lib.qg_po_state_bytes.argtypes = [vp]
    lib.qg_po_get_state.argtypes = [vp, vp]
    lib.qg_po_set_state.argtypes = [vp, vp]
    lib.qg_get_reset_streams.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
    lib.qg_set_reset_streams.argtypes = [vp, vp, C.c_uint64]
    lib.qg_po_step_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.qg_step_device_seq.argtypes = [vp, vp, vp, C.c_int32, vp]
    lib.qg_resident_start.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
    lib.qg_resident_stop.argtypes = [vp]
    lib.qg_resident_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int32)]
    lib.qg_resident_step_device.argtypes = [vp, C.c_int32, vp]
    lib.qg_resident_ensure.argtypes = [vp]
    lib.qg_resident_status.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("qg_version", "qg_build_id", "qg_last_error", "qg_time_limit_substeps", "qg_walk_state_bytes", "qg_po_state_bytes",
                        "qg_recommended_batch"):
            fn.restype = C.c_int
    _lib = lib
    return lib


# every symbol include/quadgym.h declares
EXPORTS = (
    "qg_version", "qg_build_id", "qg_last_error", "qg_device_pci_bus_id", "qg_recommended_batch", "qg_default_model", "qg_default_task", "qg_time_limit_substeps",
    "qg_create", "qg_destroy", "qg_num_envs", "qg_obs_dim", "qg_reset", "qg_step", "qg_step_device",
    "qg_step_device_packed", "qg_get_state", "qg_step_mirror", "qg_set_state", "qg_time_step_kernel", "qg_set_track_ctrl", "qg_debug_phase_times", "qg_set_task", "qg_get_task",
    "qg_uses_baked_model", "qg_set_mapping", "qg_get_mapping",
    "qg_comm_unique_id", "qg_comm_create", "qg_comm_destroy", "qg_comm_rollout", "qg_comm_synchronize",
    "qg_walk_default_params", "qg_walk_create", "qg_walk_destroy", "qg_walk_set_commands", "qg_walk_reset", "qg_walk_step",
    "qg_walk_step_device", "qg_walk_get_estimates", "qg_walk_set_command_sampler", "qg_walk_get_commands",
    "qg_po_create", "qg_po_destroy", "qg_po_obs_dim", "qg_po_reset", "qg_po_step", "qg_po_step_device",
    "qg_walk_state_bytes", "qg_walk_get_state", "qg_walk_set_state", "qg_po_state_bytes", "qg_po_get_state", "qg_po_set_state",
    "qg_get_reset_streams", "qg_set_reset_streams",
    "qg_step_device_seq", "qg_resident_start", "qg_resident_stop", "qg_resident_buffers", "qg_resident_step_device",
    "qg_resident_ensure", "qg_resident_status",
)


class QuadGymError(RuntimeError):
    pass


def check(status: int, what: str):
    if status != 0:
        msg = load_library().qg_last_error().decode("utf-8", "replace")
        raise QuadGymError(f"{what} failed ({status}): {msg}")


def recommended_batch(n_envs: int, device: int = -1) -> int:
    """The batch size at the top of the step-time stair ``n_envs`` stands on (``qg_recommended_batch``): 4096, 16 384, then multiples
    of 32 768 on an MI355X.  ``device=-1`` assumes an MI355X (no GPU needed)."""
    return int(load_library().qg_recommended_batch(int(n_envs), int(device)))


def default_model() -> QgModel:
    m = QgModel()
    check(load_library().qg_default_model(C.byref(m)), "qg_default_model")
    return m


def default_task() -> QgTask:
    t = QgTask()
    check(load_library().qg_default_task(C.byref(t)), "qg_default_task")
    return t
