"""CPU-side checks of the C-ABI boundary: the shared library loads, exports every symbol that
include/quadgym.h declares, and refuses to compute without a GPU (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from quadruped_gym_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "quadgym.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qg_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(_abi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = _abi.load_library()
    for name in _declared_functions():
        assert hasattr(lib, name), f"libquadgym.so does not export {name}"
    assert b"gfx950" in lib.qg_version()


def test_struct_layouts_match_the_c_side(oracle):
    # the oracle is compiled from the same header; its sizeof() must equal the ctypes mirrors
    L = oracle.lib()
    assert L.qgo_sizeof_model() == C.sizeof(_abi.QgModel)
    assert L.qgo_sizeof_task() == C.sizeof(_abi.QgTask)


def test_defaults_come_from_the_compiled_model(oracle):
    m, t = _abi.default_model(), _abi.default_task()
    mo = oracle.default_model()
    assert bytes(m) == bytes(mo)                       # product and oracle share include/qg_model_data.h
    assert t.frame_skip == 4 and t.max_time == 10.0 and t.use_time_limit == 1   # quadruped.py:43-44,52
    assert list(t.default_ctrl) == [0, 0, -0.5] * 4    # quadruped.py:124
    assert _abi.load_library().qg_time_limit_substeps(0.002, 10.0) == 5000
    assert _abi.load_library().qg_time_limit_substeps(0.002, 20.0) == 10001


def test_model_json_matches_header():
    import json
    j = json.load(open(os.path.join(ROOT, "quadruped-gym_amd", "model", "quadruped_model.json")))
    m = _abi.default_model()
    assert np.allclose([b["mass"] for b in j["bodies"]], list(m.body_mass))
    assert np.allclose(np.array(j["bodies"][3]["inertia"])[[0, 1, 2, 0, 0, 1], [0, 1, 2, 1, 2, 2]], list(m.body_inertia[3]))
    assert [s["adr"] for s in j["sensors"]][-7:] == [12, 15, 18, 21, 24, 27, 30]   # DOCS.md:365-400
    assert j["nsensordata"] == 33


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="this check is for hosts without a GPU")
def test_no_cpu_fallback():
    lib = _abi.load_library()
    h = C.c_void_p()
    rc = lib.qg_create(4, 0, None, None, 0, C.byref(h))
    assert rc == -2 and not h.value                    # QG_ERR_DEVICE
    assert b"no CPU backend" in lib.qg_last_error() or b"hip" in lib.qg_last_error().lower()
    from quadruped_gym_amd.sim import BatchedSim
    with pytest.raises(_abi.QuadGymError):
        BatchedSim(4)


def test_create_rejects_bad_arguments():
    lib = _abi.load_library()
    h = C.c_void_p()
    assert lib.qg_create(0, 0, None, None, 0, C.byref(h)) == -1
    m = _abi.default_model()
    m.jnt_axis[3][0] = 1.0                             # kernels assume hinge axis = link z
    assert lib.qg_create(4, 0, C.byref(m), None, 0, C.byref(h)) == -1
    assert b"hinge axis" in lib.qg_last_error()
    t = _abi.default_task()
    t.frame_skip = 0
    assert lib.qg_create(4, 0, None, C.byref(t), 0, C.byref(h)) == -1


def test_production_build_carries_no_phase_clock():
    """qg_debug_phase_times only answers in a development build (-DQG_PHASE_TIMES, tools/phase_times.sh): the library the tests and
    bench.py load must be the production one, whose kernels contain none of the timing marks."""
    lib = _abi.load_library()
    out = (C.c_uint64 * 16)()
    assert lib.qg_debug_phase_times(out) == -1         # QG_ERR_ARG
    assert b"QG_PHASE_TIMES" in lib.qg_last_error()


def test_recommended_batch_is_the_top_of_the_stair():
    """The step time is a staircase in the batch size (one wave per SIMD of the kernel AUTO picks: 4096 / 16 384 / every further
    32 768 envs on an MI355X); qg_recommended_batch rounds up to the top of the stair -- without a GPU it assumes the MI355X's 1024
    SIMDs.  And no resident-mode entry point does anything without a handle."""
    rb = _abi.recommended_batch
    assert [rb(n) for n in (1, 4096, 4097, 16384, 16385, 32768, 32769, 65536, 65537)] == \
        [4096, 4096, 16384, 16384, 32768, 32768, 65536, 65536, 98304]
    lib = _abi.load_library()
    assert lib.qg_resident_start(None, 1, 0, None, None) == -1 and lib.qg_resident_step_device(None, 1, None) == -1
    assert lib.qg_step_device_seq(None, None, None, 1, None) == -1 and lib.qg_resident_stop(None) == -1
    buf = C.create_string_buffer(8)
    assert lib.qg_device_pci_bus_id(0, buf, 8) == -1   # too small a buffer is refused before any device call
