"""Experiment (development build with -DQG_STAGGER_EXP, tools/lib_stagger.so (hipcc ... -DQG_STAGGER_EXP with the experiment patch of docs/EXPERIMENTS.md round 4; the patch is not kept in the tree)): in ONE launch of the one-link-per-lane kernel at two waves per
SIMD, the wave in the ODD wave slot of its SIMD (HW_REG_HW_ID.WAVE_ID) starts late -- does taking the two residents of a SIMD out of
lockstep recover the co-issue that two free-running launches show?"""
import os, sys, ctypes as C, collections, numpy as np, torch
sys.path.insert(0, ".")
src = open("tools/split_launch_ab.py").read().split("for n in (5120, 6144, 8192):")[0]
exec(src)
lib = _abi.load_library()
lib.qg_debug_set_stagger.argtypes = [C.c_int32, C.c_int32]
S = lambda: torch.cuda.Stream(dev)
for n in (8192, 5120):
    for ticks in (0, 25, 50, 100, 200, 400):
        lib.qg_debug_set_stagger(ticks, 1)
        w = sim_of(n, _abi.MAP_LINK); t = time_streams([(w, S())], steps)[0]
        ids = (C.c_uint * 2048)(); lib.qg_debug_get_wave_ids(ids)
        a = np.array(list(ids))[: (n // 4)]
        slots = collections.Counter((a & 0xF).tolist())
        w.close()
        print(f"{n} envs, one link launch, odd wave slots start {ticks / 100:.2f} us late: {t:.2f} us   wave slots in use {dict(sorted(slots.items()))}", flush=True)
