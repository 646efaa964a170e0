"""Where an env-step of the RESIDENT kernel goes in closed loop (one ring per env-step), from inside the kernels: the first wave of the
resident grid and the ring kernel stamp the 100 MHz clock (QG_MARK in qg_kernel_resident.hip; development build, tools/phase_times.sh).
usage (GPU box): QUADGYM_LIB=tools/lib_phase.so python tools/phase_times_resident.py [n_envs] [frame_skip]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_gym_amd import _abi  # noqa: E402
from quadruped_gym_amd._abi import check  # noqa: E402
from quadruped_gym_amd.sim import BatchedSim  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lib = _abi.load_library()
dev = torch.device("cuda:0")
t = _abi.default_task(); t.frame_skip = fs; t.auto_reset = 1
sim = BatchedSim(n, task=t)
mail_a = torch.rand((16, n, 12), device=dev) * 2 - 1
mail_p = torch.zeros((16, n, 35), device=dev)
sim.resident_start(mail_a, mail_p)
ring = sim.bind_resident_step(1)
for _ in range(2000):
    ring()
torch.cuda.current_stream().synchronize()
rows = []
for rep in range(40):
    for _ in range(64):
        ring()
    torch.cuda.current_stream().synchronize()
    out = (C.c_uint64 * 16)()
    check(lib.qg_debug_phase_times(out), "qg_debug_phase_times")      # (waits for the device: the kernel leaves, the next ring launches it again)
    rows.append(np.array(list(out), dtype=np.float64))
v = np.median(np.array(rows), axis=0) * 10.0        # ns
# the stamps of the LAST env-step of a burst: ring entered (8) -> door advanced (9) -> wave 0 sees it (1) -> action (2) -> physics (3)
# -> rows issued (4) -> drained + arrival (5) -> ring sees every shard (10)
seq = [(8, "ring kernel entered"), (9, "door advanced"), (1, "wave 0 sees the ring"), (2, "action in a register"), (3, "physics done"),
       (4, "rows issued"), (5, "rows drained, arrival issued"), (10, "ring sees every shard")]
print(f"resident closed loop, {n} envs, frame_skip {fs}: ns since the ring kernel's entry (median of 40 bursts, last env-step of each)")
prev = v[8]
for i, nm in seq:
    print(f"  {nm:30s} {v[i] - v[8]:8.0f} ns   (+{v[i] - prev:6.0f})")
    prev = v[i]
sim.resident_stop()
