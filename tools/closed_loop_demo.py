"""A policy IN the loop with the plain QuadrupedEnv step (README reward set, 33-value observation; SURVEY section 8 f3: obs -> action ->
step, /root/reference/src/train_quadruped.py:187-191): a 33-64-64-12 tanh MLP (SB3's MlpPolicy shape, random weights) reads the
step's rows and writes the next actions, everything on one stream, G closed-loop steps captured in one hipGraph.  Three ways to run
the env step inside it:
  launch    one kernel launch per env-step (qg_step_device_packed) -- the default path;
  resident  the resident step kernel, ONE ring per env-step behind the policy (qg_resident_step_device; 1-slot mailbox);
  policy    the policy alone (what the loop costs without any env step).
usage (GPU box): python tools/closed_loop_demo.py [num_envs] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from quadruped_gym_amd import _abi
from quadruped_gym_amd.sim import BatchedSim

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
G = 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = torch.nn.Sequential(torch.nn.Linear(33, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                             torch.nn.Linear(64, 12), torch.nn.Tanh()).to(dev)


def task():
    t = _abi.default_task()
    t.auto_reset, t.use_fall, t.fall_height = 1, 1, 0.05
    return t


def run(mode):
    sim = BatchedSim(n, task=task())
    sim.reset(seed=0)
    acts = torch.zeros((1, n, 12), device=dev)
    rows = torch.zeros((1, n, 35), device=dev)
    if mode == "resident":
        sim.resident_start(acts, rows, idle_timeout_us=20000)

    def one_step():
        with torch.no_grad():
            acts[0].copy_(policy(rows[0, :, :33]))
        if mode == "launch":
            sim.step_device_packed(acts[0], rows[0])
        elif mode == "resident":
            sim.resident_step(1)

    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        for _ in range(3):
            one_step()
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(G):
            one_step()
    # (a graph replays on the CURRENT stream; that stream -- not the device -- is what gets synchronised: a device-wide wait would also
    # wait for the resident kernel, i.e. until it leaves for lack of rings)
    with torch.cuda.stream(side):
        if mode == "resident":
            sim.resident_ensure()
        for _ in range(20):
            graph.replay()
        side.synchronize()
        if mode == "resident":
            sim.resident_ensure()
        t0 = time.perf_counter()
        for _ in range(steps // G):
            graph.replay()
        side.synchronize()
        dt = time.perf_counter() - t0
    k = (steps // G) * G
    extra = ""
    if mode == "resident":
        st = sim.resident_status()
        extra = f"  (rings not executed: {st['not_executed']})"
    print(f"{mode:9s}: {n} envs, hipGraph of {G} closed-loop steps: {dt / k * 1e6:7.2f} us/step  {n * k / dt / 1e6:7.1f} M env-steps/s"
          f"  finite {bool(torch.isfinite(rows).all())}{extra}", flush=True)
    if mode == "resident":
        sim.resident_stop()
    sim.close()
    return dt / k * 1e6


t_pol = run("policy")
t_launch = run("launch")
t_res = run("resident")
print(f"env step inside the loop: launch {t_launch - t_pol:6.2f} us, resident ring {t_res - t_pol:6.2f} us")
