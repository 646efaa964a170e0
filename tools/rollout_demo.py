"""Closed-loop rollout with the policy on the GPU: the reference's training setup
(``POWalkingQuadrupedEnv(obs_window=10)``, ``random_controls``, SB3 ``MlpPolicy`` = two tanh layers of 64,
``src/train_quadruped.py:16-22,50-64``) with N robots in one batch -- observation stack, policy and physics never leave
the device.  Random weights (no checkpoint offline); measures env-steps/s with the policy in the loop, eager and with the
whole step (policy + 4 launches of the env) captured in one hipGraph.

usage: python tools/rollout_demo.py [num_envs] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from quadruped_gym_amd.envs.walking import POWalkingQuadrupedVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda:0")
env = POWalkingQuadrupedVecEnv(n, obs_window=10, random_controls=True, random_init=True, device_commands=True,
                               reset_options={"min_speed": 0.0, "max_speed": 0.5}, settling_time=0.5, max_time=10.0)
torch.manual_seed(0)
policy = torch.nn.Sequential(torch.nn.Linear(env.obs_dim, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                             torch.nn.Linear(64, 12), torch.nn.Tanh()).to(dev)
obs = torch.from_numpy(env.reset()).to(dev)
rew = torch.empty(n, device=dev)
done = torch.empty(n, device=dev, dtype=torch.uint8)
act = torch.empty((n, 12), device=dev)
ret = torch.zeros(n, device=dev)


def one_step():
    with torch.no_grad():
        act.copy_(policy(obs))
    env.step_tensor(act, obs, rew, done)          # obs is overwritten in place with the next observation stack
    ret.add_(rew.nan_to_num())


def timed(fn, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for _ in range(20):
    one_step()
dt = timed(one_step, steps)
print(f"eager : {n} envs, policy in the loop: {dt / steps * 1e6:7.1f} us/step  {n * steps / dt / 1e6:7.1f} M env-steps/s")

side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for _ in range(3):
        one_step()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    one_step()
for _ in range(20):
    graph.replay()
dt = timed(graph.replay, steps)
print(f"graph : {n} envs, policy in the loop: {dt / steps * 1e6:7.1f} us/step  {n * steps / dt / 1e6:7.1f} M env-steps/s")
print(f"finite obs: {bool(torch.isfinite(obs).all())}; mean return so far {float(ret.mean()):.2f}; episodes finished this step: {int(done.sum())}")
env.close()
