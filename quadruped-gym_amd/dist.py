"""Multi-GPU rollouts: env batches shard across ranks (one process per GPU), no exchange inside the
physics; per env-step ONE gather of the packed ``[envs_per_rank, obs_dim + 2]`` f32 buffer
(obs, reward, done) to the learner rank -- RCCL over xGMI with the ``nccl`` backend, ``gloo`` on CPU
(tests).  The reference has no distributed layer (its only parallelism is SB3's process-per-env
``SubprocVecEnv``, ``src/train_quadruped.py:49-50``); this is the MI355X counterpart.

Global env index = rank * envs_per_rank + local index; per-env random streams are keyed by the global
index, so results do not depend on how the batch is sharded.
"""
from __future__ import annotations


def shard_range(total_envs: int, world: int, rank: int):
    """Contiguous partition of ``total_envs`` over ``world`` ranks (first ranks take the remainder)."""
    base, rem = divmod(int(total_envs), int(world))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


class PackedGatherer:
    """Per-step gather of the packed rollout buffer to ``dst``, double-buffered so that the gather of
    step t overlaps the physics of step t+1 when the buffers live on a GPU.

    ``submit(packed)`` enqueues the gather of this step's buffer; ``collect()`` waits for the oldest
    outstanding gather and returns the ``[world * n, row]`` tensor on ``dst`` (``None`` elsewhere).
    """

    def __init__(self, n_local: int, row: int, device, dst: int = 0, group=None):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.torch = torch
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dst = dst
        self.n, self.row = int(n_local), int(row)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.out = None
        if self.rank == dst:
            self.out = [torch.empty((self.world, self.n, self.row), device=self.device, dtype=torch.float32) for _ in range(2)]
        self.comm = torch.cuda.Stream(self.device) if self.cuda else None
        self.done = [None, None]
        self.k = 0
        self.pending = []

    def submit(self, packed):
        torch, dist = self.torch, self.dist
        assert tuple(packed.shape) == (self.n, self.row) and packed.dtype == torch.float32
        b = self.k & 1
        glist = [self.out[b][r] for r in range(self.world)] if self.rank == self.dst else None
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ready)
                dist.gather(packed, glist, dst=self.dst, group=self.group)
                ev = torch.cuda.Event()
                ev.record(self.comm)
            self.done[b] = ev
        else:
            dist.gather(packed, glist, dst=self.dst, group=self.group)
        self.pending.append(b)
        self.k += 1

    def wait_buffer_free(self, stream=None):
        """Make ``stream`` (default: current) wait until the gather that read the buffer about to be
        overwritten (two submits ago) has finished."""
        if self.cuda and self.k >= 2:
            ev = self.done[self.k & 1]
            if ev is not None:
                (stream or self.torch.cuda.current_stream(self.device)).wait_event(ev)

    def drain(self):
        """Forget the outstanding gathers without reading them (a throughput loop that never looks at the data);
        buffer reuse stays ordered by ``wait_buffer_free``.  Blocks until the last gather has finished."""
        self.pending.clear()
        if self.cuda:
            for ev in self.done:
                if ev is not None:
                    ev.synchronize()

    def collect(self):
        b = self.pending.pop(0)
        if self.cuda and self.done[b] is not None:
            self.done[b].synchronize()
        if self.rank != self.dst:
            return None
        return self.out[b].view(self.world * self.n, self.row)
