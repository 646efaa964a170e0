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


def pci_to_ints(bus_id: str):
    """'0000:05:00.0' -> (domain, bus, device, function); anything unparsable -> four -1."""
    try:
        dom, bus, rest = bus_id.strip().split(":")
        dev, fn = rest.split(".")
        return int(dom, 16), int(bus, 16), int(dev, 16), int(fn, 16)
    except Exception:
        return -1, -1, -1, -1


def describe_group(device_ordinal: int, pci_bus_id: str, tensor_device, group=None):
    """What the process group itself says about who is in it: every rank contributes (rank, device ordinal, PCI bus id) through ONE
    all-gather over the group (RCCL for the ``nccl`` backend, so the answer comes over the same communicator the rollouts use) and
    gets the same summary back -- ``world_size`` as the communicator reports it, the sorted members, how many distinct GPUs they
    sit on.  ``bench.py`` prints it as ``config.rccl`` in every N > 1 line: "did RCCL see N ranks on N devices" is answerable from
    the line."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = torch.tensor([rank, int(device_ordinal), *pci_to_ints(pci_bus_id)], dtype=torch.int64, device=tensor_device)
    everyone = torch.empty((world, 6), dtype=torch.int64, device=tensor_device)
    dist.all_gather_into_tensor(everyone.view(-1), mine, group=group)
    rows = sorted(tuple(int(x) for x in r) for r in everyone.cpu().tolist())
    members = [{"rank": r[0], "device_ordinal": r[1], "pci_bus_id": ("%04x:%02x:%02x.%x" % r[2:6]) if r[2] >= 0 else None} for r in rows]
    gpus = {m["pci_bus_id"] if m["pci_bus_id"] is not None else ("ordinal", m["device_ordinal"]) for m in members}
    return {"backend": dist.get_backend(group), "world_size": world, "ranks_seen": len({m["rank"] for m in members}),
            "distinct_gpus": len(gpus), "members": members}


class PackedGatherer:
    """Per-step exchange of the packed rollout buffer, double-buffered so that the collective of step t overlaps the
    physics of step t+1 when the buffers live on a GPU.

    ``submit(packed)`` issues the collective asynchronously (``async_op=True``: RCCL runs it on the process group's own
    stream, ordered after the work already queued on the current stream) and returns at once; ``wait_buffer_free()``
    makes the current stream wait for the collective that read the buffer about to be overwritten (two submits ago) --
    stream-level ordering on a GPU, no host synchronisation; ``collect()`` waits for the oldest outstanding collective
    and returns the ``[world * n, row]`` tensor (on ``dst`` for ``op="gather"``, on every rank for ``"all_gather"``).
    Host-side cost per step is one collective call (measured with a 1-rank RCCL group: see DESIGN.md).
    """

    def __init__(self, n_local: int, row: int, device, dst: int = 0, group=None, op: str = "gather"):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.torch = torch
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dst = dst
        self.op = op              # "gather": rows land on dst only; "all_gather": every rank receives them (one ring collective)
        self.n, self.row = int(n_local), int(row)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.out = None
        if self.rank == dst or op == "all_gather":
            self.out = [torch.empty((self.world, self.n, self.row), device=self.device, dtype=torch.float32) for _ in range(2)]
        self.flat = [o.view(self.world * self.n, self.row) for o in self.out] if self.out is not None else None
        self.lists = [[o[r] for r in range(self.world)] for o in self.out] if (self.out is not None and op == "gather") else None
        self.work = [None, None]
        self.k = 0
        self.pending = []

    def submit(self, packed):
        dist = self.dist
        b = self.k & 1
        if self.op == "all_gather":
            w = dist.all_gather_into_tensor(self.flat[b], packed, group=self.group, async_op=True)
        else:
            w = dist.gather(packed, self.lists[b] if self.rank == self.dst else None, dst=self.dst, group=self.group, async_op=True)
        self.work[b] = w
        self.pending.append(b)
        self.k += 1

    def wait_buffer_free(self, stream=None):
        """Order the current stream after the collective that read the buffer about to be overwritten (two submits ago)."""
        w = self.work[self.k & 1]
        if w is not None:
            w.wait()              # NCCL/RCCL: the current stream waits; gloo: the host waits
            self.work[self.k & 1] = None

    def drain(self):
        """Finish every outstanding collective (a throughput loop that never looks at the data)."""
        self.pending.clear()
        for b in (0, 1):
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None
        if self.cuda:
            self.torch.cuda.synchronize(self.device)

    def collect(self):
        b = self.pending.pop(0)
        if self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None
        if self.cuda:
            self.torch.cuda.current_stream(self.device).synchronize()
        if self.out is None:
            return None
        return self.flat[b]


class ActionScatterer:
    """The other direction of the per-step exchange: the learner rank holds the actions of the whole batch,
    ``[world * n, 12]`` f32 in global env order, and every rank receives the ``[n, 12]`` slice of the envs it owns.

    One ``torch.distributed.scatter`` per env-step (RCCL over xGMI on GPUs, gloo on CPU), asynchronous like
    ``PackedGatherer``: ``submit(all_actions)`` on the source rank (``None`` elsewhere) returns at once, ``wait()`` orders the
    current stream (RCCL) or the host (gloo) after the transfer and returns this rank's slice.  Double-buffered, so the
    scatter of step t+1 may be issued while the physics of step t still reads the previous slice.
    """

    def __init__(self, n_local: int, width: int, device, src: int = 0, group=None):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.src = src
        self.n, self.width = int(n_local), int(width)
        self.device = torch.device(device)
        self.local = [torch.empty((self.n, self.width), device=self.device, dtype=torch.float32) for _ in range(2)]
        self.work = [None, None]
        self.k = 0
        self.pending = []

    def submit(self, all_actions=None):
        b = self.k & 1
        chunks = None
        if self.rank == self.src:
            if all_actions is None or tuple(all_actions.shape) != (self.world * self.n, self.width):
                raise ValueError(f"the source rank must pass a [{self.world * self.n}, {self.width}] tensor")
            if all_actions.dtype != self.local[b].dtype or not all_actions.is_contiguous():
                raise ValueError("actions must be contiguous float32")
            chunks = list(all_actions.view(self.world, self.n, self.width).unbind(0))
        if len(self.pending) == 2:
            raise RuntimeError("two scatters are outstanding: call wait() before submitting a third")
        self.work[b] = self.dist.scatter(self.local[b], chunks, src=self.src, group=self.group, async_op=True)
        self.pending.append(b)
        self.k += 1

    def wait(self):
        """This rank's ``[n, width]`` slice of the oldest outstanding scatter."""
        b = self.pending.pop(0)
        self.work[b].wait()
        self.work[b] = None
        return self.local[b]


class GraphAttempt:
    """Outcome of ``negotiate_graph_replay`` on this rank (identical ``captured`` / ``agreed`` on every rank)."""

    def __init__(self):
        self.captured = False     # every rank holds a captured graph
        self.agreed = False       # every rank replayed it and finished the timed run
        self.ok = 1               # this rank's own view: 0 as soon as anything raised here
        self.note = None          # what raised here, if anything
        self.seconds = float("inf")   # the timed run, MAX over ranks (only meaningful when ``agreed``)


def make_all_ok(group=None, device=None):
    """``all_ok(flag) -> bool``: all-reduce MIN of a 0/1 flag over ``group``; on a GPU the call returns after the device has
    finished (so it doubles as barrier + synchronise)."""
    import torch
    import torch.distributed as dist
    dev = torch.device(device) if device is not None else torch.device("cpu")

    def all_ok(flag):
        t = torch.tensor([int(flag)], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        return int(t.item()) == 1
    return all_ok


def negotiate_graph_replay(all_ok, reduce_max, capture, warm_replay, timed_replay, drop_graph, clock=None):
    """The agreement protocol of ``bench.py --exchange auto``: try the K timed steps once more as hipGraph replays WITHOUT ever
    leaving two ranks inside different collectives.

    Every rank calls this with the same arguments; every collective in here is reached by every rank whatever happened locally:

    1. ``capture()`` (local, may raise)                        -> ``all_ok``: does EVERY rank hold a graph?   no -> drop, done
    2. ``warm_replay()`` (local work + local fence, may raise) -> ``all_ok``: did every rank get through one replay?  no -> drop, done
       (this all-reduce is the opening barrier + synchronise of the timed region)
    3. ``timed_replay()`` (exactly K steps + LOCAL fence, may raise) -> ``all_ok`` (the closing barrier) -> ``reduce_max(seconds)``

    ``drop_graph()`` puts this rank back on the eager path (no collective in it).  A rank whose callable raises never skips a
    collective, it only votes 0.  What the protocol cannot repair is a collective INSIDE a replayed graph that a failed peer never
    joins: that is a stalled GPU, which the caller's watchdog reports with a non-zero exit status.  Returns a ``GraphAttempt``.
    """
    import time
    clock = clock or time.perf_counter
    out = GraphAttempt()

    def note_of(e):
        return f"{type(e).__name__}: {e}"[:300]

    try:
        capture()
    except Exception as e:                       # capture refused on this rank
        out.ok, out.note = 0, note_of(e)
    out.captured = all_ok(out.ok)
    if not out.captured:
        drop_graph()
        return out
    try:
        warm_replay()
    except Exception as e:
        out.ok, out.note = 0, note_of(e)
    if not all_ok(out.ok):                       # opening barrier of the timed region
        drop_graph()
        return out
    t0 = clock()
    try:
        timed_replay()
    except Exception as e:
        out.ok, out.note = 0, note_of(e)
    out.agreed = all_ok(out.ok)                  # closing barrier: every rank's K steps are done (or some rank gave up)
    dt = clock() - t0
    if not out.agreed:
        drop_graph()
        return out
    out.seconds = float(reduce_max(dt))
    return out
