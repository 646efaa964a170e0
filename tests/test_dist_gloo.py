"""world_size-2 gloo test of the rollout gather (runs on CPU): shard ranges, the order of the gathered
rows (global env order) and the double-buffered submit/collect protocol."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from quadruped_gym_amd.dist import ActionScatterer, PackedGatherer, shard_range


def test_shard_range_partitions_contiguously():
    for total, world in [(4096, 8), (10, 3), (7, 8), (262144, 8)]:
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(n for _, n in spans) == total
        for (s0, n0), (s1, _) in zip(spans, spans[1:]):
            assert s1 == s0 + n0
    assert shard_range(262144, 8, 3) == (3 * 32768, 32768)     # BASELINE config 4


def _worker(rank, world, port, n, row, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = PackedGatherer(n, row, "cpu", dst=0)
    got = []
    for k in range(steps):
        # row i of this rank's buffer encodes (global env index, step)
        start, _ = shard_range(world * n, world, rank)
        packed = torch.empty((n, row))
        packed[:, 0] = torch.arange(start, start + n)
        packed[:, 1:] = float(k)
        g.wait_buffer_free()
        g.submit(packed)
        if k >= 1:
            out = g.collect()
            if rank == 0:
                got.append(out.clone())
    out = g.collect()
    if rank == 0:
        got.append(out.clone())
        q.put([t.numpy() for t in got])
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2_gloo():
    world, n, row, steps = 2, 5, 35, 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, row, steps, q)) for r in range(world)]
    [p.start() for p in procs]
    got = q.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert len(got) == steps
    for k, out in enumerate(got):
        assert out.shape == (world * n, row)
        assert np.array_equal(out[:, 0], np.arange(world * n))        # global env order
        assert (out[:, 1:] == k).all()                                # the k-th collect returns the k-th submit


def _scatter_worker(rank, world, port, n, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = ActionScatterer(n, 12, "cpu", src=0)
    gather = PackedGatherer(n, 14, "cpu", dst=0)
    ok = True
    for k in range(steps):
        full = None
        if rank == 0:      # the learner's actions for the whole batch: entry = global env index + step / 100 + column / 10000
            full = (torch.arange(world * n, dtype=torch.float32)[:, None] + k / 100.0 + torch.arange(12)[None, :] / 10000.0).contiguous()
        sc.submit(full)
        mine = sc.wait()
        start, _ = shard_range(world * n, world, rank)
        want = torch.arange(start, start + n, dtype=torch.float32)[:, None] + k / 100.0 + torch.arange(12)[None, :] / 10000.0
        ok = ok and bool(torch.equal(mine, want))
        # closed loop: what each rank "simulated" from its slice comes back in global env order
        packed = torch.cat([mine, torch.full((n, 2), float(rank))], dim=1)
        gather.wait_buffer_free()
        gather.submit(packed)
        out = gather.collect()
        if rank == 0:
            ok = ok and bool(torch.equal(out[:, :12], full))
    if rank == 0:
        try:
            sc.submit(torch.zeros((3, 12)))
            ok = False
        except ValueError:
            pass
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    if rank == 0:
        q.put(flags)
    dist.barrier()
    dist.destroy_process_group()


def test_action_scatter_world2_gloo():
    """SURVEY §8e downward leg: [N,12] actions from the learner rank to the rank that owns each env, and the closed
    loop scatter -> (step) -> gather returns rows in global env order."""
    world, n, steps = 2, 6, 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_scatter_worker, args=(r, world, port, n, steps, q)) for r in range(world)]
    [p.start() for p in procs]
    flags = q.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert flags == [True] * world


def _negotiate_worker(rank, world, port, fail_rank, fail_phase, q):
    """One rank of bench.py's --exchange auto attempt with mock capture / replay callables: `fail_rank` raises in `fail_phase`."""
    from quadruped_gym_amd.dist import make_all_ok, negotiate_graph_replay
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    state = {"graph": None}

    def phase(name):
        def fn():
            calls.append(name)
            if rank == fail_rank and name == fail_phase:
                raise RuntimeError(f"injected {name} failure on rank {rank}")
            if name == "capture":
                state["graph"] = "captured"
        return fn

    def drop_graph():
        calls.append("drop")
        state["graph"] = None

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    att = negotiate_graph_replay(make_all_ok(None, "cpu"), reduce_max, phase("capture"), phase("warm"), phase("timed"), drop_graph)
    # whatever happened, the ranks are still in step: the eager path's collectives that follow must match up
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    outs = [None] * world
    dist.all_gather_object(outs, {"captured": att.captured, "agreed": att.agreed, "ok": att.ok, "note": att.note, "seconds": att.seconds,
                                  "calls": calls, "graph": state["graph"], "sum": float(t.item())})
    if rank == 0:
        q.put(outs)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_phase", ["capture", "warm", "timed", None])
def test_graph_exchange_agreement_with_one_failing_rank_world2_gloo(fail_phase):
    """bench.py --exchange auto, the case a one-GPU box cannot produce: rank 1 fails to capture (or to replay) while rank 0 succeeds.
    Both ranks must leave the attempt on the eager path, in step with each other (no hang, exit status 0, later collectives match);
    with no failure both agree on the replay's time."""
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_negotiate_worker, args=(r, world, port, 1, fail_phase, q)) for r in range(world)]
    [p.start() for p in procs]
    outs = q.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    r0, r1 = outs
    assert r0["sum"] == r1["sum"] == 3.0                        # the collective after the attempt matched up
    assert r0["captured"] == r1["captured"] and r0["agreed"] == r1["agreed"]
    if fail_phase is None:
        assert r0["agreed"] and r1["agreed"] and r0["ok"] == r1["ok"] == 1
        assert r0["seconds"] == r1["seconds"] and np.isfinite(r0["seconds"])     # MAX over ranks
        assert r0["graph"] == r1["graph"] == "captured" and "drop" not in r0["calls"] + r1["calls"]
        return
    assert not r0["agreed"] and not r1["agreed"]
    assert r0["ok"] == 1 and r0["note"] is None                 # the healthy rank saw no error of its own ...
    assert r1["ok"] == 0 and "injected" in r1["note"]
    assert r0["graph"] is None and r1["graph"] is None          # ... and still went back to the eager path
    assert r0["calls"][-1] == "drop" and r1["calls"][-1] == "drop"
    assert r0["seconds"] == float("inf")
    want = {"capture": ["capture", "drop"], "warm": ["capture", "warm", "drop"], "timed": ["capture", "warm", "timed", "drop"]}[fail_phase]
    assert r0["calls"] == want and r1["calls"] == want          # nobody replays a graph its peer does not hold
    assert r0["captured"] == (fail_phase != "capture")


def _describe_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quadruped_gym_amd.dist import describe_group
    # rank 1 reports the same GPU as rank 0 on purpose: the summary must show 2 ranks on ONE device
    d = describe_group(device_ordinal=0, pci_bus_id="0000:05:00.0", tensor_device="cpu")
    q.put((rank, d))
    dist.barrier()
    dist.destroy_process_group()


def test_describe_group_world2_gloo():
    """``config.rccl`` of bench.py's N > 1 line: what the process group itself reports about its members (one all-gather over the
    group), the same on every rank -- and it makes two ranks that share a GPU visible."""
    from quadruped_gym_amd.dist import pci_to_ints
    assert pci_to_ints("0000:05:00.0") == (0, 5, 0, 0) and pci_to_ints("0001:e3:00.0") == (1, 0xE3, 0, 0) and pci_to_ints("junk") == (-1,) * 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_describe_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert got[0] == got[1]
    d = got[0]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and d["ranks_seen"] == 2 and d["distinct_gpus"] == 1
    assert [m["rank"] for m in d["members"]] == [0, 1] and d["members"][1]["pci_bus_id"] == "0000:05:00.0"
