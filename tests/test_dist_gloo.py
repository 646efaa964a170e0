"""world_size-2 gloo test of the rollout gather (runs on CPU): shard ranges, the order of the gathered
rows (global env order) and the double-buffered submit/collect protocol."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from quadruped_gym_amd.dist import PackedGatherer, shard_range


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
