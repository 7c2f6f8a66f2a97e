"""Pair sharding across ranks, exercised with two gloo processes on the CPU (the same code path
bench.py runs over RCCL with one process per GPU)."""
import os
import socket
import sys

import pytest

from cameracalibration_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blocks_cover_every_pair_once():
    for total in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, count = shard.pair_block(total, r, world)
                seen.extend(range(first, first + count))
            assert seen == list(range(total))
            counts = [shard.pair_block(total, r, world)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        shard.pair_block(4, 2, 2)


def test_seeds_are_distinct_per_pair():
    assert len({shard.pair_seed(i) for i in range(512)}) == 512
    assert shard.pair_seed(0) == 0x5EED0001 and shard.pair_seed(1) == 0x5EED0003


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total = 13
        first, count = shard.pair_block(total, rank, world)
        mine = torch.zeros(total, dtype=torch.int32)
        mine[first:first + count] = 1
        dist.all_reduce(mine)                       # test-only collective: every pair owned exactly once
        slow = shard.slowest_rank_seconds(1.0 + rank, dist)
        dist.barrier()
        q.put((rank, bool((mine == 1).all()), slow, count))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [2.0, 2.0]        # both ranks see the slowest rank's time
    assert sum(r[3] for r in res) == 13
    assert shard.job_throughput(1920 * 1080 * 256, 13, 2.0) == pytest.approx(1920 * 1080 * 256 * 13 / 2.0 / 1e6)


def test_row_bands_tile_the_image():
    for rows in (1, 7, 1080, 4320):
        for world in (1, 2, 3, 8):
            edges = [shard.row_band(rows, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == rows
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


def _band_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows, cols = 11, 5                       # odd height: the bands differ by one row
        full = torch.arange(rows * cols, dtype=torch.float32).reshape(rows, cols)
        b, e = shard.row_band(rows, rank, world)
        got = shard.gather_row_bands(full[b:e].clone(), rows, dist)
        q.put((rank, bool(torch.equal(got, full)), (b, e)))
    finally:
        dist.destroy_process_group()


def test_row_band_gather_two_ranks_gloo():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_band_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert res[0][2] == (0, 6) and res[1][2] == (6, 11)


def _stats_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 holds an empty band's sentinels in slice 1; rank 1 the larger steer maximum
        big = np.finfo(np.float32).max
        mine = np.array([[1.0, 5.0], [big, -big], [0.0, 3.0]], np.float32) if rank == 0 else \
            np.array([[2.0, 7.0], [4.0, 6.0], [1.0, 2.0]], np.float32)
        mm, sm = shard.merge_band_stats(mine, 10.5 + rank, dist)
        q.put((rank, mm.tolist(), sm))
    finally:
        dist.destroy_process_group()


def test_band_statistics_merge_two_ranks_gloo():
    # the only collective of the row-band sharding: min of minima, max of maxima (per slice) and of the steer maximum
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stats_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert r[1] == [[1.0, 7.0], [4.0, 6.0], [0.0, 3.0]]
        assert r[2] == 11.5
