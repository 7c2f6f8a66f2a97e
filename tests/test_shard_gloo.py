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
        ok = True
        for dtype in (torch.float32, torch.int16, torch.float64):        # int16: disparity maps (no 16-bit integer collectives in gloo)
            full = (torch.arange(rows * cols, dtype=torch.float32) - 20).to(dtype).reshape(rows, cols)
            b, e = shard.row_band(rows, rank, world)
            got = shard.gather_row_bands(full[b:e].clone(), rows, dist)
            ok = ok and got.dtype == dtype and bool(torch.equal(got, full))
        q.put((rank, ok, (b, e)))
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


# ---- one pair over several ranks: the sweep plan and the boundary exchange -------------------------------------------

def test_sgm_row_ranges_are_whole_bands_that_tile_the_image():
    for rows in (16, 40, 1080, 1081):
        for band in (8, 16):
            for world in (1, 2, 3, 8):
                edges = shard.sgm_row_ranges(rows, band, world)
                assert edges[0][0] == 0 and edges[-1][1] == rows
                assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
                assert all(b == e or (b % band == 0 and (e % band == 0 or e == rows)) for b, e in edges)      # ranks beyond the last band: empty


def test_more_participants_than_bands_is_an_error_on_every_rank():
    """An empty row range is refused from the ranges alone (identical on every rank), so all ranks raise together and
    nobody is left waiting for a neighbour's boundary states."""
    ranges = shard.sgm_row_ranges(40, 16, 8)          # 3 bands, 8 participants
    assert any(b >= e for b, e in ranges)
    with pytest.raises(ValueError):
        shard.check_row_ranges(ranges)
    shard.check_row_ranges(shard.sgm_row_ranges(40, 16, 3))


def test_sgm_sweep_plan_dependencies():
    for world in (1, 2, 3, 4, 8):
        plans = [shard.sgm_sweep_plan(r, world) for r in range(world)]
        for r, plan in enumerate(plans):
            assert sorted(p[0] for p in plan) == [0, 1] and [p[1] for p in plan] == [0, 1]
            for direction, second, src, dst in plan:
                assert src == (None if (r == 0 if direction == 0 else r == world - 1) else (r - 1 if direction == 0 else r + 1))
                assert dst == (None if (r == world - 1 if direction == 0 else r == 0) else (r + 1 if direction == 0 else r - 1))
        order = shard.sgm_sequential_order(world)
        assert sorted(order) == sorted((r, s) for r in range(world) for s in (0, 1))
        done = set()
        for r, step in order:
            direction, second, src, dst = plans[r][step]
            assert src is None or (src, direction) in done
            assert step == 0 or (r, plans[r][0][0]) in done
            done.add((r, direction))
        # both wave-fronts advance together: the ranks' first sweeps are all ordered before the last rank's second
        if world >= 2:
            assert order[0] == (0, 0) and order[1] == (world - 1, 0)


def _mock_costs(rows, cols):
    import numpy as np
    rng = np.random.default_rng(7)
    return rng.integers(0, 50, size=(rows, cols)).astype(np.int64)


def _mock_sweep(C, begin, end, direction, start):
    """A two-direction recursion with the shape of a vertical SGM path: L[y] = C[y] + min(L[y -+ 1], 40)."""
    import numpy as np
    ys = range(begin, end) if direction == 0 else range(end - 1, begin - 1, -1)
    prev = start
    out = np.zeros((end - begin, C.shape[1]), np.int64)
    for y in ys:
        prev = C[y] + (np.minimum(prev, 40) if prev is not None else 0)
        out[y - begin] = prev
    return out, prev


def _sweep_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows, cols, band = 44, 9, 8
        C = _mock_costs(rows, cols)
        begin, end = shard.sgm_row_ranges(rows, band, world)[rank]
        total = torch.zeros((end - begin, cols), dtype=torch.int64)
        plan = shard.sgm_sweep_plan(rank, world)

        def sweep(step, bnd):
            direction = plan[step][0]
            out, last = _mock_sweep(C, begin, end, direction, bnd.numpy().copy() if bnd is not None else None)
            total.add_(torch.from_numpy(out))
            return torch.from_numpy(last.copy())

        shard.run_sweep_plan(rank, world, sweep, shard.TorchBoundaryExchange(dist), torch.zeros(cols, dtype=torch.int64))
        q.put((rank, begin, end, total.numpy().tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sgm_boundary_exchange_reproduces_the_single_rank_result_gloo(world):
    import numpy as np
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sweep_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    C = _mock_costs(44, 9)
    want = _mock_sweep(C, 0, 44, 0, None)[0] + _mock_sweep(C, 0, 44, 1, None)[0]
    got = np.concatenate([np.array(r[3], np.int64).reshape(-1, 9) for r in res], axis=0)
    assert res[0][1] == 0 and res[-1][2] == 44
    assert np.array_equal(got, want)
