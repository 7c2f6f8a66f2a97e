#!/usr/bin/env python3
"""Large single pair, row-band sharded: 7680x4320, D=512, steer-weighted guided
aggregation (winSize 35, r2 15, eps 0.01), one process per GPU.

Every rank holds both full images.  Phase 1: each rank computes its band's share of the image-wide statistics
(per-slice cost range, steer maximum) and the shares are merged with one all-reduce of ~1000 floats.  Phase 2: each
rank computes the disparity rows of its band with mdisp_asw_gifs_band_dev; the bands are assembled with one
all_gather (outputs only).  --local-stats skips phase 1 (every rank then recomputes the statistics from the full
images).  --band K/N times one band of N on a single GPU (a projection of the per-GPU time of an N-GPU run).
Launch with torch.distributed.run for N > 1:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/c5_rowband.py
Options: --size WxH, --disp D, --backend gloo (rehearsal with ranks sharing a GPU), --check (compare with a
full-image call on rank 0)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import _lib, api, shard  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="7680x4320")
    ap.add_argument("--disp", type=int, default=512)
    ap.add_argument("--win", type=int, default=35)
    ap.add_argument("--r2", type=int, default=15)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--local-stats", action="store_true")
    ap.add_argument("--band", default="", help="K/N: run band K of N in this single process")
    args = ap.parse_args()
    W, H = (int(v) for v in args.size.split("x"))
    D = args.disp
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    ctx = _lib.Context(device=local, stream=torch.cuda.current_stream().cuda_stream)
    L, R, _ = api.synth_pair(0x5EED0001, H, W, D)
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.zeros((H, W), dtype=torch.float32, device="cuda")
    b, e = shard.row_band(H, rank, world)
    solo = None
    if args.band:
        k, n = (int(v) for v in args.band.split("/"))
        b, e = shard.row_band(H, k, n)
        solo = [shard.row_band(H, i, n) for i in range(n)]
    red = "cuda" if args.backend == "nccl" else "cpu"

    def step():
        stats = None
        if not args.local_stats:
            mm, sm = api.asw_gifs_band_stats_dev(dl.data_ptr(), dr.data_ptr(), H, W, W, args.win, 0, D - 1, True, api.SKWGIF, b, e, ctx=ctx)
            if solo:        # single-process projection: the other bands' shares, computed here but not timed as ours
                stats = step.merged
            else:
                stats = shard.merge_band_stats(mm, sm, dist if world > 1 else None, device=red)
        api.asw_gifs_band_dev(dl.data_ptr(), dr.data_ptr(), H, W, W, args.win, 0, D - 1, True, 0.01, api.SKWGIF, args.r2, 0.1,
                              b, e, out.data_ptr(), ctx=ctx, stats=stats)
        band = out[b:e]
        if solo:
            torch.cuda.synchronize()
            return out
        if args.backend != "nccl" and world > 1:
            return shard.gather_row_bands(band.cpu(), H, dist)
        return shard.gather_row_bands(band, H, dist if world > 1 else None)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if solo and not args.local_stats:
        parts = [api.asw_gifs_band_stats_dev(dl.data_ptr(), dr.data_ptr(), H, W, W, args.win, 0, D - 1, True, api.SKWGIF, bb, ee, ctx=ctx)
                 for (bb, ee) in solo]
        step.merged = (np.stack([np.min([q[0][:, 0] for q in parts], axis=0), np.max([q[0][:, 1] for q in parts], axis=0)], axis=1),
                       max(q[1] for q in parts))
    full = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full = step()
    barrier()
    dt = shard.slowest_rank_seconds((time.perf_counter() - t0) / args.steps, dist if world > 1 else None,
                                    device="cuda" if args.backend == "nccl" else "cpu")
    ok = None
    if args.check and rank == 0:
        ref = torch.zeros((H, W), dtype=torch.float32, device="cuda")
        api.asw_gifs_band_dev(dl.data_ptr(), dr.data_ptr(), H, W, W, args.win, 0, D - 1, True, 0.01, api.SKWGIF, args.r2, 0.1,
                              0, H, ref.data_ptr(), ctx=ctx)
        torch.cuda.synchronize()
        ok = bool(torch.equal(ref.cpu()[b:e], full.cpu()[b:e])) if solo else bool(torch.equal(ref.cpu(), full.cpu()))
    if rank == 0:
        print(json.dumps({"workload": f"{W}x{H} D={D} SKWGIF win {args.win} r2 {args.r2}, row bands x{world}"
                          + (f", band {args.band} only" if solo else "") + (", local statistics" if args.local_stats else ""), "n_gpus": world,
                          "ms_per_pair": dt * 1e3, "Mdisp_per_s": W * H * D / dt / 1e6, "bands_equal_full_image": ok}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
