#!/usr/bin/env python3
"""One stereo pair with its rows spread over the ranks of a torch.distributed job (one process per GPU; with fewer GPUs
than ranks the ranks share devices, which checks the protocol but cannot be faster).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
        scripts/row_shard_pair.py --rows 1080 --cols 1920 --disp 256 [--backend nccl]

Every rank runs its two sweeps (mdisp_sgbm_rows_sweep_dev) with the boundary states exchanged point to point; the row
ranges are gathered on every rank and rank 0 compares the assembled map with one mdisp_sgbm_dev call on the whole pair.
Prints one JSON line on rank 0; the exit code is 0 only when the maps are equal."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--disp", type=int, default=256)
    ap.add_argument("--P2", type=int, default=32)
    ap.add_argument("--backend", default="gloo", help="gloo: boundary states through host memory (ranks may share a GPU); nccl: RCCL, one GPU per rank")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cameracalibration_amd import _lib, api, shard

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(args.backend)
    lib = _lib.load()
    ctx = _lib.Context(device=local)
    H, W, D = args.rows, args.cols, args.disp
    params = api.sgbm_params(0, D, 8, args.P2, 10, 1, 8, 1)
    L, R, _ = api.synth_pair(0x5EED0001, H, W, D)
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    disp = torch.full((H, W), -32768, dtype=torch.int16, device="cuda")
    part = shard.RowShardedSGBM(ctx, H, W, params, rank, world)
    exchange = shard.TorchBoundaryExchange(dist, through_host=args.backend != "nccl")
    times = []
    for _ in range(args.reps):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        part.compute(dl, dr, W, disp, exchange)
        dist.barrier()
        times.append(time.perf_counter() - t0)
    # assemble: every rank contributes its rows
    ranges = shard.sgm_row_ranges(H, part.band_rows, world)
    tall = max(e - b for b, e in ranges)
    mine = torch.zeros((tall, W), dtype=torch.int16)
    mine[:part.end - part.begin] = disp[part.begin:part.end].cpu()
    mine = mine.view(torch.uint8)           # gloo has no 16-bit integer collectives: gather the bytes
    parts = [torch.empty_like(mine) for _ in range(world)]
    if args.backend == "nccl":
        gpu_parts = [p.cuda() for p in parts]
        dist.all_gather(gpu_parts, mine.cuda())
        parts = [p.cpu() for p in gpu_parts]
    else:
        dist.all_gather(parts, mine)
    ok = True
    if rank == 0:
        full = torch.cat([parts[r].view(torch.int16)[:e - b] for r, (b, e) in enumerate(ranges)], dim=0)
        ref = torch.empty((H, W), dtype=torch.int16, device="cuda")
        ws_bytes = lib.mdisp_sgbm_workspace_bytes(H, W, C.byref(params))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
        single = []
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _lib.check(lib.mdisp_sgbm_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), H, W, W, C.byref(params), ref.data_ptr(), ws.data_ptr(), ws_bytes))
            ctx.synchronize()
            single.append(time.perf_counter() - t0)
        ok = bool(torch.equal(full, ref.cpu()))
        print(json.dumps({"equal": ok, "world": world, "rows": H, "cols": W, "disp": D, "row_ranges": ranges, "backend": args.backend,
                          "ms_row_sharded": min(times) * 1e3, "ms_one_call": min(single) * 1e3,
                          "differing_pixels": int((full != ref.cpu()).sum())}), flush=True)
    flag = torch.tensor([1 if ok else 0])
    if args.backend == "nccl":
        flag = flag.cuda()
    dist.broadcast(flag, 0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) else 1)


if __name__ == "__main__":
    main()
