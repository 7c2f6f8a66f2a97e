#!/usr/bin/env python3
"""Benchmark of the dense-stereo hot path on MI355X.

Workload (the configuration the headline metric is quoted on): synthetic rectified
1920x1080 pairs, D = 256, census -> Hamming cost -> SGM 8-path aggregation -> WTA + uniqueness +
x16 sub-pixel + LR check, inputs resident in HBM, through the C ABI (mdisp_sgbm_dev).
One "step" = one batch of `--pairs` stereo pairs per GPU (default 64: a batch of 512 pairs split over 8 GPUs).  With N GPUs every rank processes its own
batch (pairs are independent: no collective on the data path) -> weak scaling.

Prints ONE JSON line on rank 0."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, D = 1920, 1080, 256
P1, P2 = 8, 32
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak
ALG_BYTES_PER_UNIT = 4.0        # u16 partial-sum volume: written once, read once
ALG_BYTES_PER_PIXEL = 8.0       # census inputs per sweep + int16 disparity


def cpu_baseline(orc, api):
    """Time the CPU oracle (restatement of the reference path, OpenMP over the host cores) on a
    bounded sample of the same workload: whole frames until about 10 s of CPU work are done."""
    L, R, _ = api.synth_pair(0x5EED0001, H, W, D)
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    frames, dt = 0, 0.0
    t0 = time.perf_counter()
    while frames < 1 or (dt < 10.0 and frames < 32):
        orc.sgm(L, R, 0, D, P1, P2, 10, 1, 8, 1)
        frames += 1
        dt = time.perf_counter() - t0
    out = {"value": frames * H * W * D / dt / 1e6, "unit": "Mdisp/s", "cores": threads, "kind": "port",
           "sample": f"{frames} frame(s) of {W}x{H}, D={D}, SGM 8-path + LR + subpixel, {dt:.1f} s of CPU work"}
    # the same routine on one thread, one frame
    os.environ["OMP_NUM_THREADS"] = "1"
    orc.set_threads(1)
    t0 = time.perf_counter()
    orc.sgm(L, R, 0, D, P1, P2, 10, 1, 8, 1)
    out["value_1_thread"] = H * W * D / (time.perf_counter() - t0) / 1e6
    orc.set_threads(threads)
    return out


def measured_traffic():
    """HBM bytes per pair from the committed rocprofv3 PMC passes of this command (FETCH_SIZE and
    WRITE_SIZE need separate runs, so they cannot be collected inside a timed run)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return float(json.load(f)["traffic_bytes_per_pair"]), os.path.relpath(files[-1], ROOT)


def measured_issue_bound():
    """VALU wave-instructions per pair from the committed SQ counter pass (scripts/sq_summary.py): the kernels are
    bound by VALU issue, not by HBM, so this is the ceiling that actually applies."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*sq_issue.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    out = {"valu_wave_instructions_per_pair": d["valu_instructions_per_pair"], "ms_per_pair_at_full_issue": d["issue_bound_ms_per_pair"],
           "source": os.path.relpath(files[-1], ROOT)}
    # where the per-opcode pricing exists (scripts/issue_bound.py: instructions of the main loops by opcode x the issue
    # cost measured for each opcode, at the kernel's own clock, scaled to the counted instructions), it replaces the
    # flat four cycles per instruction
    priced = sorted(glob.glob(os.path.join(ROOT, "profiles", "*sgm_issue_bound.json")))
    if priced:
        with open(priced[-1]) as f:
            b = json.load(f)
        if "ms_per_pair_scaled_to_counted_instructions" in b:
            out["ms_per_pair_flat_4_cycles"] = out["ms_per_pair_at_full_issue"]
            out["ms_per_pair_at_full_issue"] = b["ms_per_pair_scaled_to_counted_instructions"]
            out["pricing"] = "sum over opcodes of count x measured issue cycles at %.2f GHz" % b["kernel_clock_ghz"]
            out["source"] = os.path.relpath(priced[-1], ROOT)
    return out


def side_path(args):
    """Side measurements of the other device-resident paths (one GPU, one pair per call: what their callers do).  The
    headline line is `--path sgm`; these print the same kind of line so that they can be recorded next to it."""
    print(json.dumps(measure_side(args)), flush=True)


def side_summary():
    """The side measurements in short form, for the headline line: ms per call of the cv::StereoSGBM replacement (both modes) at
    the headline frame size and of census + box + WTA at 640x480, D=64 (each a few calls, well under a second)."""
    out = {}
    for name, path, hh in (("stereosgbm_1920x1080_D256_mode_sgbm", "stereosgbm", False), ("stereosgbm_1920x1080_D256_mode_hh", "stereosgbm", True),
                           ("census_box_wta_640x480_D64", "census_box", False)):
        a = argparse.Namespace(path=path, disp=256, paths=8, hh=hh, steps=10, warmup=3, pairs_per_call=1)
        r = measure_side(a)
        out[name] = {"ms_per_call": r["ms_per_step"], "Mdisp_per_s": r["value"], "workload": r["config"]["workload"]}
    a = argparse.Namespace(path="stereosgbm", disp=256, paths=8, hh=False, steps=4, warmup=1, pairs_per_call=8)
    r = measure_side(a)
    out["stereosgbm_1920x1080_D256_mode_sgbm_batch_of_8"] = {"ms_per_pair": r["ms_per_step"] / 8, "Mdisp_per_s": r["value"], "workload": r["config"]["workload"]}
    return out


def measure_side(args):
    import torch
    from cameracalibration_amd import _lib, api
    lib = _lib.load()
    tstream = torch.cuda.Stream()
    ctx = _lib.Context(device=0, stream=tstream.cuda_stream)
    if args.path == "stereosgbm":
        h, w, d = H, W, args.disp
        mode = 1 if args.paths == 8 and args.hh else 0
        prm = api.StereoSGBM_create(0, d, 7, 490, 1960, 1, 63, 10, 100, 32, mode)
        L, R, _ = api.synth_pair(0x5EED0001, h, w, d)
        dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
        out = torch.empty((h, w), dtype=torch.int16, device="cuda")
        nb = lib.mdisp_stereosgbm_workspace_bytes(h, w, 1, C.byref(prm))
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        workload = (f"{w}x{h} D={d} cv::StereoSGBM replacement (Birchfield-Tomasi block 7, "
                    f"{'MODE_HH' if mode else 'MODE_SGBM'}, median + speckle filter), one pair per call")
        alg = 4.0 * w * h * d + 8.0 * w * h

        def step():
            _lib.check(lib.mdisp_stereosgbm_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), h, w, 1, w, C.byref(prm), out.data_ptr(), ws.data_ptr(), nb))
        nbatch = getattr(args, "pairs_per_call", 1)
        if nbatch > 1:
            # mdisp_stereosgbm_batch_dev: the same pair nbatch times (distinct outputs), pairs side by side on the context's streams
            dl, dr = dl.unsqueeze(0).repeat(nbatch, 1, 1).contiguous(), dr.unsqueeze(0).repeat(nbatch, 1, 1).contiguous()
            out = torch.empty((nbatch, h, w), dtype=torch.int16, device="cuda")
            nb = lib.mdisp_stereosgbm_batch_workspace_bytes(h, w, 1, C.byref(prm), nbatch)
            ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            workload = workload.replace("one pair per call", f"{nbatch} pairs per call")
            alg *= nbatch

            def step():
                _lib.check(lib.mdisp_stereosgbm_batch_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), nbatch, h * w, h, w, 1, w, C.byref(prm),
                                                          out.data_ptr(), ws.data_ptr(), nb))
    else:
        h, w, d = 480, 640, 64
        prm = _lib.mdisp_param(7, 0, d - 1, 1)
        L, R, _ = api.synth_pair(0x5EED0001, h, w, d)
        dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
        out = torch.empty((h, w), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        workload = f"{w}x{h} D={d} census + 7x7 box + WTA, one pair per call"
        alg = 2.0 * w * h + 8.0 * w * h          # the views in, the double map out: the cost volume never leaves the CU

        def step():
            _lib.check(lib.mdisp_census_stereo_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), h, w, w, C.byref(prm), api.BASIC_CENSUS, api.AGG_BOX,
                                                   out.data_ptr()))
    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    per = elapsed / args.steps
    return ({
        "metric": f"Mdisp/s (WxHxD per sec) at {w}x{h}, D={d}", "value": w * h * d * getattr(args, "pairs_per_call", 1) / per / 1e6, "unit": "Mdisp/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": per * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "s16" if args.path == "stereosgbm" else "u16", "data": "synthetic",
        "config": {"workload": workload, "path": args.path},
        "roofline": {"bound": "hbm", "achieved": alg / per / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / per / 1e9 / HBM_PEAK_GBS,
                     "traffic": None, "algorithmic_bytes_per_pair": alg}})


def main():
    global D, P2
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=64, help="stereo pairs per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--paths", type=int, default=8, help="8 (the headline configuration) or 5 (StereoSGBM's default mode)")
    ap.add_argument("--disp", type=int, default=256, help="number of disparities (256 is the headline configuration; "
                                                          "other values are side measurements)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="scheduling override for side measurements, e.g. SGM_LANES16=1 or SGM_PIPELINE=1 (mdisp_tuning_set)")
    ap.add_argument("--P2", type=int, default=P2, help="SGM penalty P2 (32 is the headline configuration; values above 247 select "
                                                      "16-bit path states)")
    ap.add_argument("--path", default="sgm", choices=["sgm", "stereosgbm", "census_box"],
                    help="sgm: the headline configuration; stereosgbm / census_box: side measurements of the cv::StereoSGBM "
                         "replacement at 1920x1080 and of census + box + WTA at 640x480 (one GPU, one pair per call)")
    ap.add_argument("--pairs-per-call", type=int, default=1, help="--path stereosgbm: pairs per mdisp_stereosgbm_batch_dev call (1: the single-pair entry point)")
    ap.add_argument("--hh", action="store_true", help="--path stereosgbm: MODE_HH (eight directions) instead of MODE_SGBM")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison of the timed batch with the single-pair call "
                    "(counter passes: the single-pair call's kernels would be counted with the batch's)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only to rehearse the "
                                                      "multi-rank path on a box with fewer GPUs than ranks)")
    args = ap.parse_args()
    D = args.disp
    P2 = args.P2

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU fallback")
    if args.path != "sgm":
        if world > 1:
            raise SystemExit("--path %s is a one-GPU side measurement" % args.path)
        torch.cuda.set_device(0)
        side_path(args)
        return
    if args.backend != "nccl":
        local_rank %= torch.cuda.device_count()        # rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from cameracalibration_amd import _lib, api, shard
    lib = _lib.load()
    for kv in args.tune:
        k, v = kv.split("=")
        _lib.tuning_set(getattr(_lib, "TUNE_" + k), int(v))
    # a stream of torch's own (not the legacy default stream), handed to the library: the inputs below are complete
    # before the first call because the uploads are followed by a device-wide synchronisation
    tstream = torch.cuda.Stream()
    ctx = _lib.Context(device=local_rank, stream=tstream.cuda_stream)
    params = api.sgbm_params(0, D, P1, P2, 10, 1, args.paths, 1)

    # synthetic inputs, resident in HBM before the timed region
    B = args.pairs
    lefts = torch.empty((B, H, W), dtype=torch.uint8)
    rights = torch.empty((B, H, W), dtype=torch.uint8)
    first, count = shard.pair_block(B * world, rank, world)     # weak scaling: B pairs per rank
    assert count == B
    for i in range(B):
        L, R, _ = api.synth_pair(shard.pair_seed(first + i), H, W, D)
        lefts[i] = torch.from_numpy(L)
        rights[i] = torch.from_numpy(R)
    lefts, rights = lefts.cuda(), rights.cuda()
    disp = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
    ws_bytes = lib.mdisp_sgbm_batch_workspace_bytes(H, W, B, C.byref(params))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")

    def step():
        # one call = one batch of B independent pairs (pair i at +i*H*W bytes)
        _lib.check(lib.mdisp_sgbm_batch_dev(ctx.handle, lefts.data_ptr(), rights.data_ptr(), B, H * W, H, W, W,
                                            C.byref(params), disp.data_ptr(), ws.data_ptr(), ws_bytes))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.synchronize()           # also reads the pipeline's sticky error word: a hand-off timeout fails the run
    elapsed = shard.slowest_rank_seconds(elapsed, dist if world > 1 else None,
                                         device="cuda" if args.backend == "nccl" else "cpu")

    # per-pair device time of the hot path, HIP events on the stream the kernels run on
    _lib.check(lib.mdisp_enable_timing(ctx.handle, 1))
    call_ms, agg_ms = [], []
    for i in range(max(3, min(10, args.steps))):
        step()
        ms = C.c_float()
        _lib.check(lib.mdisp_last_kernel_ms(ctx.handle, 0, C.byref(ms)))
        call_ms.append(ms.value)
        _lib.check(lib.mdisp_last_kernel_ms(ctx.handle, 1, C.byref(ms)))
        agg_ms.append(ms.value)
    _lib.check(lib.mdisp_enable_timing(ctx.handle, 0))
    torch.cuda.synchronize()
    pair_s = float(np.mean(call_ms)) / 1e3 / B      # device time per pair, whole pipeline

    # the batch just timed against another schedule of the same library: the first and the last pair through the single-pair
    # call (whose full-size output the test suite pins to the CPU oracle bit for bit)
    verified = None
    if rank == 0 and not args.no_verify:
        single = torch.empty((H, W), dtype=torch.int16, device="cuda")
        same = True
        for i in (0, B - 1):
            _lib.check(lib.mdisp_sgbm_dev(ctx.handle, lefts[i].data_ptr(), rights[i].data_ptr(), H, W, W, C.byref(params), single.data_ptr(), None, 0))
            ctx.synchronize()
            same = same and bool(torch.equal(single, disp[i]))
        verified = "pairs 0 and %d of the timed batch equal the single-pair call bit for bit" % (B - 1) if same else "MISMATCH against the single-pair call"
        if not same:
            raise SystemExit("bench.py: the batch result differs from the single-pair call")

    # what HBM delivers on this device: streaming copy of 2 GiB between two allocations of their own (read + written bytes)
    copy_gbs = C.c_float(0)
    del ws
    csrc = torch.full((2 << 30,), 1, dtype=torch.uint8, device="cuda")
    cdst = torch.full((2 << 30,), 2, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    _lib.check(lib.mdisp_copy_bandwidth_dev(ctx.handle, cdst.data_ptr(), csrc.data_ptr(), 2 << 30, 5, C.byref(copy_gbs)))
    del csrc, cdst

    units_per_pair = W * H * D
    total_units = units_per_pair * args.pairs * args.steps * world
    alg_bytes = ALG_BYTES_PER_UNIT * units_per_pair + ALG_BYTES_PER_PIXEL * W * H
    achieved = alg_bytes / pair_s / 1e9

    if rank == 0:
        traffic, traffic_src = measured_traffic() if D == 256 and args.paths == 8 else (None, None)
        out = {
            "metric": f"Mdisp/s (WxHxD per sec) at 1920x1080, D={D}",
            "value": shard.job_throughput(units_per_pair, args.pairs * args.steps * world, elapsed),
            "unit": "Mdisp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic",
            "config": {"workload": f"1920x1080 D={D} census-Hamming cost + SGM {args.paths}-path + WTA/uniqueness + x16 subpixel + LR check",
                       "pairs_per_gpu_per_step": args.pairs, "P1": P1, "P2": P2, "parallelism": f"pair-sharded x{world}"},
            "verified": verified,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "peak_measured_stream_copy": copy_gbs.value or None,
                         "frac_of_measured": achieved / copy_gbs.value if copy_gbs.value else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "all launches of one batch call, per pair (census; ONE row-band launch with both sweep directions started from "
                                   "the two image edges, first halves then second halves + WTA; LR/sub-pixel)" if args.paths == 8 else "all launches of one batch call, per pair",
                         "ms_per_pair": pair_s * 1e3, "ms_aggregation_per_pair": float(np.mean(agg_ms)) / B,
                         "algorithmic_bytes_per_pair": alg_bytes},
        }
        issue = measured_issue_bound()
        if issue and args.paths == 8 and D == 256:
            issue["frac"] = issue["ms_per_pair_at_full_issue"] / (pair_s * 1e3)
            out["roofline"]["valu_issue_bound"] = issue
        if world == 1 and not args.no_cpu_baseline:
            from oracle import orc
            orc.build()
            out["cpu_baseline"] = cpu_baseline(orc, api)
        if world == 1 and args.paths == 8 and D == 256 and not args.no_cpu_baseline:
            out["side_measurements"] = side_summary()      # not part of the timed region: one pair per call, device-resident
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
