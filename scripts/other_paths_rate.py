#!/usr/bin/env python3
"""Throughput of the non-SGM paths (for the design notes; bench.py reports the SGM configuration only).
census + box + WTA (640x480, D=64, device-resident) and the guided-filter aggregations through the host API."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import _lib, api  # noqa: E402

lib = _lib.load()
ctx = _lib.Context(stream=torch.cuda.current_stream().cuda_stream)


def dev_rate(H, W, D, win, agg, n=50):
    L, R, _ = api.synth_pair(0x5EED0001, H, W, D)
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((H, W), dtype=torch.float64, device="cuda")
    p = _lib.mdisp_param(win, 0, D - 1, 1)

    def call():
        _lib.check(lib.mdisp_census_stereo_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), H, W, W, C.byref(p), api.BASIC_CENSUS, agg,
                                               out.data_ptr()))
    call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return dt, H * W * D / dt / 1e6


def host_rate(fn, H, W, D, n=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n
    return dt, H * W * D / dt / 1e6


for (H, W, D, win) in ((480, 640, 64, 7), (1080, 1920, 256, 7)):
    dt, r = dev_rate(H, W, D, win, api.AGG_BOX)
    print(f"census+box{win}+WTA  {W}x{H} D={D} device-resident: {dt*1e3:.3f} ms  {r:.0f} Mdisp/s")
    dt, r = dev_rate(H, W, D, win, api.AGG_GUIDED, n=5)
    print(f"census+guided{win}+WTA {W}x{H} D={D} device-resident: {dt*1e3:.3f} ms  {r:.0f} Mdisp/s")
H, W, D = 480, 640, 64
L, R, _ = api.synth_pair(0x5EED0001, H, W, D)
p = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=7, minDisparity=0, maxDisparity=D - 1, isDispLeft=True)
for name, gif in (("GIF", api.GIF), ("EGIF", api.EGIF), ("SKWGIF", api.SKWGIF), ("OURS_GIF", api.OURS_GIF), ("OURS_GIF3", api.OURS_GIF3)):
    dt, r = host_rate(lambda: api.asw_gifs(p, 0.01, gif, 15, 0.1, 0.5, api.COST_CENSUS), H, W, D)
    print(f"asw_gifs {name:7s} {W}x{H} D={D} host API (copies included): {dt*1e3:.2f} ms  {r:.0f} Mdisp/s")
dt, r = host_rate(lambda: api.computeSAD_BFOpti(p), H, W, D)
print(f"computeSAD_BFOpti  {W}x{H} D={D} host API: {dt*1e3:.2f} ms  {r:.0f} Mdisp/s")
dt, r = host_rate(lambda: api.computeSAD_inteOpti(p), H, W, D)
print(f"computeSAD_inteOpti {W}x{H} D={D} host API: {dt*1e3:.2f} ms  {r:.0f} Mdisp/s")
# the steer-weighted aggregation of the large-pair configuration (winSize 35, r2 15), at 1080p with 64 disparities
H, W, D = 1080, 1920, 64
L, R, _ = api.synth_pair(0x5EED0001, H, W, D)
p = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=35, minDisparity=0, maxDisparity=D - 1, isDispLeft=True)
for name, gif in (("GIF", api.GIF), ("SKWGIF", api.SKWGIF)):
    dt, r = host_rate(lambda: api.asw_gifs(p, 0.01, gif, 15, 0.1, 0.5, api.COST_CENSUS), H, W, D, n=2)
    print(f"asw_gifs {name:7s} win 35 r2 15 {W}x{H} D={D} host API: {dt*1e3:.2f} ms  {r:.0f} Mdisp/s")

# cv::StereoSGBM replacement (Birchfield-Tomasi block cost), device-resident
import ctypes as C
from cameracalibration_amd import _lib
lib = _lib.load()
for (H, W, args) in ((480, 640, (0, 64, 7, 490, 1960, 1, 63, 10, 100, 32, 0)), (1080, 1920, (0, 256, 7, 490, 1960, 1, 63, 10, 100, 32, 0)),
                     (1080, 1920, (0, 256, 7, 490, 1960, 1, 63, 10, 100, 32, 1))):
    L, R, _ = api.synth_pair(0x5EED0001, H, W, args[1])
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((H, W), dtype=torch.int16, device="cuda")
    p = api.StereoSGBM_create(*args)
    ctx = _lib.Context(stream=torch.cuda.current_stream().cuda_stream)
    nb = lib.mdisp_stereosgbm_workspace_bytes(H, W, 1, C.byref(p))
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    def run():
        _lib.check(lib.mdisp_stereosgbm_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), H, W, 1, W, C.byref(p), out.data_ptr(), ws.data_ptr(), nb))
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"StereoSGBM (BT, block 7, mode {args[-1]}) {W}x{H} D={args[1]} device-resident: {dt*1e3:.2f} ms  {H*W*args[1]/dt/1e6:.0f} Mdisp/s")

# cv::StereoBM replacement, device-resident: the reference's getDisparity_BM parameters (144 candidates from 186, 35x35
# windows) at its 1280x720 working size, and a VGA / 64 / 21x21 case
for (H, W, nd, bs, mind) in ((720, 1280, 144, 35, 186), (480, 640, 64, 21, 0), (1080, 1920, 256, 21, 0)):
    L, R, _ = api.synth_pair(0x5EED0001, H, W, nd)
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((H, W), dtype=torch.int16, device="cuda")
    p = _lib.mdisp_stereobm_params(mind, nd, bs, 31, 10, 25, 0, 32, -1)
    nb = lib.mdisp_stereobm_workspace_bytes(H, W, C.byref(p))
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    def run():
        _lib.check(lib.mdisp_stereobm_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), H, W, W, C.byref(p), out.data_ptr(), ws.data_ptr(), nb))
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"StereoBM (block {bs}, minD {mind}) {W}x{H} D={nd} device-resident: {dt*1e3:.2f} ms  {H*W*nd/dt/1e6:.0f} Mdisp/s")

# postProcess_: the WLS post-filter (fast global smoother) on one float map at main()'s working size, host API
H, W = 720, 1280
rng = np.random.default_rng(1)
guide = api.synth_pair(0x5EED0001, H, W, 64)[0]
dmap = (rng.random((H, W)) * 64).astype(np.float32)
dt, _ = host_rate(lambda: api.fastGlobalSmootherFilter(guide, dmap, 8000.0, 0.5), H, W, 1, n=5)
print(f"fastGlobalSmootherFilter (lambda 8000, sigma 0.5, 3 iterations) {W}x{H} host API: {dt*1e3:.2f} ms")

# the guided-filter matchers of stereomatch_1 at the reference's own parameters (methods/method_StereoMatching.h:179-187:
# winSize 35, minDisparity 186, numDisparity 144) on a 1280x720 colour pair, computeSimilarity, and the float-table SAD
H, W, D = 720, 1280, 144
Lg, Rg, _ = api.synth_pair(0x5EED0001, H, W, 256)
rng = np.random.default_rng(7)
gain = rng.integers(200, 256, 3)
Lc = np.stack([(Lg.astype(np.int32) * g // 255).astype(np.uint8) for g in gain], axis=-1)
Rc = np.stack([(Rg.astype(np.int32) * g // 255).astype(np.uint8) for g in gain], axis=-1)
sm = api.stereomatch_1
for name, fn in (("computeAdaptiveWeight_GuidedF", lambda: sm.computeAdaptiveWeight_GuidedF(Lc, Rc)),
                 ("computeAdaptiveWeight_GuidedF_2", lambda: sm.computeAdaptiveWeight_GuidedF_2(Lc, Rc)),
                 ("computeAdaptiveWeight_GuidedF_3", lambda: sm.computeAdaptiveWeight_GuidedF_3(Lc, Rc))):
    dt, r = host_rate(fn, H, W, D, n=2)
    print(f"{name} (win 35, minD 186, 144 candidates) {W}x{H} colour, host API: {dt*1e3:.1f} ms  {r:.0f} Mdisp/s")
costs = []
dt, r = host_rate(lambda: sm.computeSimilarity(Lc, Rc, costs, 0.1, 7.0, 2.0, api.DISPARITY_LEFT, None, 0, 30), H, W, 30, n=2)
print(f"computeSimilarity (30 candidates) {W}x{H} colour, host API (30 float slices copied out): {dt*1e3:.1f} ms  {r:.0f} Mdisp/s")
p = api.StereoMatchParam(imgLeft=Lg, imgRight=Rg, winSize=7, minDisparity=0, maxDisparity=255, isDispLeft=True)
for name, mode in (("float tables (the reference's CV_32F integral)", api.SAT_F32), ("integer tables", api.SAT_INT)):
    dt, r = host_rate(lambda: api.computeSAD_inteOpti(p, None, mode), H, W, 256)
    print(f"computeSAD_inteOpti, {name}, win 7 {W}x{H} D=256 host API: {dt*1e3:.2f} ms  {r:.0f} Mdisp/s")
