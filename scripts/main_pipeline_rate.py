#!/usr/bin/env python3
"""The sequence the reference's main() runs (cameraCalibration.cpp:163-208) at its own parameters: 1280x720 views,
asw_gifs(minDisparity 50, maxDisparity 800, winSize 35, eps 0.01, OURS_GIF3, r2 15, 0.01, 0.5) for the left and the right
map, then postProcess_ on both.  The matching cost is the census-Hamming cost (the dense-SIFT cost of the reference is
not built); host API, copies included."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import api  # noqa: E402

H, W = 720, 1280
L, R, _ = api.synth_pair(0x5EED0001, H, W, 256)


def run():
    pl = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=35, minDisparity=50, maxDisparity=800, isDispLeft=True)
    pr = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=35, minDisparity=50, maxDisparity=800, isDispLeft=False)
    t0 = time.perf_counter()
    dl = api.asw_gifs(pl, 0.01, api.OURS_GIF3, 15, 0.01, 0.5, api.COST_CENSUS)
    dr = api.asw_gifs(pr, 0.01, api.OURS_GIF3, 15, 0.01, 0.5, api.COST_CENSUS)
    t1 = time.perf_counter()
    fl, fr = api.postProcess_(pl, dl, dr)
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, fl, fr


run()
a, b, fl, fr = run()
D = 751
print(f"asw_gifs OURS_GIF3 x2 (1280x720, D={D}, win 35, r2 15): {a*1e3:.1f} ms = {2*H*W*D/a/1e6:.0f} Mdisp/s; postProcess_: {b*1e3:.1f} ms; "
      f"outputs {fl.dtype} {fl.shape}, finite: {bool(np.isfinite(fl).all() and np.isfinite(fr).all())}")
