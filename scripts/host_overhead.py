#!/usr/bin/env python3
"""Where does the time of a host-buffer call go?  Times a trivial call (census of a small image)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import api  # noqa: E402

for (H, W) in ((48, 64), (480, 640), (1080, 1920)):
    img = np.random.default_rng(0).integers(0, 256, (H, W), dtype=np.uint8)
    api.countCensusImg(img)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        api.countCensusImg(img)
        ts.append(time.perf_counter() - t0)
    print(f"countCensusImg {W}x{H}: min {min(ts)*1e3:.3f} ms  median {sorted(ts)[5]*1e3:.3f} ms")
L, R, _ = api.synth_pair(1, 480, 640, 64)
p = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=7, minDisparity=0, maxDisparity=63, isDispLeft=True)
for name, fn in (("censusStereo box", lambda: api.censusStereo(p, api.BASIC_CENSUS, api.AGG_BOX)), ("SAD_BF", lambda: api.computeSAD_BFOpti(p))):
    fn()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    print(f"{name}: min {min(ts)*1e3:.3f} ms  median {sorted(ts)[5]*1e3:.3f} ms")
