"""Time of the WLS post-filter (fast global smoother, lambda 8000, sigma 0.5, 3 iterations) on one 1280x720 float map, host API."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import api  # noqa: E402

H, W = 720, 1280
L, R, gt = api.synth_pair(0x5EED0001, H, W, 64)
disp = gt.astype(np.float32)
api.fastGlobalSmootherFilter(L, disp, 8000.0, 0.5)
t0 = time.perf_counter()
for _ in range(10):
    api.fastGlobalSmootherFilter(L, disp, 8000.0, 0.5)
print("%dx%d: %.2f ms per call" % (W, H, (time.perf_counter() - t0) * 100))
