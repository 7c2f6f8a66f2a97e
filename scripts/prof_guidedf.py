#!/usr/bin/env python3
"""Profiling target: stereomatch_1::computeAdaptiveWeight_GuidedF / _2 / _3 at the reference's parameters (1280x720 colour pair,
winSize 35, minDisparity 186, 144 candidates).  usage: prof_guidedf.py [1|2|3]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import api  # noqa: E402

which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, W = 720, 1280
Lg, Rg, _ = api.synth_pair(0x5EED0001, H, W, 256)
gain = np.random.default_rng(7).integers(200, 256, 3)
Lc = np.stack([(Lg.astype(np.int32) * g // 255).astype(np.uint8) for g in gain], axis=-1)
Rc = np.stack([(Rg.astype(np.int32) * g // 255).astype(np.uint8) for g in gain], axis=-1)
sm = api.stereomatch_1
fn = {1: sm.computeAdaptiveWeight_GuidedF, 2: sm.computeAdaptiveWeight_GuidedF_2, 3: sm.computeAdaptiveWeight_GuidedF_3}[which]
fn(Lc, Rc)
t0 = time.perf_counter()
for _ in range(3):
    out = fn(Lc, Rc)
dt = (time.perf_counter() - t0) / 3
print(f"computeAdaptiveWeight_GuidedF{'' if which == 1 else '_%d' % which}: {dt * 1e3:.1f} ms, finite {bool(np.isfinite(out).all())}")
