#!/usr/bin/env python3
"""Profiling target: census + guided-filter aggregation + WTA at 1920x1080, D=256 (device-resident)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import _lib, api  # noqa: E402

H, W, D = 1080, 1920, 256
lib = _lib.load()
ctx = _lib.Context(stream=torch.cuda.current_stream().cuda_stream)
L, R, _ = api.synth_pair(0x5EED0001, H, W, D)
dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
out = torch.empty((H, W), dtype=torch.float64, device="cuda")
p = _lib.mdisp_param(7, 0, D - 1, 1)
for _ in range(3):
    _lib.check(lib.mdisp_census_stereo_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), H, W, W, C.byref(p), api.BASIC_CENSUS, api.AGG_GUIDED,
                                           out.data_ptr()))
torch.cuda.synchronize()
