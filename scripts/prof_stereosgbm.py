"""rocprofv3 target: a few cv::StereoSGBM-replacement calls at 640x480, D=64 (the reference's call-site shape), or with
the argument `1080p` at 1920x1080, D=256."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import _lib, api  # noqa: E402

lib = _lib.load()
H, W = 480, 640
args = (0, 64, 7, 490, 1960, 1, 63, 10, 100, 32, 0)
if len(sys.argv) > 1 and sys.argv[1] == "1080p":          # the benchmark's frame size and range
    H, W = 1080, 1920
    args = (0, 256, 7, 490, 1960, 1, 63, 10, 100, 32, 0)
L, R, _ = api.synth_pair(0x5EED0001, H, W, args[1])
dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
out = torch.empty((H, W), dtype=torch.int16, device="cuda")
p = api.StereoSGBM_create(*args)
ctx = _lib.Context(stream=torch.cuda.current_stream().cuda_stream)
nb = lib.mdisp_stereosgbm_workspace_bytes(H, W, 1, C.byref(p))
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
for _ in range(5):
    _lib.check(lib.mdisp_stereosgbm_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), H, W, 1, W, C.byref(p), out.data_ptr(), ws.data_ptr(), nb))
torch.cuda.synchronize()
# the context owns its stream when torch's current stream is the default one: time with the host clock around a synchronise
import time
_lib.check(lib.mdisp_synchronize(ctx.handle))
t0 = time.perf_counter()
for _ in range(10):
    _lib.check(lib.mdisp_stereosgbm_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), H, W, 1, W, C.byref(p), out.data_ptr(), ws.data_ptr(), nb))
_lib.check(lib.mdisp_synchronize(ctx.handle))
print("%dx%d D=%d: %.3f ms per call" % (W, H, args[1], (time.perf_counter() - t0) * 100))
