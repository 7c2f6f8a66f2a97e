#!/usr/bin/env python3
"""One 7680x4320 pair, 512 disparities, eight-path SGM through mdisp_sgbm_dev against the CPU oracle (about 70 GB of host
memory and a minute of CPU time for the oracle: a one-off check, not part of the test suite)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import _lib, api  # noqa: E402
from oracle import orc  # noqa: E402

H, W, D = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4320, 7680, 512)
lib = _lib.load()
ctx = _lib.Context()
prm = api.sgbm_params(0, D, 8, 32, 10, 1, 8, 1)
L, R, _ = api.synth_pair(0x5EED0001, H, W, D)
dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
out = torch.empty((H, W), dtype=torch.int16, device="cuda")
n = lib.mdisp_sgbm_workspace_bytes(H, W, C.byref(prm))
ws = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter()
    _lib.check(lib.mdisp_sgbm_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), H, W, W, C.byref(prm), out.data_ptr(), ws.data_ptr(), n))
    ctx.synchronize()
    dt = time.perf_counter() - t0
print(f"GPU: {dt*1e3:.1f} ms = {H*W*D/dt/1e6:.0f} Mdisp/s, workspace {n/2**30:.1f} GiB", flush=True)
orc.build()
t0 = time.perf_counter()
exp = orc.sgm(L, R, 0, D, 8, 32, 10, 1, 8, 1)
print(f"oracle: {time.perf_counter()-t0:.1f} s", flush=True)
got = out.cpu().numpy()
print("equal:", bool(np.array_equal(got, exp)), "differing pixels:", int((got != exp).sum()))
sys.exit(0 if np.array_equal(got, exp) else 1)
