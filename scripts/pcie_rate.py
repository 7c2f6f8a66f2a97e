#!/usr/bin/env python3
"""Rate of the host-buffer entry point (mdisp_sgbm: H2D copies + kernels + D2H copy) at 1920x1080, D=256,
for comparison with the HBM-resident rate bench.py reports."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cameracalibration_amd import api  # noqa: E402

H, W, D = 1080, 1920, 256
L, R, _ = api.synth_pair(0x5EED0001, H, W, D)
p = api.sgbm_params(0, D, 8, 32, 10, 1, 8, 1)
api.sgbm_compute(L, R, p)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    api.sgbm_compute(L, R, p)
dt = (time.perf_counter() - t0) / n
print(f"host-buffer mdisp_sgbm: {dt*1e3:.2f} ms per pair, {H*W*D/dt/1e6:.0f} Mdisp/s (PCIe copies and sync included)")

# the streaming batch entry point: 64 host pairs, 16 per chunk, copies overlapped with the computation
import numpy as np  # noqa: E402
N = 64
Ls = np.stack([L] * N)
Rs = np.stack([R] * N)
api.sgbm_compute_batch(Ls[:16], Rs[:16], p, 16)
for chunk in (8, 16, 32):
    t0 = time.perf_counter()
    out = api.sgbm_compute_batch(Ls, Rs, p, chunk)
    dt = (time.perf_counter() - t0) / N
    print(f"host-buffer mdisp_sgbm_batch, {N} pairs, chunk {chunk}: {dt*1e3:.2f} ms per pair, {H*W*D/dt/1e6:.0f} Mdisp/s "
          f"(pageable host memory, PCIe copies and sync included)")

# the multi-device entry point with the host images page-locked for the call (one or two workers on this GPU)
for devices, pin in (([0], False), ([0], True), ([0, 0], True)):
    for chunk in (16, 32):
        api.sgbm_compute_batch_multi(devices, Ls[:chunk], Rs[:chunk], p, chunk, pin_host=pin)
        t0 = time.perf_counter()
        out = api.sgbm_compute_batch_multi(devices, Ls, Rs, p, chunk, pin_host=pin)
        dt = (time.perf_counter() - t0) / N
        print(f"mdisp_sgbm_batch_multi, {len(devices)} worker(s), {'pinned' if pin else 'pageable'} host memory, chunk {chunk}: "
              f"{dt*1e3:.2f} ms per pair, {H*W*D/dt/1e6:.0f} Mdisp/s (registration, PCIe copies and sync included)")
