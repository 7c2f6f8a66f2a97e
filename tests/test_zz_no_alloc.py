"""The entry point that allocates nothing (mdisp_census_stereo_dev_noalloc): through the C ABI, against the existing
mdisp_census_stereo_dev on the same device buffers, bit for bit, on two calls in a row of which the second is smaller
(the shared scratch array keeps the first call's contents), plus the oracle and the size limit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(lib, _lib, ctx, fn, L, R, win, minD, maxD, left, method, agg):
    import torch
    H, W = L.shape
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.full((H, W), -7.0, dtype=torch.float64, device="cuda")
    par = _lib.mdisp_param(win, minD, maxD, 1 if left else 0)
    _lib.check(fn(ctx.handle, dl.data_ptr(), dr.data_ptr(), H, W, W, C.byref(par), method, agg, out.data_ptr()))
    ctx.synchronize()
    return out.cpu().numpy()


def test_no_alloc_entry_point_matches_the_allocating_one(api, orc):
    from cameracalibration_amd import _lib
    lib = _lib.load()
    ctx = _lib.Context()
    cases = [  # (H, W, numDisp for the generator, win, minD, maxD, left, census method, aggregation)
        (480, 640, 64, 7, 0, 63, True, api.BASIC_CENSUS, api.AGG_BOX),          # the 640x480, D = 64 configuration
        (96, 160, 32, 5, 2, 33, False, api.UNIFORM_CENSUS, api.AGG_SQRBOX),    # smaller: stale bytes of the first call remain behind it
        (60, 88, 16, 9, 0, 15, True, api.STATISTIC_MULTISCALE_CENSUS, api.AGG_NONE),   # uses the zero-initialised histogram
        (480, 640, 64, 7, 0, 63, True, api.BASIC_CENSUS, api.AGG_BOX),
    ]
    # a context of its own for the no-allocation calls: it must never acquire device memory
    bare = _lib.Context()
    assert lib.mdisp_context_device_bytes(bare.handle) == 0
    for (H, W, nd, win, minD, maxD, left, method, agg) in cases:
        L, R, _ = api.synth_pair(0x5EED0001 + H, H, W, nd)
        got = _run(lib, _lib, bare, lib.mdisp_census_stereo_dev_noalloc, L, R, win, minD, maxD, left, method, agg)
        assert lib.mdisp_context_device_bytes(bare.handle) == 0, "the no-allocation entry point grew the context's workspace"
        ref = _run(lib, _lib, ctx, lib.mdisp_census_stereo_dev, L, R, win, minD, maxD, left, method, agg)
        assert np.array_equal(got, ref)
        if method == api.BASIC_CENSUS:
            assert np.array_equal(got, orc.census_stereo(L, R, win, minD, maxD, left, orc.BASIC_CENSUS, orc.AGG_BOX))
    assert lib.mdisp_context_device_bytes(ctx.handle) > 0      # the allocating entry point did use the workspace
    bare.close()
    ctx.close()


def test_no_alloc_entry_point_rejects_what_does_not_fit(api):
    import torch
    from cameracalibration_amd import _lib
    lib = _lib.load()
    ctx = _lib.Context()
    H, W = 2200, 2000                      # 4 * 2200 * 2304 bytes > 16 MiB
    d = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    out = torch.zeros((H, W), dtype=torch.float64, device="cuda")
    par = _lib.mdisp_param(7, 0, 255, 1)
    rc = lib.mdisp_census_stereo_dev_noalloc(ctx.handle, d.data_ptr(), d.data_ptr(), H, W, W, C.byref(par), 0, 1, out.data_ptr())
    assert rc == _lib.E_RANGE and b"scratch" in lib.mdisp_last_error()
    par = _lib.mdisp_param(7, 0, 15, 1)
    rc = lib.mdisp_census_stereo_dev_noalloc(ctx.handle, d.data_ptr(), d.data_ptr(), 64, 64, W, C.byref(par), 0, 3, out.data_ptr())
    assert rc == _lib.E_UNSUPPORTED
    ctx.close()
