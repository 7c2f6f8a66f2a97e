"""One pair with its rows spread over several participants (mdisp_sgbm_rows_sweep_dev): the assembled map must equal the
map of one mdisp_sgbm_dev call bit for bit, whatever the number of participants and the order the plan runs them in."""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _single(lib, _lib, ctx, torch, dl, dr, H, W, params):
    ref = torch.empty((H, W), dtype=torch.int16, device="cuda")
    n = lib.mdisp_sgbm_workspace_bytes(H, W, C.byref(params))
    ws = torch.empty(n, dtype=torch.uint8, device="cuda")
    _lib.check(lib.mdisp_sgbm_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), H, W, W, C.byref(params), ref.data_ptr(), ws.data_ptr(), n))
    ctx.synchronize()
    return ref.cpu().numpy()


CASES = [
    # H, W, D, minD, P2, world
    (100, 150, 64, 0, 32, 2),       # fewer disparities than slots: the masked tail
    (67, 200, 256, 0, 32, 2),       # height not a multiple of the band: the mirrored sweep pads its first band
    (67, 200, 256, 3, 100, 3),      # three participants: the middle one receives in both sweeps; two-byte four-path sums
    (90, 130, 200, 0, 32, 4),
    (50, 120, 512, 0, 32, 2),       # 32 disparities per lane, 8 rows per band
    (16, 64, 128, 0, 32, 1),        # one participant: both image edges
]


@pytest.mark.parametrize("H,W,D,minD,P2,world", CASES)
def test_row_sharded_pair_equals_the_single_call(H, W, D, minD, P2, world):
    _run_case(H, W, D, minD, P2, world)


def test_row_sharded_pair_with_the_smallest_hand_off_ring():
    """Hand-off entries (and the entry a boundary is delivered into) are reused every second band."""
    from cameracalibration_amd import _lib
    _lib.tuning_reset()
    _lib.tuning_set(_lib.TUNE_SGM_HAND_RING, 2)
    try:
        _run_case(150, 130, 200, 0, 32, 3)
    finally:
        _lib.tuning_reset()


def _run_case(H, W, D, minD, P2, world):
    import torch
    from cameracalibration_amd import _lib, api, shard
    lib = _lib.load()
    ctx = _lib.Context()
    params = api.sgbm_params(minD, D, 8, P2, 10, 1, 8, 1)
    L, R, _ = api.synth_pair(0xABC0 + H + D, H, W, min(D, W // 2))
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    want = _single(lib, _lib, ctx, torch, dl, dr, H, W, params)
    # every participant with its own workspace and output map, run one after the other in a dependency order
    parts = [shard.RowShardedSGBM(ctx, H, W, params, r, world) for r in range(world)]
    maps = [torch.full((H, W), -32768, dtype=torch.int16, device="cuda") for _ in range(world)]
    mail = {}
    torch.cuda.synchronize()              # the fills above ran on torch's stream, the sweeps run on the context's
    for r, step in shard.sgm_sequential_order(world):
        direction, second, src, dst = shard.sgm_sweep_plan(r, world)[step]
        bnd = mail.pop((src, r, direction)) if src is not None else None
        out = parts[r].sweep(step, dl, dr, W, maps[r], bnd)
        if dst is not None:
            mail[(r, dst, direction)] = out          # consumed by a later sweep on the same stream
    ctx.synchronize()
    assert not mail
    got = np.full((H, W), -32768, np.int16)
    for r, p in enumerate(parts):
        m = maps[r].cpu().numpy()
        got[p.begin:p.end] = m[p.begin:p.end]
        untouched = np.ones(H, bool)
        untouched[p.begin:p.end] = False
        assert (m[untouched] == -32768).all()          # a participant writes its own rows only
    assert np.array_equal(got, want), f"{(got != want).sum()} pixels differ"


def test_row_sharded_full_frame_two_participants():
    import torch
    from cameracalibration_amd import _lib, api, shard
    lib = _lib.load()
    ctx = _lib.Context()
    H, W, D = 1080, 1920, 256
    params = api.sgbm_params(0, D, 8, 32, 10, 1, 8, 1)
    L, R, _ = api.synth_pair(0x5EED0001, H, W, D)
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    want = _single(lib, _lib, ctx, torch, dl, dr, H, W, params)
    parts = [shard.RowShardedSGBM(ctx, H, W, params, r, 2) for r in range(2)]
    disp = torch.full((H, W), -32768, dtype=torch.int16, device="cuda")
    mail = {}
    torch.cuda.synchronize()
    for r, step in shard.sgm_sequential_order(2):
        direction, second, src, dst = shard.sgm_sweep_plan(r, 2)[step]
        out = parts[r].sweep(step, dl, dr, W, disp, mail.pop((src, r, direction)) if src is not None else None)
        if dst is not None:
            mail[(r, dst, direction)] = out          # consumed by a later sweep on the same stream
    ctx.synchronize()
    assert np.array_equal(disp.cpu().numpy(), want)


def test_row_sharded_rejects_bad_ranges_and_missing_boundaries():
    import torch
    from cameracalibration_amd import _lib, api
    lib = _lib.load()
    ctx = _lib.Context()
    H, W, D = 64, 96, 64
    params = api.sgbm_params(0, D, 8, 32, 10, 1, 8, 1)
    band, bnd, wsb = C.c_int(), C.c_size_t(), C.c_size_t()
    _lib.check(lib.mdisp_sgbm_rows_geometry(H, W, C.byref(params), C.byref(band), C.byref(bnd), C.byref(wsb)))
    assert band.value == 16 and bnd.value == W * 3 * 256
    img = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    disp = torch.zeros((H, W), dtype=torch.int16, device="cuda")
    ws = torch.empty(wsb.value, dtype=torch.uint8, device="cuda")

    def call(r0, r1, direction, second, bin_=None, ws_bytes=wsb.value):
        return lib.mdisp_sgbm_rows_sweep_dev(ctx.handle, img.data_ptr(), img.data_ptr(), H, W, W, C.byref(params), r0, r1, direction, second,
                                             bin_, None, disp.data_ptr(), ws.data_ptr(), ws_bytes)
    assert call(8, 64, 0, 0) == _lib.E_BADARG                 # not on a band edge
    assert call(0, 40, 0, 0) == _lib.E_BADARG
    assert call(16, 64, 0, 0) == _lib.E_BADARG                # a down sweep that starts inside the image needs the states of row 15
    assert call(0, 32, 1, 0) == _lib.E_BADARG                 # and an up sweep that ends inside those of row 32
    assert call(0, 64, 0, 0, ws_bytes=1024) == _lib.E_BADARG
    assert call(0, 64, 2, 0) == _lib.E_BADARG
    wide = api.sgbm_params(0, D, 8, 300, 10, 1, 8, 1)          # 16-bit path states: not built for the split
    assert lib.mdisp_sgbm_rows_geometry(H, W, C.byref(wide), None, None, None) == _lib.E_UNSUPPORTED
    ctx.synchronize()


def test_two_processes_exchange_boundaries_over_gloo():
    """Two ranks (sharing this box's GPU) with the boundary states exchanged over torch.distributed (gloo, through host memory)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "scripts", "row_shard_pair.py"), "--rows", "270", "--cols", "480", "--disp", "128",
           "--backend", "gloo", "--reps", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["equal"] and out["world"] == 2 and out["row_ranges"] == [[0, 144], [144, 270]]
