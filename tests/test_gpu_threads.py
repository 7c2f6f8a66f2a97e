"""Two host threads, one context each, running the left and the right disparity at the same time, as the
reference's main() does with std::async (cameraCalibration.cpp:163-190).  Results must equal the sequential ones."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_left_and_right_routines_run_concurrently(api):
    from cameracalibration_amd import _lib
    H, W, D = 120, 200, 32
    L, R, _ = api.synth_pair(0x5EED0031, H, W, D)
    pL = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=7, minDisparity=0, maxDisparity=D - 1, isDispLeft=True)
    pR = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=7, minDisparity=0, maxDisparity=D - 1, isDispLeft=False)
    sp = api.sgbm_params(0, D, 8, 32, 10, 1, 8, 1)

    def work(ctx, p):
        return (api.asw_gifs(p, 0.01, api.SKWGIF, 15, 0.01, 0.5, ctx=ctx), api.asw_gifs(p, 0.01, api.OURS_GIF3, 15, 0.01, 0.5, ctx=ctx),
                api.censusStereo(p, api.BASIC_CENSUS, ctx=ctx), api.sgbm_compute(L, R, sp, ctx=ctx),
                api.stereosgbm_compute(L, R, api.StereoSGBM_create(0, 32, 7, 490, 1960, 1, 63, 10, 100, 32), ctx=ctx))

    seq = [work(None, pL), work(None, pR)]
    for _ in range(3):
        ctxs = [_lib.Context(), _lib.Context()]
        out = [None, None]
        err = []

        def run(i, p):
            try:
                out[i] = work(ctxs[i], p)
            except Exception as e:          # noqa: BLE001
                err.append(e)

        ts = [threading.Thread(target=run, args=(0, pL)), threading.Thread(target=run, args=(1, pR))]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
        assert not err, err
        for i in range(2):
            for a, b in zip(out[i], seq[i]):
                assert np.array_equal(a, b)


@pytest.mark.parametrize("shape", [(200, 320, 64, 12), (1080, 1920, 256, 8)])
def test_two_contexts_run_batched_row_bands_concurrently(api, shape):
    """Two contexts, two host threads, each a batch large enough for the one-launch row-band schedule (whose workgroups
    wait on tickets and progress words inside the launch): the two launches share the device, neither may starve the
    other, the results equal the sequential ones and the contexts' sticky error words stay clear."""
    from cameracalibration_amd import _lib
    # the second shape is the benchmark's frame: 8 pairs x 68 bands x 2 directions = 1088 workgroups per launch, more than
    # the device holds at once, so the two launches compete for residency
    H, W, D, n = shape
    sp = api.sgbm_params(0, D, 8, 32, 10, 1, 8, 1)
    sets = []
    for k in range(2):
        pairs = [api.synth_pair(0x5EED0101 + 100 * k + 2 * i, H, W, D)[:2] for i in range(n)]
        sets.append(([a for a, _ in pairs], [b for _, b in pairs]))
    seq = [api.sgbm_compute_batch(ls, rs, sp, chunk=n) for ls, rs in sets]
    for _ in range(3):
        ctxs = [_lib.Context(), _lib.Context()]
        out, err = [None, None], []

        def run(i):
            try:
                out[i] = api.sgbm_compute_batch(sets[i][0], sets[i][1], sp, chunk=n, ctx=ctxs[i])
                ctxs[i].synchronize()           # raises if a band of this context's launches timed out waiting
            except Exception as e:          # noqa: BLE001
                err.append(e)

        ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=100)
        assert not any(t.is_alive() for t in ts), "a batched call did not return"
        assert not err, err
        for i in range(2):
            assert np.array_equal(out[i], seq[i])
        for c in ctxs:
            c.close()
