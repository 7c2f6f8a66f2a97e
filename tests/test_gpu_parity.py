"""GPU parity: every HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs.  Integer / index results must be bit-exact."""
import numpy as np
import pytest

from conftest import rand_pair

pytestmark = pytest.mark.gpu


def _param(api, L, R, win, minD, maxD, left=True):
    return api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=win, minDisparity=minD, maxDisparity=maxD,
                                isDispLeft=left)


SIZES = [(48, 64), (37, 53), (5, 9), (130, 300), (16, 257)]


@pytest.mark.parametrize("shape", SIZES)
def test_census_basic_and_luts(api, orc, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(api.countCensusImg(img), orc.census(img, orc.BASIC_CENSUS))
    assert np.array_equal(api.countCensusImg_rotationInv(img), orc.census(img, orc.ROTATION_INVARIANT_CENSUS))
    assert np.array_equal(api.countCensusImg_uniform(img), orc.census(img, orc.UNIFORM_CENSUS))


def test_census_known_answers_on_gpu(api):
    img = np.full((9, 11), 77, np.uint8)
    c = api.countCensusImg(img)
    assert (c[1:-1, 1:-1] == 0xFF).all() and not c[0].any() and not c[:, 0].any() and not c[-1].any() and not c[:, -1].any()
    ramp = np.tile(np.arange(20, dtype=np.uint8) * 3, (7, 1))
    assert (api.countCensusImg(ramp)[1:-1, 1:-1] == 0x7C).all()


@pytest.mark.parametrize("shape", [(48, 64), (37, 53), (130, 300)])
def test_census_multiscale_and_circle(api, orc, shape):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    for scale in (7, 16):
        assert np.array_equal(api.countCensusImg_multiScale(img, scale), orc.census(img, orc.MULTISCALE_CENSUS, scale))
    for radius in (2, 5):
        assert np.array_equal(api.countCensusImg_circle(img, radius), orc.census(img, orc.CIRCLE_CENSUS, radius))
    for scale in (7, 16):
        assert np.array_equal(api.countCensusImg_multiScale2(img, scale), orc.census(img, orc.STATISTIC_MULTISCALE_CENSUS, scale))


def test_census_statistic_multiscale_ties(api, orc):
    # few distinct codes: many histogram bins tie at zero, which exercises the "later rank overwrites" rule
    img = np.zeros((40, 60), dtype=np.uint8)
    img[:, 30:] = 200
    img[10:20, 5:25] = 90
    assert np.array_equal(api.countCensusImg_multiScale2(img, 9), orc.census(img, orc.STATISTIC_MULTISCALE_CENSUS, 9))


def test_census_bgr_input_goes_through_gray(api, orc):
    rng = np.random.default_rng(8)
    bgr = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    assert np.array_equal(api.countCensusImg(bgr), orc.census(orc.bgr2gray(bgr)))


def test_hamming(api, orc):
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, (33, 70), dtype=np.uint8)
    b = rng.integers(0, 256, (33, 70), dtype=np.uint8)
    assert np.array_equal(api.countHummingDist(a, b), orc.hamming(a, b))


@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("agg", [0, 1, 2])
@pytest.mark.parametrize("shape,win,minD,maxD", [((48, 64), 7, 0, 15), ((37, 53), 5, 2, 20), ((70, 150), 9, 0, 31),
                                                  ((20, 24), 35, 0, 7)])
def test_census_stereo_integer_modes(api, orc, left, agg, shape, win, minD, maxD):
    L, R = rand_pair(11, shape[0], shape[1], maxD + 1)
    got = api.censusStereo(_param(api, L, R, win, minD, maxD, left), api.BASIC_CENSUS, agg)
    exp = orc.census_stereo(L, R, win, minD, maxD, left, orc.BASIC_CENSUS, agg)
    assert got.dtype == np.float64
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("method", [2, 3, 4, 1, 5])
def test_census_stereo_variants(api, orc, method):
    L, R = rand_pair(12, 60, 90, 16)
    for left in (True, False):
        got = api.censusStereo(_param(api, L, R, 7, 0, 15, left), method, api.AGG_BOX)
        exp = orc.census_stereo(L, R, 7, 0, 15, left, method, orc.AGG_BOX)
        assert np.array_equal(got, exp)


# the marching box kernel: chunk overlap (D not a multiple of the chunk), several waves per workgroup (D up to 300), the
# smallest D it takes (16) and the tile kernel below it, windows 1 / 33 (its largest) / 35 (tile kernel again), strips and
# bands that end inside the image, an image narrower than a strip
@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("shape,win,minD,maxD,agg", [((70, 200), 7, 0, 16, 1), ((70, 200), 7, 3, 18, 2), ((33, 130), 3, 0, 99, 1),
                                                      ((40, 400), 5, 1, 300, 1), ((150, 90), 33, 0, 40, 1), ((150, 90), 35, 0, 40, 1),
                                                      ((300, 50), 9, 0, 70, 0), ((90, 37), 15, 2, 33, 2), ((64, 64), 17, 0, 20, 2)])
def test_census_stereo_marching_box_edges(api, orc, left, shape, win, minD, maxD, agg):
    L, R = rand_pair(41, shape[0], shape[1], min(maxD + 1, shape[1]))
    got = api.censusStereo(_param(api, L, R, win, minD, maxD, left), api.BASIC_CENSUS, agg)
    assert np.array_equal(got, orc.census_stereo(L, R, win, minD, maxD, left, orc.BASIC_CENSUS, agg))


@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("shape,win,minD,maxD", [((70, 200), 7, 0, 16), ((40, 300), 33, 0, 150), ((200, 120), 21, 5, 75), ((60, 90), 35, 0, 30)])
def test_sad_box_marching_edges(api, orc, left, shape, win, minD, maxD):
    # squared absolute differences: 32-bit sums, compared as floats
    L, R = rand_pair(42, shape[0], shape[1], min(maxD + 1, shape[1]))
    got = api.computeSAD_BFOpti(_param(api, L, R, win, minD, maxD, left))
    assert np.array_equal(got, orc.sad_box(L, R, win, minD, maxD, left))


def test_census_stereo_config2_bit_exact(api, orc):
    # 640x480, D = 64, BASIC_CENSUS, winSize 7 on the synthetic pair
    L, R, _ = api.synth_pair(0x5EED0001, 480, 640, 64)
    for agg in (api.AGG_NONE, api.AGG_BOX):
        got = api.censusStereo(_param(api, L, R, 7, 0, 63, True), api.BASIC_CENSUS, agg)
        exp = orc.census_stereo(L, R, 7, 0, 63, True, orc.BASIC_CENSUS, agg)
        assert np.array_equal(got, exp)


def test_bad_parameters_return_none(api, capsys):
    img = np.zeros((8, 8), np.uint8)
    assert api.censusStereo(_param(api, img, img, 4, 0, 3), api.BASIC_CENSUS, api.AGG_BOX) is None
    assert api.computeSAD_BFOpti(_param(api, img, img[:, :6].copy(), 5, 0, 3)) is None
    assert api.computeSAD_inteOpti(_param(api, None, img, 5, 0, 3)) is None
    assert "bad function parameters" in capsys.readouterr().out


@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("shape,win,minD,maxD", [((48, 64), 7, 0, 15), ((37, 53), 5, 3, 20), ((90, 200), 11, 0, 40)])
def test_sad_box(api, orc, left, shape, win, minD, maxD):
    L, R = rand_pair(13, shape[0], shape[1], maxD + 1)
    got = api.computeSAD_BFOpti(_param(api, L, R, win, minD, maxD, left))
    assert np.array_equal(got, orc.sad_box(L, R, win, minD, maxD, left))


def test_sad_box_wraps_modulo_256(api, orc):
    L, R = rand_pair(14, 24, 40, 8)
    got = api.computeSAD_BFOpti(_param(api, L, R, 5, 250, 262, True))
    assert np.array_equal(got, orc.sad_box(L, R, 5, 250, 262, True))
    assert got.min() < 7   # 256..262 wrapped


@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("shape,win,minD,maxD", [((48, 64), 7, 0, 15), ((37, 53), 5, 3, 20)])
def test_sad_inte(api, orc, left, shape, win, minD, maxD):
    L, R = rand_pair(15, shape[0], shape[1], maxD + 1)
    got = api.computeSAD_inteOpti(_param(api, L, R, win, minD, maxD, left))
    assert np.array_equal(got, orc.sad_inte(L, R, win, minD, maxD, left, orc.SAT_INT))
    # tables stay below 2^24 at these sizes, so this is also what the float table gives
    assert np.array_equal(got, orc.sad_inte(L, R, win, minD, maxD, left, orc.SAT_F32))
    assert np.array_equal(api.computeSAD_inteOpti(_param(api, L, R, win, minD, maxD, left), tables=api.SAT_INT), got)


@pytest.mark.parametrize("left", [True, False])
@pytest.mark.parametrize("shape,win,maxD", [((720, 1280), 7, 23), ((1080, 1920), 35, 15)])
def test_sad_inte_float_tables_at_the_reference_sizes(api, orc, left, shape, win, maxD):
    """main() works at 1280 x 720 (cameraCalibration.cpp:125-130): the CV_32F integral tables pass 2^24 there, their
    entries round and the winner follows the rounding.  The default mode reproduces that arithmetic bit for bit; the
    integer mode reproduces the exact-sum restatement; the two differ at these sizes (rate printed with -s)."""
    H, W = shape
    L, R, _ = api.synth_pair(0x5EED0001 + H, H, W, 64)
    p = _param(api, L, R, win, 0, maxD, left)
    f32 = orc.sad_inte(L, R, win, 0, maxD, left, orc.SAT_F32)
    exact = orc.sad_inte(L, R, win, 0, maxD, left, orc.SAT_INT)
    assert np.array_equal(api.computeSAD_inteOpti(p), f32)
    assert np.array_equal(api.computeSAD_inteOpti(p, tables=api.SAT_F32), f32)
    assert np.array_equal(api.computeSAD_inteOpti(p, tables=api.SAT_INT), exact)
    differ = float((f32 != exact).mean())
    print("SAT_F32 vs SAT_INT at %dx%d, win %d, %s: %.4f %% of the pixels differ" % (W, H, win, "left" if left else "right", 100 * differ))
    assert differ > 0, "the tables never passed 2^24: this size does not exercise the float rounding"


def test_sad_inte_integer_tables_through_the_tuning_switch(api, orc):
    from cameracalibration_amd import _lib
    L, R, _ = api.synth_pair(0x5EED0003, 720, 1280, 64)
    p = _param(api, L, R, 9, 0, 11, True)
    try:
        _lib.tuning_set(_lib.TUNE_SAT_INT_TABLES, 1)
        got = api.computeSAD_inteOpti(p)
    finally:
        _lib.tuning_reset()
    assert np.array_equal(got, orc.sad_inte(L, R, 9, 0, 11, True, orc.SAT_INT))


SGM_CASES = [
    # H, W, minD, D, P1, P2, uniq, maxDiff, paths, subpixel
    (24, 40, 0, 16, 8, 32, 10, 1, 8, 1),
    (37, 53, 0, 64, 8, 32, 10, 1, 8, 1),
    (30, 70, 2, 48, 3, 20, 5, 2, 5, 1),
    (33, 65, 0, 100, 8, 32, 15, 1, 4, 0),
    (20, 90, 0, 128, 8, 32, 10, 1, 8, 1),
    (16, 130, 0, 256, 8, 32, 10, 1, 8, 1),
    (12, 80, 3, 300, 10, 40, 0, 1, 8, 1),
    (3, 3, 0, 4, 0, 0, -1, 0, 8, 1),
    (35, 49, 0, 80, 20, 300, 10, 1, 8, 1),      # path values above 255: 16-bit horizontal volumes
    (70, 33, 1, 256, 8, 32, 10, 2, 8, 1),
    (17, 16, 0, 32, 8, 32, 10, 1, 8, 1),
    (45, 61, 0, 64, 10, 100, 10, 1, 8, 1),      # byte path states but 16-bit four-path partial sums
    (150, 20, 0, 16, 8, 32, 10, 1, 8, 1),       # taller than wide: most sheared strips are partial
    (40, 7, 0, 8, 8, 32, 10, 1, 8, 1),          # narrower than one strip
    (3, 50, 0, 16, 8, 32, 10, 1, 8, 1),         # the smallest height the census frame allows
    (64, 200, 0, 256, 8, 47, 10, 1, 8, 1),      # largest P2 with byte partial sums
    (64, 200, 0, 256, 8, 48, 10, 1, 8, 1),      # smallest P2 without
    (90, 310, 0, 128, 8, 247, 10, 1, 8, 1),     # largest P2 with byte path states
    (60, 300, 0, 256, 8, 32, 10, 1, 5, 1),      # five paths (StereoSGBM's default mode): one fused down sweep
    (75, 40, 1, 20, 2, 5, 10, 1, 5, 0),
    (50, 80, 0, 64, 20, 300, 10, 1, 5, 1),      # five paths with 16-bit path values: row bands with 16-bit states
    (120, 260, 0, 64, 8, 32, 10, 1, 8, 1),      # four lanes per pixel, several 64-column strips
    (90, 200, 1, 65, 8, 32, 10, 1, 8, 1),       # just above the four-lane limit: eight lanes, masked tail
    (100, 330, 0, 96, 6, 60, 10, 2, 8, 1),      # eight lanes per pixel, 16-bit four-path sums
    (80, 150, 0, 40, 8, 32, 10, 1, 5, 1),       # four lanes per pixel, five paths
    (50, 120, 0, 64, 8, 200, 10, 1, 5, 1),      # five paths, byte path values whose pairwise sums exceed a byte
    (50, 120, 0, 64, 8, 44, 10, 1, 8, 1),       # eight paths: byte four-path sums, but horizontal + sum exceeds a byte
    (50, 120, 0, 64, 8, 43, 10, 1, 8, 1),       # ... and the largest P2 for which it does not
    (40, 700, 0, 512, 8, 32, 10, 1, 8, 1),      # 512 disparities: sixteen lanes x 32 per lane, row bands of eight rows
    (37, 420, 3, 300, 6, 60, 10, 2, 8, 1),      # ... with a masked tail and 16-bit four-path sums
    (30, 350, 0, 257, 8, 247, 10, 1, 8, 0),     # ... just above 256, largest P2 with byte path states
    (45, 300, 0, 200, 8, 32, 10, 1, 8, 1),      # sixteen lanes with a masked tail
    (45, 300, 0, 129, 8, 100, 10, 1, 8, 1),     # ... just above 128, 16-bit four-path sums
    (60, 130, 1, 256, 8, 32, 10, 1, 4, 1),      # four paths (horizontal + vertical): row bands without the diagonal steps
    (40, 210, 0, 200, 5, 300, 10, 1, 4, 1),      # ... with 16-bit path states and a masked tail
    (44, 90, 0, 48, 8, 32, 10, 2, 4, 0),        # ... few disparities (still sixteen lanes per pixel: most of them masked)
    (70, 200, 0, 256, 8, 300, 10, 1, 8, 1),     # 16-bit path states: row bands with 16-bit hand-off (batches), upright strips (latency)
    (50, 160, 2, 200, 20, 500, 10, 1, 8, 1),    # ... with a masked tail, close to the largest P2 (8 * (8 + P2) < 4096)
    (33, 90, 0, 128, 8, 248, 5, 2, 8, 0),       # ... the smallest P2 whose path values leave the byte range
]


# how a single pair is scheduled: "latency" = the default for small batches (two sweeps side by side on two
# streams + summing winner-take-all; for one pair the three paths of a sweep on three groups of waves, "latency, one
# wave group" forces the kernel that two or three pairs get), "pipeline" = the sheared strip pipeline that batches use,
# with the strip width of small batches (512 lanes per workgroup) or of large ones (256); "row bands" = the pipeline of
# row bands with four paths per sweep that batches of eight-path problems take, here one sweep after the other;
# "row bands, both edges" = the row bands as batches run them: both sweeps start at once, each from its image edge;
# "row bands, ring of two" = the same with the smallest hand-off ring, so that entries are reused many times
@pytest.fixture(params=["latency", "latency, one wave group", "pipeline", "pipeline, narrow strips", "row bands", "row bands, both edges",
                        "row bands, ring of two"])
def sgm_schedule(request):
    from cameracalibration_amd import _lib
    _lib.tuning_reset()
    if request.param == "latency, one wave group":
        _lib.tuning_set(_lib.TUNE_SGM_PATH_WAVES_MAX, 0)
    elif request.param.startswith("row bands"):
        _lib.tuning_set(_lib.TUNE_SGM_SPLIT_MAX, 0)
        if request.param == "row bands":
            _lib.tuning_set(_lib.TUNE_SGM_MEET_MAX, 0)
        if "ring" in request.param:
            _lib.tuning_set(_lib.TUNE_SGM_HAND_RING, 2)
    elif request.param != "latency":
        _lib.tuning_set(_lib.TUNE_SGM_SPLIT_MAX, 0)
        _lib.tuning_set(_lib.TUNE_SGM_PIPELINE, 1)
        _lib.tuning_set(_lib.TUNE_SGM_STRIP_LANES, 256 if "narrow" in request.param else 512)
    yield request.param
    _lib.tuning_reset()


BATCH_CASES = [
    # B, H, W, pitch, D, minD, P2, tuning
    (9, 70, 150, 150, 256, 0, 32, {}),                          # row bands, several pairs per band level
    (9, 70, 150, 150, 256, 0, 32, {"SGM_ONE_LAUNCH": 0}),       # both edges at once as two launches per phase instead of one launch
    (7, 100, 130, 130, 256, 0, 32, {}),                         # one launch, an odd number of bands (the middle band changes roles between the directions)
    (7, 128, 130, 130, 256, 0, 32, {"SGM_HAND_RING": 2}),       # one launch, an even number of bands, the smallest hand-off ring
    (9, 70, 150, 150, 256, 0, 32, {"SGM_PAIR_GROUP": 4}),       # pairs dispatched in groups, the last group partial
    (5, 45, 300, 300, 512, 0, 32, {}),                          # 32 disparities per lane
    (3, 90, 210, 210, 300, 2, 60, {}),                          # a few pairs at D > 256: sweeps from both image edges, 16-bit sums
    (7, 50, 120, 120, 128, 0, 300, {}),                         # 16-bit path states on the row bands
    (6, 40, 75, 83, 64, 1, 32, {}),                             # rows that are not dword-aligned: the census reads bytes
    (6, 40, 76, 80, 64, 0, 32, {"SGM_PIPELINE": 1}),            # the sheared strips
    (2, 130, 90, 92, 200, 0, 32, {"SGM_SPLIT_MAX": 0, "SGM_HAND_RING": 4}),
    (9, 70, 150, 150, 256, 0, 32, {"SGM_MEET_MAX": 0}),         # one sweep after the other instead of both edges at once
    (6, 70, 140, 140, 96, 0, 32, {}),                           # eight lanes per pixel, both edges
    (6, 75, 100, 100, 48, 3, 32, {}),                           # four lanes per pixel, both edges, masked tail
    (7, 80, 120, 120, 128, 0, 300, {"SGM_MEET_MAX": 0}),        # 16-bit path states, one sweep after the other
    (9, 70, 150, 150, 256, 0, 32, {"SGM_RING_HANDOFF": 1}),     # one launch without the workgroup barrier per step: rows handed on inside a wave and through LDS rings between waves
    (7, 100, 130, 130, 256, 0, 32, {"SGM_RING_HANDOFF": 1}),
    (7, 128, 130, 130, 256, 0, 32, {"SGM_RING_HANDOFF": 1, "SGM_HAND_RING": 2}),
    (5, 37, 5, 8, 256, 0, 32, {"SGM_RING_HANDOFF": 1}),         # ... rows shorter than the ring between two waves
    (5, 52, 9, 12, 256, 0, 32, {"SGM_RING_HANDOFF": 1}),
    (5, 37, 5, 8, 256, 0, 32, {}),
]
BATCH5_CASES = [
    (6, 70, 150, 150, 256, 0, 32), (6, 60, 140, 140, 100, 2, 32), (5, 50, 90, 90, 48, 0, 60), (6, 64, 120, 120, 128, 0, 300), (1, 40, 100, 100, 64, 0, 400),
]


@pytest.mark.parametrize("B,H,W,pitch,D,minD,P2,tuning", BATCH_CASES)
def test_sgm_batch_variants(api, orc, B, H, W, pitch, D, minD, P2, tuning):
    """mdisp_sgbm_batch_dev (pitched device input, pair stride with slack) == the oracle on every pair."""
    import ctypes as C
    import torch
    from cameracalibration_amd import _lib
    lib = _lib.load()
    _lib.tuning_reset()
    for k, v in tuning.items():
        _lib.tuning_set(getattr(_lib, "TUNE_" + k), v)
    try:
        params = api.sgbm_params(minD, D, 8, P2, 10, 1, 8, 1)
        stride = pitch * H + 52                                 # pairs need not be packed (52: keeps dword alignment, breaks 16-byte alignment)
        bufL = np.full(B * stride, 0xAB, np.uint8)
        bufR = np.full(B * stride, 0xCD, np.uint8)
        Ls, Rs = [], []
        for i in range(B):
            L, R = rand_pair(1000 + 7 * i + H, H, W, min(D, W))
            Ls.append(L)
            Rs.append(R)
            bufL[i * stride:i * stride + pitch * H].reshape(H, pitch)[:, :W] = L
            bufR[i * stride:i * stride + pitch * H].reshape(H, pitch)[:, :W] = R
        dl, dr = torch.from_numpy(bufL).cuda(), torch.from_numpy(bufR).cuda()
        out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        ctx = _lib.Context()
        _lib.check(lib.mdisp_sgbm_batch_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), B, stride, H, W, pitch, C.byref(params),
                                            out.data_ptr(), None, 0))
        ctx.synchronize()
        got = out.cpu().numpy()
        for i in range(B):
            assert np.array_equal(got[i], orc.sgm(Ls[i], Rs[i], minD, D, 8, P2, 10, 1, 8, 1)), f"pair {i}"
    finally:
        _lib.tuning_reset()


@pytest.mark.gpu
@pytest.mark.parametrize("tuning", [{}, {"SGM_ONE_LAUNCH": 0}, {"SGM_MEET_MAX": 0}, {"SGM_PIPELINE": 1}])
def test_sgm_batch_ignores_what_the_workspace_held(api, orc, tuning):
    """A caller's workspace arrives with whatever the previous user left in it: the maps must not depend on it (every word
    the kernels read -- tickets, progress and done words, hand-off rows, sums, records -- is written or cleared by the call)."""
    import ctypes as C
    import torch
    from cameracalibration_amd import _lib
    lib = _lib.load()
    _lib.tuning_reset()
    try:
        for k, v in tuning.items():
            _lib.tuning_set(getattr(_lib, "TUNE_" + k), v)
        B, H, W, D = 12, 150, 260, 256
        params = api.sgbm_params(0, D, 8, 32, 10, 1, 8, 1)
        Ls, Rs = zip(*[rand_pair(3000 + 5 * i, H, W, 200) for i in range(B)])
        dl, dr = torch.from_numpy(np.stack(Ls)).cuda(), torch.from_numpy(np.stack(Rs)).cuda()
        nbytes = lib.mdisp_sgbm_batch_workspace_bytes(H, W, B, C.byref(params))
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        ctx = _lib.Context()
        maps = []
        for fill in (0, 255, None):
            if fill is None:
                ws.random_(0, 255)
            else:
                ws.fill_(fill)
            out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
            torch.cuda.synchronize()                           # the context runs on a stream of its own: the fill has to be complete
            _lib.check(lib.mdisp_sgbm_batch_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), B, H * W, H, W, W, C.byref(params),
                                                out.data_ptr(), ws.data_ptr(), nbytes))
            ctx.synchronize()
            maps.append(out.cpu().numpy())
        assert np.array_equal(maps[0][0], orc.sgm(Ls[0], Rs[0], 0, D, 8, 32, 10, 1, 8, 1))
        assert np.array_equal(maps[0], maps[1]) and np.array_equal(maps[0], maps[2])
    finally:
        _lib.tuning_reset()


@pytest.mark.parametrize("B,H,W,pitch,D,minD,P2", BATCH5_CASES)
def test_sgm_five_path_batches(api, orc, B, H, W, pitch, D, minD, P2):
    """Five paths on the row bands: four paths in one sweep, then the opposite horizontal path with the winner-take-all."""
    import ctypes as C
    import torch
    from cameracalibration_amd import _lib
    lib = _lib.load()
    _lib.tuning_reset()
    params = api.sgbm_params(minD, D, 8, P2, 10, 1, 5, 1)
    Ls, Rs = zip(*[rand_pair(2000 + 3 * i + H, H, W, min(D, W)) for i in range(B)])
    dl, dr = torch.from_numpy(np.stack(Ls)).cuda(), torch.from_numpy(np.stack(Rs)).cuda()
    out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx = _lib.Context()
    _lib.check(lib.mdisp_sgbm_batch_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), B, H * W, H, W, pitch, C.byref(params), out.data_ptr(), None, 0))
    ctx.synchronize()
    got = out.cpu().numpy()
    for i in range(B):
        assert np.array_equal(got[i], orc.sgm(Ls[i], Rs[i], minD, D, 8, P2, 10, 1, 5, 1)), f"pair {i}"


@pytest.mark.parametrize("case", SGM_CASES)
def test_sgm_bit_exact(api, orc, case, sgm_schedule):
    H, W, minD, D, P1, P2, uniq, maxDiff, paths, sub = case
    L, R = rand_pair(H * 31 + W, H, W, min(D, W))
    got = api.sgbm_compute(L, R, api.sgbm_params(minD, D, P1, P2, uniq, maxDiff, paths, sub))
    exp = orc.sgm(L, R, minD, D, P1, P2, uniq, maxDiff, paths, sub)
    assert got.dtype == np.int16
    assert np.array_equal(got, exp)


def test_sgm_synthetic_pair_recovers_ground_truth(api, orc):
    L, R, gt = api.synth_pair(0x5EED0001, 120, 320, 64)
    got = api.sgbm_compute(L, R, api.sgbm_params(0, 64, 8, 32, 10, 1, 8, 1))
    exp = orc.sgm(L, R, 0, 64, 8, 32, 10, 1, 8, 1)
    assert np.array_equal(got, exp)
    valid = got >= 0
    assert valid.mean() > 0.5
    err = np.abs(got[valid] / 16.0 - gt[valid])
    assert np.median(err) <= 1.0


def test_sgm_variants_of_census(api, orc):
    L, R = rand_pair(21, 30, 60, 32)
    for m in (api.ROTATION_INVARIANT_CENSUS, api.UNIFORM_CENSUS):
        got = api.sgbm_compute(L, R, api.sgbm_params(0, 32, 8, 32, 10, 1, 8, 1, 1, m))
        exp = orc.sgm(L, R, 0, 32, 8, 32, 10, 1, 8, 1, 1, m)
        assert np.array_equal(got, exp)


def test_sgm_rejects_overflowing_penalties(api):
    from cameracalibration_amd._lib import MdispError, E_RANGE
    img = np.zeros((8, 8), np.uint8)
    with pytest.raises(MdispError) as e:
        api.sgbm_compute(img, img, api.sgbm_params(0, 8, 100, 9000))
    assert e.value.code == E_RANGE


def test_lr_check(api, orc):
    rng = np.random.default_rng(16)
    dL = rng.integers(0, 20, (30, 50)).astype(np.float32) + rng.random((30, 50), dtype=np.float32)
    dR = rng.integers(0, 20, (30, 50)).astype(np.float32)
    vl, vr = api.lr_check(dL, dR, 1.0)
    el, er = orc.lr_check(dL, dR, 1.0)
    assert np.array_equal(vl, el) and np.array_equal(vr, er)


def test_remap_fixed_point(api, orc):
    rng = np.random.default_rng(17)
    for cn in (1, 3):
        shape = (40, 60) if cn == 1 else (40, 60, 3)
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        K = np.array([[80.0, 0, 29.5], [0, 80.0, 19.5], [0, 0, 1]])
        dist = np.array([-0.2, 0.05, 0.001, -0.002, 0.01])
        ang = 0.03
        R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
        P = np.array([[75.0, 0, 31.0, 0], [0, 75.0, 21.0, 0], [0, 0, 1, 0]])
        m1, m2 = api.initUndistortRectifyMap(K, dist, R, P, (60, 40))
        o1, o2 = orc.init_undistort_rectify_map(K, dist, R, P, (60, 40))
        assert np.array_equal(m1, o1) and np.array_equal(m2, o2)
        assert np.array_equal(api.remap(img, m1, m2), orc.remap(img, m1, m2))


@pytest.mark.parametrize("B", [3, 6])        # a small batch (latency schedule) and one the strip pipeline takes
def test_sgm_batch_matches_single_calls(api, orc, B):
    """mdisp_sgbm_batch_dev over B pairs == the oracle on every pair."""
    import ctypes as C
    import torch
    from cameracalibration_amd import _lib
    lib = _lib.load()
    H, W, D = 40, 72, 96
    params = api.sgbm_params(0, D, 8, 32, 10, 1, 8, 1)
    Ls, Rs = [], []
    for i in range(B):
        L, R, _ = api.synth_pair(0x5EED0001 + 2 * i, H, W, D)
        Ls.append(L)
        Rs.append(R)
    dl = torch.from_numpy(np.stack(Ls)).cuda()
    dr = torch.from_numpy(np.stack(Rs)).cuda()
    out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
    ctx = _lib.Context(stream=torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.mdisp_sgbm_batch_dev(ctx.handle, dl.data_ptr(), dr.data_ptr(), B, H * W, H, W, W, C.byref(params),
                                        out.data_ptr(), None, 0))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i in range(B):
        assert np.array_equal(got[i], orc.sgm(Ls[i], Rs[i], 0, D, 8, 32, 10, 1, 8, 1))


def test_sgm_volume_bit_exact(api, orc, sgm_schedule):
    for (H, W, minD, D, P1, P2) in [(24, 40, 0, 16, 8, 32), (37, 53, 0, 64, 8, 32), (16, 130, 0, 256, 8, 32),
                                    (40, 70, 3, 200, 5, 30), (33, 17, 0, 256, 8, 32), (29, 45, 0, 96, 20, 300),
                                    (21, 90, 0, 512, 8, 32), (18, 75, 1, 400, 8, 100)]:
        L, R = rand_pair(H * 31 + W, H, W, min(D, W))
        S = api.sgbm_volume(L, R, api.sgbm_params(minD, D, P1, P2, 10, 1, 8, 1))
        _, E = orc.sgm(L, R, minD, D, P1, P2, 10, 1, 8, 1, want_S=True)
        assert np.array_equal(S, E)
        S5 = api.sgbm_volume(L, R, api.sgbm_params(minD, D, P1, P2, 10, 1, 5, 1))
        _, E5 = orc.sgm(L, R, minD, D, P1, P2, 10, 1, 5, 1, want_S=True)
        assert np.array_equal(S5, E5)


@pytest.mark.gpu
def test_sgm_full_size_frame_bit_exact(api, orc, sgm_schedule):
    # the benchmark configuration itself: 1920x1080, D = 256, P1 = 8, P2 = 32, 8 paths (oracle: about a second)
    L, R, gt = api.synth_pair(0x5EED0001, 1080, 1920, 256)
    p = api.sgbm_params(0, 256, 8, 32, 10, 1, 8, 1)
    got = api.sgbm_compute(L, R, p)
    exp = orc.sgm(L, R, 0, 256, 8, 32, 10, 1, 8, 1)
    assert np.array_equal(got, exp)
    # size-independent property: away from occlusions the planted disparity is recovered
    valid = got >= 0
    assert valid.mean() > 0.6
    assert np.median(np.abs(got[valid] / 16.0 - gt[valid])) <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("tuning", [{}, {"SGM_ONE_LAUNCH": 0}, {"SGM_RING_HANDOFF": 1}])
def test_sgm_batch_full_size_fills_the_gpu(api, orc, tuning):
    """A batch of full-size frames that occupies every workgroup slot several times over (the one-launch schedule hands
    four-path sums between workgroups of one launch, on whatever part of the chip they run): every pair equals the oracle
    for the first frames and the single-pair call, which another test pins to the oracle, for the rest."""
    from cameracalibration_amd import _lib
    B, H, W, D = 20, 1080, 1920, 256
    par = api.sgbm_params(0, D, 8, 32, 10, 1, 8, 1)
    pairs = [api.synth_pair(0x5EED0101 + 2 * i, H, W, D)[:2] for i in range(B)]
    _lib.tuning_reset()
    try:
        for k, v in tuning.items():
            _lib.tuning_set(getattr(_lib, "TUNE_" + k), v)
        got = api.sgbm_compute_batch([a for a, _ in pairs], [b for _, b in pairs], par, chunk=B)
    finally:
        _lib.tuning_reset()
    for i, (a, b) in enumerate(pairs):
        exp = orc.sgm(a, b, 0, D, 8, 32, 10, 1, 8, 1) if i < 3 else api.sgbm_compute(a, b, par)
        assert np.array_equal(got[i], exp), "pair %d" % i


@pytest.mark.gpu
def test_reprojection_to_3d(api, orc):
    rng = np.random.default_rng(5)
    H, W = 45, 70
    disp = rng.integers(0, 64, (H, W)).astype(np.float32)
    disp[rng.random((H, W)) < 0.2] = 0.0                      # holes stay at the origin
    disp[3, 4] = 0.01                                          # far point: beyond the 20000 cut of the triangulation
    Q = np.array([[1, 0, 0, -31.5], [0, 1, 0, -22.25], [0, 0, 0, 855.95], [0, 0, 6.55e-3, 0.0]])
    assert np.array_equal(api.reprojectQ(disp, Q), orc.reproject_q(disp, Q))
    K = np.array([[855.95, 0, 31.5], [0, 860.1, 22.25], [0, 0, 1.0]])
    bgr = rng.integers(0, 3, (H, W, 3)).astype(np.float32)     # some pixels are black
    for colour in (None, bgr):
        got = api.reprojectTriangulate(disp, K, 152.63, colour)
        exp = orc.reproject_triangulate(disp, K, 152.63, colour)
        assert np.array_equal(got, exp)
    assert not got[3, 4].any()


@pytest.mark.gpu
@pytest.mark.parametrize("n,chunk,W", [(7, 3, 72), (5, 0, 70), (1, 4, 72), (12, 6, 64)])
def test_sgm_host_batch_streams_chunks(api, orc, n, chunk, W):
    # pairs in host memory, processed chunk by chunk with the copies on a second stream; ragged last chunk,
    # a width that is not a multiple of 4 (pitched copies), a chunk larger than the sequence
    H, D = 36, 48
    p = api.sgbm_params(0, D, 8, 32, 10, 1, 8, 1)
    Ls, Rs = [], []
    for i in range(n):
        L, R, _ = api.synth_pair(0x5EED0101 + 2 * i, H, W, D)
        Ls.append(L)
        Rs.append(R)
    got = api.sgbm_compute_batch(Ls, Rs, p, chunk)
    assert got.shape == (n, H, W)
    for i in range(n):
        assert np.array_equal(got[i], orc.sgm(Ls[i], Rs[i], 0, D, 8, 32, 10, 1, 8, 1))


@pytest.mark.gpu
def test_rectification_maps_built_on_the_gpu(api, orc):
    # device-resident maps: every entry equals the host construction and the oracle, at a small size with all eight
    # distortion terms and at the calibrated 2560x1440 camera of the reference (unditortStereoCalib.xml values)
    import json
    import os
    rng = np.random.default_rng(23)
    K = np.array([[80.0, 0, 29.5], [0, 81.0, 19.5], [0, 0, 1]])
    ang = 0.03
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    P = np.array([[75.0, 0, 31.0, 0], [0, 75.0, 21.0, 0], [0, 0, 1, 0]])
    for dist in (np.zeros(0), np.array([-0.2, 0.05, 0.001, -0.002]), np.array([-0.2, 0.05, 0.001, -0.002, 0.01]),
                 np.array([-0.2, 0.05, 0.001, -0.002, 0.01, 0.02, -0.01, 0.003])):
        maps = api.RectifyMaps(K, dist, R, P, (61, 43))
        g1, g2 = maps.download()
        h1, h2 = api.initUndistortRectifyMap(K, dist, R, P, (61, 43))
        o1, o2 = orc.init_undistort_rectify_map(K, dist, R, P, (61, 43))
        assert np.array_equal(g1, h1) and np.array_equal(g2, h2)
        assert np.array_equal(g1, o1) and np.array_equal(g2, o2)
        for shape in ((40, 60), (40, 60, 3)):
            img = rng.integers(0, 256, shape, dtype=np.uint8)
            assert np.array_equal(maps.remap(img), orc.remap(img, o1, o2))
        maps.close()
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stereo_rectify_2560x1440.json")))
    m = {k: (np.array(v["data"]).reshape(v["rows"], v["cols"]) if isinstance(v, dict) else v) for k, v in g.items()}
    maps = api.RectifyMaps(m["StereoCalib_K1"], m["StereoCalib_D1"], m["Rectify_R1"], m["Rectify_P1"], (2560, 1440))
    g1, g2 = maps.download()
    h1, h2 = api.initUndistortRectifyMap(m["StereoCalib_K1"], m["StereoCalib_D1"], m["Rectify_R1"], m["Rectify_P1"], (2560, 1440))
    assert np.array_equal(g1, h1) and np.array_equal(g2, h2)


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", [0, 1])
def test_sgm_handoff_timeout_is_not_lost_between_chunks(api, orc, pipeline):
    # a hand-off timeout in the FIRST chunk of a chunked batch must still fail the call: the error word is owned by the
    # context and no later chunk clears it.  The word is planted before chunk 0's kernels; afterwards the context works again.
    from cameracalibration_amd import _lib
    from cameracalibration_amd._lib import MdispError, E_HIP
    H, W, D, n = 36, 72, 48, 9
    p = api.sgbm_params(0, D, 8, 32, 10, 1, 8, 1)
    Ls, Rs = [], []
    for i in range(n):
        L, R, _ = api.synth_pair(0x5EED0201 + 2 * i, H, W, D)
        Ls.append(L)
        Rs.append(R)
    _lib.tuning_reset()
    try:
        _lib.tuning_set(_lib.TUNE_SGM_SPLIT_MAX, 0)          # every chunk goes through a pipelined schedule
        _lib.tuning_set(_lib.TUNE_SGM_PIPELINE, pipeline)
        _lib.tuning_set(_lib.TUNE_SGM_PLANT_ERROR, 1)
        with pytest.raises(MdispError) as e:
            api.sgbm_compute_batch(Ls, Rs, p, 3)
        assert e.value.code == E_HIP
        assert _lib.load().mdisp_tuning_get(_lib.TUNE_SGM_PLANT_ERROR) == -1
        got = api.sgbm_compute_batch(Ls, Rs, p, 3)
        for i in range(n):
            assert np.array_equal(got[i], orc.sgm(Ls[i], Rs[i], 0, D, 8, 32, 10, 1, 8, 1))
    finally:
        _lib.tuning_reset()
