"""GPU parity of the guided-filter matcher of namespace stereomatch_1 (methods/method_StereoMatching.cpp:3288-3349,
3573-3672, 3685-3781) and of getGuidedFilter with a multi-channel guide (disp_method/gifs.cpp:8-135) against the CPU
oracle.  Float work: the filtered slices agree to 1e-3 on the normalised cost (the bar of the single-channel guided
filter), the winner-take-all map may differ only where the oracle's best and runner-up are closer than 1e-5."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["marching box kernels", "LDS-tile box kernels"])
def mc_kernels(request):
    """Every test runs on both implementations of the box filters (the marching kernels are what the library uses)."""
    from cameracalibration_amd import _lib
    _lib.tuning_reset()
    if request.param == "LDS-tile box kernels":
        _lib.tuning_set(_lib.TUNE_GIF_MC_TILES, 1)
    yield request.param
    _lib.tuning_reset()


def _colour_pair(seed, H, W, shift):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(H, W + 2 * shift + 8, 3)).astype(np.int32)
    sm = (base + np.roll(base, 1, 1) + np.roll(base, -1, 1) + np.roll(base, 1, 0)) // 4
    right = sm[:, shift:shift + W].astype(np.uint8)
    left = np.clip(sm[:, :W] + rng.integers(-2, 3, size=(H, W, 3)), 0, 255).astype(np.uint8)      # left(x) = right(x - shift)
    return np.ascontiguousarray(left), np.ascontiguousarray(right)


@pytest.mark.parametrize("is_left", [True, False])
@pytest.mark.parametrize("shape,win,disp", [((40, 70), 7, 3), ((33, 129), 15, 0), ((20, 64), 1, 5), ((50, 45), 35, 9)])
def test_cost_sad_d(api, orc, is_left, shape, win, disp):
    H, W = shape
    L, R = _colour_pair(H + W, H, W, 6)
    pad = 12
    if is_left:
        l, r = L[..., 1].copy(), np.pad(R[..., 1], ((0, 0), (pad, 0)), mode="symmetric")
    else:
        l, r = np.pad(L[..., 1], ((0, 0), (0, pad)), mode="symmetric"), R[..., 1].copy()
    dt = api.DISPARITY_LEFT if is_left else api.DISPARITY_RIGHT
    got = api.stereomatch_1.getCostSAD_d(l, r, disp, dt, win)
    exp = orc.cost_sad_d(l, r, disp, is_left, win)
    assert got.dtype == np.float32 and np.array_equal(got, exp)          # integer window sums scaled once: exact


def test_cost_sad_d_colour_input_and_refusals(api, orc):
    L, R = _colour_pair(3, 24, 40, 4)
    Rb = np.ascontiguousarray(np.pad(R, ((0, 0), (8, 0), (0, 0)), mode="symmetric"))
    got = api.stereomatch_1.getCostSAD_d(L, Rb, 2, api.DISPARITY_LEFT, 5)
    gray = lambda c: ((c[..., 0].astype(np.int64) * 1868 + c[..., 1].astype(np.int64) * 9617 + c[..., 2].astype(np.int64) * 4899 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(got, orc.cost_sad_d(gray(L), gray(Rb), 2, True, 5))
    assert api.stereomatch_1.getCostSAD_d(L, Rb, 2, api.DISPARITY_LEFT, 6) is None        # even window
    assert api.stereomatch_1.getCostSAD_d(L, R, 2, api.DISPARITY_LEFT, 5) is None         # the right view was not bordered


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("shape,cn,r,eps", [((40, 70), 3, 7, 0.01), ((37, 130), 6, 15, 0.01), ((70, 33), 6, 5, 1e-4), ((30, 64), 3, 1, 0.01)])
def test_guided_filter_multi_channel_guide(api, orc, variant, shape, cn, r, eps):
    H, W = shape
    rng = np.random.default_rng(H * 31 + W + cn)
    L, R = _colour_pair(H + cn, H, W, 3)
    guide = np.ascontiguousarray(np.concatenate([L, R], axis=2)[..., :cn])
    guide[5:12, 8:20] = 77                                   # a constant patch: zero variance (variant 1 regularises with 0 there)
    P = rng.random((H, W)).astype(np.float32) * 40 + np.linspace(0, 9, W, dtype=np.float32)[None, :]
    f = api.stereomatch_1.getGuidedFilter if variant == 1 else api.getGuidedFilter_multi
    got = f(guide, P, r, eps)
    exp = orc.guided_filter_mc(guide, P, r, eps, variant)
    assert np.isfinite(exp).all() and np.isfinite(got).all()
    assert np.abs(got - exp).max() <= 1e-3
    P8 = rng.integers(0, 256, (H, W), dtype=np.uint8)
    assert np.abs(f(guide, P8, r, eps) - orc.guided_filter_mc(guide, P8, r, eps, variant)).max() <= 1e-3
    assert f(guide, P[:-1], r, eps) is None                  # size mismatch: empty result


def _tie_aware_equal(api, orc, got, L, R, is_left, eps, win, minD, numD, exp):
    """Pixels where the maps differ must be near-ties of the oracle's filtered slices (gap below 1e-5)."""
    bad = np.argwhere(got != exp)
    if len(bad) == 0:
        return 0
    assert len(bad) <= 0.002 * got.size, "%d pixels differ" % len(bad)
    # recompute the two competing slices with the oracle's pieces and compare their values at the differing pixels
    maxoff = minD + numD - 1
    gray = lambda c: ((c[..., 0].astype(np.int64) * 1868 + c[..., 1].astype(np.int64) * 9617 + c[..., 2].astype(np.int64) * 4899 + 8192) >> 14).astype(np.uint8)
    W = L.shape[1]
    if is_left:
        othP = np.ascontiguousarray(np.pad(R, ((0, 0), (maxoff, 0), (0, 0)), mode="symmetric"))
    else:
        othP = np.ascontiguousarray(np.pad(L, ((0, 0), (0, maxoff), (0, 0)), mode="symmetric"))
    slices = {}

    def slice_of(d):
        if d not in slices:
            off = maxoff - d if is_left else d
            crop = np.ascontiguousarray(othP[:, off:off + W])
            if is_left:
                cost = orc.cost_sad_d(gray(L), gray(othP), d, True, win)
                guide = np.concatenate([L, crop], axis=2)
            else:
                cost = orc.cost_sad_d(gray(othP), gray(R), d, False, win)
                guide = np.concatenate([crop, R], axis=2)
            slices[d] = orc.guided_filter_mc(np.ascontiguousarray(guide), cost, win, eps, 1)
        return slices[d]

    for y, x in bad[:50]:
        a, b = slice_of(int(got[y, x]))[y, x], slice_of(int(exp[y, x]))[y, x]
        assert abs(float(a) - float(b)) < 1e-5, (y, x, got[y, x], exp[y, x], a, b)
    return len(bad)


@pytest.mark.parametrize("is_left", [True, False])
@pytest.mark.parametrize("shape,eps,win,minD,numD", [((40, 90), 0.01, 7, 0, 12), ((33, 70), 0.01, 15, 2, 9), ((60, 140), 1e-3, 5, 4, 20)])
def test_adaptive_weight_guidedf(api, orc, is_left, shape, eps, win, minD, numD):
    H, W = shape
    L, R = _colour_pair(H * 3 + W, H, W, minD + numD // 2)
    dt = api.DISPARITY_LEFT if is_left else api.DISPARITY_RIGHT
    got = api.stereomatch_1.computeAdaptiveWeight_GuidedF(L, R, dt, eps, win, minD, numD)
    exp = orc.adaptive_weight_guidedf(L, R, is_left, eps, win, minD, numD)
    assert got.dtype == np.float32 and got.shape == (H, W)
    _tie_aware_equal(api, orc, got, L, R, is_left, eps, win, minD, numD, exp)
    if is_left:
        inner = got[:, minD + numD:]
        assert (inner == minD + numD // 2).mean() > 0.7                 # finds the synthetic shift


@pytest.mark.parametrize("which", [1, 2, 3])
def test_guided_filter_matchers_at_the_reference_size(api, which):
    """1280 x 720 colour pair at the reference's own default arguments (methods/method_StereoMatching.h:179-187: window 35,
    minDisparity 186, 144 candidates) -- too large for the oracle in a test, so a size-independent property: the views
    differ by a constant shift inside the candidate range, and away from the border the matchers on a dissimilarity recover it.
    (_3 takes the MINIMUM of the reference's correlation measure, as the reference does, so it does not find the shift: its
    map only has to be finite and inside the candidate range.)"""
    H, W, shift = 720, 1280, 250
    L, R = _colour_pair(1234, H, W, shift)
    sm = api.stereomatch_1
    fn = {1: sm.computeAdaptiveWeight_GuidedF, 2: sm.computeAdaptiveWeight_GuidedF_2, 3: sm.computeAdaptiveWeight_GuidedF_3}[which]
    got = fn(L, R)
    assert got.dtype == np.float32 and got.shape == (H, W) and np.isfinite(got).all()
    assert got.min() >= 0 and got.max() <= 186 + 143
    inner = got[40:-40, 186 + 144 + 40:-40]
    if which != 3:
        assert (np.abs(inner - shift) <= 1).mean() > 0.9


def test_stereo_matching_dispatches_the_guided_filter_entry(api, orc):
    """stereoMatching(ADAPTIVE_WEIGHT_GUIDED_FILTER) = computeAdaptiveWeight_GuidedF(L, R, DISPARITY_LEFT, 0.01, 15, 10, 150)
    (methods/methods.cpp:1801-1803)."""
    H, W = 48, 260
    L, R = _colour_pair(11, H, W, 40)
    got = api.stereoMatching(L, R, api.ADAPTIVE_WEIGHT_GUIDED_FILTER)
    exp = orc.adaptive_weight_guidedf(L, R, True, 0.01, 15, 10, 150)
    _tie_aware_equal(api, orc, got, L, R, True, 0.01, 15, 10, 150, exp)


def test_guidedf_refuses_what_the_reference_cannot_run(api):
    from cameracalibration_amd._lib import MdispError, E_UNSUPPORTED
    g = np.zeros((20, 30), np.uint8)
    with pytest.raises(MdispError) as e:
        api.stereomatch_1.computeAdaptiveWeight_GuidedF(g, g, api.DISPARITY_LEFT, 0.01, 7, 0, 8)     # 2-channel guide
    assert e.value.code == E_UNSUPPORTED


# ---- computeSimilarity and the matchers that filter other cost volumes ------------------------------------------------

@pytest.mark.parametrize("shape,reg,tc,tg,minD,numD", [((40, 70), 0.4, 10, 50, 0, 12), ((33, 129), 0.25, 3, 20, 2, 9), ((20, 45), 0.4, 300, 5000, 0, 5)])
def test_compute_similarity(api, orc, shape, reg, tc, tg, minD, numD):
    """Per-pixel float arithmetic in the reference's order: the slices equal the restatement bit for bit."""
    H, W = shape
    L, R = _colour_pair(H + 2 * W, H, W, 4)
    slices = ["stale"]
    api.stereomatch_1.computeSimilarity(L, R, slices, reg, tc, tg, api.DISPARITY_LEFT, None, minD, numD)
    exp = orc.compute_similarity(L, R, True, reg, tc, tg, minD, numD)
    assert len(slices) == numD and np.array_equal(np.stack(slices), exp)
    # the ten-argument overload borders every slice with BORDER_REFLECT
    api.stereomatch_1.computeSimilarity(L, R, slices, reg, tc, tg, api.DISPARITY_LEFT, 7, minD, numD)
    assert np.array_equal(np.stack(slices), np.pad(exp, ((0, 0), (3, 3), (3, 3)), mode="symmetric"))
    keep = ["kept"]
    api.stereomatch_1.computeSimilarity(L, R, keep, reg, tc, tg, api.DISPARITY_LEFT, 6, minD, numD)       # even window: returns before anything
    assert keep == ["kept"]
    api.stereomatch_1.computeSimilarity(L, R[:-1], keep, reg, tc, tg, api.DISPARITY_LEFT, None, minD, numD)   # size mismatch: cleared only
    assert keep == []


def test_compute_similarity_refuses_what_fails_in_the_reference(api):
    from cameracalibration_amd._lib import MdispError, E_UNSUPPORTED
    L, R = _colour_pair(9, 20, 40, 3)
    for args in ((L, R, api.DISPARITY_RIGHT), (L[..., 0].copy(), R[..., 0].copy(), api.DISPARITY_LEFT)):
        with pytest.raises(MdispError) as e:
            api.stereomatch_1.computeSimilarity(args[0], args[1], [], 0.4, 10, 50, args[2], None, 0, 8)
        assert e.value.code == E_UNSUPPORTED


def _near_tie_check(got, exp, limit=0.003):
    assert got.dtype == np.float32 and got.shape == exp.shape
    assert float((got != exp).mean()) <= limit, "%d pixels differ" % int((got != exp).sum())


@pytest.mark.parametrize("which,is_left,shape,eps,win,minD,numD", [
    (2, True, (40, 90), 0.01, 7, 0, 12), (2, True, (60, 140), 0.01, 13, 4, 20),
    (3, True, (40, 90), 1e-6, 7, 0, 12), (3, False, (40, 90), 1e-6, 7, 0, 12), (3, True, (36, 100), 1e-4, 15, 2, 9)])
def test_adaptive_weight_guidedf_2_and_3(api, orc, which, is_left, shape, eps, win, minD, numD):
    H, W = shape
    L, R = _colour_pair(H * 5 + W + which, H, W, minD + numD // 2)
    f = api.stereomatch_1.computeAdaptiveWeight_GuidedF_2 if which == 2 else api.stereomatch_1.computeAdaptiveWeight_GuidedF_3
    got = f(L, R, api.DISPARITY_LEFT if is_left else api.DISPARITY_RIGHT, eps, win, minD, numD)
    exp = orc.adaptive_weight_guidedf_23(L, R, which, is_left, eps, win, minD, numD)
    # float winner-take-all over filtered slices: only near-ties may differ (eps = 1e-6 amplifies last-bit differences of the
    # variance, so the third matcher gets a wider share)
    _near_tie_check(got, exp, 0.003 if which == 2 else 0.02)
    if which == 2:
        assert (got[:, minD + numD:] == minD + numD // 2).mean() > 0.7


def test_stereo_matching_dispatches_guided_filter_2_and_3(api, orc):
    H, W = 40, 240
    L, R = _colour_pair(21, H, W, 30)
    _near_tie_check(api.stereoMatching(L, R, api.ADAPTIVE_WEIGHT_GUIDED_FILTER_2), orc.adaptive_weight_guidedf_23(L, R, 2, True, 0.01, 13, 10, 150))
    _near_tie_check(api.stereoMatching(L, R, api.ADAPTIVE_WEIGHT_GUIDED_FILTER_3), orc.adaptive_weight_guidedf_23(L, R, 3, True, 1e-6, 15, 0, 64), 0.02)
    from cameracalibration_amd._lib import MdispError
    with pytest.raises(MdispError):
        api.stereomatch_1.computeAdaptiveWeight_GuidedF_2(L, R, api.DISPARITY_RIGHT, 0.01, 7, 0, 8)
