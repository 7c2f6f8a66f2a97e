"""GPU parity of the float paths (guided-filter family, steer filter, guided / asw aggregation)
against the CPU oracle.  Tolerances: filtered slices max-abs <= 1e-3 on the [0,1]-normalised cost;
float WTA may differ from the oracle only where the oracle's best-vs-runner-up gap is < 1e-5."""
import numpy as np
import pytest

from conftest import rand_pair

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _textured(seed, H, W):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (H, W)).astype(np.float64)
    from scipy import ndimage
    a = ndimage.uniform_filter(a, 3, mode="mirror")
    a = (a - a.min()) / (a.max() - a.min()) * 255
    return a.astype(np.uint8)


def _tie_aware_equal(got, exp, costs, minD, gap=1e-5):
    """disparities equal, or the oracle's two best filtered costs are within `gap` there"""
    bad = got != exp
    if not bad.any():
        return True
    ys, xs = np.nonzero(bad)
    for y, x in zip(ys, xs):
        c = costs[:, y, x]
        if abs(c[int(got[y, x]) - minD] - c[int(exp[y, x]) - minD]) >= gap:
            return False
    return True


@pytest.mark.parametrize("shape,r", [((48, 64), 7), ((37, 53), 5), ((90, 140), 15), ((60, 80), 35)])
@pytest.mark.parametrize("dtype", [np.uint8, np.int32, np.float32])
def test_guided_filter_gif(api, orc, shape, r, dtype):
    guide = _textured(1, *shape)
    rng = np.random.default_rng(2)
    P = rng.integers(0, 9, shape).astype(dtype)
    if dtype == np.float32:
        P = P + rng.random(shape, dtype=np.float32)
    for eps in (1e-2, 1e-6):
        got = api.getGuidedFilter(guide, P, r, eps)
        exp = orc.guided_filter(guide, P, r, eps, orc.GIF)
        assert np.abs(got - exp).max() <= TOL


@pytest.mark.parametrize("shape,r", [((48, 64), 7), ((37, 53), 9)])
def test_guided_filter_wgif_egif(api, orc, shape, r):
    guide = _textured(3, *shape)
    rng = np.random.default_rng(4)
    P = rng.integers(0, 256, shape, dtype=np.uint8)
    got = api.getGuidedFilter_wgif(guide, P, r, 0.01)
    assert np.abs(got - orc.guided_filter(guide, P, r, 0.01, orc.WGIF)).max() <= TOL
    got, beta = api.getGuidedFilter_egif(guide, P, r, 0.01)
    exp, ebeta = orc.guided_filter(guide, P, r, 0.01, orc.EGIF, want_beta=True)
    assert np.abs(got - exp).max() <= TOL
    assert np.abs(beta - ebeta).max() <= TOL * max(1.0, np.abs(ebeta).max())


@pytest.mark.parametrize("shape", [(48, 64), (37, 53), (20, 150)])
def test_steer_filter(api, orc, shape):
    img = _textured(5, *shape)
    got = api.steerFilter(img)
    exp = orc.steer_filter(img)
    assert got.dtype == np.float64
    assert np.abs(got - exp).max() <= 1e-9 * 120


@pytest.mark.parametrize("shape,r,r2", [((48, 64), 7, 5), ((40, 56), 9, 15)])
def test_guided_filter_skwgif2(api, orc, shape, r, r2):
    guide = _textured(6, *shape)
    rng = np.random.default_rng(7)
    P = rng.integers(0, 9, shape, dtype=np.uint8)
    sw = orc.steer_filter(guide)
    got = api.getGuidedFilter_skwgif2(guide, P, sw, r, r2, 0.01, 0.1)
    exp = orc.guided_filter(guide, P, r, 0.01, orc.SKWGIF, steerW=sw, r2=r2, namuda=0.1)
    assert np.abs(got - exp).max() <= TOL


@pytest.mark.parametrize("shape,r,r2", [((48, 64), 7, 5), ((40, 56), 9, 15), ((70, 45), 5, 7), ((30, 100), 15, 3)])
def test_guided_filter_ours(api, orc, shape, r, r2):
    # bilateral-smoothed coefficients (gifs.cpp:620-804); the reference's main() runs OURS_GIF3
    guide = _textured(8, *shape)
    rng = np.random.default_rng(9)
    P = rng.integers(0, 9, shape, dtype=np.uint8)
    for eps in (1e-2, 1e-4):
        got = api.getGuidedFilter_ours_gif(guide, P, r, r2, eps, 0.01)
        assert np.abs(got - orc.guided_filter(guide, P, r, eps, orc.OURS_GIF, r2=r2, namuda=0.01)).max() <= TOL
        got = api.getGuidedFilter_ours_gif2(guide, P, r, r2, eps, 0.01)
        assert np.abs(got - orc.guided_filter(guide, P, r, eps, orc.OURS_GIF2, r2=r2, namuda=0.01)).max() <= TOL
        got, beta = api.getGuidedFilter_ours_gif3(guide, P, r, r2, eps, 0.01)
        exp, ebeta = orc.guided_filter(guide, P, r, eps, orc.OURS_GIF3, r2=r2, namuda=0.01, want_beta=True)
        assert np.abs(got - exp).max() <= TOL
        ok = np.abs(ebeta) < 1e3                      # a_bil close to 1 makes beta = a/(1-a) ill-conditioned
        assert np.abs(beta - ebeta)[ok].max() <= TOL * max(1.0, np.abs(ebeta[ok]).max())


def test_guided_filter_ours_flat_slice(api, orc):
    # a constant slice gives a = 0 everywhere: cv::bilateralFilter copies a flat image
    guide = _textured(10, 40, 50)
    P = np.full((40, 50), 3, np.uint8)
    for fn, t in ((api.getGuidedFilter_ours_gif, orc.OURS_GIF), (api.getGuidedFilter_ours_gif2, orc.OURS_GIF2)):
        got = fn(guide, P, 7, 5, 0.01, 0.01)
        assert np.abs(got - orc.guided_filter(guide, P, 7, 0.01, t, r2=5, namuda=0.01)).max() <= TOL


@pytest.mark.parametrize("left", [True, False])
def test_census_stereo_guided_is_the_reference_default(api, orc, left):
    L, R, _ = api.synth_pair(0x5EED0001, 96, 160, 32)
    p = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=7, minDisparity=0, maxDisparity=31, isDispLeft=left)
    got = api.censusStereo(p, api.BASIC_CENSUS)          # agg defaults to AGG_GUIDED
    exp, costs = orc.census_stereo(L, R, 7, 0, 31, left, orc.BASIC_CENSUS, orc.AGG_GUIDED, want_costs=True)
    assert got.dtype == np.float64
    assert (got == exp).mean() > 0.999
    assert _tie_aware_equal(got, exp, costs, 0)


@pytest.mark.parametrize("gif", [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("source", ["census", "absdiff", "volume"])
def test_asw_gifs(api, orc, gif, source):
    H, W, minD, maxD, win = 64, 96, 2, 17, 7
    L, R, _ = api.synth_pair(0x5EED0003, H, W, 16)
    D = maxD - minD + 1
    pad = orc.pad_reflect_lr(R, maxD, 0)
    if source == "census":
        cL, cP = orc.census(L), orc.census(pad)
        costs = np.stack([orc.hamming(cL, np.ascontiguousarray(cP[:, maxD - d:maxD - d + W])) for d in range(minD, maxD + 1)])
        src = api.COST_CENSUS
    else:
        costs = np.stack([np.abs(L.astype(np.int32) - pad[:, maxD - d:maxD - d + W].astype(np.int32))
                          for d in range(minD, maxD + 1)])
        src = api.COST_ABSDIFF
    costs = costs.astype(np.float32)
    p = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=win, minDisparity=minD, maxDisparity=maxD, isDispLeft=True)
    if source == "volume":
        got = api.asw_gifs(p, 0.01, gif, 5, 0.1, 0.5, api.COST_VOLUME, costs)
    else:
        got = api.asw_gifs(p, 0.01, gif, 5, 0.1, 0.5, src)
    exp, filt = orc.asw_gifs(L, costs, minD, win, 0.01, gif, 5, 0.1, want_filtered=True)
    assert got.dtype == np.float32 and got.shape == (H, W)
    assert (got == exp).mean() > 0.999
    assert _tie_aware_equal(got, exp, filt, minD)
    assert D == filt.shape[0]


@pytest.mark.gpu
@pytest.mark.parametrize("gif,win,r2", [(0, 7, 5), (3, 9, 15), (2, 11, 5), (3, 35, 15), (4, 7, 15), (6, 9, 15)])
def test_row_bands_equal_the_full_image(api, gif, win, r2):
    # the row-band entry point (one GPU's share of a large pair) reproduces the rows of a full call bit for bit
    import torch
    H, W, D = 97, 150, 24
    L, R, _ = api.synth_pair(0x5EED0001, H, W, D)
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    pitch = W
    if pitch % 4:
        pitch = (W + 3) // 4 * 4
        dl = torch.nn.functional.pad(dl, (0, pitch - W)).contiguous()
        dr = torch.nn.functional.pad(dr, (0, pitch - W)).contiguous()

    def run(bands):
        out = torch.full((H, W), -1.0, dtype=torch.float32, device="cuda")
        for (b, e) in bands:
            api.asw_gifs_band_dev(dl.data_ptr(), dr.data_ptr(), H, W, pitch, win, 0, D - 1, True, 0.01, gif, r2, 0.1, b, e, out.data_ptr())
        torch.cuda.synchronize()
        return out.cpu().numpy()

    full = run([(0, H)])
    assert full.min() >= 0
    p = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=win, minDisparity=0, maxDisparity=D - 1, isDispLeft=True)
    assert np.array_equal(full, api.asw_gifs(p, 0.01, gif, r2, 0.1, 0.5, api.COST_CENSUS))
    for bands in ([(0, 48), (48, 97)], [(0, 3), (3, 50), (50, 96), (96, 97)], [(0, 33), (33, 65), (65, 97)]):
        assert np.array_equal(run(bands), full)


@pytest.mark.gpu
@pytest.mark.parametrize("gif,win,r2", [(0, 7, 5), (3, 9, 15), (2, 11, 5), (3, 35, 15), (6, 9, 15)])
def test_row_bands_with_exchanged_statistics(api, gif, win, r2):
    # each band computes its share of the image-wide statistics (per-slice cost range, steer maximum), the shares are
    # merged (what the all-reduce of the multi-GPU run does), and the bands still reproduce the full image bit for bit
    import torch
    from cameracalibration_amd import shard
    H, W, D = 97, 152, 24
    L, R, _ = api.synth_pair(0x5EED0021, H, W, D)
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    p = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=win, minDisparity=0, maxDisparity=D - 1, isDispLeft=True)
    full = api.asw_gifs(p, 0.01, gif, r2, 0.1, 0.5, api.COST_CENSUS)
    for bands in ([(0, 48), (48, 97)], [(0, 3), (3, 50), (50, 96), (96, 97)], [(0, 97), (97, 97)]):
        parts = [api.asw_gifs_band_stats_dev(dl.data_ptr(), dr.data_ptr(), H, W, W, win, 0, D - 1, True, gif, b, e) for (b, e) in bands]
        mm = np.stack([np.min([q[0][:, 0] for q in parts], axis=0), np.max([q[0][:, 1] for q in parts], axis=0)], axis=1)
        sm = max(q[1] for q in parts)
        one = shard.merge_band_stats(parts[0][0], parts[0][1])          # single process: unchanged
        assert np.array_equal(one[0], parts[0][0]) and one[1] == parts[0][1]
        out = torch.full((H, W), -1.0, dtype=torch.float32, device="cuda")
        for (b, e) in bands:
            api.asw_gifs_band_dev(dl.data_ptr(), dr.data_ptr(), H, W, W, win, 0, D - 1, True, 0.01, gif, r2, 0.1, b, e, out.data_ptr(),
                                  stats=(mm, sm))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), full)


# ---- the large-pair configuration's own parameters: SKWGIF, winSize 35, r2 15, eps 0.01, 512 disparities ----

@pytest.mark.parametrize("shape", [(60, 80), (96, 160), (41, 203)])
def test_guided_filter_skwgif2_wide_windows(api, orc, shape):
    # the wide-window marching kernel (row re-fetch + prefix sums) under the steer-weighted variant
    guide = _textured(11, *shape)
    rng = np.random.default_rng(12)
    sw = orc.steer_filter(guide)
    for P in (rng.integers(0, 9, shape, dtype=np.uint8), rng.integers(0, 256, shape, dtype=np.uint8)):
        got = api.getGuidedFilter_skwgif2(guide, P, sw, 35, 15, 0.01, 0.1)
        exp = orc.guided_filter(guide, P, 35, 0.01, orc.SKWGIF, steerW=sw, r2=15, namuda=0.1)
        assert np.abs(got - exp).max() <= TOL


def test_asw_gifs_large_pair_parameters_against_oracle(api, orc):
    # asw_gifs(param{0..511, win 35}, 0.01, SKWGIF, r2 15) at a size the oracle affords: 512 slices of 96x160
    # (the REFLECT pad of the right image makes any width legal, methods_disp.cpp:1005)
    H, W, minD, maxD, win, r2 = 96, 160, 0, 511, 35, 15
    L, R, _ = api.synth_pair(0x5EED0005, H, W, 64)
    pad = orc.pad_reflect_lr(R, maxD, 0)
    cL, cP = orc.census(L), orc.census(pad)
    costs = np.stack([orc.hamming(cL, np.ascontiguousarray(cP[:, maxD - d:maxD - d + W])) for d in range(minD, maxD + 1)]).astype(np.float32)
    p = api.StereoMatchParam(imgLeft=L, imgRight=R, winSize=win, minDisparity=minD, maxDisparity=maxD, isDispLeft=True)
    got = api.asw_gifs(p, 0.01, api.SKWGIF, r2, 0.1, 0.5, api.COST_CENSUS)
    exp, filt = orc.asw_gifs(L, costs, minD, win, 0.01, orc.SKWGIF, r2, 0.1, want_filtered=True)
    assert filt.shape[0] == 512
    assert got.min() >= minD and got.max() <= maxD
    assert (got == exp).mean() > 0.999
    assert _tie_aware_equal(got, exp, filt, minD)


def test_large_pair_row_band_at_full_size(api):
    # 7680x4320, D = 512, SKWGIF, winSize 35, r2 15: one GPU's band of a row-band split, checked by size-independent
    # properties -- its rows equal the rows of the full-image call bit for bit, every value is a candidate disparity,
    # and a finer split of the same rows gives the same map
    import torch
    H, W, D, win, r2 = 4320, 7680, 512, 35, 15
    L, R, _ = api.synth_pair(0x5EED0001, H, W, D)
    dl, dr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()

    def run(bands):
        out = torch.full((H, W), -1.0, dtype=torch.float32, device="cuda")
        for (b, e) in bands:
            api.asw_gifs_band_dev(dl.data_ptr(), dr.data_ptr(), H, W, W, win, 0, D - 1, True, 0.01, api.SKWGIF, r2, 0.1, b, e, out.data_ptr())
        torch.cuda.synchronize()
        return out.cpu().numpy()

    b0, b1 = 3 * (H // 8), 4 * (H // 8)              # band 3 of 8
    full = run([(0, H)])
    assert full.min() >= 0 and full.max() <= D - 1
    band = run([(b0, b1)])
    assert (band[:b0] == -1).all() and (band[b1:] == -1).all()
    assert np.array_equal(band[b0:b1], full[b0:b1])
    split = run([(b0, b0 + 100), (b0 + 100, b0 + 101), (b0 + 101, b1)])
    assert (split[b0:b1] == band[b0:b1]).mean() >= 0.999
    assert np.array_equal(split[b0:b1], band[b0:b1])
