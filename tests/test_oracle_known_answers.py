"""Pins for the CPU oracle: hand-derivable known answers from the reference source text and
independent numpy / scipy re-derivations of the OpenCV primitives it restates.

The reference ships no tests or golden vectors, so these are the strongest pins available for the
reference-owned arithmetic (census, Hamming, tables, WTA rule)."""
import numpy as np
import pytest
from scipy import ndimage

from conftest import rand_pair


def test_census_constant_image_is_ff_with_zero_frame(orc):
    # `nbr >= centre` on a constant image sets all 8 bits; the 1-px frame stays 0
    # (methods_disp.cpp:428-443)
    img = np.full((9, 11), 77, np.uint8)
    c = orc.census(img)
    assert (c[1:-1, 1:-1] == 0xFF).all()
    assert not c[0].any() and not c[-1].any() and not c[:, 0].any() and not c[:, -1].any()


def test_census_x_ramp_is_7c(orc):
    # strictly increasing in x, constant in y: T,TR,R,BR,B set (equal counts as set), TL,BL,L clear
    img = np.tile(np.arange(20, dtype=np.uint8) * 3, (7, 1))
    c = orc.census(img)
    assert (c[1:-1, 1:-1] == 0x7C).all()


def test_census_bit_order(orc):
    # a single bright pixel at each neighbour position sets exactly the documented bit
    order = {(-1, -1): 7, (-1, 0): 6, (-1, 1): 5, (0, 1): 4, (1, 1): 3, (1, 0): 2, (1, -1): 1, (0, -1): 0}
    for (dy, dx), bit in order.items():
        img = np.zeros((5, 5), np.uint8)
        img[2, 2] = 10
        img[2 + dy, 2 + dx] = 20
        assert orc.census(img)[2, 2] == 1 << bit


def test_rotation_invariant_table_is_shift(orc):
    assert np.array_equal(orc.ri_table(), np.arange(256) >> 1)


def test_uniform_table(orc):
    t = orc.uniform_table()
    # 58 codes have <= 2 transitions in hopCount's circular pattern (a[0] = 0 followed by the low 7 bits)
    assert np.count_nonzero(t) == 58
    assert t[0] == 1 and t[255] == 58


def test_uniform_table_matches_python_restatement(orc):
    def hop(i):
        a = [0] * 8
        k = 7
        while k:
            a[k] = i & 1
            i >>= 1
            k -= 1
        return sum(a[j] != a[(j + 1) % 8] for j in range(8))
    exp = np.zeros(256, np.uint8)
    temp = 1
    for i in range(256):
        if hop(i) <= 2:
            exp[i] = temp & 0xFF
            temp += 1
    assert np.array_equal(orc.uniform_table(), exp)


def test_hamming_is_popcount(orc):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (13, 17), dtype=np.uint8)
    b = rng.integers(0, 256, (13, 17), dtype=np.uint8)
    exp = np.unpackbits((a ^ b)[..., None], axis=-1).sum(-1)
    assert np.array_equal(orc.hamming(a, b), exp)


def test_reflect_padding_matches_numpy_symmetric(orc):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (6, 10), dtype=np.uint8)
    assert np.array_equal(orc.pad_reflect_lr(img, 7, 0), np.pad(img, ((0, 0), (7, 0)), mode="symmetric"))
    assert np.array_equal(orc.pad_reflect_lr(img, 0, 23), np.pad(img, ((0, 0), (0, 23)), mode="symmetric"))


def test_identical_pair_wta_is_min_disparity(orc):
    # identical L/R: slice d=0 is zero and first-min keeps minDisparity -- except at column 0, where
    # the left census is the zero frame while the right census is taken on the PADDED image
    # (methods_disp.cpp:862,875) and so holds a real code; box windows spread that by win/2
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (24, 40), dtype=np.uint8)
    for agg, margin in ((orc.AGG_NONE, 1), (orc.AGG_BOX, 3), (orc.AGG_SQRBOX, 3)):
        d = orc.census_stereo(img, img, 5, 0, 7, True, orc.BASIC_CENSUS, agg)
        assert (d[:, margin:] == 0).all()


def test_bad_parameters_return_empty(orc):
    img = np.zeros((8, 8), np.uint8)
    assert orc.census_stereo(img, img, 4, 0, 3) is None      # even window
    assert orc.sad_box(img, img, 6, 0, 3) is None
    assert orc.sad_inte(img, img, 2, 0, 3) is None


def test_box_filter_matches_scipy_mirror(orc):
    rng = np.random.default_rng(4)
    src = rng.random((31, 45), dtype=np.float32)
    for k in (3, 7, 15):
        exp = ndimage.uniform_filter(src.astype(np.float64), size=k, mode="mirror")
        got = orc.box_filter_f32(src, k, k, True)
        assert np.abs(got - exp).max() < 1e-6


def test_sad_box_matches_numpy(orc):
    L, R = rand_pair(5, 30, 50, 12)
    win, minD, maxD = 5, 0, 11
    pad = np.pad(R, ((0, 0), (maxD, 0)), mode="symmetric").astype(np.int64)
    costs = []
    for d in range(minD, maxD + 1):
        diff = np.abs(L.astype(np.int64) - pad[:, maxD - d:maxD - d + L.shape[1]])
        s = ndimage.uniform_filter((diff * diff).astype(np.float64), size=win, mode="mirror") * win * win
        costs.append(np.rint(s).astype(np.int64).astype(np.float32))
    exp = (np.argmin(np.stack(costs), axis=0) + minD).astype(np.uint8)
    assert np.array_equal(orc.sad_box(L, R, win, minD, maxD, True), exp)


def test_sad_inte_int_equals_f32_on_small_images(orc):
    # every integral entry < 2^24 here, so the float table is exact and both modes agree
    L, R = rand_pair(6, 40, 60, 10)
    a = orc.sad_inte(L, R, 7, 0, 9, True, orc.SAT_INT)
    b = orc.sad_inte(L, R, 7, 0, 9, True, orc.SAT_F32)
    assert np.array_equal(a, b)


def test_sad_inte_footprint_is_2h_by_2h(orc):
    # interior pixels: window rows y-h..y+h-1, cols x-h..x+h-1 (methods_disp.cpp:253-256)
    L, R = rand_pair(7, 32, 48, 8)
    win, h, maxD = 5, 2, 7
    pad = np.pad(R, ((0, 0), (maxD, 0)), mode="symmetric").astype(np.int64)
    H, W = L.shape
    costs = np.zeros((maxD + 1, H, W), np.int64)
    for d in range(maxD + 1):
        diff = np.abs(L.astype(np.int64) - pad[:, maxD - d:maxD - d + W])
        I = np.zeros((H + 1, W + 1), np.int64)
        I[1:, 1:] = diff.cumsum(0).cumsum(1)

        def g(x, y):
            return I[y, x] if 0 <= x <= W and 0 <= y <= H else 0
        for y in range(H):
            for x in range(W):
                costs[d, y, x] = g(x + h, y + h) - g(x + h, y - h) - g(x - h, y + h) + g(x - h, y - h)
    exp = np.argmin(costs, axis=0).astype(np.uint8)
    assert np.array_equal(orc.sad_inte(L, R, win, 0, maxD, True, orc.SAT_INT), exp)


def test_sgm_single_path_recursion_matches_python(orc):
    # tiny image, 4 paths: compare the aggregated volume with a direct transcription of the recursion
    L, R = rand_pair(8, 6, 9, 5)
    D, P1, P2 = 5, 3, 11
    _, S = orc.sgm(L, R, 0, D, P1, P2, paths=4, want_S=True)
    maxD = D - 1
    cL = orc.census(L)
    cR = orc.census(orc.pad_reflect_lr(R, maxD, 0))
    H, W = L.shape
    Cv = np.zeros((H, W, D), np.int64)
    for d in range(D):
        Cv[:, :, d] = orc.hamming(cL, np.ascontiguousarray(cR[:, maxD - d:maxD - d + W]))
    exp = np.zeros((H, W, D), np.int64)
    for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
        Lr = np.zeros((H, W, D), np.int64)
        ys = range(H) if dy >= 0 else range(H - 1, -1, -1)
        xs = range(W) if dx >= 0 else range(W - 1, -1, -1)
        for y in ys:
            for x in xs:
                px, py = x - dx, y - dy
                if 0 <= px < W and 0 <= py < H:
                    prev = Lr[py, px]
                    mp = prev.min()
                    for d in range(D):
                        c = [prev[d], mp + P2]
                        if d > 0:
                            c.append(prev[d - 1] + P1)
                        if d < D - 1:
                            c.append(prev[d + 1] + P1)
                        Lr[y, x, d] = Cv[y, x, d] + min(c) - mp
                else:
                    Lr[y, x] = Cv[y, x]
        exp += Lr
    assert np.array_equal(S.astype(np.int64), exp)


def test_sgm_identical_pair_gives_zero_disparity(orc):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (20, 36), dtype=np.uint8)
    d = orc.sgm(img, img, 0, 8, 8, 32, uniq=0, paths=8)
    inv = (0 - 1) * 16
    # column 0 compares the zero census frame with a real code of the padded right image
    assert set(np.unique(d[:, 1:]).tolist()) <= {0, inv}
    assert (d == 0).mean() > 0.9


def test_remap_identity_and_half_pixel(orc):
    rng = np.random.default_rng(10)
    img = rng.integers(0, 256, (12, 16), dtype=np.uint8)
    xs, ys = np.meshgrid(np.arange(16), np.arange(12))
    m1 = np.stack([xs, ys], -1).astype(np.int16)
    assert np.array_equal(orc.remap(img, m1, np.zeros((12, 16), np.uint16)), img)
    # fx = 16/32: average of horizontal neighbours, rounded half up; the last column blends with 0
    out = orc.remap(img, m1, np.full((12, 16), 16, np.uint16))
    nxt = np.concatenate([img[:, 1:], np.zeros((12, 1), np.uint8)], 1).astype(np.int32)
    exp = (img.astype(np.int32) * 16384 + nxt * 16384 + 16384) >> 15
    assert np.array_equal(out, exp.astype(np.uint8))


def test_guided_filter_constant_slice_is_constant(orc):
    rng = np.random.default_rng(11)
    guide = rng.integers(0, 256, (24, 32), dtype=np.uint8)
    P = np.full((24, 32), 5, np.uint8)       # zero range -> normalises to all zeros
    out = orc.guided_filter(guide, P, 7, 1e-6)
    assert np.abs(out).max() < 1e-6


def test_guided_filter_matches_float64_numpy(orc):
    rng = np.random.default_rng(12)
    guide = rng.integers(0, 256, (40, 56), dtype=np.uint8)
    P = rng.integers(0, 9, (40, 56), dtype=np.uint8)
    r, eps = 7, 1e-2
    G = (guide.astype(np.float64) - guide.min()) / (int(guide.max()) - int(guide.min()))
    Pn = (P.astype(np.float64) - P.min()) / (int(P.max()) - int(P.min()))
    box = lambda a: ndimage.uniform_filter(a, size=r, mode="mirror")
    mG, mP = box(G), box(Pn)
    a = (box(G * Pn) - mG * mP) / (box(G * G) - mG * mG + eps)
    b = mP - a * mG
    exp = box(a) * G + box(b)
    got = orc.guided_filter(guide, P, r, eps)
    assert np.abs(got - exp).max() < 1e-4


def test_steer_filter_range_and_flat_region(orc):
    rng = np.random.default_rng(13)
    img = rng.integers(0, 256, (30, 34), dtype=np.uint8)
    s = orc.steer_filter(img)
    assert s.shape == img.shape and s.min() >= 1e-6 and s.max() <= 120
    # a vertical step edge has a much stronger orientation response than the flat areas beside it
    step = np.zeros((40, 40), np.uint8)
    step[:, 20:] = 200
    e = orc.steer_filter(step)
    assert e[20, 19:21].max() > 50 * e[20, 5]


def test_lr_check_simple(orc):
    dL = np.full((4, 10), 2, np.float32)
    dR = np.full((4, 10), 2, np.float32)
    vl, vr = orc.lr_check(dL, dR)
    assert vl.all() and vr.all()
    dR[:, 3] = 7
    vl, _ = orc.lr_check(dL, dR)
    assert not vl[:, 5].any() and vl[:, 6].all()


def test_reprojection_known_answers(orc):
    # one pixel by hand: Q * (row, col, d, 1) with the row index first, then / W / 100
    Q = np.array([[1, 0, 0, -10.0], [0, 1, 0, -20.0], [0, 0, 0, 800.0], [0, 0, 0.01, 0.0]])
    disp = np.zeros((4, 5), np.float32)
    disp[2, 3] = 8.0
    xyz = orc.reproject_q(disp, Q)
    w = 0.01 * 8.0
    assert np.allclose(xyz[2, 3], [(2 - 10.0) / w / 100, (3 - 20.0) / w / 100, 800.0 / w / 100], rtol=1e-6)
    assert not xyz[0, 0].any()
    # triangulation on the optical axis: Z = baseline * f0 / d
    K = np.array([[500.0, 0, 3.0], [0, 500.0, 2.0], [0, 0, 1.0]])
    p = orc.reproject_triangulate(disp, K, 100.0)[2, 3]
    assert np.allclose(p, [0.0, 0.0, 100.0 * 500.0 / 8.0], rtol=1e-6)


def test_bilateral_restatement_against_direct_formula(orc):
    # cv::bilateralFilter restated with its 4096-bin colour table: within 1e-5 of the closed-form
    # weights exp(-r^2/2ss^2) * exp(-dv^2/2sc^2) over the circular footprint; flat images are copied
    rng = np.random.default_rng(0)
    a = (rng.random((20, 30)) * 3).astype(np.float32)
    got = orc.bilateral_filter(a, 5, 2, 3)
    H, W = a.shape
    R = 2
    pad = np.pad(a, R, mode="reflect")
    exp = np.zeros_like(a)
    for y in range(H):
        for x in range(W):
            s = ws = 0.0
            for i in range(-R, R + 1):
                for j in range(-R, R + 1):
                    if (i * i + j * j) ** 0.5 > R:
                        continue
                    v = float(pad[y + R + i, x + R + j])
                    w = np.exp(-(i * i + j * j) / 18.0) * np.exp(-((v - float(a[y, x])) ** 2) / 8.0)
                    s += v * w
                    ws += w
            exp[y, x] = s / ws
    assert np.abs(got - exp).max() < 1e-5
    flat = np.full((9, 11), 0.25, np.float32)
    assert np.array_equal(orc.bilateral_filter(flat, 7, 2, 3), flat)


def test_stereosgbm_restatement_known_answers(orc):
    # identical images: every block cost at d = 0 is zero, so the winner is 0 wherever the range is computed;
    # columns left of minDisparity + numDisparities are invalid ((minD - 1) * 16)
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (24, 80)).astype(np.uint8)
    d, cost = orc.sgbm_bt(img, img, 0, 16, 5, want_cost=True)
    assert (cost[:, :, 0] == 0).all()
    assert (d[:, :16] == -16).all()
    assert (d[:, 16:] == 0).all()
    # a pure shift by 5 columns is found exactly away from the borders
    R = rng.integers(0, 256, (24, 120)).astype(np.uint8)
    L = np.roll(R, 5, axis=1)
    d = orc.sgbm_bt(L, R, 0, 16, 5)
    assert (np.abs(d[2:-2, 24:-8] - 5 * 16) < 8).all()           # within the sub-pixel parabola's half step
    # a range wider than the image leaves nothing to compute
    assert (orc.sgbm_bt(img[:, :30], img[:, :30], 0, 32, 5) == -16).all()


def test_median_and_speckle_restatements(orc):
    a = np.arange(30, dtype=np.int16).reshape(5, 6)
    m = orc.median3_s16(a)
    assert m[2, 2] == a[2, 2] and m[0, 0] == a[0, 1]          # replicated borders: {0,0,1,0,0,1,6,6,7} -> 1
    img = np.full((8, 8), 160, np.int16)
    img[2:4, 2:4] = 800                                        # a 4-pixel island
    img[6, 6] = -16                                            # already invalid
    out = orc.filter_speckles_s16(img, -16, 4, 16)
    assert (out[2:4, 2:4] == -16).all() and out[0, 0] == 160 and out[6, 6] == -16
    assert np.array_equal(orc.filter_speckles_s16(img, -16, 3, 16), img)
    assert (orc.filter_speckles_s16(img, -16, 4, 1000) == img).all()    # one big region when everything connects


def test_legacy_matchers_known_answers(orc):
    # a pure shift by 5 columns: every window matcher's first zero-cost candidate is index 5 -> (char)(16 * 5) inside
    # the evaluated region; the frame the reference never writes stays 0
    rng = np.random.default_rng(11)
    R = rng.integers(0, 256, (30, 90)).astype(np.uint8)
    L = np.roll(R, 5, axis=1)
    for m in (orc.LEGACY_SAD, orc.LEGACY_SAD_INTE, orc.LEGACY_SSD, orc.LEGACY_SSD_INTE, orc.LEGACY_SSD_BF, orc.LEGACY_STAD_BF):
        d = orc.legacy_disparity(L, R, m, 7, 30, 0, 16)
        assert (d[3:-3, 24:-3] == 80).all(), m
        assert (d[:3] == 0).all() and (d[:, :3] == 0).all()
    assert (orc.legacy_disparity(L, R, orc.LEGACY_SAD_BF, 7, 30, 0, 16)[4:-4, 24:-4] == 80).all()
    # the T variants clamp the winning index, not the cost
    assert (orc.legacy_disparity(L, R, orc.LEGACY_STAD, 7, 3, 0, 16)[3:-3, 24:-3] == 48).all()
    # AD reads the left pixel through at<char>: values above 127 count as 0, and candidate k is column i - numDisparity + k
    Lc = np.full((4, 40), 200, np.uint8)
    Rc = np.tile(np.arange(40, dtype=np.uint8), (4, 1))
    d = orc.legacy_disparity(Lc, Rc, orc.LEGACY_AD, 1, 0, 0, 8)
    assert (d[:, 8:] == 0).all()               # |0 - R| is smallest at the first candidate (smallest column)
    assert orc.legacy_disparity(L, R, orc.LEGACY_SAD, 6, 30, 0, 16) is None      # even window


def test_stereobm_oracle_known_answers(orc):
    # prefilter: a constant image has no x-gradient -> `cap` everywhere; a steep ramp saturates at 2*cap inside the frame
    flat = np.full((6, 9), 77, np.uint8)
    assert (orc.bm_prefilter_xsobel(flat, 31) == 31).all()
    ramp = np.tile((np.arange(9) * 20).astype(np.uint8), (6, 1))
    pf = orc.bm_prefilter_xsobel(ramp, 31)
    assert (pf[:, 1:-1] == 62).all() and (pf[:, 0] == 31).all() and (pf[:, -1] == 31).all()
    odd = orc.bm_prefilter_xsobel(np.tile((np.arange(9) * 20).astype(np.uint8), (5, 1)), 31)
    assert (odd[4] == 31).all() and (odd[:4, 1:-1] == 62).all()          # odd height: the unpaired last row stays flat
    # a pair that differs by a pure shift: every pixel of the valid rectangle reports that shift; the window sum is 0
    # there, so the sub-pixel term (p - n) * 256 / (2 max(p, n)) stays within half a pixel
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (40, 160)).astype(np.uint8)
    R = base[:, 20:140]
    L = base[:, 20 - 9:140 - 9]                 # L(x) = R(x - 9)
    d = orc.stereobm(L, R, 16, 7, uniquenessRatio=0, textureThreshold=0)
    FILTERED = -16
    inner = d[3:-3, 15 + 3:-3]
    assert (d[:3] == FILTERED).all() and (d[:, :15 + 3] == FILTERED).all() and (d[:, -3:] == FILTERED).all()
    assert (inner != FILTERED).all()
    assert (np.abs(inner - 9 * 16) <= 8).all()
    # arguments the library refuses
    for nd, bs in ((20, 7), (16, 6), (16, 3)):
        with pytest.raises(ValueError):
            orc.stereobm(L, R, nd, bs)


def test_fgs_oracle_known_answers(orc):
    from scipy.linalg import solve_banded
    rng = np.random.default_rng(11)
    g = rng.integers(0, 256, (24, 37), dtype=np.uint8)
    f = (rng.random((24, 37)) * 90).astype(np.float32)
    # lambda = 0: the system is the identity
    assert np.array_equal(orc.fgs_filter(g, f, 0.0, 0.5), f)
    # a constant map is a fixed point, and every solve preserves the sum (the smoothness term has zero row sums)
    c = orc.fgs_filter(g, np.full(g.shape, 7.0, np.float32), 8000.0, 5.0)
    assert np.abs(c - 7.0).max() < 1e-2
    o = orc.fgs_filter(g, f, 8000.0, 5.0)
    assert abs(float(o.astype(np.float64).sum()) - float(f.astype(np.float64).sum())) < 1e-3 * f.size
    assert o.std() < f.std()
    # a single row, one iteration: the tridiagonal system solved independently in float64
    row = rng.integers(0, 256, (1, 50), dtype=np.uint8)
    fr = (rng.random((1, 50)) * 40).astype(np.float32)
    lam, sig = 30.0, 20.0
    w = np.exp(-np.abs(np.diff(row[0].astype(np.float64))) / sig)           # weight between x and x+1
    n = row.shape[1]
    ab = np.zeros((3, n))
    ab[0, 1:] = -lam * w
    ab[2, :-1] = -lam * w
    ab[1] = 1.0
    ab[1, :-1] += lam * w
    ab[1, 1:] += lam * w
    exp = solve_banded((1, 1), ab, fr[0].astype(np.float64))
    got = orc.fgs_filter(row, fr, lam, sig, 0.25, 1)
    assert np.abs(got[0] - exp).max() < 1e-3
    # three channels: the colour distance is summed over the channels
    g3 = np.stack([g, g, g], axis=-1)
    a = orc.fgs_filter(g3, f, 100.0, 30.0)
    b = orc.fgs_filter(g, f, 100.0, 30.0 / np.sqrt(3.0))
    assert np.abs(a - b).max() < 1e-2
    # int16 maps come back rounded
    d16 = (f * 16).astype(np.int16)
    o16 = orc.fgs_filter(g, d16, 8000.0, 5.0)
    assert o16.dtype == np.int16 and np.abs(o16 - np.rint(orc.fgs_filter(g, d16.astype(np.float32), 8000.0, 5.0))).max() == 0
