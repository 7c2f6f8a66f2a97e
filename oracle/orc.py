"""ctypes binding of the CPU oracle (oracle/_build/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg, never by the shipped package.  Parity is unpinned against the real reference
(it cannot be built here; see oracle/orc.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liborc.so")

BASIC_CENSUS, CIRCLE_CENSUS, ROTATION_INVARIANT_CENSUS, UNIFORM_CENSUS, MULTISCALE_CENSUS, \
    STATISTIC_MULTISCALE_CENSUS, CENSUS_2017 = range(7)
AGG_NONE, AGG_BOX, AGG_SQRBOX, AGG_GUIDED = range(4)
SAT_F32, SAT_INT = 0, 1
GIF, WGIF, EGIF, SKWGIF, OURS_GIF, OURS_GIF2, OURS_GIF3 = range(7)


class orc_sgm_params(C.Structure):
    _fields_ = [("minDisparity", C.c_int), ("numDisparities", C.c_int), ("P1", C.c_int), ("P2", C.c_int),
                ("uniquenessRatio", C.c_int), ("disp12MaxDiff", C.c_int), ("paths", C.c_int), ("subpixel", C.c_int),
                ("costWin", C.c_int), ("censusMethod", C.c_int)]


class orc_sgbm_bt_params(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff", "preFilterCap",
                                       "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode")]


MODE_SGBM, MODE_HH = 0, 1


class orc_bm_params(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("minDisparity", "numDisparities", "blockSize", "preFilterCap", "textureThreshold",
                                       "uniquenessRatio", "speckleWindowSize", "speckleRange", "disp12MaxDiff")]


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


def set_threads(n):
    """OpenMP threads used by the row-parallel loops from now on."""
    lib().orc_set_threads(int(n))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def pad_reflect_lr(src, left, right):
    src = _u8(src)
    H, W = src.shape
    dst = np.empty((H, W + left + right), np.uint8)
    lib().orc_pad_reflect_lr(_p(src), H, W, left, right, _p(dst))
    return dst


def bgr2gray(bgr):
    bgr = _u8(bgr)
    H, W, _ = bgr.shape
    g = np.empty((H, W), np.uint8)
    lib().orc_bgr2gray(_p(bgr), H, W, _p(g))
    return g


def census(src, method=BASIC_CENSUS, arg=0):
    src = _u8(src)
    H, W = src.shape
    dst = np.empty((H, W), np.uint8)
    L = lib()
    if method == BASIC_CENSUS:
        L.orc_census_basic(_p(src), H, W, _p(dst))
    elif method == ROTATION_INVARIANT_CENSUS:
        L.orc_census_rotation_inv(_p(src), H, W, _p(dst))
    elif method == UNIFORM_CENSUS:
        L.orc_census_uniform(_p(src), H, W, _p(dst))
    elif method == MULTISCALE_CENSUS:
        L.orc_census_multiscale(_p(src), H, W, arg, _p(dst))
    elif method == STATISTIC_MULTISCALE_CENSUS:
        L.orc_census_multiscale2(_p(src), H, W, arg, _p(dst))
    elif method == CIRCLE_CENSUS:
        L.orc_census_circle(_p(src), H, W, arg, 8, _p(dst))
    else:
        raise ValueError("unsupported census method")
    return dst


def ri_table():
    t = np.empty(256, np.uint8)
    lib().orc_ri_table(_p(t))
    return t


def uniform_table():
    t = np.empty(256, np.uint8)
    lib().orc_uniform_table(_p(t))
    return t


def hamming(a, b):
    a, b = _u8(a), _u8(b)
    dst = np.empty(a.shape, np.int32)
    lib().orc_hamming(_p(a), _p(b), C.c_long(a.size), _p(dst))
    return dst


def census_stereo(L, R, winSize, minD, maxD, isDispLeft=True, method=BASIC_CENSUS, agg=AGG_GUIDED, want_costs=False):
    L, R = _u8(L), _u8(R)
    H, W = L.shape
    disp = np.empty((H, W), np.float64)
    costs = np.empty((maxD - minD + 1, H, W), np.float32) if want_costs else None
    rc = lib().orc_census_stereo(_p(L), _p(R), H, W, winSize, minD, maxD, int(isDispLeft), method, agg, _p(disp),
                                 _p(costs) if want_costs else None)
    if rc:
        return (None, None) if want_costs else None
    return (disp, costs) if want_costs else disp


def sad_inte(L, R, winSize, minD, maxD, isDispLeft=True, sat=SAT_INT):
    L, R = _u8(L), _u8(R)
    H, W = L.shape
    disp = np.empty((H, W), np.uint8)
    rc = lib().orc_sad_inte(_p(L), _p(R), H, W, winSize, minD, maxD, int(isDispLeft), sat, _p(disp))
    return None if rc else disp


def sad_box(L, R, winSize, minD, maxD, isDispLeft=True):
    L, R = _u8(L), _u8(R)
    H, W = L.shape
    disp = np.empty((H, W), np.uint8)
    rc = lib().orc_sad_box(_p(L), _p(R), H, W, winSize, minD, maxD, int(isDispLeft), _p(disp))
    return None if rc else disp


def box_filter_f32(src, kw, kh, normalize=True):
    src = np.ascontiguousarray(src, np.float32)
    H, W = src.shape
    dst = np.empty((H, W), np.float32)
    lib().orc_box_filter_f32(_p(src), H, W, kw, kh, int(normalize), _p(dst))
    return dst


def normalize_minmax(src):
    src = np.ascontiguousarray(src)
    dst = np.empty(src.shape, np.float32)
    n = C.c_long(src.size)
    if src.dtype == np.uint8:
        lib().orc_normalize_minmax_u8(_p(src), n, _p(dst))
    elif src.dtype == np.int32:
        lib().orc_normalize_minmax_s32(_p(src), n, _p(dst))
    elif src.dtype == np.float32:
        lib().orc_normalize_minmax_f32(_p(src), n, _p(dst))
    else:
        raise TypeError(src.dtype)
    return dst


def steer_filter(src):
    src = _u8(src)
    H, W = src.shape
    dst = np.empty((H, W), np.float64)
    lib().orc_steer_filter(_p(src), H, W, _p(dst))
    return dst


def guided_filter(guide, P, r, eps, gif_type=GIF, steerW=None, r2=5, namuda=0.1, want_beta=False):
    """guide: uint8, P: uint8 / int32 / float32 (normalised here as the reference does)."""
    G = normalize_minmax(_u8(guide))
    Pn = normalize_minmax(np.ascontiguousarray(P))
    H, W = G.shape
    out = np.empty((H, W), np.float32)
    L = lib()
    d = C.c_double
    if gif_type == GIF:
        L.orc_guided_filter(_p(G), _p(Pn), H, W, r, d(eps), _p(out))
    elif gif_type == WGIF:
        L.orc_guided_filter_wgif(_p(G), _p(Pn), H, W, r, d(eps), _p(out))
    elif gif_type == EGIF:
        beta = np.empty((H, W), np.float32)
        L.orc_guided_filter_egif(_p(G), _p(Pn), H, W, r, d(eps), _p(out), _p(beta))
        if want_beta:
            return out, beta
    elif gif_type == SKWGIF:
        sw = np.ascontiguousarray(steerW, np.float64)
        L.orc_guided_filter_skwgif2(_p(G), _p(Pn), _p(sw), H, W, r, r2, d(eps), d(namuda), _p(out))
    elif gif_type in (OURS_GIF, OURS_GIF2, OURS_GIF3):
        beta = np.empty((H, W), np.float32)
        L.orc_guided_filter_ours(_p(G), _p(Pn), H, W, r, r2, d(eps), d(namuda), gif_type - OURS_GIF + 1, _p(out), _p(beta))
        if want_beta:
            return out, beta
    else:
        raise ValueError(gif_type)
    return out


def bilateral_filter(src, d, sigma_color, sigma_space):
    src = np.ascontiguousarray(src, np.float32)
    H, W = src.shape
    dst = np.empty((H, W), np.float32)
    lib().orc_bilateral_f32(_p(src), H, W, d, C.c_double(sigma_color), C.c_double(sigma_space), _p(dst))
    return dst


def asw_gifs(guide, costs, minD, winSize, eps, gif_type=GIF, r2=5, namuda=0.1, want_filtered=False):
    guide = _u8(guide)
    costs = np.ascontiguousarray(costs, np.float32)
    D, H, W = costs.shape
    disp = np.empty((H, W), np.float32)
    filt = np.empty((D, H, W), np.float32) if want_filtered else None
    rc = lib().orc_asw_gifs(_p(guide), _p(costs), H, W, D, minD, winSize, C.c_double(eps), gif_type, r2,
                            C.c_double(namuda), _p(disp), _p(filt) if want_filtered else None)
    if rc:
        return None
    return (disp, filt) if want_filtered else disp


def sgm(L, R, minD=0, D=64, P1=0, P2=0, uniq=-1, maxDiff=0, paths=8, subpixel=1, costWin=1, census_method=BASIC_CENSUS,
        want_S=False):
    L, R = _u8(L), _u8(R)
    H, W = L.shape
    p = orc_sgm_params(minD, D, P1, P2, uniq, maxDiff, paths, subpixel, costWin, census_method)
    disp = np.empty((H, W), np.int16)
    S = np.empty((H, W, D), np.uint16) if want_S else None
    rc = lib().orc_sgm(_p(L), _p(R), H, W, C.byref(p), _p(disp), _p(S) if want_S else None)
    if rc:
        raise RuntimeError(f"orc_sgm failed: {rc}")
    return (disp, S) if want_S else disp


def lr_check(dL, dR, thresh=1.0):
    dL = np.ascontiguousarray(dL, np.float32)
    dR = np.ascontiguousarray(dR, np.float32)
    H, W = dL.shape
    vl = np.empty((H, W), np.uint8)
    vr = np.empty((H, W), np.uint8)
    lib().orc_lr_check(_p(dL), _p(dR), H, W, C.c_float(thresh), _p(vl), _p(vr))
    return vl, vr


def init_undistort_rectify_map(K, dist, R, P, size):
    W, H = size
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    P = np.ascontiguousarray(P, np.float64)
    if P.size == 9:
        P = np.hstack([P.reshape(3, 3), np.zeros((3, 1))])
    P = np.ascontiguousarray(P.reshape(12))
    dist = np.ascontiguousarray(dist, np.float64).reshape(-1)
    m1 = np.empty((H, W, 2), np.int16)
    m2 = np.empty((H, W), np.uint16)
    lib().orc_init_undistort_rectify_map(_p(K), _p(dist), dist.size, _p(R), _p(P), W, H, _p(m1), _p(m2))
    return m1, m2


def remap(src, map1, map2):
    src = _u8(src)
    cn = 1 if src.ndim == 2 else src.shape[2]
    m1 = np.ascontiguousarray(map1, np.int16)
    m2 = np.ascontiguousarray(map2, np.uint16)
    dH, dW = m2.shape
    dst = np.empty((dH, dW) + ((cn,) if src.ndim == 3 else ()), np.uint8)
    lib().orc_remap_bilinear_u8(_p(src), src.shape[0], src.shape[1], cn, _p(m1), _p(m2), dH, dW, _p(dst))
    return dst


def stereo_rectify(K1, D1, K2, D2, size, R, T, alpha=-1.0, zero_disparity=True, global_rotation=0):
    """cv::stereoRectify restatement -> dict(R1, R2, P1, P2, Q, roi1, roi2)."""
    W, H = size
    f = lambda a, n: np.ascontiguousarray(a, np.float64).reshape(n)  # noqa: E731
    K1, K2, R, T = f(K1, 9), f(K2, 9), f(R, 9), f(T, 3)
    D1 = np.ascontiguousarray(D1, np.float64).reshape(-1)
    D2 = np.ascontiguousarray(D2, np.float64).reshape(-1)
    assert D1.size == D2.size
    R1, R2, P1, P2, Q = (np.empty(n, np.float64) for n in (9, 9, 12, 12, 16))
    roi1, roi2 = np.empty(4, np.int32), np.empty(4, np.int32)
    fn = lib().orc_stereo_rectify
    fn.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int] + [C.c_void_p] * 7
    fn.restype = None
    fn(_p(K1), _p(D1), _p(K2), _p(D2), D1.size, W, H, _p(R), _p(T), float(alpha), 1 if zero_disparity else 0, int(global_rotation),
       _p(R1), _p(R2), _p(P1), _p(P2), _p(Q), _p(roi1), _p(roi2))
    return {"R1": R1.reshape(3, 3), "R2": R2.reshape(3, 3), "P1": P1.reshape(3, 4), "P2": P2.reshape(3, 4),
            "Q": Q.reshape(4, 4), "roi1": tuple(int(v) for v in roi1), "roi2": tuple(int(v) for v in roi2)}


def reproject_q(disp, Q):
    disp = np.ascontiguousarray(disp, np.float32)
    Q = np.ascontiguousarray(Q, np.float64).reshape(16)
    H, W = disp.shape
    xyz = np.empty((H, W, 3), np.float32)
    lib().orc_reproject_q(_p(disp), H, W, _p(Q), _p(xyz))
    return xyz


def reproject_triangulate(disp, K, baseline, bgr=None):
    disp = np.ascontiguousarray(disp, np.float32)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    H, W = disp.shape
    xyz = np.empty((H, W, 3), np.float32)
    fn = lib().orc_reproject_triangulate
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    fn.restype = None
    c = np.ascontiguousarray(bgr, np.float32) if bgr is not None else None
    fn(_p(disp), H, W, _p(K), float(baseline), _p(c) if c is not None else None, _p(xyz))
    return xyz


def sgbm_bt(L, R, minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0, preFilterCap=0,
            uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=MODE_SGBM, want_cost=False, want_raw=False):
    """cv::StereoSGBM::create(...)->compute(L, R) restated (parity unpinned); images HxW or HxWx3 uint8."""
    L, R = np.ascontiguousarray(L, np.uint8), np.ascontiguousarray(R, np.uint8)
    H, W = L.shape[:2]
    cn = 1 if L.ndim == 2 else L.shape[2]
    p = orc_sgbm_bt_params(minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap, uniquenessRatio,
                           speckleWindowSize, speckleRange, mode)
    disp = np.empty((H, W), np.int16)
    W1 = max(W - minDisparity - numDisparities, 0)
    cost = np.empty((H, W1, numDisparities), np.int16) if want_cost else None
    raw = np.empty((H, W), np.int16) if want_raw else None
    rc = lib().orc_sgbm_bt(_p(L), _p(R), H, W, cn, C.byref(p), _p(disp), _p(cost) if want_cost else None,
                           _p(raw) if want_raw else None)
    if rc:
        raise ValueError(f"orc_sgbm_bt rc={rc}")
    out = [disp]
    if want_cost:
        out.append(cost)
    if want_raw:
        out.append(raw)
    return out[0] if len(out) == 1 else tuple(out)


def median3_s16(src):
    src = np.ascontiguousarray(src, np.int16)
    dst = np.empty_like(src)
    lib().orc_median3_s16(_p(src), src.shape[0], src.shape[1], _p(dst))
    return dst


def filter_speckles_s16(img, newVal, maxSpeckleSize, maxDiff):
    out = np.ascontiguousarray(img, np.int16).copy()
    lib().orc_filter_speckles_s16(_p(out), out.shape[0], out.shape[1], newVal, maxSpeckleSize, maxDiff)
    return out


(LEGACY_AD, LEGACY_SAD, LEGACY_SAD_INTE, LEGACY_SAD_BF, LEGACY_TAD, LEGACY_STAD, LEGACY_STAD_INTE, LEGACY_STAD_BF,
 LEGACY_SD, LEGACY_SSD, LEGACY_SSD_INTE, LEGACY_SSD_BF) = range(12)


def legacy_disparity(L, R, method, winSize=7, threshold_T=30, minDisparity=0, numDisparity=30):
    """stereomatch_1::computeAD ... computeSSD_BFOpti (methods/method_StereoMatching.cpp:32-869) -> uint8, or None
    where the reference would refuse or throw."""
    L, R = _u8(L), _u8(R)
    H, W = L.shape
    out = np.empty((H, W), np.uint8)
    rc = lib().orc_legacy_disparity(_p(L), _p(R), H, W, method, winSize, threshold_T, minDisparity, numDisparity, _p(out))
    return None if rc else out


def legacy_inte_f32(L, R, method, winSize=7, threshold_T=30, minDisparity=0, numDisparity=30):
    """The _inteOpti kinds with the reference's CV_32F integral tables (rounding above 2^24 included) -> uint8."""
    L, R = _u8(L), _u8(R)
    H, W = L.shape
    out = np.empty((H, W), np.uint8)
    rc = lib().orc_legacy_inte_f32(_p(L), _p(R), H, W, method, winSize, threshold_T, minDisparity, numDisparity, _p(out))
    return None if rc else out


def cost_sad_d(left, right, disparity, is_left, winSize):
    """stereomatch_1::getCostSAD_d on gray images of which one is bordered (wider) -> float32, or None where the
    reference returns an empty image."""
    left, right = _u8(left), _u8(right)
    H = left.shape[0]
    W = left.shape[1] if is_left else right.shape[1]
    out = np.empty((H, W), np.float32)
    rc = lib().orc_cost_sad_d(_p(left), left.shape[1], _p(right), right.shape[1], H, disparity, int(is_left), winSize, _p(out))
    return None if rc else out


def guided_filter_mc(guide, P, r, eps, variant=0):
    """getGuidedFilter with an H x W x cn uint8 guide (normalised here with one min / max over all channels) and a
    single-channel P; variant 0: gifs.cpp, 1: stereomatch_1 -> float32."""
    g = np.ascontiguousarray(guide, np.uint8)
    H, W, cn = g.shape
    mn, mx = int(g.min()), int(g.max())
    scale = 1.0 / (mx - mn) if mx - mn > np.finfo(np.float64).eps else 0.0
    a, b = np.float32(scale), np.float32(0.0 - mn * scale)
    G = np.ascontiguousarray(np.moveaxis(g.astype(np.float32) * a + b, 2, 0))
    Pn = normalize_minmax(np.ascontiguousarray(P))
    out = np.empty((H, W), np.float32)
    lib().orc_guided_filter_mc(_p(G), cn, _p(Pn), H, W, r, C.c_double(eps), variant, _p(out))
    return out


def adaptive_weight_guidedf(L, R, is_left=True, eps=0.01, winSize=15, minDisparity=10, numDisparity=150):
    """stereomatch_1::computeAdaptiveWeight_GuidedF on H x W x 3 uint8 images -> float32 map, or None where the
    reference fails (gray input)."""
    L = np.ascontiguousarray(L, np.uint8)
    R = np.ascontiguousarray(R, np.uint8)
    cn = 1 if L.ndim == 2 else L.shape[2]
    H, W = L.shape[:2]
    out = np.empty((H, W), np.float32)
    rc = lib().orc_adaptive_weight_guidedf(_p(L), _p(R), H, W, cn, int(is_left), C.c_double(eps), winSize, minDisparity, numDisparity,
                                           _p(out))
    return None if rc else out


def compute_similarity(L, R, is_left=True, regularity=0.4, thresC=10.0, thresG=50.0, minDisparity=0, numDisparity=30):
    """stereomatch_1::computeSimilarity (nine-argument overload) on H x W x 3 uint8 images -> float32 [numDisparity, H, W],
    or None where the reference fails or returns nothing."""
    L = np.ascontiguousarray(L, np.uint8)
    R = np.ascontiguousarray(R, np.uint8)
    cn = 1 if L.ndim == 2 else L.shape[2]
    H, W = L.shape[:2]
    out = np.empty((numDisparity, H, W), np.float32)
    rc = lib().orc_compute_similarity(_p(L), _p(R), H, W, cn, int(is_left), C.c_double(regularity), C.c_double(thresC), C.c_double(thresG),
                                      minDisparity, numDisparity, _p(out))
    return None if rc else out


def adaptive_weight_guidedf_23(L, R, which, is_left=True, eps=0.01, winSize=15, minDisparity=10, numDisparity=150):
    """stereomatch_1::computeAdaptiveWeight_GuidedF_2 (which = 2) / _3 (which = 3) -> float32 map, or None where the reference fails."""
    L = np.ascontiguousarray(L, np.uint8)
    R = np.ascontiguousarray(R, np.uint8)
    cn = 1 if L.ndim == 2 else L.shape[2]
    H, W = L.shape[:2]
    out = np.empty((H, W), np.float32)
    rc = lib().orc_adaptive_weight_guidedf_23(_p(L), _p(R), H, W, cn, which, int(is_left), C.c_double(eps), winSize, minDisparity,
                                              numDisparity, _p(out))
    return None if rc else out


def ncc_disparity(L, R, is_left=True, winSize=7, minDisparity=0, numDisparity=30):
    """stereomatch_1::computeNCC, map overload (methods/method_StereoMatching.cpp:915-1015) -> float32, or None where the
    reference refuses (even window)."""
    L, R = _u8(L), _u8(R)
    H, W = L.shape
    out = np.empty((H, W), np.float32)
    rc = lib().orc_ncc_disparity(_p(L), _p(R), H, W, 1 if is_left else 0, winSize, minDisparity, numDisparity, _p(out))
    return None if rc else out


def ncc_volume(L, R, is_left=True, winSize=7, minDisparity=0, numDisparity=30):
    """stereomatch_1::computeNCC, cost-volume overload (methods/method_StereoMatching.cpp:1017-1106) -> float32
    [numDisparity, H, W], every slice normalised to [0, 1]."""
    L, R = _u8(L), _u8(R)
    H, W = L.shape
    out = np.empty((numDisparity, H, W), np.float32)
    rc = lib().orc_ncc_volume(_p(L), _p(R), H, W, 1 if is_left else 0, winSize, minDisparity, numDisparity, _p(out))
    return None if rc else out


def bm_prefilter_xsobel(img, cap=31):
    img = _u8(img)
    out = np.empty_like(img)
    lib().orc_bm_prefilter_xsobel(_p(img), img.shape[0], img.shape[1], cap, _p(out))
    return out


def stereobm(L, R, numDisparities=64, blockSize=21, minDisparity=0, preFilterCap=31, textureThreshold=10, uniquenessRatio=15,
             speckleWindowSize=0, speckleRange=0, disp12MaxDiff=-1):
    """cv::StereoBM::create(numDisparities, blockSize) + setters -> compute(): CV_16S disparity x16."""
    L, R = _u8(L), _u8(R)
    H, W = L.shape
    p = orc_bm_params(minDisparity, numDisparities, blockSize, preFilterCap, textureThreshold, uniquenessRatio, speckleWindowSize,
                      speckleRange, disp12MaxDiff)
    disp = np.empty((H, W), np.int16)
    rc = lib().orc_stereobm(_p(L), _p(R), H, W, C.byref(p), _p(disp))
    if rc:
        raise ValueError("StereoBM: bad parameters")
    return disp


def fgs_filter(guide, src, lambda_=8000.0, sigma_color=0.5, attenuation=0.25, iters=3):
    """ximgproc::fastGlobalSmootherFilter(guide, src, dst, lambda, sigma_color, lambda_attenuation, num_iter); src float32 or
    int16 HxW, dst of the same type."""
    g = _u8(guide)
    H, W = g.shape[:2]
    cn = 1 if g.ndim == 2 else g.shape[2]
    if src.dtype == np.int16:
        s = np.ascontiguousarray(src)
        out = np.empty_like(s)
        lib().orc_fgs_filter_s16(_p(g), H, W, cn, _p(s), C.c_double(lambda_), C.c_double(sigma_color), C.c_double(attenuation), iters, _p(out))
        return out
    s = np.ascontiguousarray(src, dtype=np.float32)
    out = np.empty_like(s)
    lib().orc_fgs_filter_f32(_p(g), H, W, cn, _p(s), C.c_double(lambda_), C.c_double(sigma_color), C.c_double(attenuation), iters, _p(out))
    return out


def postProcess_(imgLeft, imgRight, dispL, dispR):
    """methods_disp.cpp:1299-1313: both maps through the generic WLS filter without confidence (lambda 8000, sigma 0.5)."""
    return fgs_filter(imgLeft, dispL), fgs_filter(imgRight, dispR)
