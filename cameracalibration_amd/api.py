"""Host-side mirror of the reference's function-level stereo API, on top of the C ABI.

Names, argument meaning and error behaviour follow cameraCalibration/disp_method/methods_disp.h,
gifs.h, steerFilter.h and methods/methods.h; a numpy array plays the role of cv::Mat.  On bad
parameters the reference prints a message and returns an empty Mat -- here the same message is
printed and None is returned.  All compute happens in libmdisp.so on the GPU."""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from ._lib import MdispError, check, image, load, mdisp_param, mdisp_sgbm_params, mdisp_stereobm_params, mdisp_stereosgbm_params

# CENSUS_ALGORITHM, methods_disp.h:18-27
BASIC_CENSUS, CIRCLE_CENSUS, ROTATION_INVARIANT_CENSUS, UNIFORM_CENSUS, MULTISCALE_CENSUS, \
    STATISTIC_MULTISCALE_CENSUS, CENSUS_2017 = range(7)
# aggregation of the Hamming slices in censusStereo
AGG_NONE, AGG_BOX, AGG_SQRBOX, AGG_GUIDED = range(4)
# GIF_TYPE, gifs.h:4-13
GIF, WGIF, EGIF, SKWGIF, OURS_GIF, OURS_GIF2, OURS_GIF3 = range(7)
COST_VOLUME, COST_CENSUS, COST_ABSDIFF = range(3)
# StereoMatchingAlgorithms, methods/parametersStereo.h:11-24
(BM, SGBM, ADAPTIVE_WEIGHT, ADAPTIVE_WEIGHT_8DIRECT, ADAPTIVE_WEIGHT_GEODESIC, ADAPTIVE_WEIGHT_BILATERAL_GRID, ADAPTIVE_WEIGHT_BLO1,
 ADAPTIVE_WEIGHT_GUIDED_FILTER, ADAPTIVE_WEIGHT_GUIDED_FILTER_2, ADAPTIVE_WEIGHT_GUIDED_FILTER_3, ADAPTIVE_WEIGHT_MEDIAN) = range(11)


@dataclass
class StereoMatchParam:
    """methods_disp.h:5-16.  maxDisparity is inclusive."""
    imgLeft: Optional[np.ndarray] = None
    imgRight: Optional[np.ndarray] = None
    imgLeft_C: Optional[np.ndarray] = None
    imgRight_C: Optional[np.ndarray] = None
    winSize: int = 7
    minDisparity: int = 0
    maxDisparity: int = 30
    isDispLeft: bool = True

    def pod(self):
        return mdisp_param(self.winSize, self.minDisparity, self.maxDisparity, 1 if self.isDispLeft else 0)


_BAD = "bad function parameters, please check image and window size."


def _ctx(ctx):
    return (ctx or _lib.default_context()).handle


def _u8(a):
    a = np.asarray(a)
    if a.dtype != np.uint8:
        raise TypeError("expected an 8-bit image")
    return np.ascontiguousarray(a)


def _pair_call(fn, param, out_dtype, ctx, *extra):
    if param.imgLeft is None or param.imgRight is None:
        print("empty image error")
        print(_BAD)
        return None
    L, R = _u8(param.imgLeft), _u8(param.imgRight)
    out = np.empty(L.shape[:2], dtype=out_dtype)
    p = param.pod()
    rc = fn(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), C.byref(p), *extra, C.byref(image(out)))
    if rc == _lib.E_BADARG:
        print(_BAD)
        return None
    check(rc)
    return out


SAT_F32, SAT_INT = 0, 1


def computeSAD_inteOpti(param, ctx=None, tables=None):
    """methods_disp.h:43 -> CV_8U disparity.  tables: None = the library's default (the reference's CV_32F integral tables),
    SAT_F32 or SAT_INT (exact integer tables) per call."""
    if tables is None:
        return _pair_call(load().mdisp_sad_inte, param, np.uint8, ctx)
    return _pair_call(load().mdisp_sad_inte_mode, param, np.uint8, ctx, int(tables))


def computeSAD_BFOpti(param, ctx=None):
    """methods_disp.h:44 -> CV_8U disparity."""
    return _pair_call(load().mdisp_sad_box, param, np.uint8, ctx)


def censusStereo(param, method=BASIC_CENSUS, agg=AGG_GUIDED, ctx=None):
    """methods_disp.h:62 -> CV_64F disparity.  `agg` selects the slice aggregation; the reference
    hard-codes the guided filter (AGG_GUIDED)."""
    lib = load()

    def fn(h, l, r, p, o):
        return lib.mdisp_census_stereo(h, l, r, p, method, agg, o)
    return _pair_call(fn, param, np.float64, ctx)


def _census(src, method, arg, ctx):
    src = _u8(src)
    dst = np.empty(src.shape[:2], dtype=np.uint8)
    check(load().mdisp_census_transform(_ctx(ctx), C.byref(image(src)), method, arg, C.byref(image(dst))))
    return dst


def countCensusImg(src, ctx=None):
    return _census(src, BASIC_CENSUS, 0, ctx)


def countCensusImg_circle(src, radius, samplePtNum=8, ctx=None):
    if samplePtNum != 8:
        raise ValueError("only 8 sample points are supported")
    return _census(src, CIRCLE_CENSUS, radius, ctx)


def countCensusImg_rotationInv(src, ctx=None):
    return _census(src, ROTATION_INVARIANT_CENSUS, 0, ctx)


def countCensusImg_uniform(src, ctx=None):
    return _census(src, UNIFORM_CENSUS, 0, ctx)


def countCensusImg_multiScale(src, scale, ctx=None):
    return _census(src, MULTISCALE_CENSUS, scale, ctx)


def countCensusImg_multiScale2(src, scale, ctx=None):
    """methods_disp.h:56: multi-scale census remapped to the rank of its code in the image's histogram."""
    return _census(src, STATISTIC_MULTISCALE_CENSUS, scale, ctx)


def countHummingDist(src1, src2, ctx=None):
    a, b = _u8(src1), _u8(src2)
    dst = np.empty(a.shape, dtype=np.int32)
    check(load().mdisp_hamming_dist(_ctx(ctx), C.byref(image(a)), C.byref(image(b)), C.byref(image(dst))))
    return dst


def _guided(guide, P, r, eps, gif_type, steerW, r2, namuda, want_beta, ctx):
    guide = _u8(guide)
    P = np.ascontiguousarray(P)
    out = np.empty(guide.shape, dtype=np.float32)
    beta = np.empty(guide.shape, dtype=np.float32) if want_beta else None
    sw = None
    if steerW is not None:
        sw = np.ascontiguousarray(steerW, dtype=np.float64)
    check(load().mdisp_guided_filter(_ctx(ctx), C.byref(image(guide)), C.byref(image(P)), r, eps, gif_type,
                                     C.byref(image(sw)) if sw is not None else None, r2, namuda,
                                     C.byref(image(out)), C.byref(image(beta)) if want_beta else None))
    return (out, beta) if want_beta else out


def getGuidedFilter(guidedImg, inputP, r, eps, ctx=None):
    """gifs.h:17"""
    return _guided(guidedImg, inputP, r, eps, GIF, None, 0, 0.0, False, ctx)


def getGuidedFilter_wgif(guidedImg, inputP, r, eps, ctx=None):
    return _guided(guidedImg, inputP, r, eps, WGIF, None, 0, 0.0, False, ctx)


def getGuidedFilter_egif(guidedImg, inputP, r, eps, ctx=None):
    """gifs.h:21 -> (filtered, beta)"""
    return _guided(guidedImg, inputP, r, eps, EGIF, None, 0, 0.0, True, ctx)


def getGuidedFilter_skwgif2(guidedImg, inputP, steerFilterW, r, r2, eps, namuda, h=0.5, ctx=None):
    """gifs.h:33 (`h` is accepted and, as in the reference, unused)."""
    return _guided(guidedImg, inputP, r, eps, SKWGIF, steerFilterW, r2, namuda, False, ctx)


def getGuidedFilter_ours_gif(guidedImg, inputP, r, r2, eps, namuda, h=0.5, ctx=None):
    """gifs.h:36: WGIF-like coefficients (gamma from the r2-window variance), bilateral-smoothed."""
    return _guided(guidedImg, inputP, r, eps, OURS_GIF, None, r2, namuda, False, ctx)


def getGuidedFilter_ours_gif2(guidedImg, inputP, r, r2, eps, namuda, h=0.5, ctx=None):
    """gifs.h:37: plain guided-filter coefficients, bilateral-smoothed."""
    return _guided(guidedImg, inputP, r, eps, OURS_GIF2, None, r2, namuda, False, ctx)


def getGuidedFilter_ours_gif3(guidedImg, inputP, r, r2, eps, namuda, h=0.5, ctx=None):
    """gifs.h:38: EGIF coefficients, bilateral-smoothed with diameter r -> (filtered, beta)."""
    return _guided(guidedImg, inputP, r, eps, OURS_GIF3, None, r2, namuda, True, ctx)


def steerFilter(src, ctx=None):
    """steerFilter.h:39 -> CV_64F orientation-strength map."""
    src = _u8(src)
    dst = np.empty(src.shape, dtype=np.float64)
    check(load().mdisp_steer_filter(_ctx(ctx), C.byref(image(src)), C.byref(image(dst))))
    return dst


def asw_gifs(param, eps, gifType=GIF, r2=5, namda=0.1, h=0.5, cost_source=COST_CENSUS, costs=None, ctx=None):
    """methods_disp.h:66 -> CV_32F disparity.  The reference feeds dense-SIFT costs
    (computeDisp_Feature); pass them as `costs` (numDisp x H x W float32) with
    cost_source=COST_VOLUME, or let the GPU build census / absdiff slices."""
    lib = load()
    cptr = None
    if cost_source == COST_VOLUME:
        costs = np.ascontiguousarray(costs, dtype=np.float32)
        cptr = costs.ctypes.data

    def fn(hd, l, r, p, o):
        return lib.mdisp_asw_gifs(hd, l, r, p, eps, gifType, r2, namda, h, cost_source, cptr, o)
    return _pair_call(fn, param, np.float32, ctx)


def asw_gifs_band_dev(d_left, d_right, rows, cols, pitch, winSize, minDisparity, maxDisparity, isDispLeft, eps, gifType, r2, namda,
                      row_begin, row_end, d_disp, cost_source=COST_CENSUS, ctx=None, stats=None):
    """asw_gifs on device-resident images (raw device pointers), output rows [row_begin, row_end) only:
    one GPU's share of a pair that is sharded by horizontal row bands.  d_disp: float32 rows x cols.
    stats: (cost_minmax, steer_max) merged over all bands (asw_gifs_band_stats_dev + element-wise min / max), or
    None to recompute the image-wide statistics locally."""
    p = mdisp_param(winSize, minDisparity, maxDisparity, 1 if isDispLeft else 0)
    mm = sm = None
    if stats is not None:
        mm = np.ascontiguousarray(stats[0], dtype=np.float32).reshape(-1)
        if mm.size != 2 * (maxDisparity - minDisparity + 1):
            raise ValueError("cost_minmax must hold (min, max) for every disparity")
        sm = C.c_double(float(stats[1]))
    check(load().mdisp_asw_gifs_band_dev(_ctx(ctx), d_left, d_right, rows, cols, pitch, C.byref(p), eps, gifType, r2, namda,
                                         cost_source, row_begin, row_end, d_disp, mm.ctypes.data if mm is not None else None,
                                         C.byref(sm) if sm is not None else None))


def asw_gifs_band_stats_dev(d_left, d_right, rows, cols, pitch, winSize, minDisparity, maxDisparity, isDispLeft, gifType,
                            row_begin, row_end, cost_source=COST_CENSUS, ctx=None):
    """One band's share of the image-wide statistics of asw_gifs: (cost_minmax float32 [numDisp, 2], steer_max float)."""
    p = mdisp_param(winSize, minDisparity, maxDisparity, 1 if isDispLeft else 0)
    D = maxDisparity - minDisparity + 1
    mm = np.empty((D, 2), dtype=np.float32)
    sm = C.c_double(0.0)
    check(load().mdisp_asw_gifs_band_stats_dev(_ctx(ctx), d_left, d_right, rows, cols, pitch, C.byref(p), gifType, cost_source,
                                               row_begin, row_end, mm.ctypes.data, C.byref(sm)))
    return mm, sm.value


def sgbm_params(minDisparity=0, numDisparities=64, P1=0, P2=0, uniquenessRatio=-1, disp12MaxDiff=0, paths=8,
                subpixel=1, costWin=1, censusMethod=BASIC_CENSUS):
    return mdisp_sgbm_params(minDisparity, numDisparities, P1, P2, uniquenessRatio, disp12MaxDiff, paths, subpixel,
                             costWin, censusMethod)


def sgbm_compute(srcLeft, srcRight, params, ctx=None):
    """cv::StereoSGBM::compute replacement -> CV_16S disparity x16."""
    L, R = _u8(srcLeft), _u8(srcRight)
    out = np.empty(L.shape[:2], dtype=np.int16)
    check(load().mdisp_sgbm(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), C.byref(params), C.byref(image(out))))
    return out


def sgbm_compute_batch(lefts, rights, params, chunk=0, ctx=None):
    """mdisp_sgbm over a sequence of host pairs (arrays N x H x W or lists of H x W uint8), with the copies
    overlapped with the computation -> int16 N x H x W."""
    Ls = [np.ascontiguousarray(a, dtype=np.uint8) for a in lefts]
    Rs = [np.ascontiguousarray(a, dtype=np.uint8) for a in rights]
    n = len(Ls)
    if n != len(Rs):
        raise ValueError("left / right count mismatch")
    if n == 0:
        return np.empty((0, 0, 0), np.int16)
    H, W = Ls[0].shape
    out = np.empty((n, H, W), dtype=np.int16)
    ptrs = C.c_void_p * n
    pl = ptrs(*[a.ctypes.data for a in Ls])
    pr = ptrs(*[a.ctypes.data for a in Rs])
    po = ptrs(*[out[i].ctypes.data for i in range(n)])
    check(load().mdisp_sgbm_batch(_ctx(ctx), pl, pr, n, H, W, W, C.byref(params), po, chunk))
    return out


MULTI_PIN_HOST = 1


def sgbm_compute_batch_multi(devices, lefts, rights, params, chunk=0, pin_host=True):
    """mdisp_sgbm_batch_multi: the pairs of a host batch spread over `devices` (HIP device indices, repeats allowed), one
    host thread and one context per entry -> int16 N x H x W."""
    Ls = [np.ascontiguousarray(a, dtype=np.uint8) for a in lefts]
    Rs = [np.ascontiguousarray(a, dtype=np.uint8) for a in rights]
    n = len(Ls)
    if n != len(Rs):
        raise ValueError("left / right count mismatch")
    if n == 0:
        return np.empty((0, 0, 0), np.int16)
    H, W = Ls[0].shape
    out = np.empty((n, H, W), dtype=np.int16)
    ptrs = C.c_void_p * n
    pl = ptrs(*[a.ctypes.data for a in Ls])
    pr = ptrs(*[a.ctypes.data for a in Rs])
    po = ptrs(*[out[i].ctypes.data for i in range(n)])
    dev = (C.c_int * len(devices))(*devices)
    check(load().mdisp_sgbm_batch_multi(dev, len(devices), pl, pr, n, H, W, W, C.byref(params), po, chunk, MULTI_PIN_HOST if pin_host else 0))
    return out


def sgbm_compute_rows_multi(devices, srcLeft, srcRight, params):
    """mdisp_sgbm_rows_multi: one pair, its rows spread over `devices` -> CV_16S disparity x16."""
    L, R = _u8(srcLeft), _u8(srcRight)
    out = np.empty(L.shape[:2], dtype=np.int16)
    dev = (C.c_int * len(devices))(*devices)
    check(load().mdisp_sgbm_rows_multi(dev, len(devices), C.byref(image(L)), C.byref(image(R)), C.byref(params), C.byref(image(out))))
    return out


def sgbm_volume(srcLeft, srcRight, params, ctx=None):
    """Aggregated SGM cost volume S (uint16, H x W x numDisparities, biased by +P2 per path)."""
    L, R = _u8(srcLeft), _u8(srcRight)
    S = np.empty(L.shape[:2] + (params.numDisparities,), dtype=np.uint16)
    check(load().mdisp_sgbm_volume(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), C.byref(params), S.ctypes.data))
    return S


MODE_SGBM, MODE_HH = 0, 1


def StereoSGBM_create(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0, preFilterCap=0,
                      uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=MODE_SGBM):
    """cv::StereoSGBM::create with its argument order and defaults -> parameter block."""
    return mdisp_stereosgbm_params(minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap, uniquenessRatio,
                                   speckleWindowSize, speckleRange, mode)


def _u8_image(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim not in (2, 3):
        raise ValueError("expected HxW or HxWxC uint8")
    return a


def stereosgbm_compute(left, right, params, want_cost=False, want_raw=False, ctx=None):
    """cv::StereoSGBM::compute (Birchfield-Tomasi block cost, MODE_SGBM / MODE_HH, median, speckles) -> CV_16S x16.
    want_cost: also the block cost volume int16 [H][W - minD - D][D]; want_raw: also the map before the median."""
    L, R = _u8_image(left), _u8_image(right)
    H, W = L.shape[:2]
    out = np.empty((H, W), dtype=np.int16)
    if not (want_cost or want_raw):
        check(load().mdisp_stereosgbm_compute(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), C.byref(params), C.byref(image(out))))
        return out
    W1 = max(W - params.minDisparity - params.numDisparities, 0)
    cost = np.empty((H, W1, params.numDisparities), dtype=np.int16) if want_cost else None
    raw = np.empty((H, W), dtype=np.int16) if want_raw else None
    check(load().mdisp_stereosgbm_stages(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), C.byref(params), C.byref(image(out)),
                                         cost.ctypes.data if want_cost else None, C.byref(image(raw)) if want_raw else None))
    res = [out]
    if want_cost:
        res.append(cost)
    if want_raw:
        res.append(raw)
    return tuple(res)


def medianBlur3(src, ctx=None):
    """cv::medianBlur(src, dst, 3) on CV_16SC1."""
    a = np.ascontiguousarray(src, dtype=np.int16)
    dst = np.empty_like(a)
    check(load().mdisp_median_blur3_s16(_ctx(ctx), C.byref(image(a)), C.byref(image(dst))))
    return dst


def filterSpeckles(img, newVal, maxSpeckleSize, maxDiff, ctx=None):
    """cv::filterSpeckles on CV_16SC1; returns the filtered copy."""
    a = np.ascontiguousarray(img, dtype=np.int16).copy()
    check(load().mdisp_filter_speckles_s16(_ctx(ctx), C.byref(image(a)), newVal, maxSpeckleSize, maxDiff))
    return a


def getDisparity_SGBM(srcLeft, srcRight, ctx=None):
    """methods/methods.cpp:1883-1921: StereoSGBM::create(0, 48, 7)->compute, then `disp / 16`
    (a CV_16S MatExpr divide: round half to even)."""
    if srcLeft is None or srcRight is None:
        print(" --(!) Error reading images ")
        return None
    d16 = stereosgbm_compute(srcLeft, srcRight, StereoSGBM_create(0, 48, 7), ctx=ctx)
    return np.rint(d16.astype(np.float64) / 16.0).astype(np.int16)


class StereoBM:
    """cv::StereoBM as the reference drives it (methods/methods.cpp:1205-1213, 1829-1838): create(numDisparities, blockSize),
    the library's defaults for everything else, setters with its names, compute() -> CV_16S disparity x16."""

    def __init__(self, numDisparities=0, blockSize=21):
        self.p = mdisp_stereobm_params(0, numDisparities if numDisparities > 0 else 64, blockSize, 31, 10, 15, 0, 0, -1)

    @staticmethod
    def create(numDisparities=0, blockSize=21):
        return StereoBM(numDisparities, blockSize)

    def setMinDisparity(self, v): self.p.minDisparity = v
    def setNumDisparities(self, v): self.p.numDisparities = v
    def setBlockSize(self, v): self.p.blockSize = v
    def setPreFilterCap(self, v): self.p.preFilterCap = v
    def setTextureThreshold(self, v): self.p.textureThreshold = v
    def setUniquenessRatio(self, v): self.p.uniquenessRatio = v
    def setSpeckleWindowSize(self, v): self.p.speckleWindowSize = v
    def setSpeckleRange(self, v): self.p.speckleRange = v
    def setDisp12MaxDiff(self, v): self.p.disp12MaxDiff = v

    def compute(self, left, right, ctx=None):
        L, R = _u8_image(left), _u8_image(right)       # three channels: converted with COLOR_BGR2GRAY first
        out = np.empty(L.shape[:2], dtype=np.int16)
        check(load().mdisp_stereobm_compute(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), C.byref(self.p), C.byref(image(out))))
        return out


def getDisparity_BM(srcLeft, srcRight, ctx=None):
    """methods/methods.cpp:1825-1881: StereoBM(144, 35), minDisparity 186, textureThreshold 10, disp12MaxDiff -1, preFilterCap 31,
    uniquenessRatio 25, speckleRange 32, speckleWindowSize 0 -> CV_16S disparity x16 (colour input goes through BGR2GRAY)."""
    if srcLeft is None or srcRight is None or srcLeft.size == 0 or srcRight.size == 0:
        print(" --(!) Error reading images ")
        return None
    sbm = StereoBM.create(16 * 9, 35)
    sbm.setMinDisparity(186)
    sbm.setTextureThreshold(10)
    sbm.setDisp12MaxDiff(-1)
    sbm.setPreFilterCap(31)
    sbm.setUniquenessRatio(25)
    sbm.setSpeckleRange(32)
    sbm.setSpeckleWindowSize(0)
    return sbm.compute(srcLeft, srcRight, ctx)


def stereoMatching(srcLeft, srcRight, algorithmType=SGBM, ctx=None):
    """methods/methods.cpp:1776-1813: the BM, SGBM and ADAPTIVE_WEIGHT_GUIDED_FILTER entries (the other adaptive-weight
    research variants are not built)."""
    if algorithmType == BM:
        return getDisparity_BM(srcLeft, srcRight, ctx)
    if algorithmType == ADAPTIVE_WEIGHT_GUIDED_FILTER:
        return stereomatch_1.computeAdaptiveWeight_GuidedF(srcLeft, srcRight, DISPARITY_LEFT, 0.01, 15, 10, 150, ctx=ctx)   # :1801-1803
    if algorithmType == ADAPTIVE_WEIGHT_GUIDED_FILTER_2:
        return stereomatch_1.computeAdaptiveWeight_GuidedF_2(srcLeft, srcRight, DISPARITY_LEFT, 0.01, 13, 10, 150, ctx=ctx)  # :1804-1806
    if algorithmType == ADAPTIVE_WEIGHT_GUIDED_FILTER_3:
        return stereomatch_1.computeAdaptiveWeight_GuidedF_3(srcLeft, srcRight, DISPARITY_LEFT, 1e-6, 15, 0, 64, ctx=ctx)    # :1807-1809
    if algorithmType != SGBM:
        raise NotImplementedError("only the BM, SGBM and ADAPTIVE_WEIGHT_GUIDED_FILTER* entries of stereoMatching are built")
    return getDisparity_SGBM(srcLeft, srcRight, ctx)


def fastGlobalSmootherFilter(guide, src, lambda_=8000.0, sigma_color=0.5, lambda_attenuation=0.25, num_iter=3, ctx=None):
    """ximgproc::fastGlobalSmootherFilter: guide uint8 HxW or HxWx3, src float32 or int16 HxW -> same type as src."""
    g = _u8_image(guide)
    s = np.ascontiguousarray(src) if np.asarray(src).dtype == np.int16 else np.ascontiguousarray(src, dtype=np.float32)
    out = np.empty_like(s)
    check(load().mdisp_fgs_filter(_ctx(ctx), C.byref(image(g)), C.byref(image(s)), lambda_, sigma_color, lambda_attenuation, num_iter,
                                  C.byref(image(out))))
    return out


def postProcess_(param, dispL, dispR, ctx=None):
    """disp_method/methods_disp.cpp:1299-1313: createDisparityWLSFilterGeneric(false), lambda 8000, sigmaColor 0.5, whole-frame
    ROI, applied to the left map with the left view and to the right map with the right view -> (filteredDispL, filteredDispR)."""
    return (fastGlobalSmootherFilter(param.imgLeft, dispL, 8000.0, 0.5, ctx=ctx),
            fastGlobalSmootherFilter(param.imgRight, dispR, 8000.0, 0.5, ctx=ctx))


def lr_check(dispL, dispR, thresh=1.0, ctx=None):
    """PatchMatch::postProcess validity masks, methods/patchmatch.cpp:817-828."""
    a = np.ascontiguousarray(dispL, dtype=np.float32)
    b = np.ascontiguousarray(dispR, dtype=np.float32)
    vl = np.empty(a.shape, dtype=np.uint8)
    vr = np.empty(a.shape, dtype=np.uint8)
    check(load().mdisp_lr_check(_ctx(ctx), C.byref(image(a)), C.byref(image(b)), thresh, C.byref(image(vl)),
                                C.byref(image(vr))))
    return vl, vr


def initUndistortRectifyMap(K, dist, R, P, size):
    """cv::initUndistortRectifyMap(..., CV_16SC2): returns (map1 int16 HxWx2, map2 uint16 HxW)."""
    W, H = size
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
    P = np.ascontiguousarray(P, dtype=np.float64)
    if P.size == 9:
        P = np.hstack([P.reshape(3, 3), np.zeros((3, 1))])
    P = np.ascontiguousarray(P.reshape(12))
    dist = np.ascontiguousarray(dist, dtype=np.float64).reshape(-1)
    m1 = np.empty((H, W, 2), dtype=np.int16)
    m2 = np.empty((H, W), dtype=np.uint16)
    dp = C.POINTER(C.c_double)
    check(load().mdisp_build_rectify_maps(K.ctypes.data_as(dp), dist.ctypes.data_as(dp), dist.size,
                                          R.ctypes.data_as(dp), P.ctypes.data_as(dp), W, H, m1.ctypes.data,
                                          m2.ctypes.data))
    return m1, m2


class RectifyMaps:
    """Device-resident maps of cv::initUndistortRectifyMap(K, dist, R, P, size, CV_16SC2), built on the GPU once and
    reused for every frame: `maps.remap(img)` = cv::remap(img, ..., INTER_LINEAR)."""

    def __init__(self, K, dist, R, P, size, ctx=None):
        W, H = size
        self.size = (W, H)
        self._ctx = ctx
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
        P = np.ascontiguousarray(P, dtype=np.float64)
        if P.size == 9:
            P = np.hstack([P.reshape(3, 3), np.zeros((3, 1))])
        P = np.ascontiguousarray(P.reshape(12))
        dist = np.ascontiguousarray(dist, dtype=np.float64).reshape(-1)
        dp = C.POINTER(C.c_double)
        self._h = C.c_void_p()
        check(load().mdisp_rectify_maps_create(_ctx(ctx), K.ctypes.data_as(dp), dist.ctypes.data_as(dp), dist.size,
                                               R.ctypes.data_as(dp), P.ctypes.data_as(dp), W, H, C.byref(self._h)))

    def download(self):
        W, H = self.size
        m1 = np.empty((H, W, 2), dtype=np.int16)
        m2 = np.empty((H, W), dtype=np.uint16)
        check(load().mdisp_rectify_maps_download(_ctx(self._ctx), self._h, m1.ctypes.data, m2.ctypes.data))
        return m1, m2

    def remap(self, src):
        src = _u8(src)
        W, H = self.size
        dst = np.empty((H, W) + ((src.shape[2],) if src.ndim == 3 else ()), dtype=np.uint8)
        check(load().mdisp_remap_with(_ctx(self._ctx), self._h, C.byref(image(src)), C.byref(image(dst))))
        return dst

    def remap_dev(self, d_src, src_pitch, src_rows, src_cols, channels, d_dst, dst_pitch):
        check(load().mdisp_remap_dev(_ctx(self._ctx), self._h, d_src, src_pitch, src_rows, src_cols, channels, d_dst, dst_pitch))

    def close(self):
        if self._h:
            load().mdisp_rectify_maps_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stereoRectify(K1, D1, K2, D2, size, R, T, zero_disparity=True, alpha=-1.0):
    """cv::stereoRectify as called at methods/methods.cpp:1691-1694 (newImageSize = imageSize).
    Returns (R1, R2, P1, P2, Q, roi1, roi2)."""
    W, H = size
    dp = C.POINTER(C.c_double)
    f = lambda a, n: np.ascontiguousarray(a, dtype=np.float64).reshape(n)  # noqa: E731
    K1, K2, R, T = f(K1, 9), f(K2, 9), f(R, 9), f(T, 3)
    D1 = np.ascontiguousarray(D1, dtype=np.float64).reshape(-1)
    D2 = np.ascontiguousarray(D2, dtype=np.float64).reshape(-1)
    if D1.size != D2.size:
        raise ValueError("distortion vectors must have the same length")
    R1, R2, P1, P2, Q = (np.empty(n, dtype=np.float64) for n in (9, 9, 12, 12, 16))
    roi1, roi2 = (C.c_int * 4)(), (C.c_int * 4)()
    p = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    check(load().mdisp_stereo_rectify(p(K1), p(D1), p(K2), p(D2), D1.size, W, H, p(R), p(T), 1 if zero_disparity else 0,
                                      float(alpha), p(R1), p(R2), p(P1), p(P2), p(Q), roi1, roi2))
    return R1.reshape(3, 3), R2.reshape(3, 3), P1.reshape(3, 4), P2.reshape(3, 4), Q.reshape(4, 4), tuple(roi1), tuple(roi2)


def remap(src, map1, map2, ctx=None):
    """cv::remap(src, dst, map1, map2, INTER_LINEAR), constant-0 border."""
    src = _u8(src)
    m1 = np.ascontiguousarray(map1, dtype=np.int16)
    m2 = np.ascontiguousarray(map2, dtype=np.uint16)
    shape = m2.shape + ((src.shape[2],) if src.ndim == 3 else ())
    dst = np.empty(shape, dtype=np.uint8)
    check(load().mdisp_remap(_ctx(ctx), C.byref(image(src)), m1.ctypes.data, m2.ctypes.data, C.byref(image(dst))))
    return dst


def reprojectQ(disparityMap, Q, ctx=None):
    """methods/methods.cpp:2039-2076: Q * (row, col, d, 1), divided by W and by 100 -> H x W x 3 float32."""
    d = np.ascontiguousarray(disparityMap, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float64).reshape(16)
    xyz = np.empty(d.shape + (3,), dtype=np.float32)
    check(load().mdisp_reproject_q(_ctx(ctx), C.byref(image(d)), Q.ctypes.data_as(C.POINTER(C.c_double)), C.byref(image(xyz))))
    return xyz


def reprojectTriangulate(disparityMap, K_new, baseline, originBGRMap=None, ctx=None):
    """methods/methods.cpp:2300-2352 -> H x W x 3 float32 (points that the reference skips stay zero)."""
    d = np.ascontiguousarray(disparityMap, dtype=np.float32)
    K = np.ascontiguousarray(K_new, dtype=np.float64).reshape(9)
    xyz = np.empty(d.shape + (3,), dtype=np.float32)
    bgr = np.ascontiguousarray(originBGRMap, dtype=np.float32) if originBGRMap is not None else None
    check(load().mdisp_reproject_triangulate(_ctx(ctx), C.byref(image(d)), K.ctypes.data_as(C.POINTER(C.c_double)), float(baseline),
                                             C.byref(image(bgr)) if bgr is not None else None, C.byref(image(xyz))))
    return xyz


def synth_pair(seed, rows, cols, numDisp):
    """Synthetic rectified pair (left, right, ground-truth disparity)."""
    left = np.empty((rows, cols), dtype=np.uint8)
    right = np.empty((rows, cols), dtype=np.uint8)
    gt = np.empty((rows, cols), dtype=np.uint16)
    check(load().mdisp_synth_pair(seed, rows, cols, numDisp, left.ctypes.data, right.ctypes.data, gt.ctypes.data))
    return left, right, gt


# ---- namespace stereomatch_1 (methods/method_StereoMatching.h:94-112): fixed-window matchers ----
(LEGACY_AD, LEGACY_SAD, LEGACY_SAD_INTE, LEGACY_SAD_BF, LEGACY_TAD, LEGACY_STAD, LEGACY_STAD_INTE, LEGACY_STAD_BF,
 LEGACY_SD, LEGACY_SSD, LEGACY_SSD_INTE, LEGACY_SSD_BF) = range(12)


def _guided_multi(guidedImg, inputP, r, eps, variant, ctx):
    g = np.ascontiguousarray(guidedImg, np.uint8)
    P = np.ascontiguousarray(inputP)
    if P.dtype != np.uint8:
        P = np.ascontiguousarray(P, np.float32)
    if g.shape[:2] != P.shape[:2]:
        return None
    out = np.empty(g.shape[:2], np.float32)
    check(load().mdisp_guided_filter_multi(_ctx(ctx), C.byref(image(g)), C.byref(image(P)), r, eps, variant, C.byref(image(out))))
    return out


def getGuidedFilter_multi(guidedImg, inputP, r, eps, ctx=None):
    """disp_method/gifs.cpp:47-135 with a 3- or 6-channel uint8 guide -> float32, None on a size mismatch."""
    return _guided_multi(guidedImg, inputP, r, eps, 0, ctx)


def _legacy(method, leftImg, rightImg, winSize, threshold_T, minDisparity, numDisparity, ctx):
    L, R = np.ascontiguousarray(leftImg, np.uint8), np.ascontiguousarray(rightImg, np.uint8)
    if method not in (LEGACY_AD, LEGACY_TAD, LEGACY_SD) and winSize % 2 == 0:
        print("winsize must be odd")               # method_StereoMatching.cpp:99-103
        return None
    out = np.empty(L.shape[:2], dtype=np.uint8)
    check(load().mdisp_legacy_disparity(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), method, winSize, threshold_T, minDisparity,
                                        numDisparity, C.byref(image(out))))
    return out


DISPARITY_LEFT, DISPARITY_RIGHT = 0, 1          # methods/parametersStereo.h:5-9


class stereomatch_1:
    """The reference's namespace of the same name: same function names, argument order and defaults."""

    @staticmethod
    def computeSimilarity(leftImg, rightImg, cost_d_imgs, regularity, thresC, thresG, dispType=DISPARITY_LEFT, winSize=None,
                          minDisparity=0, numDisparity=30, ctx=None):
        """Both overloads of computeSimilarity (methods/method_StereoMatching.h): `cost_d_imgs` (a list) is cleared and
        receives numDisparity float32 slices -- bordered by winSize // 2 when a window is given (the ten-argument overload).
        Left untouched for an even window, cleared only for mismatching sizes, as in the reference; inputs on which the
        reference fails (gray views, DISPARITY_RIGHT) raise."""
        L, R = np.ascontiguousarray(leftImg, np.uint8), np.ascontiguousarray(rightImg, np.uint8)
        if winSize is not None and winSize % 2 == 0:
            return
        del cost_d_imgs[:]
        if L.shape != R.shape:
            return
        H, W = L.shape[:2]
        b = 0 if winSize is None else winSize // 2
        vol = np.empty((numDisparity, H + 2 * b, W + 2 * b), np.float32)
        check(load().mdisp_compute_similarity(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), 1 if dispType == DISPARITY_LEFT else 0,
                                              regularity, thresC, thresG, 0 if winSize is None else winSize, minDisparity, numDisparity,
                                              vol.ctypes.data))
        cost_d_imgs.extend(vol[k] for k in range(numDisparity))

    @staticmethod
    def _guidedf_n(which, leftImg, rightImg, dispType, eps, winSize, minDisparity, numDisparity, ctx):
        L, R = np.ascontiguousarray(leftImg, np.uint8), np.ascontiguousarray(rightImg, np.uint8)
        out = np.empty(L.shape[:2], np.float32)
        check(load().mdisp_adaptive_weight_guidedf_n(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), which, 1 if dispType == DISPARITY_LEFT else 0,
                                                     eps, winSize, minDisparity, numDisparity, C.byref(image(out))))
        return out

    @staticmethod
    def computeAdaptiveWeight_GuidedF_2(leftImg, rightImg, dispType=DISPARITY_LEFT, eps=0.01, winSize=35, minDisparity=186, numDisparity=144,
                                        ctx=None):
        """methods/method_StereoMatching.h:182-184."""
        return stereomatch_1._guidedf_n(2, leftImg, rightImg, dispType, eps, winSize, minDisparity, numDisparity, ctx)

    @staticmethod
    def computeAdaptiveWeight_GuidedF_3(leftImg, rightImg, dispType=DISPARITY_LEFT, eps=1e-6, winSize=35, minDisparity=186, numDisparity=144,
                                        ctx=None):
        """methods/method_StereoMatching.h:185-187."""
        return stereomatch_1._guidedf_n(3, leftImg, rightImg, dispType, eps, winSize, minDisparity, numDisparity, ctx)

    @staticmethod
    def getCostSAD_d(leftImg, rightImg, disparity, dispType=DISPARITY_LEFT, winSize=35, ctx=None):
        """methods/method_StereoMatching.h:171: one of the views is the bordered (wider) one -> float32 slice, or None where
        the reference returns an empty image."""
        L, R = np.ascontiguousarray(leftImg, np.uint8), np.ascontiguousarray(rightImg, np.uint8)
        if winSize % 2 == 0:
            print("winsize must be odd")
            return None
        ref, oth = (L, R) if dispType == DISPARITY_LEFT else (R, L)
        if oth.shape[1] <= ref.shape[1]:
            return None
        out = np.empty(ref.shape[:2], np.float32)
        check(load().mdisp_cost_sad_d(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), disparity, 1 if dispType == DISPARITY_LEFT else 0,
                                      winSize, C.byref(image(out))))
        return out

    @staticmethod
    def getGuidedFilter(guidedImg, inputP, r, eps, ctx=None):
        """methods/method_StereoMatching.h:178 with a 3- or 6-channel uint8 guide -> float32, None on a size mismatch."""
        return _guided_multi(guidedImg, inputP, r, eps, 1, ctx)

    @staticmethod
    def computeAdaptiveWeight_GuidedF(leftImg, rightImg, dispType=DISPARITY_LEFT, eps=0.01, winSize=35, minDisparity=186, numDisparity=144,
                                      ctx=None):
        """methods/method_StereoMatching.h:179-181 on H x W x 3 uint8 images -> float32 map."""
        L, R = np.ascontiguousarray(leftImg, np.uint8), np.ascontiguousarray(rightImg, np.uint8)
        out = np.empty(L.shape[:2], np.float32)
        check(load().mdisp_adaptive_weight_guidedf(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), 1 if dispType == DISPARITY_LEFT else 0,
                                                   eps, winSize, minDisparity, numDisparity, C.byref(image(out))))
        return out

    @staticmethod
    def computeNCC(leftImg, rightImg, dispType=DISPARITY_LEFT, winSize=7, minDisparity=0, numDisparity=30, ctx=None, cost_ds=None):
        """Both overloads of computeNCC (methods/method_StereoMatching.h:116-120).  Without `cost_ds` the float32 map is
        returned (None where the reference returns an empty image); with `cost_ds` (a list) the list is cleared and
        receives numDisparity normalised float32 slices, and nothing is returned."""
        L, R = np.ascontiguousarray(leftImg, np.uint8), np.ascontiguousarray(rightImg, np.uint8)
        if L.shape[:2] != R.shape[:2] or winSize % 2 == 0:
            return None
        H, W = L.shape[:2]
        if cost_ds is None:
            out = np.empty((H, W), np.float32)
            check(load().mdisp_ncc(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), 1 if dispType == DISPARITY_LEFT else 0, winSize,
                                   minDisparity, numDisparity, C.byref(image(out)), None))
            return out
        vol = np.empty((numDisparity, H, W), np.float32)
        check(load().mdisp_ncc(_ctx(ctx), C.byref(image(L)), C.byref(image(R)), 1 if dispType == DISPARITY_LEFT else 0, winSize,
                               minDisparity, numDisparity, None, vol.ctypes.data))
        del cost_ds[:]
        cost_ds.extend(vol[k] for k in range(numDisparity))
        return None

    @staticmethod
    def computeAD(leftImg, rightImg, minDisparity=0, numDisparity=30, ctx=None):
        return _legacy(LEGACY_AD, leftImg, rightImg, 1, 0, minDisparity, numDisparity, ctx)

    @staticmethod
    def computeSAD(leftImg, rightImg, winSize=7, minDisparity=0, numDisparity=30, ctx=None):
        return _legacy(LEGACY_SAD, leftImg, rightImg, winSize, 0, minDisparity, numDisparity, ctx)

    @staticmethod
    def computeSAD_inteOpti(leftImg, rightImg, winSize=7, minDisparity=0, numDisparity=30, ctx=None):
        return _legacy(LEGACY_SAD_INTE, leftImg, rightImg, winSize, 0, minDisparity, numDisparity, ctx)

    @staticmethod
    def computeSAD_BFOpti(leftImg, rightImg, winSize=7, minDisparity=0, numDisparity=30, ctx=None):
        return _legacy(LEGACY_SAD_BF, leftImg, rightImg, winSize, 0, minDisparity, numDisparity, ctx)

    @staticmethod
    def computeTAD(leftImg, rightImg, threshold_T=30, minDisparity=0, numDisparity=30, ctx=None):
        return _legacy(LEGACY_TAD, leftImg, rightImg, 1, threshold_T, minDisparity, numDisparity, ctx)

    @staticmethod
    def computeSTAD(leftImg, rightImg, winSize=7, threshold_T=30, minDisparity=0, numDisparity=30, ctx=None):
        return _legacy(LEGACY_STAD, leftImg, rightImg, winSize, threshold_T, minDisparity, numDisparity, ctx)

    @staticmethod
    def computeSTAD_inteOpti(leftImg, rightImg, winSize=7, threshold_T=30, minDisparity=0, numDisparity=30, ctx=None):
        return _legacy(LEGACY_STAD_INTE, leftImg, rightImg, winSize, threshold_T, minDisparity, numDisparity, ctx)

    @staticmethod
    def computeSTAD_BFOpti(leftImg, rightImg, winSize=7, threshold_T=30, minDisparity=0, numDisparity=30, ctx=None):
        return _legacy(LEGACY_STAD_BF, leftImg, rightImg, winSize, threshold_T, minDisparity, numDisparity, ctx)

    @staticmethod
    def computeSD(leftImg, rightImg, minDisparity=0, numDisparity=30, ctx=None):
        return _legacy(LEGACY_SD, leftImg, rightImg, 1, 0, minDisparity, numDisparity, ctx)

    @staticmethod
    def computeSSD(leftImg, rightImg, winSize=7, minDisparity=0, numDisparity=30, ctx=None):
        return _legacy(LEGACY_SSD, leftImg, rightImg, winSize, 0, minDisparity, numDisparity, ctx)

    @staticmethod
    def computeSSD_inteOpti(leftImg, rightImg, winSize=7, minDisparity=0, numDisparity=30, ctx=None):
        return _legacy(LEGACY_SSD_INTE, leftImg, rightImg, winSize, 0, minDisparity, numDisparity, ctx)

    @staticmethod
    def computeSSD_BFOpti(leftImg, rightImg, winSize=7, minDisparity=0, numDisparity=30, ctx=None):
        return _legacy(LEGACY_SSD_BF, leftImg, rightImg, winSize, 0, minDisparity, numDisparity, ctx)
