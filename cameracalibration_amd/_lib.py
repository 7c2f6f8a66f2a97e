"""ctypes binding of libmdisp.so (the C ABI declared in include/mdisp.h).

The library is the product; this module only loads it and declares the prototypes.  It fails
loudly when the shared object is missing -- there is no Python or CPU fallback."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MDISP_LIB_PATH: a differently built library (kernel experiments, scripts/exp_build.sh); the default is the in-tree build
LIB_PATH = os.environ.get("MDISP_LIB_PATH") or os.path.join(HERE, "lib", "libmdisp.so")

MDISP_8U, MDISP_8S, MDISP_16U, MDISP_16S, MDISP_32S, MDISP_32F, MDISP_64F = range(7)
_DEPTH_OF = {np.dtype(np.uint8): MDISP_8U, np.dtype(np.int8): MDISP_8S, np.dtype(np.uint16): MDISP_16U,
             np.dtype(np.int16): MDISP_16S, np.dtype(np.int32): MDISP_32S, np.dtype(np.float32): MDISP_32F,
             np.dtype(np.float64): MDISP_64F}

E_BADARG, E_NODEVICE, E_HIP, E_UNSUPPORTED, E_RANGE, E_NOMEM = -1, -2, -3, -4, -5, -6
# keys of mdisp_tuning_set (include/mdisp.h)
(TUNE_SGM_SPLIT_MAX, TUNE_SGM_PATH_WAVES_MAX, TUNE_SGM_STRIP_LANES, TUNE_SGM_FORCE_PREPASS, TUNE_SGM_LANES16,
 TUNE_SGM_PER_DIRECTION, TUNE_SGBM_BT_SEQUENTIAL, TUNE_SGM_PIPELINE, TUNE_SGM_PLANT_ERROR, TUNE_SGM_BAND_SLACK, TUNE_SGM_PAIR_GROUP,
 TUNE_SGM_HAND_RING, TUNE_SGM_MEET_MAX, TUNE_SGM_ONE_LAUNCH, TUNE_SAT_INT_TABLES, TUNE_SGM_RING_HANDOFF, TUNE_SGBM_BATCH_LANES,
 TUNE_GIF_MC_TILES, TUNE_COUNT) = range(19)


class mdisp_image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("channels", C.c_int),
                ("depth", C.c_int), ("step_bytes", C.c_size_t)]


class mdisp_param(C.Structure):
    _fields_ = [("winSize", C.c_int), ("minDisparity", C.c_int), ("maxDisparity", C.c_int), ("isDispLeft", C.c_int)]


class mdisp_stereosgbm_params(C.Structure):
    """cv::StereoSGBM::create arguments, in its order"""
    _fields_ = [(n, C.c_int) for n in ("minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff", "preFilterCap",
                                       "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode")]


class mdisp_stereobm_params(C.Structure):
    """cv::StereoBM::create(numDisparities, blockSize) and its setters"""
    _fields_ = [(n, C.c_int) for n in ("minDisparity", "numDisparities", "blockSize", "preFilterCap", "textureThreshold",
                                       "uniquenessRatio", "speckleWindowSize", "speckleRange", "disp12MaxDiff")]


class mdisp_sgbm_params(C.Structure):
    _fields_ = [("minDisparity", C.c_int), ("numDisparities", C.c_int), ("P1", C.c_int), ("P2", C.c_int),
                ("uniquenessRatio", C.c_int), ("disp12MaxDiff", C.c_int), ("paths", C.c_int), ("subpixel", C.c_int),
                ("costWin", C.c_int), ("censusMethod", C.c_int)]


# every symbol include/mdisp.h declares
EXPORTS = [
    "mdisp_version", "mdisp_last_error", "mdisp_device_count", "mdisp_create", "mdisp_destroy", "mdisp_synchronize",
    "mdisp_census_transform", "mdisp_hamming_dist", "mdisp_census_stereo", "mdisp_sad_inte", "mdisp_sad_box",
    "mdisp_guided_filter", "mdisp_steer_filter", "mdisp_asw_gifs", "mdisp_sgbm", "mdisp_sgbm_volume", "mdisp_lr_check",
    "mdisp_build_rectify_maps", "mdisp_stereo_rectify", "mdisp_remap", "mdisp_reproject_q", "mdisp_reproject_triangulate", "mdisp_sgbm_workspace_bytes", "mdisp_sgbm_dev",
    "mdisp_sgbm_batch_workspace_bytes", "mdisp_sgbm_batch_dev", "mdisp_sgbm_rows_geometry", "mdisp_sgbm_rows_sweep_dev",
    "mdisp_census_stereo_dev", "mdisp_asw_gifs_band_dev", "mdisp_last_kernel_ms", "mdisp_enable_timing", "mdisp_synth_pair",
    "mdisp_stereosgbm_compute", "mdisp_stereosgbm_stages", "mdisp_median_blur3_s16", "mdisp_filter_speckles_s16",
    "mdisp_stereosgbm_workspace_bytes", "mdisp_stereosgbm_dev", "mdisp_stereosgbm_batch_workspace_bytes", "mdisp_stereosgbm_batch_dev", "mdisp_sgbm_batch", "mdisp_legacy_disparity", "mdisp_ncc",
    "mdisp_asw_gifs_band_stats_dev", "mdisp_rectify_maps_create", "mdisp_rectify_maps_destroy", "mdisp_rectify_maps_download", "mdisp_remap_with", "mdisp_remap_dev",
    "mdisp_census_stereo_dev_noalloc", "mdisp_create_on_stream", "mdisp_tuning_set", "mdisp_tuning_get",
    "mdisp_copy_bandwidth_dev", "mdisp_stereobm_compute", "mdisp_stereobm_workspace_bytes", "mdisp_stereobm_dev", "mdisp_fgs_filter",
    "mdisp_context_device_bytes", "mdisp_sad_inte_mode",
    "mdisp_cost_sad_d", "mdisp_guided_filter_multi", "mdisp_adaptive_weight_guidedf",
    "mdisp_sgbm_batch_multi", "mdisp_sgbm_rows_multi", "mdisp_compute_similarity", "mdisp_adaptive_weight_guidedf_n",
]

_lib = None


def load():
    """Load libmdisp.so once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m cameracalibration_amd.build` (needs hipcc). "
                           "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    img, par, sgp = P(mdisp_image), P(mdisp_param), P(mdisp_sgbm_params)
    vp = C.c_void_p
    lib.mdisp_version.restype = C.c_int
    lib.mdisp_last_error.restype = C.c_char_p
    lib.mdisp_device_count.restype = C.c_int
    lib.mdisp_create.argtypes = [P(vp), C.c_int, vp]
    lib.mdisp_create_on_stream.argtypes = [P(vp), C.c_int, vp]
    lib.mdisp_tuning_set.argtypes = [C.c_int, C.c_int]
    lib.mdisp_tuning_get.argtypes = [C.c_int]
    lib.mdisp_destroy.argtypes = [vp]
    lib.mdisp_destroy.restype = None
    lib.mdisp_synchronize.argtypes = [vp]
    lib.mdisp_census_transform.argtypes = [vp, img, C.c_int, C.c_int, img]
    lib.mdisp_hamming_dist.argtypes = [vp, img, img, img]
    lib.mdisp_census_stereo.argtypes = [vp, img, img, par, C.c_int, C.c_int, img]
    lib.mdisp_sad_inte.argtypes = [vp, img, img, par, img]
    lib.mdisp_sad_inte_mode.argtypes = [vp, img, img, par, C.c_int, img]
    lib.mdisp_sad_box.argtypes = [vp, img, img, par, img]
    lib.mdisp_guided_filter.argtypes = [vp, img, img, C.c_int, C.c_double, C.c_int, img, C.c_int, C.c_double, img, img]
    lib.mdisp_steer_filter.argtypes = [vp, img, img]
    lib.mdisp_asw_gifs.argtypes = [vp, img, img, par, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                   vp, img]
    lib.mdisp_sgbm.argtypes = [vp, img, img, sgp, img]
    lib.mdisp_sgbm_volume.argtypes = [vp, img, img, sgp, vp]
    lib.mdisp_lr_check.argtypes = [vp, img, img, C.c_float, img, img]
    lib.mdisp_build_rectify_maps.argtypes = [P(C.c_double), P(C.c_double), C.c_int, P(C.c_double), P(C.c_double),
                                             C.c_int, C.c_int, vp, vp]
    dp = P(C.c_double)
    lib.mdisp_stereo_rectify.argtypes = [dp, dp, dp, dp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_int, C.c_double,
                                         dp, dp, dp, dp, dp, P(C.c_int), P(C.c_int)]
    lib.mdisp_reproject_q.argtypes = [vp, img, dp, img]
    lib.mdisp_reproject_triangulate.argtypes = [vp, img, dp, C.c_double, img, img]
    lib.mdisp_remap.argtypes = [vp, img, vp, vp, img]
    lib.mdisp_sgbm_workspace_bytes.argtypes = [C.c_int, C.c_int, sgp]
    lib.mdisp_sgbm_workspace_bytes.restype = C.c_size_t
    ssp = P(mdisp_stereosgbm_params)
    lib.mdisp_stereosgbm_compute.argtypes = [vp, img, img, ssp, img]
    lib.mdisp_stereosgbm_stages.argtypes = [vp, img, img, ssp, img, vp, img]
    lib.mdisp_median_blur3_s16.argtypes = [vp, img, img]
    lib.mdisp_filter_speckles_s16.argtypes = [vp, img, C.c_int, C.c_int, C.c_int]
    lib.mdisp_stereosgbm_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, ssp]
    lib.mdisp_stereosgbm_workspace_bytes.restype = C.c_size_t
    lib.mdisp_stereosgbm_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, ssp, vp, vp, C.c_size_t]
    lib.mdisp_stereosgbm_batch_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, ssp, C.c_int]
    lib.mdisp_stereosgbm_batch_workspace_bytes.restype = C.c_size_t
    lib.mdisp_stereosgbm_batch_dev.argtypes = [vp, vp, vp, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_size_t, ssp, vp, vp, C.c_size_t]
    dp = C.POINTER(C.c_double)
    lib.mdisp_rectify_maps_create.argtypes = [vp, dp, dp, C.c_int, dp, dp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.mdisp_rectify_maps_destroy.argtypes = [vp]
    lib.mdisp_rectify_maps_destroy.restype = None
    lib.mdisp_rectify_maps_download.argtypes = [vp, vp, vp, vp]
    lib.mdisp_remap_with.argtypes = [vp, vp, img, img]
    lib.mdisp_remap_dev.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
    lib.mdisp_legacy_disparity.argtypes = [vp, img, img, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, img]
    lib.mdisp_ncc.argtypes = [vp, img, img, C.c_int, C.c_int, C.c_int, C.c_int, img, vp]
    lib.mdisp_sgbm_batch.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, sgp, vp, C.c_int]
    lib.mdisp_sgbm_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, sgp, vp, vp, C.c_size_t]
    lib.mdisp_sgbm_batch_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, sgp]
    lib.mdisp_sgbm_batch_workspace_bytes.restype = C.c_size_t
    lib.mdisp_sgbm_batch_dev.argtypes = [vp, vp, vp, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_size_t, sgp, vp, vp, C.c_size_t]
    lib.mdisp_sgbm_rows_geometry.argtypes = [C.c_int, C.c_int, sgp, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.mdisp_sgbm_rows_sweep_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, sgp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_size_t]
    lib.mdisp_census_stereo_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, par, C.c_int, C.c_int, vp]
    lib.mdisp_census_stereo_dev_noalloc.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, par, C.c_int, C.c_int, vp]
    lib.mdisp_asw_gifs_band_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, par, C.c_double, C.c_int, C.c_int, C.c_double,
                                            C.c_int, C.c_int, C.c_int, vp, vp, vp]
    lib.mdisp_asw_gifs_band_stats_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, par, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.mdisp_last_kernel_ms.argtypes = [vp, C.c_int, P(C.c_float)]
    lib.mdisp_enable_timing.argtypes = [vp, C.c_int]
    lib.mdisp_context_device_bytes.argtypes = [vp]
    lib.mdisp_compute_similarity.argtypes = [vp, img, img, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, vp]
    lib.mdisp_adaptive_weight_guidedf_n.argtypes = [vp, img, img, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, img]
    lib.mdisp_sgbm_batch_multi.argtypes = [P(C.c_int), C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, P(mdisp_sgbm_params), vp,
                                           C.c_int, C.c_int]
    lib.mdisp_sgbm_rows_multi.argtypes = [P(C.c_int), C.c_int, img, img, P(mdisp_sgbm_params), img]
    lib.mdisp_cost_sad_d.argtypes = [vp, img, img, C.c_int, C.c_int, C.c_int, img]
    lib.mdisp_guided_filter_multi.argtypes = [vp, img, img, C.c_int, C.c_double, C.c_int, img]
    lib.mdisp_adaptive_weight_guidedf.argtypes = [vp, img, img, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, img]
    lib.mdisp_context_device_bytes.restype = C.c_size_t
    sbp = P(mdisp_stereobm_params)
    lib.mdisp_stereobm_compute.argtypes = [vp, img, img, sbp, img]
    lib.mdisp_stereobm_workspace_bytes.argtypes = [C.c_int, C.c_int, sbp]
    lib.mdisp_stereobm_workspace_bytes.restype = C.c_size_t
    lib.mdisp_stereobm_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, sbp, vp, vp, C.c_size_t]
    lib.mdisp_fgs_filter.argtypes = [vp, img, img, C.c_double, C.c_double, C.c_double, C.c_int, img]
    lib.mdisp_copy_bandwidth_dev.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, P(C.c_float)]
    lib.mdisp_synth_pair.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    _lib = lib
    return lib


class MdispError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mdisp error {code}: {msg}")
        self.code = code
        self.msg = msg


def check(rc):
    if rc != 0:
        raise MdispError(rc, load().mdisp_last_error().decode("utf-8", "replace"))


def image(arr):
    """View a C-contiguous-rows numpy array (H x W or H x W x C) as an mdisp_image."""
    if arr.ndim == 2:
        rows, cols, ch = arr.shape[0], arr.shape[1], 1
    elif arr.ndim == 3:
        rows, cols, ch = arr.shape
    else:
        raise ValueError("image must be 2-D or 3-D")
    if arr.strides[-1] != arr.itemsize or (arr.ndim == 3 and arr.strides[1] != arr.itemsize * ch):
        raise ValueError("image rows must be contiguous")
    return mdisp_image(arr.ctypes.data, rows, cols, ch, _DEPTH_OF[arr.dtype], arr.strides[0])


class Context:
    """Owns one mdisp_ctx (one HIP stream + workspace)."""

    def __init__(self, device=-1, stream=None):
        """stream=None: the context creates its own stream; an integer (0 = the legacy default stream included):
        the hipStream_t to enqueue on."""
        self._h = C.c_void_p()
        if stream is None:
            check(load().mdisp_create(C.byref(self._h), device, None))
        else:
            check(load().mdisp_create_on_stream(C.byref(self._h), device, C.c_void_p(stream)))

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        check(load().mdisp_synchronize(self._h))

    def close(self):
        if self._h:
            load().mdisp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tuning_set(key, value):
    check(load().mdisp_tuning_set(key, value))


def tuning_reset():
    for k in range(TUNE_COUNT):
        check(load().mdisp_tuning_set(k, -1))


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx
