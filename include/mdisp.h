/*
 * mdisp -- C ABI of the MI355X (gfx950) dense-stereo disparity library, libmdisp.so.
 *
 * Every entry point replaces one function (or one OpenCV call site) of the reference's dense
 * stereo path; the reference interface is cited as file:line relative to
 * cameraCalibration/ in the reference repository.  Signatures use plain pointers and sizes only.
 *
 * Conventions
 *   - all functions return 0 on success or a negative MDISP_E_* code; they never throw.
 *   - "host" entry points take host buffers (what a cv::Mat holds), copy in, compute on the GPU and
 *     copy the result back into a caller-allocated output image.
 *   - "_dev" entry points take device pointers that are already resident in HBM and enqueue the
 *     work on the context's stream without synchronising; outputs stay on the device.
 *   - a context owns one HIP stream and a grow-only device workspace; calls on one context are
 *     serialised by that stream, two contexts may be used concurrently from two host threads (the
 *     reference calls its L and R disparity routines from two std::async threads,
 *     cameraCalibration.cpp:163-190).
 *   - there is no CPU fallback: without a HIP device every compute entry point fails with
 *     MDISP_E_NODEVICE.
 */
#ifndef MDISP_H
#define MDISP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ---- */
#define MDISP_OK            0
#define MDISP_E_BADARG     -1   /* null pointer, size mismatch, even winSize, bad enum ...        */
#define MDISP_E_NODEVICE   -2   /* no HIP device / HIP runtime error at context creation        */
#define MDISP_E_HIP        -3   /* a HIP call failed; see mdisp_last_error()                     */
#define MDISP_E_UNSUPPORTED -4  /* e.g. CENSUS_2017, which is undefined behaviour in the reference */
#define MDISP_E_RANGE      -5   /* parameters overflow the 16-bit path-cost range                */
#define MDISP_E_NOMEM      -6

/* ---- pixel depths (numeric values equal OpenCV's CV_8U ... CV_64F) ---- */
#define MDISP_8U  0
#define MDISP_8S  1
#define MDISP_16U 2
#define MDISP_16S 3
#define MDISP_32S 4
#define MDISP_32F 5
#define MDISP_64F 6

/* POD view of a cv::Mat (data, rows, cols, channels(), depth(), step[0]). */
typedef struct mdisp_image {
    void* data;
    int rows, cols, channels, depth;
    size_t step_bytes;
} mdisp_image;

/* StereoMatchParam without the images, disp_method/methods_disp.h:5-16.
 * maxDisparity is INCLUSIVE: numDisp = maxDisparity - minDisparity + 1 (methods_disp.cpp:219). */
typedef struct mdisp_param {
    int winSize;
    int minDisparity;
    int maxDisparity;
    int isDispLeft;
} mdisp_param;

/* CENSUS_ALGORITHM, disp_method/methods_disp.h:18-27 (same numeric values). */
enum {
    MDISP_BASIC_CENSUS = 0,
    MDISP_CIRCLE_CENSUS = 1,
    MDISP_ROTATION_INVARIANT_CENSUS = 2,
    MDISP_UNIFORM_CENSUS = 3,
    MDISP_MULTISCALE_CENSUS = 4,
    MDISP_STATISTIC_MULTISCALE_CENSUS = 5,
    MDISP_CENSUS_2017 = 6
};

/* Aggregation of each Hamming slice inside censusStereo (methods_disp.cpp:969-982).  GUIDED is
 * what the reference runs (getGuidedFilter(img, slice, winSize, 1e-6), :978); SQRBOX is the
 * alternative left in its comments (:979-980); BOX and NONE are the integer-exact modes. */
enum {
    MDISP_AGG_NONE = 0,
    MDISP_AGG_BOX = 1,
    MDISP_AGG_SQRBOX = 2,
    MDISP_AGG_GUIDED = 3
};

/* GIF_TYPE, disp_method/gifs.h:4-13 (same numeric values).  OURS_GIF / OURS_GIF2 / OURS_GIF3
 * (disp_method/gifs.cpp:620-669, 672-725, 728-804) smooth the coefficients a, b with
 * cv::bilateralFilter(sigmaColor 2, sigmaSpace 3, diameter r2 / r2 / r) instead of a box mean. */
enum {
    MDISP_GIF = 0,
    MDISP_WGIF = 1,
    MDISP_EGIF = 2,
    MDISP_SKWGIF = 3,
    MDISP_OURS_GIF = 4,
    MDISP_OURS_GIF2 = 5,
    MDISP_OURS_GIF3 = 6
};

/* cost source of mdisp_asw_gifs */
enum {
    MDISP_COST_VOLUME = 0,   /* caller supplies D float slices (what computeDisp_Feature returns)  */
    MDISP_COST_CENSUS = 1,   /* Hamming distance of BASIC_CENSUS codes, as in censusStereo         */
    MDISP_COST_ABSDIFF = 2   /* |L - R shifted|, as in computeSAD_* (methods_disp.cpp:232-234)      */
};

/* Parameters of cv::StereoSGBM::create as used at methods/methods.cpp:1893 and :1216-1219, plus
 * the choice of matching cost.  Zero / negative values select StereoSGBM's defaults
 * (P1 -> 2, P2 -> max(5, P1+1), uniquenessRatio -> 10, disp12MaxDiff -> 1). */
typedef struct mdisp_sgbm_params {
    int minDisparity;
    int numDisparities;     /* a COUNT (unlike mdisp_param.maxDisparity)                         */
    int P1, P2;
    int uniquenessRatio;
    int disp12MaxDiff;
    int paths;              /* 8 (all directions), 5 (MODE_SGBM set) or 4 (axis-aligned)         */
    int subpixel;           /* 1: x16 parabola refinement, 0: disparity*16                        */
    int costWin;            /* 1: per-pixel Hamming cost; odd >1: box-summed over costWin^2       */
    int censusMethod;       /* MDISP_BASIC_CENSUS / ROTATION_INVARIANT / UNIFORM                  */
} mdisp_sgbm_params;

/* The fixed-window matchers of namespace stereomatch_1, methods/method_StereoMatching.h:94-112
 * (method_StereoMatching.cpp:32-869). */
enum {
    MDISP_LEGACY_AD = 0, MDISP_LEGACY_SAD = 1, MDISP_LEGACY_SAD_INTE = 2, MDISP_LEGACY_SAD_BF = 3,
    MDISP_LEGACY_TAD = 4, MDISP_LEGACY_STAD = 5, MDISP_LEGACY_STAD_INTE = 6, MDISP_LEGACY_STAD_BF = 7,
    MDISP_LEGACY_SD = 8, MDISP_LEGACY_SSD = 9, MDISP_LEGACY_SSD_INTE = 10, MDISP_LEGACY_SSD_BF = 11
};

/* Arguments of cv::StereoSGBM::create, in its order (methods/methods.cpp:1893: create(0, 48, 7);
 * :1216-1219, :1565-1568: create(0,64,7,490,1960,1,63,10,100,32,MODE_SGBM)), for the entry points
 * that run StereoSGBM with its own Birchfield-Tomasi block cost. */
enum { MDISP_SGBM_MODE_SGBM = 0, MDISP_SGBM_MODE_HH = 1 };   /* StereoSGBM::MODE_SGBM (5 paths), MODE_HH (8 paths) */
typedef struct mdisp_stereosgbm_params {
    int minDisparity;       /* >= 0                                                                */
    int numDisparities;     /* positive multiple of 16                                             */
    int blockSize;          /* odd; <= 0 selects 5                                                 */
    int P1, P2;             /* <= 0 select 2 and 5; P2 is raised to P1 + 1                         */
    int disp12MaxDiff;      /* <= 0 selects 1                                                      */
    int preFilterCap;       /* clip of the x-Sobel response: max(preFilterCap, 15) | 1             */
    int uniquenessRatio;    /* < 0 selects 10; above 32767: MDISP_E_RANGE                          */
    int speckleWindowSize;  /* 0 disables the speckle filter                                       */
    int speckleRange;
    int mode;               /* MDISP_SGBM_MODE_*                                                   */
} mdisp_stereosgbm_params;

typedef struct mdisp_ctx mdisp_ctx;

/* ---- library / context ---- */
int         mdisp_version(void);
const char* mdisp_last_error(void);              /* thread-local message of the last failure    */
int         mdisp_device_count(void);
/* device < 0 selects the current HIP device.  stream: a hipStream_t to enqueue on, or NULL to
 * let the context create its own. */
int  mdisp_create(mdisp_ctx** ctx, int device, void* stream);
/* Same, but `stream` is always the stream to enqueue on, the legacy default stream (NULL) included: for callers
 * whose producers run on the default stream and rely on its ordering.  No counterpart in the reference. */
int  mdisp_create_on_stream(mdisp_ctx** ctx, int device, void* stream);
void mdisp_destroy(mdisp_ctx* ctx);
int  mdisp_synchronize(mdisp_ctx* ctx);
/* Bytes of device memory the context has allocated so far for its grow-only workspace and copy staging (0 for a fresh
 * context).  Diagnostic, no counterpart in the reference: lets a caller verify that a call allocated nothing. */
size_t mdisp_context_device_bytes(mdisp_ctx* ctx);

/* ---- census family: countCensusImg*, methods_disp.cpp:415-714 ----
 * dst is CV_8U of src's size.  `arg` is the radius for CIRCLE and the window for the
 * MULTISCALE variants, ignored otherwise. */
int mdisp_census_transform(mdisp_ctx*, const mdisp_image* src, int method, int arg, mdisp_image* dst);

/* countHummingDist, methods_disp.cpp:808-838: two CV_8U images -> CV_32S popcount(a ^ b). */
int mdisp_hamming_dist(mdisp_ctx*, const mdisp_image* a, const mdisp_image* b, mdisp_image* dst);

/* censusStereo(StereoMatchParam, CENSUS_ALGORITHM), methods_disp.h:62, methods_disp.cpp:845-1145.
 * L/R: CV_8UC1 or CV_8UC3 (BGR, converted as checkImg does, :130-147).  out: CV_64F HxW. */
int mdisp_census_stereo(mdisp_ctx*, const mdisp_image* L, const mdisp_image* R, const mdisp_param* p,
                        int census_method, int agg_mode, mdisp_image* out);

/* computeSAD_inteOpti / computeSAD_BFOpti, methods_disp.h:43-44, methods_disp.cpp:207-408.
 * out: CV_8U HxW (value stored modulo 256 like the reference's .at<char>).
 * mdisp_sad_inte follows the reference's integral(..., CV_32F) tables (:238, :282) in their own arithmetic: per row a
 * float running sum s, sum[y+1][x+1] = sum[y][x+1] + s, the four corners combined left to right in float (:253-256),
 * zero outside the table (getMatVal, :159-171).  Such tables round once an entry passes 2^24 (a 255-valued difference
 * image does after 65793 pixels; main() works at 1280x720) and the winner follows the rounding -- reproduced here.
 * mdisp_sad_inte_mode selects the arithmetic per call: MDISP_SAT_F32 (the above) or MDISP_SAT_INT, exact 32-bit integer
 * tables with the same 2h x 2h footprint and zero-outside rule; the two agree while every table entry stays below 2^24.
 * MDISP_TUNE_SAT_INT_TABLES switches mdisp_sad_inte and the stereomatch_1 *_inteOpti matchers to the integer tables. */
enum { MDISP_SAT_F32 = 0, MDISP_SAT_INT = 1 };
int mdisp_sad_inte(mdisp_ctx*, const mdisp_image* L, const mdisp_image* R, const mdisp_param* p, mdisp_image* out);
int mdisp_sad_inte_mode(mdisp_ctx*, const mdisp_image* L, const mdisp_image* R, const mdisp_param* p, int table_mode, mdisp_image* out);
int mdisp_sad_box(mdisp_ctx*, const mdisp_image* L, const mdisp_image* R, const mdisp_param* p, mdisp_image* out);

/* getGuidedFilter / _wgif / _egif / _skwgif2, gifs.h:17-33, gifs.cpp:47-135,227-273,376-409,537-617.
 * guide: CV_8UC1, P: CV_8U / CV_32S / CV_32F single channel, out (and beta): CV_32F.
 * steerW: CV_64F map from mdisp_steer_filter (SKWGIF only).  beta may be NULL. */
int mdisp_guided_filter(mdisp_ctx*, const mdisp_image* guide, const mdisp_image* P, int r, double eps,
                        int gif_type, const mdisp_image* steerW, int r2, double namuda,
                        mdisp_image* out, mdisp_image* beta);

/* steerFilter(src, dst), steerFilter.h:39, steerFilter.cpp:239-446.  src CV_8UC1, dst CV_64F. */
int mdisp_steer_filter(mdisp_ctx*, const mdisp_image* src, mdisp_image* dst);

/* asw_gifs(StereoMatchParam, eps, GIF_TYPE, r2, namda, h), methods_disp.h:66, methods_disp.cpp:1147-1297.
 * cost_source MDISP_COST_VOLUME: `costs` points to numDisp contiguous CV_32F HxW slices (the
 * output of computeDisp_Feature, which stays on the caller's side); otherwise costs is NULL and
 * the slices are computed on the GPU.  out: CV_32F HxW. */
int mdisp_asw_gifs(mdisp_ctx*, const mdisp_image* L, const mdisp_image* R, const mdisp_param* p,
                   double eps, int gif_type, int r2, double namda, double h,
                   int cost_source, const float* costs, mdisp_image* out);

/* getDisparity_SGBM / cv::StereoSGBM::compute call sites, methods/methods.cpp:1883-1921,1216-1219,1291.
 * L/R CV_8UC1; out CV_16S HxW, fixed point x16, invalid = (minDisparity-1)*16. */
int mdisp_sgbm(mdisp_ctx*, const mdisp_image* L, const mdisp_image* R, const mdisp_sgbm_params* p, mdisp_image* out);

/* mdisp_sgbm over a sequence of pairs held in host memory (the batch configuration): CV_8UC1 images of
 * rows x cols with row stride step_bytes, disps[i] = rows*cols int16, contiguous.  Pairs are processed
 * `chunk` at a time (<= 0 selects 16); uploads of the next chunk and downloads of the previous one run on
 * a second stream while a chunk is computed.  Results equal npairs calls of mdisp_sgbm.  The reference
 * processes one pair per call (methods/methods.cpp:1883-1921) and has no batch counterpart. */
int mdisp_sgbm_batch(mdisp_ctx*, const uint8_t* const* lefts, const uint8_t* const* rights, int npairs, int rows, int cols,
                     size_t step_bytes, const mdisp_sgbm_params* p, int16_t* const* disps, int chunk);

/* The aggregated SGM cost volume S(y,x,d) = sum over paths of L_r, biased by +P2 per path (see the
 * recursion: L'(p,d) = C + min(L'(p-r,d), L'(p-r,d+-1)+P1, min L'(p-r)+P2) - min L'(p-r)), as uint16 [rows][cols][numDisparities] in host
 * memory.  No counterpart call in the reference (StereoSGBM keeps S private); exported so that
 * the path recursion can be checked value by value. */
int mdisp_sgbm_volume(mdisp_ctx*, const mdisp_image* L, const mdisp_image* R, const mdisp_sgbm_params* p, uint16_t* S);

/* cv::StereoSGBM::create(...)->compute(left, right, disp) as called at methods/methods.cpp:1912, :1291,
 * :1650: Birchfield-Tomasi block cost on the x-Sobel-prefiltered and raw images, MODE_SGBM / MODE_HH
 * aggregation, uniqueness, x16 sub-pixel, disp12MaxDiff, then medianBlur(3) and filterSpeckles.
 * left/right CV_8UC1 or CV_8UC3 of equal size; out CV_16S.  OpenCV itself is not part of the
 * reference tree: the arithmetic is restated from its published algorithm and is not pinned by
 * any fixture of the reference. */
int mdisp_stereosgbm_compute(mdisp_ctx*, const mdisp_image* left, const mdisp_image* right,
                             const mdisp_stereosgbm_params* p, mdisp_image* out);
/* Same, with the intermediate results a parity check wants: the block cost volume C as int16
 * [rows][cols - minDisparity - numDisparities][numDisparities] (NULL to skip) and the disparity map
 * before the median filter (CV_16S, NULL to skip).  No counterpart call in the reference. */
int mdisp_stereosgbm_stages(mdisp_ctx*, const mdisp_image* left, const mdisp_image* right,
                            const mdisp_stereosgbm_params* p, mdisp_image* out, int16_t* cost, mdisp_image* raw);
/* cv::medianBlur(src, dst, 3) on CV_16SC1 (the filter StereoSGBM::compute applies to its map). */
int mdisp_median_blur3_s16(mdisp_ctx*, const mdisp_image* src, mdisp_image* dst);
/* cv::filterSpeckles(img, newVal, maxSpeckleSize, maxDiff) on CV_16SC1, in place. */
int mdisp_filter_speckles_s16(mdisp_ctx*, mdisp_image* img, int newVal, int maxSpeckleSize, int maxDiff);

/* stereomatch_1::computeAD / SAD / SAD_inteOpti / SAD_BFOpti / TAD / STAD* / SD / SSD*
 * (methods/method_StereoMatching.cpp:32-869).  left/right CV_8UC1 or BGR (converted to gray as the reference
 * does); numDisparity is a COUNT; out CV_8U = (char)(16 * winning index), 0 outside the evaluated region.
 * winSize is ignored by AD / TAD / SD, threshold_T by everything but the T variants (where it clamps the
 * winning index, as the reference does).  An even winSize fails with MDISP_E_BADARG ("winsize must be odd"). */
int mdisp_legacy_disparity(mdisp_ctx*, const mdisp_image* left, const mdisp_image* right, int method, int winSize,
                           int threshold_T, int minDisparity, int numDisparity, mdisp_image* out);

/* stereomatch_1::computeNCC (methods/method_StereoMatching.cpp:871-1106; declarations method_StereoMatching.h:115-120).
 * left/right CV_8UC1 or 3-channel (converted with COLOR_RGB2GRAY as the reference does: the first channel is weighted as
 * red); numDisparity is a COUNT.  The score is sum(l.*r) / (sum(l.*l) * sum(r.*r)) over winSize x winSize windows of the
 * REFLECT-padded view minus the REFLECT_101 box mean -- the reference's expression, product of energies included.
 *   out  (CV_32F, or NULL): the map overload.  is_left: offsets minDisparity .. minDisparity + numDisparity - 2 (the
 *        reference never tries the last one), the strictly smallest score wins, 0 where none was accepted.  The
 *        reference's right map is never assigned (it compares with "> DBL_MAX") and returns uninitialised memory: here
 *        it is all zeros.
 *   cost (host floats numDisparity x rows x cols, or NULL): the cost-volume overload, every offset, every slice
 *        normalised to [0, 1] (cv::normalize NORM_MINMAX), for both map types.
 * An even winSize or a size mismatch fails with MDISP_E_BADARG (the reference returns an empty result).
 * Parity unpinned: sums follow a CPU restatement (row-major over the window, float products, double sums); the library
 * routines the reference calls may round differently in the last bit, which can flip the strict '<' between near-equal
 * scores. */
int mdisp_ncc(mdisp_ctx*, const mdisp_image* left, const mdisp_image* right, int is_left, int winSize, int minDisparity,
              int numDisparity, mdisp_image* out, float* cost);

/* LR consistency check of PatchMatch::postProcess, methods/patchmatch.cpp:817-828.
 * dL/dR CV_32F, validL/validR CV_8U (1 = consistent within `thresh`). */
int mdisp_lr_check(mdisp_ctx*, const mdisp_image* dL, const mdisp_image* dR, float thresh,
                   mdisp_image* validL, mdisp_image* validR);

/* cv::initUndistortRectifyMap(K, D, R, P, size, CV_16SC2, map1, map2), methods/methods.cpp:1705-1706.
 * Host-side double arithmetic.  map1: int16 x2 per pixel, map2: uint16 per pixel. */
int mdisp_build_rectify_maps(const double K[9], const double* dist, int ndist, const double R[9],
                             const double P[12], int width, int height, int16_t* map1, uint16_t* map2);

/* cv::stereoRectify(K1, D1, K2, D2, size, R, T, R1, R2, P1, P2, Q, flags, alpha, size, &roi1, &roi2),
 * methods/methods.cpp:1691-1694 (also :1143, :1512).  Host-side double arithmetic, row-major
 * matrices, newImageSize = imageSize.  zero_disparity = 1 is CALIB_ZERO_DISPARITY; alpha < 0 keeps
 * the averaged focal length, 0 <= alpha <= 1 scales it between "valid pixels only" and "all source
 * pixels".  With alpha < 0 the results reproduce the known answer the reference stores in
 * unditortStereoCalib.xml (Rectify_R1/R2/P1/P2/Q).  Q, roi1, roi2 may be NULL; roi = x, y, w, h. */
int mdisp_stereo_rectify(const double K1[9], const double* D1, const double K2[9], const double* D2, int ndist,
                         int width, int height, const double R[9], const double T[3], int zero_disparity,
                         double alpha, double R1[9], double R2[9], double P1[12], double P2[12], double Q[16],
                         int roi1[4], int roi2[4]);

/* cv::remap(src, dst, map1, map2, INTER_LINEAR), methods/methods.cpp:1738,1745.
 * src/dst CV_8UC1 or CV_8UC3; maps as produced above, of dst's size. */
int mdisp_remap(mdisp_ctx*, const mdisp_image* src, const int16_t* map1, const uint16_t* map2, mdisp_image* dst);

/* Device-resident maps of cv::initUndistortRectifyMap(K, D, R, P, size, CV_16SC2, ...) (methods/methods.cpp:1705-1706),
 * built ON THE GPU once per calibration and reused for every frame (mdisp_remap uploads host maps on every call).
 * Entries are bit-identical to mdisp_build_rectify_maps. */
typedef struct mdisp_rectify_maps mdisp_rectify_maps;
int  mdisp_rectify_maps_create(mdisp_ctx*, const double K[9], const double* dist, int ndist, const double R[9],
                               const double P[12], int width, int height, mdisp_rectify_maps** maps);
void mdisp_rectify_maps_destroy(mdisp_rectify_maps* maps);
int  mdisp_rectify_maps_download(mdisp_ctx*, const mdisp_rectify_maps* maps, int16_t* map1, uint16_t* map2);
/* cv::remap(src, dst, map1, map2, INTER_LINEAR) (methods/methods.cpp:1709-1710) with such maps: host images ... */
int  mdisp_remap_with(mdisp_ctx*, const mdisp_rectify_maps* maps, const mdisp_image* src, mdisp_image* dst);
/* ... and device-resident interleaved 8-bit images (no copies, no sync); dst has the size of the maps. */
int  mdisp_remap_dev(mdisp_ctx*, const mdisp_rectify_maps* maps, const uint8_t* d_src, size_t src_pitch, int src_rows, int src_cols,
                     int channels, uint8_t* d_dst, size_t dst_pitch);


/* Disparity -> 3-D points, the step downstream of the disparity routines.
 * mdisp_reproject_q: methods/methods.cpp:2039-2076 -- [X Y Z W] = Q * [row col d 1] (row index first, as the
 *   reference writes it), point = (X/W, Y/W, Z/W) / 100; zero where the disparity is 0.
 * mdisp_reproject_triangulate: methods/methods.cpp:2300-2352 -- depth from the baseline and K_new with the
 *   row-dependent focal coefficient; points beyond 20000 or (when bgr is given) with a black colour stay zero.
 * disp CV_32FC1, bgr CV_32FC3 or NULL, xyz CV_32FC3 (x, y, z interleaved, what the PCD writer stores). */
int mdisp_reproject_q(mdisp_ctx*, const mdisp_image* disp, const double Q[16], mdisp_image* xyz);
int mdisp_reproject_triangulate(mdisp_ctx*, const mdisp_image* disp, const double K[9], double baseline,
                                const mdisp_image* bgr, mdisp_image* xyz);

/* cv::StereoBM as the reference configures it in getDisparity_BM (methods/methods.cpp:1825-1881:
 * StereoBM::create(144, 35) + setMinDisparity(186), setTextureThreshold(10), setDisp12MaxDiff(-1), setPreFilterCap(31),
 * setUniquenessRatio(25), setSpeckleRange(32), setSpeckleWindowSize(0)) and at methods/methods.cpp:1205-1213.
 * The prefilter is PREFILTER_XSOBEL (the library's default; the reference never changes it). */
typedef struct mdisp_stereobm_params {
    int minDisparity;       /* >= 0                                                            */
    int numDisparities;     /* positive multiple of 16, at most 512                            */
    int blockSize;          /* SADWindowSize: odd, 5..255, smaller than the image              */
    int preFilterCap;       /* 1..63                                                           */
    int textureThreshold;   /* >= 0                                                            */
    int uniquenessRatio;    /* >= 0; 0 disables the test                                       */
    int speckleWindowSize;  /* 0 disables the speckle filter                                   */
    int speckleRange;
    int disp12MaxDiff;      /* < 0 disables the left-right validation                          */
} mdisp_stereobm_params;

/* cv::StereoBM::compute(left, right, disp): 8-bit single-channel pair in host memory -> out CV_16S HxW, disparity x16,
 * (minDisparity - 1) * 16 where the matcher rejects a pixel or cannot place its window.  Arguments the library
 * refuses with an exception return MDISP_E_BADARG. */
int mdisp_stereobm_compute(mdisp_ctx*, const mdisp_image* left, const mdisp_image* right, const mdisp_stereobm_params* p,
                           mdisp_image* out);

/* ximgproc::fastGlobalSmootherFilter(guide, src, dst, lambda, sigma_color, lambda_attenuation, num_iter) — what
 * postProcess_ (disp_method/methods_disp.cpp:1299-1313) runs on both disparity maps through
 * createDisparityWLSFilterGeneric(false)->filter(disp, view, out, Mat(), whole-frame ROI) with lambda 8000, sigma 0.5
 * (attenuation 0.25, 3 iterations: the library's defaults).  guide: CV_8UC1 or CV_8UC3; src: CV_32FC1 or CV_16SC1 of the
 * same size; dst: same type as src.  Float arithmetic in the library's elimination order; tolerance 1e-3 relative. */
int mdisp_fgs_filter(mdisp_ctx*, const mdisp_image* guide, const mdisp_image* src, double lambda, double sigma_color,
                     double lambda_attenuation, int num_iter, mdisp_image* dst);

/* ---- device-resident entry points (inputs already in HBM, nothing copied, no sync) ---- */

/* bytes of device workspace mdisp_sgbm_dev needs for one pair */
size_t mdisp_sgbm_workspace_bytes(int rows, int cols, const mdisp_sgbm_params* p);

/* cv::StereoSGBM::compute (see mdisp_stereosgbm_compute) on device-resident interleaved 8-bit images
 * with `channels` = 1 or 3; d_disp: device int16 HxW; workspace as for mdisp_sgbm_dev. */
size_t mdisp_stereosgbm_workspace_bytes(int rows, int cols, int channels, const mdisp_stereosgbm_params* p);
int mdisp_stereosgbm_dev(mdisp_ctx*, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, int channels, size_t pitch,
                         const mdisp_stereosgbm_params* p, int16_t* d_disp, void* d_workspace, size_t workspace_bytes);

/* The same for n_pairs pairs (methods/methods.cpp:1216-1219, 1883-1921 call the matcher once per pair of a sequence): pair i at
 * d_left / d_right + i * pair_stride_bytes, its map at d_disp + i * rows * cols.  Up to four pairs run side by side on streams
 * of the context (a pair's launches are partly bound by the serial chains of its scan lines, and the others fill the
 * gaps); everything is ordered after the work already on the context's stream and before what is enqueued on it next.
 * Every map equals mdisp_stereosgbm_dev's.  d_workspace: mdisp_stereosgbm_batch_workspace_bytes(), or NULL (the context's). */
size_t mdisp_stereosgbm_batch_workspace_bytes(int rows, int cols, int channels, const mdisp_stereosgbm_params* p, int n_pairs);
int mdisp_stereosgbm_batch_dev(mdisp_ctx*, const uint8_t* d_left, const uint8_t* d_right, int n_pairs, size_t pair_stride_bytes,
                               int rows, int cols, int channels, size_t pitch, const mdisp_stereosgbm_params* p,
                               int16_t* d_disp, void* d_workspace, size_t workspace_bytes);

/* cv::StereoBM::compute (see mdisp_stereobm_compute) on device-resident CV_8UC1 images; d_disp: device int16 HxW;
 * d_workspace: at least mdisp_stereobm_workspace_bytes() of device memory, or NULL to use the context's own. */
size_t mdisp_stereobm_workspace_bytes(int rows, int cols, const mdisp_stereobm_params* p);
int mdisp_stereobm_dev(mdisp_ctx*, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, size_t pitch,
                       const mdisp_stereobm_params* p, int16_t* d_disp, void* d_workspace, size_t workspace_bytes);

/* d_left/d_right: device CV_8UC1 with the given pitch; d_disp: device int16 HxW (pitch = cols*2).
 * d_workspace: at least mdisp_sgbm_workspace_bytes() of device memory, or NULL to use the
 * context's own workspace. */
int mdisp_sgbm_dev(mdisp_ctx*, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, size_t pitch,
                   const mdisp_sgbm_params* p, int16_t* d_disp, void* d_workspace, size_t workspace_bytes);

/* Batch of independent pairs in one call (one launch covers all pairs, which is what fills the
 * GPU): pair i lives at d_left + i*pair_stride / d_right + i*pair_stride (bytes); d_disp receives
 * batch contiguous int16 HxW maps. */
size_t mdisp_sgbm_batch_workspace_bytes(int rows, int cols, int batch, const mdisp_sgbm_params* p);
int mdisp_sgbm_batch_dev(mdisp_ctx*, const uint8_t* d_left, const uint8_t* d_right, int batch, size_t pair_stride,
                         int rows, int cols, size_t pitch, const mdisp_sgbm_params* p, int16_t* d_disp,
                         void* d_workspace, size_t workspace_bytes);

/* ONE pair whose rows are spread over several devices (a single large pair, where the latency of the pair counts and
 * pair-sharding has nothing to shard).  The semi-global paths cross every row boundary, so the devices run the two
 * sweeps of the aggregation over their own row range and hand the path states of the boundary row to the neighbour:
 *   mdisp_sgbm_rows_geometry    band_rows: row ranges start at multiples of it (and end at one, or at the last row);
 *                               boundary_bytes: size of one boundary-state buffer; workspace_bytes: device workspace a
 *                               participant needs (it keeps the four-path sums between its two sweeps there, so the
 *                               same workspace must be passed to both sweeps of a pair)
 *   mdisp_sgbm_rows_sweep_dev   one sweep over rows [row_begin, row_end) of the pair.  direction 0 runs top to bottom
 *                               (paths from the left, the top-left, the top and the top-right), 1 bottom to top (the
 *                               four opposite paths).  second = 0: the first of the participant's two sweeps; second =
 *                               1: its other sweep, which also writes rows [row_begin, row_end) of d_disp (a full HxW
 *                               int16 map; the other rows are left alone).  d_boundary_in: the states of the row just
 *                               before the range in sweep order, as the neighbour's d_boundary_out delivered them (NULL
 *                               at the image edge); d_boundary_out: receives the states of the range's last row in
 *                               sweep order (NULL when nobody needs them).
 * Both views must be present on every participant (the cost of a row reads that row only, the census its two
 * neighbours).  Any order of sweeps that respects the boundary dependencies gives the map of mdisp_sgbm_dev bit for
 * bit.  Needs eight paths and 8 + P2 <= 255.  The reference computes a pair on one thread (stereoMatching.cpp:2110
 * computeDisparity); it has no counterpart of the split. */
int mdisp_sgbm_rows_geometry(int rows, int cols, const mdisp_sgbm_params* p, int* band_rows, size_t* boundary_bytes, size_t* workspace_bytes);
int mdisp_sgbm_rows_sweep_dev(mdisp_ctx*, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, size_t pitch,
                              const mdisp_sgbm_params* p, int row_begin, int row_end, int direction, int second,
                              const void* d_boundary_in, void* d_boundary_out, int16_t* d_disp, void* d_workspace, size_t workspace_bytes);

/* ---- several GPUs from one process (host-side; no counterpart in the reference, whose only concurrency is the two
 * std::async calls of cameraCalibration.cpp:163-190).  One host thread and one context per entry of devices[]; an
 * entry may repeat (several workers on one GPU).  No collective: pairs are independent, and the rows of one pair hand
 * their boundary states from neighbour to neighbour. ----
 *
 * mdisp_sgbm_batch_multi: mdisp_sgbm_batch over devices[]: worker r takes a contiguous block of the pairs (blocks differ
 * by at most one pair) and runs the chunked, double-buffered host batch on its device.  flags & MDISP_MULTI_PIN_HOST:
 * the images and maps of the call are page-locked (hipHostRegister) while it runs, so uploads and downloads are DMA
 * transfers that overlap the kernels.  Maps equal mdisp_sgbm_batch's bit for bit.
 * mdisp_sgbm_rows_multi: ONE pair (CV_8UC1 views, CV_16SC1 map), its rows spread over devices[] in whole bands
 * (mdisp_sgbm_rows_geometry): every worker holds both views, runs first the sweep whose boundary states arrive first,
 * and delivers the states of its last row with hipMemcpyPeerAsync (a device copy when both workers share a GPU).
 * The map equals mdisp_sgbm's bit for bit.  More workers than row bands: MDISP_E_BADARG. */
enum { MDISP_MULTI_PIN_HOST = 1 };
int mdisp_sgbm_batch_multi(const int* devices, int ndevices, const uint8_t* const* lefts, const uint8_t* const* rights, int npairs,
                           int rows, int cols, size_t step_bytes, const mdisp_sgbm_params* p, int16_t* const* disps, int chunk,
                           int flags);
int mdisp_sgbm_rows_multi(const int* devices, int ndevices, const mdisp_image* left, const mdisp_image* right,
                          const mdisp_sgbm_params* p, mdisp_image* out);

/* census stereo on device buffers; d_disp: device double HxW */
int mdisp_census_stereo_dev(mdisp_ctx*, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, size_t pitch,
                            const mdisp_param* p, int census_method, int agg_mode, double* d_disp);

/* mdisp_census_stereo_dev for callers that cannot allocate: code under HIP graph capture, tight serving loops and
 * harnesses that forbid hipMalloc.  It has no counterpart in the reference.  Same arguments and results as
 * mdisp_census_stereo_dev, but every intermediate plane lives in a fixed device array of MDISP_NOALLOC_SCRATCH_BYTES
 * inside the library (held by every GPU the library runs on until the process ends), and the call performs no
 * hipMalloc / hipFree.  Needs 4 * rows * roundup(cols + maxDisparity, 256) + 2048 bytes of it: larger problems fail
 * with MDISP_E_RANGE, MDISP_AGG_GUIDED with MDISP_E_UNSUPPORTED.  All calls on one GPU share the array: calls that
 * overlap in time on one GPU (two contexts, two streams) race; calls on one context are ordered by its stream. */
#define MDISP_NOALLOC_SCRATCH_BYTES (16u << 20)
int mdisp_census_stereo_dev_noalloc(mdisp_ctx*, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, size_t pitch,
                                    const mdisp_param* p, int census_method, int agg_mode, double* d_disp);

/* asw_gifs (methods_disp.h:66) on device buffers, restricted to the output rows [row_begin, row_end):
 * the share of one GPU when a large pair is split into horizontal row bands.  d_left / d_right hold
 * the WHOLE images (every GPU keeps a copy); d_disp is a float rows x cols map of which only the
 * band's rows are written.  Rows are bit-identical to those of a full-image call.  cost_source is
 * MDISP_COST_CENSUS or MDISP_COST_ABSDIFF.
 * cost_minmax (host, 2*numDisp floats: min, max per slice) and steer_max (host, one double) carry the
 * statistics that span the whole image.  With NULL they are recomputed locally from the full images (no
 * exchange at all, at the price of doing that part once per GPU); otherwise pass the element-wise min / max
 * over all bands of what mdisp_asw_gifs_band_stats_dev returned (one small all-reduce), and the band computes
 * only its own share. */
int mdisp_asw_gifs_band_dev(mdisp_ctx*, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, size_t pitch,
                            const mdisp_param* p, double eps, int gif_type, int r2, double namda, int cost_source,
                            int row_begin, int row_end, float* d_disp, const float* cost_minmax, const double* steer_max);

/* One band's share of those statistics: per-slice min / max of the matching cost over rows [row_begin, row_end)
 * (+FLT_MAX / -FLT_MAX for an empty band) and the maximum of the steer filter's orientation strength over the same
 * rows (0 unless gif_type is MDISP_SKWGIF).  Results go to host memory; the call synchronises the stream.
 * The reference computes these over the whole image (gifs.cpp:56, steerFilter.cpp:400-407). */
int mdisp_asw_gifs_band_stats_dev(mdisp_ctx*, const uint8_t* d_left, const uint8_t* d_right, int rows, int cols, size_t pitch,
                                  const mdisp_param* p, int gif_type, int cost_source, int row_begin, int row_end,
                                  float* cost_minmax, double* steer_max);

/* ---- the guided-filter matcher of namespace stereomatch_1 (methods/method_StereoMatching.h, .cpp:3288-3781) ----
 *
 * mdisp_cost_sad_d: stereomatch_1::getCostSAD_d (.cpp:3288-3349).  One of the two views is the one the caller bordered with
 * copyMakeBorder(BORDER_REFLECT) and is wider: the right one for is_left (compared at column cols_right - cols - disparity),
 * the left one otherwise (at column disparity).  CV_8U, 1 or 3 channels (BGR -> gray as the reference does); out: CV_32F of
 * the narrower view's size = boxFilter(|difference|, winSize x winSize, normalised).  MDISP_E_BADARG where the reference
 * returns an empty image (even window, other view not wider). */
int mdisp_cost_sad_d(mdisp_ctx*, const mdisp_image* left, const mdisp_image* right, int disparity, int is_left, int winSize,
                     mdisp_image* out);

/* getGuidedFilter with a 3- or 6-channel CV_8U guide (single-channel guides: mdisp_guided_filter).  P: CV_8U or CV_32F,
 * one channel; out: CV_32F.  Guide and P are min-max normalised first (one minimum / maximum over all guide channels),
 * every channel is filtered on its own, a = cov / (var + eps) per channel, b = mean(P) - dot(a, mean(G)),
 * out = dot(box(a), G) + box(b).
 * MDISP_GIF_MC_GIFS: disp_method/gifs.cpp:47-135 with multiChl_to_oneChl_mul (:8-45).
 * MDISP_GIF_MC_STEREOMATCH: the twin methods/method_StereoMatching.cpp:3612-3672, whose regulariser is
 * (var / var) * eps -- 0 where the variance is 0 (cv::divide), and a = 0 there. */
enum { MDISP_GIF_MC_GIFS = 0, MDISP_GIF_MC_STEREOMATCH = 1 };
int mdisp_guided_filter_multi(mdisp_ctx*, const mdisp_image* guide, const mdisp_image* P, int r, double eps, int variant,
                              mdisp_image* out);

/* stereomatch_1::computeAdaptiveWeight_GuidedF (.cpp:3685-3781; stereoMatching(ADAPTIVE_WEIGHT_GUIDED_FILTER) calls it with
 * DISPARITY_LEFT, 0.01, 15, 10, 150, methods/methods.cpp:1801-1803).  L / R: CV_8UC3 of equal size; per candidate the
 * getCostSAD_d slice is filtered with the six-channel guide [left, right] of the two views aligned at the candidate, then
 * the first minimum wins; out: CV_32F map holding minDisparity + index.  Gray input: MDISP_E_UNSUPPORTED (the reference's
 * 2-channel guide fails inside multiChl_to_oneChl_mul). */
int mdisp_adaptive_weight_guidedf(mdisp_ctx*, const mdisp_image* L, const mdisp_image* R, int is_left, double eps, int winSize,
                                  int minDisparity, int numDisparity, mdisp_image* out);

/* stereomatch_1::computeSimilarity (methods/method_StereoMatching.cpp:1108-1348; .h both overloads): colour + gradient cost
 * slices.  The reference function runs for CV_8UC3 views and DISPARITY_LEFT only -- its other branches multiply a CV_32F
 * matrix by a CV_8U mask (:1218, :1257, :1292), which cv::multiply refuses -- so every other input gives
 * MDISP_E_UNSUPPORTED.  Slice k (candidate minDisparity + k), as the library evaluates the expressions:
 *   colour   = round(saturate(c0 + c1) / 3 + c2 / 3) of the per-channel |L - R|; cost = saturate(colour + thresC) where colour >
 *              thresC, else 0 (the reference's mask arithmetic, :1152-1158)
 *   gradient = |filter2D([-3 0 3; -10 0 10; -3 0 3])| differences, (g0 + g1) / 3 + g2 / 3; cost = g + 254 * thresG where g >
 *              thresG, else 255 * thresG (:1166-1175: bitwise_not of a 0 / 1 mask)
 *   slice    = (1 - regularity) * colour cost + regularity * gradient cost, float.
 * winSize = 0: the nine-argument overload, numDisparity slices of rows x cols floats.  Odd winSize: the ten-argument
 * overload (:1331-1348), every slice bordered by winSize / 2 cells of BORDER_REFLECT on all sides.  Even winSize:
 * MDISP_E_BADARG (the reference returns without slices).  cost: host floats. */
int mdisp_compute_similarity(mdisp_ctx*, const mdisp_image* L, const mdisp_image* R, int is_left, double regularity, double thresC,
                             double thresG, int winSize, int minDisparity, int numDisparity, float* cost);

/* stereomatch_1::computeAdaptiveWeight_GuidedF (which = 1), _GuidedF_2 (2: computeSimilarity(0.4, 10, 50) slices filtered with
 * the reference view as 3-channel guide, .cpp:3794-3868; DISPARITY_RIGHT: MDISP_E_UNSUPPORTED, as computeSimilarity) and
 * _GuidedF_3 (3: computeNCC cost-volume slices; DISPARITY_LEFT filters with the six-channel guide, DISPARITY_RIGHT with the
 * right view, .cpp:3881-3955).  stereoMatching calls them with (0.01, 15, 10, 150), (0.01, 13, 10, 150) and
 * (1e-6, 15, 0, 64), methods/methods.cpp:1801-1809. */
int mdisp_adaptive_weight_guidedf_n(mdisp_ctx*, const mdisp_image* L, const mdisp_image* R, int which, int is_left, double eps,
                                    int winSize, int minDisparity, int numDisparity, mdisp_image* out);

/* time of the most recent call's dominant kernel on this context, measured with HIP events on
 * the context's stream (milliseconds); which = 0: whole call, 1: SGM path aggregation (every scan
 * and sweep kernel of the call), 2: the separate winner-take-all kernel of the per-direction
 * fallback (not recorded when winner-take-all is fused into the upward sweep) */
int mdisp_last_kernel_ms(mdisp_ctx*, int which, float* ms);

/* Measurement aid, no counterpart in the reference: copies `bytes` (a multiple of 16) from d_src to d_dst `reps`
 * times with a streaming kernel on the context's stream and reports read + written GB/s (HIP events).  bench.py
 * quotes it next to the nominal HBM peak. */
int mdisp_copy_bandwidth_dev(mdisp_ctx*, void* d_dst, const void* d_src, size_t bytes, int reps, float* gb_per_s);
int mdisp_enable_timing(mdisp_ctx*, int on);

/* Diagnostic aid, no counterpart in the reference: process-wide scheduling overrides that let the tests reach every
 * schedule of the SGM code at small image sizes (by default the library picks a schedule from the image and batch
 * size alone; it never reads the environment).  value < 0 restores the default.  Not synchronised with calls that
 * are in flight on other threads: set it while no compute call is running. */
enum {
    MDISP_TUNE_SGM_SPLIT_MAX = 0,      /* largest batch on the latency schedule (default 4)                         */
    MDISP_TUNE_SGM_PATH_WAVES_MAX = 1, /* largest batch whose sweeps run one path per wave group (default 4)        */
    MDISP_TUNE_SGM_STRIP_LANES = 2,    /* 256 or 512 compute lanes per sheared strip                                */
    MDISP_TUNE_SGM_FORCE_PREPASS = 3,  /* 1: upright strips + diagonal pre-pass even when byte path states fit      */
    MDISP_TUNE_SGM_LANES16 = 4,        /* 1: sixteen lanes per pixel whatever the disparity count                   */
    MDISP_TUNE_SGM_PER_DIRECTION = 5,  /* 1: one kernel per path direction (the fallback of unsupported shapes)     */
    MDISP_TUNE_SGBM_BT_SEQUENTIAL = 6, /* 1: StereoSGBM replacement runs its directions one launch after the other (what large problems do); 2: the same with the selection as a kernel of its own instead of inside the last direction; 3: the same as 1 with the vertical window sum and the downward path as two kernels */
    MDISP_TUNE_SGM_PIPELINE = 7,       /* batches: 0 = row bands with four paths per sweep (default), 1 = sheared strips with a separate horizontal scan */
    MDISP_TUNE_SGM_PLANT_ERROR = 8,    /* 1: the next pipelined SGM call finds its hand-off error word already set  */
    MDISP_TUNE_SGM_BAND_SLACK = 9,     /* row bands: pixels a band stays behind its predecessor beyond the minimum  */
    MDISP_TUNE_SGM_PAIR_GROUP = 10,    /* row bands: pairs whose bands are dispatched together (band-major inside a group) */
    MDISP_TUNE_SGM_HAND_RING = 11,     /* row bands: entries of the hand-off ring (2..32, a power of two; default 32) */
    MDISP_TUNE_SGM_MEET_MAX = 12,      /* row bands: largest batch whose two sweeps start from both image edges at once (default: any) */
    MDISP_TUNE_SGM_ONE_LAUNCH = 13,    /* row bands: 0 = the two sweep directions as two launches per phase, 1 = both directions and both roles in one launch; default: one launch up to 96 pairs */
    MDISP_TUNE_SAT_INT_TABLES = 14,    /* 1: mdisp_sad_inte and the stereomatch_1 _inteOpti matchers sum in exact integer tables instead of the reference's CV_32F ones */
    MDISP_TUNE_SGM_RING_HANDOFF = 15,  /* row bands, one-launch schedule: 0 = one workgroup barrier per step (default), 1 = barrier-free waves with LDS rings between them (measured slower) */
    MDISP_TUNE_SGBM_BATCH_LANES = 16,  /* mdisp_stereosgbm_batch_dev: pairs that run side by side (1..4; default 4) */
    MDISP_TUNE_GIF_MC_TILES = 17,      /* guided filter with a multi-channel guide: 1 = the LDS-tile kernels (windows up to 47) instead of the marching ones */
    MDISP_TUNE_COUNT = 18
};
int mdisp_tuning_set(int key, int value);
int mdisp_tuning_get(int key);

/* ---- synthetic rectified pair generator used by the tests and the benchmark (host) ----
 * right = box-filtered white noise, left(x,y) = right(reflect(x - d_gt), y) + noise in [-2,2];
 * gt (nullable) receives the piecewise-planar ground-truth disparity. */
int mdisp_synth_pair(uint32_t seed, int rows, int cols, int numDisp, uint8_t* left, uint8_t* right, uint16_t* gt);

#ifdef __cplusplus
}
#endif
#endif /* MDISP_H */
