/*
 * CPU oracle for the dense-stereo hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's CPU algorithms (census /
 * Hamming / SAD / box and guided aggregation / WTA / SGM recursion / LR check /
 * remap).  It is never linked into, imported by or called from the shipped
 * library; only the test-suite, the smoke check and the benchmark's CPU-baseline
 * leg use it, and there only as the checker.
 *
 * PARITY UNPINNED: the reference cannot be compiled here (OpenCV, PCL and CImg
 * are absent) and it ships no tests, golden vectors or expected outputs.  The
 * functions below follow the reference source text line by line where the
 * reference owns the arithmetic, and the documented OpenCV 4.1.0 semantics where
 * it only calls into OpenCV.  What pins them: hand-derivable known answers
 * (constant image, ramps, identical pairs), independent numpy/scipy
 * re-derivations in tests/, and the stereoRectify outputs stored in
 * cameraCalibration/unditortStereoCalib.xml.
 */
#ifndef ORC_H
#define ORC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* census variants, cameraCalibration/disp_method/methods_disp.h:18-27 */
enum {
    ORC_BASIC_CENSUS = 0,
    ORC_CIRCLE_CENSUS = 1,
    ORC_ROTATION_INVARIANT_CENSUS = 2,
    ORC_UNIFORM_CENSUS = 3,
    ORC_MULTISCALE_CENSUS = 4,
    ORC_STATISTIC_MULTISCALE_CENSUS = 5,
    ORC_CENSUS_2017 = 6
};

/* aggregation applied to each Hamming slice inside census stereo */
enum {
    ORC_AGG_NONE = 0,      /* raw per-pixel Hamming distance                              */
    ORC_AGG_BOX = 1,       /* integer box sum, win x win, BORDER_REFLECT_101              */
    ORC_AGG_SQRBOX = 2,    /* integer box sum of squares (the commented sqrBoxFilter)     */
    ORC_AGG_GUIDED = 3     /* getGuidedFilter(guide, slice, win, 1e-6): the active path   */
};

enum { ORC_SAT_F32 = 0, ORC_SAT_INT = 1 };

enum {
    ORC_GIF = 0, ORC_WGIF = 1, ORC_EGIF = 2, ORC_SKWGIF = 3, ORC_OURS_GIF = 4, ORC_OURS_GIF2 = 5, ORC_OURS_GIF3 = 6
};

typedef struct {
    int minDisparity;
    int numDisparities;
    int P1, P2;
    int uniquenessRatio;   /* percent; <0 disables the test                                */
    int disp12MaxDiff;     /* <0 disables the LR check                                     */
    int paths;             /* 8, 5 (MODE_SGBM order) or 4                                  */
    int subpixel;          /* 1: x16 parabola refinement                                   */
    int costWin;           /* odd box window summed over the Hamming cost (1 = per pixel)  */
    int censusMethod;
} orc_sgm_params;

/* fixed-window matchers of namespace stereomatch_1 (methods/method_StereoMatching.cpp:32-869) */
enum {
    ORC_LEGACY_AD = 0, ORC_LEGACY_SAD = 1, ORC_LEGACY_SAD_INTE = 2, ORC_LEGACY_SAD_BF = 3,
    ORC_LEGACY_TAD = 4, ORC_LEGACY_STAD = 5, ORC_LEGACY_STAD_INTE = 6, ORC_LEGACY_STAD_BF = 7,
    ORC_LEGACY_SD = 8, ORC_LEGACY_SSD = 9, ORC_LEGACY_SSD_INTE = 10, ORC_LEGACY_SSD_BF = 11
};
int orc_legacy_disparity(const uint8_t* L, const uint8_t* R, int H, int W, int method, int winSize, int T, int minD, int numD,
                         uint8_t* out);
/* the three _inteOpti kinds through CV_32F summed-area tables in the library's accumulation order */
int orc_legacy_inte_f32(const uint8_t* L, const uint8_t* R, int H, int W, int method, int winSize, int T, int minD, int numD,
                        uint8_t* out);

/* stereomatch_1::getCostSAD_d, getGuidedFilter with a multi-channel guide, computeAdaptiveWeight_GuidedF
 * (methods/method_StereoMatching.cpp:3288-3349, 3612-3672, 3685-3781; disp_method/gifs.cpp:8-135) */
int orc_cost_sad_d(const uint8_t* left, int leftW, const uint8_t* right, int rightW, int H, int disparity, int isLeft, int winSize,
                   float* out);
void orc_guided_filter_mc(const float* G /* cn planes, normalised */, int cn, const float* P, int H, int W, int r, double eps,
                          int variant, float* out);
int orc_adaptive_weight_guidedf(const uint8_t* L, const uint8_t* R, int H, int W, int cn, int isLeft, double eps, int winSize, int minD,
                                int numD, float* depth);

/* stereomatch_1::computeSimilarity (colour, DISPARITY_LEFT) and computeAdaptiveWeight_GuidedF_2 / _3
 * (methods/method_StereoMatching.cpp:1108-1180, 3794-3955) */
int orc_compute_similarity(const uint8_t* L, const uint8_t* R, int H, int W, int cn, int isLeft, double regularity, double thresC,
                           double thresG, int minD, int numD, float* cost);
int orc_adaptive_weight_guidedf_23(const uint8_t* L, const uint8_t* R, int H, int W, int cn, int which, int isLeft, double eps, int winSize,
                                   int minD, int numD, float* depth);

/* cv::StereoSGBM parameters (create() argument order) */
enum { ORC_SGBM_MODE_SGBM = 0, ORC_SGBM_MODE_HH = 1 };
typedef struct {
    int minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap, uniquenessRatio,
        speckleWindowSize, speckleRange, mode;
} orc_sgbm_bt_params;

/* ---- borders / colour ---- */
int  orc_reflect(int p, int n);        /* BORDER_REFLECT     fedcba|abcdef|fedcba */
int  orc_reflect101(int p, int n);     /* BORDER_REFLECT_101 gfedcb|abcdef|edcba  */
void orc_pad_reflect_lr(const uint8_t* src, int H, int W, int left, int right, uint8_t* dst);
void orc_bgr2gray(const uint8_t* bgr, int H, int W, uint8_t* gray);

/* ---- census family / Hamming ---- */
void orc_census_basic(const uint8_t* src, int H, int W, uint8_t* dst);
void orc_census_rotation_inv(const uint8_t* src, int H, int W, uint8_t* dst);
void orc_census_uniform(const uint8_t* src, int H, int W, uint8_t* dst);
void orc_census_multiscale(const uint8_t* src, int H, int W, int scale, uint8_t* dst);
void orc_census_multiscale2(const uint8_t* src, int H, int W, int scale, uint8_t* dst);
void orc_census_circle(const uint8_t* src, int H, int W, int radius, int npts, uint8_t* dst);
int  orc_census(const uint8_t* src, int H, int W, int method, int winSize, int isPaddedView, uint8_t* dst);
void orc_ri_table(uint8_t* t256);
void orc_uniform_table(uint8_t* t256);
int  orc_hop_count(uint8_t v);
void orc_hamming(const uint8_t* a, const uint8_t* b, long n, int32_t* dst);

/* ---- stereo drivers ---- */
int orc_census_stereo(const uint8_t* L, const uint8_t* R, int H, int W, int winSize, int minD, int maxD,
                      int isDispLeft, int method, int agg, double* disp, float* costVolume /*nullable: D*H*W*/);
int orc_sad_inte(const uint8_t* L, const uint8_t* R, int H, int W, int winSize, int minD, int maxD,
                 int isDispLeft, int satMode, uint8_t* disp);
int orc_sad_box(const uint8_t* L, const uint8_t* R, int H, int W, int winSize, int minD, int maxD,
                int isDispLeft, uint8_t* disp);

/* ---- filters ---- */
void orc_box_filter_f32(const float* src, int H, int W, int kw, int kh, int normalize, float* dst);
void orc_box_sum_u32(const uint32_t* src, int H, int W, int kw, int kh, uint32_t* dst);
void orc_normalize_minmax_u8(const uint8_t* src, long n, float* dst);
void orc_normalize_minmax_s32(const int32_t* src, long n, float* dst);
void orc_normalize_minmax_f32(const float* src, long n, float* dst);

/* orc_ncc.c: stereomatch_1::computeNCC, map and cost-volume overloads (methods/method_StereoMatching.cpp:871-1106) */
int orc_ncc_disparity(const uint8_t* L, const uint8_t* R, int H, int W, int is_left, int win, int minD, int numD, float* out);
int orc_ncc_volume(const uint8_t* L, const uint8_t* R, int H, int W, int is_left, int win, int minD, int numD, float* cost);
void orc_guided_filter(const float* guideN, const float* pN, int H, int W, int r, double eps, float* out);
void orc_guided_filter_egif(const float* guideN, const float* pN, int H, int W, int r, double eps, float* out, float* beta);
void orc_guided_filter_wgif(const float* guideN, const float* pN, int H, int W, int r, double eps, float* out);
void orc_guided_filter_skwgif2(const float* guideN, const float* pN, const double* steerW, int H, int W,
                               int r, int r2, double eps, double namuda, float* out);
void orc_steer_filter(const uint8_t* src, int H, int W, double* dst);
void orc_bilateral_f32(const float* src, int H, int W, int d, double sigma_color, double sigma_space, float* dst);
void orc_guided_filter_ours(const float* guideN, const float* pN, int H, int W, int r, int r2, double eps, double namuda,
                            int variant /* 1, 2, 3 */, float* out, float* beta /* nullable */);
int  orc_asw_gifs(const uint8_t* guide, const float* costs /*D*H*W f32*/, int H, int W, int D, int minD,
                  int winSize, double eps, int gifType, int r2, double namuda, float* disp, float* filtered /*nullable*/);

/* ---- SGM ---- */
int orc_sgm(const uint8_t* L, const uint8_t* R, int H, int W, const orc_sgm_params* p,
            int16_t* disp, uint16_t* Svol /*nullable: H*W*D*/);

/* cv::StereoSGBM::compute with its own Birchfield-Tomasi block cost (parity unpinned), cn = 1 or 3 */
int orc_sgbm_bt(const uint8_t* L, const uint8_t* R, int H, int W, int cn, const orc_sgbm_bt_params* p, int16_t* disp,
                int16_t* Cvol /* nullable */, int16_t* raw /* nullable */);
void orc_median3_s16(const int16_t* src, int H, int W, int16_t* dst);
void orc_filter_speckles_s16(int16_t* img, int H, int W, int newVal, int maxSpeckleSize, int maxDiff);

/* ---- LR check (patchmatch.cpp:817-828) ---- */
void orc_lr_check(const float* dL, const float* dR, int H, int W, float thresh, uint8_t* validL, uint8_t* validR);

/* ---- rectification ---- */
void orc_init_undistort_rectify_map(const double K[9], const double* dist, int ndist, const double R[9],
                                    const double P[12], int W, int H, int16_t* map1, uint16_t* map2);
void orc_remap_bilinear_u8(const uint8_t* src, int sH, int sW, int cn, const int16_t* map1, const uint16_t* map2,
                           int dH, int dW, uint8_t* dst);
void orc_stereo_rectify(const double K1[9], const double* D1, const double K2[9], const double* D2, int ndist,
                        int W, int H, const double R[9], const double T[3], double alpha, int zero_disparity,
                        int global_rotation, double R1[9], double R2[9], double P1[12], double P2[12], double Q[16],
                        int roi1[4], int roi2[4]);

/* ---- disparity -> 3-D points ---- */
void orc_reproject_q(const float* disp, int H, int W, const double Q[16], float* xyz);
void orc_reproject_triangulate(const float* disp, int H, int W, const double K[9], double baseline, const float* bgr, float* xyz);

/* ---- cv::StereoBM (orc_bm.c) ---- */
typedef struct {
    int minDisparity, numDisparities, blockSize, preFilterCap, textureThreshold, uniquenessRatio, speckleWindowSize, speckleRange,
        disp12MaxDiff;
} orc_bm_params;
void orc_bm_prefilter_xsobel(const uint8_t* src, int H, int W, int cap, uint8_t* dst);
/* 8-bit single-channel pair -> CV_16S disparity x16; returns -1 for arguments the library rejects */
int orc_stereobm(const uint8_t* L, const uint8_t* R, int H, int W, const orc_bm_params* p, int16_t* disp);

/* ---- fast global smoother / disparity WLS post-filter (orc_fgs.c) ---- */
void orc_fgs_weights_lut(double sigmaColor, float* lut);
void orc_fgs_filter_f32(const uint8_t* guide, int H, int W, int cn, const float* src, double lambda, double sigmaColor,
                        double attenuation, int iters, float* dst);
void orc_fgs_filter_s16(const uint8_t* guide, int H, int W, int cn, const int16_t* src, double lambda, double sigmaColor,
                        double attenuation, int iters, int16_t* dst);

/* number of OpenMP threads the row-parallel loops use from now on (bench.py times one thread beside all cores) */
void orc_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
