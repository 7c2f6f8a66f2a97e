// Shared declarations of libmdisp.so (MI355X / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "../../include/mdisp.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;

void mdisp_set_error(const char* fmt, ...);
// process-wide scheduling override (mdisp_tuning_set): the value, or -1 when the default applies
int md_tune(int key);
void md_tune_clear(int key);
int md_tune_take(int key);      // value of a one-shot override, cleared by the same atomic exchange

#define MD_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            mdisp_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return MDISP_E_HIP;                                                           \
        }                                                                                 \
    } while (0)

#define MD_REQUIRE(cond, code, ...)                  \
    do {                                             \
        if (!(cond)) {                               \
            mdisp_set_error(__VA_ARGS__);            \
            return (code);                           \
        }                                            \
    } while (0)

#define MD_TRY(expr)                 \
    do {                             \
        int rc__ = (expr);           \
        if (rc__) return rc__;       \
    } while (0)

struct mdisp_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    void* ws;            // grow-only device workspace
    size_t ws_bytes;
    bool timing;
    hipEvent_t ev[6];    // [0,1] whole call, [2,3] SGM aggregation, [4,5] separate WTA kernel
    bool ev_valid[3];
    void* stage;         // grow-only device staging buffer of the host<->device image copies
    size_t stage_bytes;
    hipStream_t aux;     // second stream for independent SGM stages of small batches (created on first use)
    hipEvent_t aux_ev[3];
    bool aux_ready;
    int* sgm_err;        // device word owned by the context: set (never cleared) by a pipelined SGM kernel whose hand-off wait ran out
    bool sgm_pending;    // a pipelined SGM call has been enqueued since the word was last read
    hipStream_t lane[3]; // further streams of the batched StereoSGBM entry point (pairs side by side; created on first use)
    hipEvent_t lane_ev[4];
    bool lanes_ready;
    u8* noalloc_scratch; // device address, on this context's GPU, of the file-scope scratch array of the no-allocation entry point
};
// after the stream has been synchronised: fails if the last SGM call reported a hand-off timeout
int md_check_sgm_error(mdisp_ctx* ctx);

// Image copies between host memory (row stride hstep) and pitched device memory, enqueued on the
// context's stream.  A 2-D copy from pageable host memory is executed row by row by the runtime
// (milliseconds for a VGA image); contiguous host images therefore travel as ONE linear copy to or
// from a device staging buffer plus a device-side 2-D copy.
int md_h2d_2d(mdisp_ctx* c, void* d, size_t dpitch, const void* h, size_t hstep, size_t rowBytes, size_t rows);
int md_d2h_2d(mdisp_ctx* c, void* h, size_t hstep, const void* d, size_t dpitch, size_t rowBytes, size_t rows);

// grow-only workspace; returns nullptr on failure (error set)
void* mdisp_ws(mdisp_ctx* ctx, size_t bytes);

static inline size_t md_align(size_t v, size_t a) { return (v + a - 1) / a * a; }
__host__ __device__ static inline int md_align_dev(int v, int a) { return (v + a - 1) / a * a; }

// ---- kernels' host launchers (all enqueue on `s`, no sync) ----
// census.hip
// ncc.hip
size_t md_ncc_bytes(int H, int W, int minD, int numD);
int md_ncc(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, int H, int W, int isLeft, int win, int minD, int numD, void* ws,
           float* dOut, float* dCost);
int md_census_pairs(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, size_t pairStride, int batch, int H, int W, int left, int method,
                    u8* cL, size_t pitchL, size_t strideL, u8* cR, size_t pitchR, size_t strideR);
int md_pad_reflect_lr(hipStream_t s, const u8* src, size_t spitch, int H, int W, int left, int right, u8* dst, size_t dpitch);
int md_census(hipStream_t s, const u8* src, size_t spitch, int H, int W, int method, int arg, u8* dst, size_t dpitch,
              u8* scratch /* H*dpitch (rounded up to 256) + MD_CENSUS_SCRATCH_TAIL bytes, needed by the MULTISCALE variants */);
#define MD_CENSUS_SCRATCH_TAIL 2048
int md_hamming(hipStream_t s, const u8* a, size_t ap, const u8* b, size_t bp, int H, int W, int32_t* dst, size_t dp);
int md_bgr2gray(hipStream_t s, const u8* bgr, size_t spitch, int H, int W, u8* gray, size_t dpitch, int rgbOrder = 0);

// sgm.hip
struct md_sgm_plan {
    int H, W, D, minD, maxD, Wp;
    int P1, P2, uniq, maxDiff, npaths, subpixel, costWin, census;
    size_t off_cL, off_cR, off_pad, off_scratch, off_S, off_aux, off_bnd, total;
    size_t pitchL, pitchR;
    int batch;        // stereo pairs processed per call
    size_t strideS;   // u16 elements between the S volumes of consecutive pairs
    bool use_v2;      // strip-fused two-sweep aggregation (sgm2.hip) instead of per-direction scans
    int sstride;      // u16 elements per pixel in the S volume
    bool rowBands;    // the row-band schedule whatever the batch size (row-sharded calls)
    bool exportVolume; // the caller wants the full 16-bit sums, not disparity maps
};
// second stream + fork/join events of a context (created on first use); null members = run everything on one stream
struct md_aux_streams { hipStream_t aux; hipEvent_t ev[3]; };
int md_ctx_aux(mdisp_ctx* ctx, md_aux_streams* out);

// sgm2.hip
bool md_sgm2_supported(const md_sgm_plan& pl);
size_t md_sgm2_extra_bytes(const md_sgm_plan& pl, int batch);
int md_sgm2_slots(const md_sgm_plan& pl);   // u16 elements per pixel of the S volume (64, 128 or 256)
bool md_sgm2_u8(const md_sgm_plan& pl);
size_t md_sgm2_partial_sum_bytes(const md_sgm_plan& pl);   // bytes per slot of the volume between the sweeps when winner-take-all is fused (1 or 2)
int md_sgm2_aggregate(hipStream_t s, const md_sgm_plan& pl, const u8* cL, size_t strideL, const u8* cR, size_t strideR,
                      u16* S, size_t strideS, u8* extra, int batch, int naturalOut, int16_t* dDisp, int* errWord, bool* usedErrWord,
                      const md_aux_streams* aux = nullptr);
int md_sgm_plan_make(int H, int W, const mdisp_sgbm_params* p, int batch, md_sgm_plan* plan, bool exportVolume = false, bool rowBands = false);
// one sweep of the row-band schedule over the image rows [r0, r1) of one pair (sgm2.hip / sgm.hip)
int md_sgm2_band_rows(const md_sgm_plan& pl);            // rows per band: r0 must be a multiple, r1 a multiple or H
size_t md_sgm2_boundary_bytes(const md_sgm_plan& pl);    // bytes of the path states that cross a band edge
int md_sgm2_rows_sweep(hipStream_t s, const md_sgm_plan& pl, const u8* cL, const u8* cR, u16* S, u8* extra, int r0, int r1, bool up, bool second,
                       const u8* bndIn, u8* bndOut, int16_t* dDisp, int* errWord, bool* usedErrWord);
int md_sgm_rows_sweep(mdisp_ctx* ctx, const md_sgm_plan& pl, const u8* dL, const u8* dR, size_t pitch, void* ws, int r0, int r1, bool up, bool second,
                      const u8* bndIn, u8* bndOut, int16_t* dDisp);
int md_sgm_run(mdisp_ctx* ctx, const md_sgm_plan& plan, const u8* dL, const u8* dR, size_t pitch, size_t pairStride,
               int16_t* dDisp, void* ws);
// aggregation only; *S receives the device volume (pixel stride plan.sstride)
int md_sgm_aggregate_only(mdisp_ctx* ctx, const md_sgm_plan& plan, const u8* dL, const u8* dR, size_t pitch, void* ws, const u16** S);

// sgbm_bt.hip: cv::StereoSGBM with its own Birchfield-Tomasi block cost
size_t md_stereosgbm_bytes(int H, int W, int cn, const mdisp_stereosgbm_params* p);
int md_stereosgbm_run(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, int H, int W, int cn,
                      const mdisp_stereosgbm_params* p, int16_t* dDisp, void* ws, int16_t* dCost, int16_t* dRaw);
int md_median3_s16(hipStream_t s, const int16_t* src, int H, int W, int16_t* dst);
size_t md_speckle_scratch_bytes(int H, int W);
int md_filter_speckles_s16(hipStream_t s, int16_t* img, int H, int W, int newVal, int maxSize, int maxDiff, void* scratch);

// bm.hip: cv::StereoBM
int md_stereobm_check(int H, int W, const mdisp_stereobm_params* p);
size_t md_stereobm_bytes(int H, int W, const mdisp_stereobm_params* p);
int md_stereobm_run(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, int H, int W, const mdisp_stereobm_params* p,
                    int16_t* dDisp, void* ws);

// fgs.hip: fast global smoother (the WLS post-filter of postProcess_)
size_t md_fgs_bytes(int H, int W);
int md_fgs_filter(hipStream_t s, const u8* dGuide, size_t gpitch, int H, int W, int cn, float* dU, double lambda, double sigmaColor,
                  double attenuation, int iters, void* ws);
int md_fgs_convert(hipStream_t s, const int16_t* src, float* dst, size_t n);
int md_fgs_convert_back(hipStream_t s, const float* src, int16_t* dst, size_t n);

// legacy.hip: fixed-window matchers of namespace stereomatch_1
// I / best: the integral-table scratch of md_sad_inte (used by the _inteOpti kinds only, may be null otherwise)
int md_legacy_disparity(hipStream_t s, const u8* dL, size_t lp, const u8* dR, size_t rp, int H, int W, int method, int winSize,
                        int T, int minD, int numD, u8* dOut, size_t op, int* I, int* best);
bool md_legacy_uses_tables(int method);

// stereo.hip
int md_cost_box_wta(hipStream_t s, const u8* ref, size_t rp, const u8* pad, size_t pp, int H, int W,
                    int minD, int maxD, int off0, int offStep, int win, int costKind, int outKind, void* out, size_t op);
// I: md_sad_inte_tables() tables of (H+1)*(W+1) words, best: 2*H*W words of device scratch.  floatTables: the reference's
// CV_32F table arithmetic (rounds above 2^24), else exact 32-bit integer tables.
int md_sad_inte_tables(int H, int W, int D);
int md_sad_inte(hipStream_t s, const u8* ref, size_t rp, const u8* pad, size_t pp, int H, int W,
                int minD, int maxD, int off0, int offStep, int win, int floatTables, u8* out, size_t op, int* I, int* best);
int md_legacy_inte(hipStream_t s, const u8* L, size_t lp, const u8* R, size_t rp, int H, int W, int minD, int numD, int win,
                   int square, int clampT, int T, int floatTables, u8* out, size_t op, int* I, int* best);
int md_lr_check(hipStream_t s, const float* dL, const float* dR, int H, int W, float thresh, u8* vL, u8* vR);
int md_remap(hipStream_t s, const u8* src, size_t sp, int sH, int sW, int cn, const int16_t* m1, const u16* m2,
             int dH, int dW, u8* dst, size_t dp);

int md_rectify_maps(hipStream_t s, const double ir[9], const double k[8], double fx, double fy, double u0, double v0, int W, int H,
                    int16_t* m1, u16* m2);
// host.hip: the inverse of P[:3,:3]*R and the padded distortion vector of initUndistortRectifyMap
int md_rectify_map_setup(const double K[9], const double* dist, int ndist, const double R[9], const double P[12], double ir[9], double k[8]);
int md_reproject_q(hipStream_t s, const float* disp, int H, int W, const double Q[16], float* xyz);
int md_reproject_tri(hipStream_t s, const float* disp, int H, int W, const double K[9], double baseline, const float* bgr, float* xyz);

// filters.hip: guided-filter aggregation of D cost slices + first-minimum WTA
size_t md_gif_scratch_bytes(int H, int W, int D);
int md_gif_stereo(mdisp_ctx* c, void* scratch, const u8* dGuide, size_t gpitch, int H, int W, int D, int minD, int r, double eps,
                  int gif_type, int r2, double namuda, const u8* ref, size_t rp, const u8* pad, size_t pp, int off0, int offStep,
                  int costKind, const float* hostCosts, int normMode, float* dDisp32, double* dDisp64,
                  int yA = -1, int yB = -1 /* output rows [yA, yB); default: the whole image */,
                  const float* extMinMax = nullptr, const double* extSteerMax = nullptr /* image-wide statistics, host memory */);
int md_gif_band_stats(mdisp_ctx* c, void* scratch, const u8* dGuide, size_t gpitch, int H, int W, int D, int gif_type,
                      const u8* ref, size_t rp, const u8* pad, size_t pp, int off0, int offStep, int costKind, int yA, int yB,
                      float* minmax, double* steerMax);

// gif_mc.hip: guided filter with a multi-channel guide, getCostSAD_d, computeAdaptiveWeight_GuidedF
int md_mc_check_window(int r);
size_t md_mc_scratch_bytes(int H, int W, int cn, int n, int Wa, int Wb);
size_t md_asw_guidedf_bytes(int H, int W, int cn, int minD, int numD);
int md_asw_guidedf(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, int H, int W, int cn, int isLeft, double eps, int win,
                   int minD, int numD, float* dDisp, void* ws);
int md_cost_sad_d(hipStream_t s, const u8* ref, size_t rp, const u8* oth, size_t op, int H, int W, int off, int win, float* out);
int md_guided_filter_mc(hipStream_t s, const u8* guide, size_t gp, int cn, const float* dP, int H, int W, int r, double eps, int variant,
                        float* dOut, void* scratch);

size_t md_similarity_bytes(int H, int W, int minD, int numD);
int md_compute_similarity(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, int H, int W, double regularity, double thresC, double thresG,
                          int minD, int numD, int border, float* dCost, void* ws);
size_t md_asw_guidedf23_bytes(int H, int W, int which, int isLeft, int minD, int numD);
int md_asw_guidedf23(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, int H, int W, int which, int isLeft, double eps, int win,
                     int minD, int numD, float* dDisp, void* ws);

// cost kinds of md_cost_box_wta
enum { MD_COST_HAMMING = 0, MD_COST_HAMMING_SQR = 1, MD_COST_ABSDIFF_SQR = 2, MD_COST_ABSDIFF = 3 };
// output kinds
enum { MD_OUT_U8 = 0, MD_OUT_F64 = 1, MD_OUT_F32 = 2 };
