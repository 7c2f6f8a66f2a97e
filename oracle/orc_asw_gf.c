/*
 * CPU oracle (test infrastructure only, parity unpinned -- see orc.h): the guided-filter matcher of namespace
 * stereomatch_1 and what it is made of.
 *
 *  - orc_cost_sad_d: stereomatch_1::getCostSAD_d, cameraCalibration/methods/method_StereoMatching.cpp:3288-3349.
 *  - orc_guided_filter_mc: getGuidedFilter with a multi-channel guide, disp_method/gifs.cpp:47-135 (variant 0) and its
 *    twin methods/method_StereoMatching.cpp:3612-3672 (variant 1), with multiChl_to_oneChl_mul (gifs.cpp:8-45,
 *    method_StereoMatching.cpp:3573-3610).
 *  - orc_adaptive_weight_guidedf: stereomatch_1::computeAdaptiveWeight_GuidedF, method_StereoMatching.cpp:3685-3781.
 *
 * OpenCV semantics assumed (4.1.0, not part of the reference tree): cv::normalize(NORM_MINMAX) of a multi-channel
 * image takes ONE minimum / maximum over all channels; cv::boxFilter filters every channel on its own
 * (RowSum<float,double>, ColumnSum<double,float>, BORDER_REFLECT_101); Mat expressions round to float per
 * operation; cv::divide writes 0 where the divisor is 0; Vec::dot accumulates in float, channel 0 first.
 */
#include "orc.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float* fnew(long n) { return (float*)malloc(sizeof(float) * (size_t)n); }

/*
 * getCostSAD_d (:3288-3349).  One of the two gray images is wider (bordered by the caller with BORDER_REFLECT):
 * DISPARITY_LEFT: the right one, compared at column offset rightW - W - disparity (:3323); DISPARITY_RIGHT: the left
 * one at offset disparity (:3338).  |difference| as float, then boxFilter(normalize = true) (:3326, :3341).
 * Returns -1 where the reference returns an empty image (even window, bordered image not wider).
 */
int orc_cost_sad_d(const uint8_t* left, int leftW, const uint8_t* right, int rightW, int H, int disparity, int isLeft, int winSize,
                   float* out)
{
    if (winSize % 2 == 0) return -1;
    const int W = isLeft ? leftW : rightW;
    if (isLeft ? rightW <= W : leftW <= W) return -1;
    const int off = isLeft ? rightW - W - disparity : disparity;
    if (off < 0 || off + W > (isLeft ? rightW : leftW)) return -2;      /* the reference's ROI would throw */
    float* d = fnew((long)H * W);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int a = isLeft ? left[(long)y * leftW + x] : left[(long)y * leftW + off + x];
            const int b = isLeft ? right[(long)y * rightW + off + x] : right[(long)y * rightW + x];
            d[(long)y * W + x] = (float)(a > b ? a - b : b - a);
        }
    orc_box_filter_f32(d, H, W, winSize, winSize, 1, out);
    free(d);
    return 0;
}

/*
 * Guided filter with a cn-channel guide.  G: cn planes of H*W, already normalised; P: the normalised slice.
 *   meanG_c, meanP, corrGP_c = box(G_c * P), corrG_c = box(G_c * G_c), var_c = corrG_c - meanG_c^2,
 *   cov_c = corrGP_c - meanG_c * meanP                                                  (gifs.cpp:58-96)
 *   variant 0: a_c = cov_c / (var_c + eps)                                              (gifs.cpp:127)
 *   variant 1: a_c = cov_c / (var_c + (var_c / var_c) * eps), divisions by zero give 0   (method_StereoMatching.cpp:3663)
 *   b = meanP - dot(a, meanG); a_c, b box-filtered; out = dot(a, G) + b                  (gifs.cpp:128-134)
 */
void orc_guided_filter_mc(const float* G, int cn, const float* P, int H, int W, int r, double eps, int variant, float* out)
{
    const long N = (long)H * W;
    float* meanG = fnew(N * cn);
    float* a = fnew(N * cn);
    float* meanP = fnew(N);
    float* t = fnew(N);
    float* c = fnew(N);
    const float epsf = (float)eps;
    orc_box_filter_f32(P, H, W, r, r, 1, meanP);
    for (int ch = 0; ch < cn; ch++) {
        const float* g = G + ch * N;
        float* mg = meanG + ch * N;
        orc_box_filter_f32(g, H, W, r, r, 1, mg);
        for (long i = 0; i < N; i++) t[i] = g[i] * P[i];
        orc_box_filter_f32(t, H, W, r, r, 1, c);                      /* corrGuidP channel */
        float* cov = a + ch * N;
        for (long i = 0; i < N; i++) { volatile float m = mg[i] * meanP[i]; cov[i] = c[i] - m; }
        for (long i = 0; i < N; i++) t[i] = g[i] * g[i];
        orc_box_filter_f32(t, H, W, r, r, 1, c);                      /* corrGuid channel */
        for (long i = 0; i < N; i++) {
            volatile float mm = mg[i] * mg[i];
            volatile float var = c[i] - mm;
            volatile float den;
            if (variant == 0) den = var + epsf;
            else {
                /* (varGuid / varGuid) * eps is one scaled division: var * eps / var, 0 where var == 0 */
                volatile float ve = var * epsf;
                volatile float q = var != 0.f ? ve / var : 0.f;
                den = var + q;
            }
            cov[i] = den != 0.f ? cov[i] / den : 0.f;                  /* a */
        }
    }
    float* b = fnew(N);
    for (long i = 0; i < N; i++) {
        volatile float s = 0.f;
        for (int ch = 0; ch < cn; ch++) { volatile float p = a[ch * N + i] * meanG[ch * N + i]; s = s + p; }
        b[i] = meanP[i] - s;
    }
    float* mb = fnew(N);
    orc_box_filter_f32(b, H, W, r, r, 1, mb);
    for (int ch = 0; ch < cn; ch++) {
        orc_box_filter_f32(a + ch * N, H, W, r, r, 1, t);
        memcpy(a + ch * N, t, sizeof(float) * (size_t)N);
    }
    for (long i = 0; i < N; i++) {
        volatile float s = 0.f;
        for (int ch = 0; ch < cn; ch++) { volatile float p = a[ch * N + i] * G[ch * N + i]; s = s + p; }
        out[i] = s + mb[i];
    }
    free(meanG); free(a); free(meanP); free(t); free(c); free(b); free(mb);
}

/* the planes of the slice's guide: the reference view's channels, then the other view's at the slice's offset, normalised
 * with one minimum / maximum over all of them (cv::normalize on the merged image, :3620) */
static void guide_planes(const uint8_t* ref, const uint8_t* oth, int othW, int off, int H, int W, int cn, int refFirst, float* G)
{
    const long N = (long)H * W;
    int mn = 255, mx = 0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int ch = 0; ch < cn; ch++) {
                const int a = ref[((long)y * W + x) * cn + ch], b = oth[((long)y * othW + off + x) * cn + ch];
                if (a < mn) mn = a;
                if (a > mx) mx = a;
                if (b < mn) mn = b;
                if (b > mx) mx = b;
            }
    const double scale = ((double)mx - mn > DBL_EPSILON) ? 1.0 / ((double)mx - mn) : 0.0;
    const double shift = 0.0 - mn * scale;
    const float sa = (float)scale, sb = (float)shift;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int ch = 0; ch < cn; ch++) {
                volatile float t0 = (float)ref[((long)y * W + x) * cn + ch] * sa;
                volatile float t1 = (float)oth[((long)y * othW + off + x) * cn + ch] * sa;
                G[(long)((refFirst ? 0 : cn) + ch) * N + (long)y * W + x] = t0 + sb;
                G[(long)((refFirst ? cn : 0) + ch) * N + (long)y * W + x] = t1 + sb;
            }
}

static void pad_reflect_cn(const uint8_t* src, int H, int W, int cn, int left, int right, uint8_t* dst)
{
    const int Wp = W + left + right;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < Wp; x++) {
            const int sx = orc_reflect(x - left, W);
            memcpy(dst + ((long)y * Wp + x) * cn, src + ((long)y * W + sx) * cn, (size_t)cn);
        }
}

/*
 * computeAdaptiveWeight_GuidedF (:3685-3781).  L, R: interleaved 8-bit images with cn channels.  Per candidate
 * d = minDisparity + i: SAD slice (getCostSAD_d on the gray images), guide = the 2 * cn channels [left, right] of the two
 * views aligned at d (left view first in both branches, :3725-3730, :3742-3747), getGuidedFilter(guide, slice, winSize, eps),
 * then the first minimum over the slices (strict '<' from DBL_MAX), map value d as float.
 * Returns -2 where the reference throws: gray input makes a 2-channel guide, for which multiChl_to_oneChl_mul returns an
 * empty image (:3578-3609) and the subtraction from it fails.
 */
int orc_adaptive_weight_guidedf(const uint8_t* L, const uint8_t* R, int H, int W, int cn, int isLeft, double eps, int winSize, int minD,
                                int numD, float* depth)
{
    if (!L || !R || !depth || numD <= 0 || minD < 0) return -1;
    if (cn != 3) return -2;
    if (winSize % 2 == 0) return -3;
    const long N = (long)H * W;
    const int maxOff = minD + numD - 1, Wp = W + maxOff;
    uint8_t* refC = (uint8_t*)(isLeft ? L : R);
    uint8_t* othP = (uint8_t*)malloc((size_t)H * Wp * cn);
    if (isLeft) pad_reflect_cn(R, H, W, cn, maxOff, 0, othP); else pad_reflect_cn(L, H, W, cn, 0, maxOff, othP);
    uint8_t* refG = (uint8_t*)malloc((size_t)N);
    uint8_t* othG = (uint8_t*)malloc((size_t)H * Wp);
    orc_bgr2gray(refC, H, W, refG);
    orc_bgr2gray(othP, H, Wp, othG);
    float* cost = fnew(N); float* Pn = fnew(N); float* G = fnew(N * 2 * cn); float* out = fnew(N);
    double* best = (double*)malloc(sizeof(double) * (size_t)N);
    for (long i = 0; i < N; i++) { best[i] = DBL_MAX; depth[i] = 0.f; }
    int rc = 0;
    for (int i = 0; i < numD && !rc; i++) {
        const int d = minD + i;
        if (isLeft) rc = orc_cost_sad_d(refG, W, othG, Wp, H, d, 1, winSize, cost);
        else rc = orc_cost_sad_d(othG, Wp, refG, W, H, d, 0, winSize, cost);
        if (rc) break;
        orc_normalize_minmax_f32(cost, N, Pn);
        /* left: right_border(Rect(numDisparity - i - 1)) = the cost's own offset Wp - W - d; right: left_border(Rect(i + minDisparity)) */
        const int off = isLeft ? numD - i - 1 : i + minD;
        guide_planes(refC, othP, Wp, off, H, W, cn, isLeft, G);
        orc_guided_filter_mc(G, 2 * cn, Pn, H, W, winSize, eps, 1, out);
        for (long p = 0; p < N; p++)
            if ((double)out[p] < best[p]) { best[p] = (double)out[p]; depth[p] = (float)d; }
    }
    free(othP); free(refG); free(othG); free(cost); free(Pn); free(G); free(out); free(best);
    return rc;
}

/*
 * stereomatch_1::computeSimilarity, nine-argument overload, colour input with DISPARITY_LEFT
 * (method_StereoMatching.cpp:1108-1180) -- the only branch of that function that runs: the others multiply a CV_32F
 * matrix by a CV_8U mask (:1218, :1257, :1292), which the library refuses.  As the expressions evaluate in OpenCV 4.1:
 *   colour: |L - R(x - d)| per channel (8-bit); (c0 + c1 + c2) / 3 = addWeighted(saturate(c0 + c1), 1/3, c2, 1/3) in float,
 *           rounded to 8 bits (:1152); where that exceeds thresC the cost is saturate(colour + thresC), elsewhere 0
 *           (:1155-1157: colour.mul(mask / 255) + thresC * (mask / 255)), then float.
 *   gradient: filter2D with [-3 0 3; -10 0 10; -3 0 3] (BORDER_REFLECT_101; the right view is filtered after its
 *           BORDER_REFLECT padding) as float, |difference| per channel, (g0 + g1) then addWeighted(., 1/3, g2, 1/3) in
 *           float (:1166); where that exceeds thresG the cost is g + 254 * thresG, elsewhere 255 * thresG (:1170-1175:
 *           the bitwise_not of a 0 / 1 mask is 255 / 254).
 *   slice = addWeighted(colour cost, 1 - regularity, gradient cost, regularity) in float (:1177).
 * L, R: interleaved 3-channel 8-bit images; cost: numD slices of H * W floats.  Returns -2 for the inputs on which the
 * reference fails (gray images, DISPARITY_RIGHT) and -1 for a size mismatch (it returns without slices).
 */
static void sobel3(const uint8_t* img, int H, int W, float* out /* H*W*3 */)
{
    static const int kx[3] = {-3, -10, -3};
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < 3; c++) {
                int s = 0;
                for (int i = -1; i <= 1; i++) {
                    const int yy = orc_reflect101(y + i, H), xl = orc_reflect101(x - 1, W), xr = orc_reflect101(x + 1, W);
                    s += kx[i + 1] * ((int)img[((long)yy * W + xl) * 3 + c] - (int)img[((long)yy * W + xr) * 3 + c]);
                }
                out[((long)y * W + x) * 3 + c] = (float)s;
            }
}

int orc_compute_similarity(const uint8_t* L, const uint8_t* R, int H, int W, int cn, int isLeft, double regularity, double thresC,
                           double thresG, int minD, int numD, float* cost)
{
    if (!L || !R || !cost || numD <= 0 || minD < 0) return -1;
    if (cn != 3 || !isLeft) return -2;
    const long N = (long)H * W;
    const int maxOff = minD + numD - 1, Wp = W + maxOff;
    uint8_t* Rb = (uint8_t*)malloc((size_t)H * Wp * 3);
    pad_reflect_cn(R, H, W, 3, maxOff, 0, Rb);
    float* sL = fnew(N * 3); float* sR = fnew((long)H * Wp * 3);
    sobel3(L, H, W, sL);
    sobel3(Rb, H, Wp, sR);
    const float third = (float)(1.0 / 3.0), betaC = (float)(thresC / 255.0), tG = (float)thresG;
    const float wC = (float)(1.0 - regularity), wG = (float)regularity;
    for (int k = 0; k < numD; k++) {
        const int off = maxOff - (minD + k);
        float* out = cost + (long)k * N;
#pragma omp parallel for
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                int c[3]; float g[3];
                for (int ch = 0; ch < 3; ch++) {
                    const int a = L[((long)y * W + x) * 3 + ch], b = Rb[((long)y * Wp + off + x) * 3 + ch];
                    c[ch] = a > b ? a - b : b - a;
                    g[ch] = fabsf(sL[((long)y * W + x) * 3 + ch] - sR[((long)y * Wp + off + x) * 3 + ch]);
                }
                int m1 = c[0] + c[1]; if (m1 > 255) m1 = 255;
                volatile float t0 = (float)m1 * third, t1 = (float)c[2] * third;
                volatile float cs = t0 + t1;
                int colour = (int)lrintf(cs); if (colour > 255) colour = 255;
                float cc = 0.f;
                if ((double)colour > thresC) {
                    volatile float u0 = 255.0f * betaC;
                    volatile float u1 = u0 + (float)colour;
                    int v = (int)lrintf(u1); if (v > 255) v = 255; if (v < 0) v = 0;
                    cc = (float)v;
                }
                volatile float g01 = g[0] + g[1];
                volatile float h0 = g01 * third, h1 = g[2] * third;
                volatile float grad = h0 + h1;
                float cg;
                if (grad > tG) { volatile float p = 254.0f * tG; cg = p + grad; }
                else { volatile float p = 255.0f * tG; cg = p; }
                volatile float e0 = cc * wC, e1 = cg * wG;
                out[(long)y * W + x] = e0 + e1;
            }
    }
    free(Rb); free(sL); free(sR);
    return 0;
}

/* COLOR_RGB2GRAY on a BGR image, as computeNCC converts its colour input (method_StereoMatching.cpp:934-942): the first
 * channel gets the red weight */
static void rgb_order_gray(const uint8_t* img, long n, uint8_t* gray)
{
    for (long i = 0; i < n; i++)
        gray[i] = (uint8_t)((img[3 * i] * 4899 + img[3 * i + 1] * 9617 + img[3 * i + 2] * 1868 + 8192) >> 14);
}

/*
 * computeAdaptiveWeight_GuidedF_2 (:3794-3868): slices from computeSimilarity(L, R, 0.4, 10, 50, dispType, minD, numD),
 * each filtered with getGuidedFilter(left view (3 channels), slice, winSize, eps), first minimum.  DISPARITY_RIGHT fails
 * inside computeSimilarity (-2).
 * computeAdaptiveWeight_GuidedF_3 (:3881-3955): slices from the cost-volume overload of computeNCC; DISPARITY_LEFT filters
 * with the six-channel guide of computeAdaptiveWeight_GuidedF, DISPARITY_RIGHT with the right view alone (:3924).
 */
int orc_adaptive_weight_guidedf_23(const uint8_t* L, const uint8_t* R, int H, int W, int cn, int which, int isLeft, double eps, int winSize,
                                   int minD, int numD, float* depth)
{
    if (!L || !R || !depth || numD <= 0 || minD < 0) return -1;
    if (cn != 3) return -2;
    if (winSize % 2 == 0) return -3;
    const long N = (long)H * W;
    const int maxOff = minD + numD - 1, Wp = W + maxOff;
    float* vol = fnew(N * numD);
    int rc;
    if (which == 2) rc = orc_compute_similarity(L, R, H, W, cn, isLeft, 0.4, 10.0, 50.0, minD, numD, vol);
    else {
        uint8_t* gl = (uint8_t*)malloc((size_t)N); uint8_t* gr = (uint8_t*)malloc((size_t)N);
        rgb_order_gray(L, N, gl); rgb_order_gray(R, N, gr);
        rc = orc_ncc_volume(gl, gr, H, W, isLeft, winSize, minD, numD, vol);
        free(gl); free(gr);
    }
    if (rc) { free(vol); return rc < 0 ? rc : -3; }
    const int six = which == 3 && isLeft;
    uint8_t* othP = NULL;
    if (six) { othP = (uint8_t*)malloc((size_t)H * Wp * cn); pad_reflect_cn(R, H, W, cn, maxOff, 0, othP); }
    const int gcn = six ? 2 * cn : cn;
    float* G = fnew(N * gcn); float* Pn = fnew(N); float* out = fnew(N);
    double* best = (double*)malloc(sizeof(double) * (size_t)N);
    for (long i = 0; i < N; i++) { best[i] = DBL_MAX; depth[i] = 0.f; }
    const uint8_t* single = which == 2 ? (isLeft ? L : R) : R;      /* :3822 / :3837; :3924 */
    if (!six) {
        int mn = 255, mx = 0;
        for (long i = 0; i < N * cn; i++) { if (single[i] < mn) mn = single[i]; if (single[i] > mx) mx = single[i]; }
        const double scale = ((double)mx - mn > DBL_EPSILON) ? 1.0 / ((double)mx - mn) : 0.0;
        const float sa = (float)scale, sb = (float)(0.0 - mn * scale);
        for (long i = 0; i < N; i++)
            for (int ch = 0; ch < cn; ch++) { volatile float t = (float)single[i * cn + ch] * sa; G[(long)ch * N + i] = t + sb; }
    }
    for (int i = 0; i < numD; i++) {
        if (six) guide_planes(L, othP, Wp, numD - i - 1, H, W, cn, 1, G);
        orc_normalize_minmax_f32(vol + (long)i * N, N, Pn);
        orc_guided_filter_mc(G, gcn, Pn, H, W, winSize, eps, 1, out);
        for (long p = 0; p < N; p++)
            if ((double)out[p] < best[p]) { best[p] = (double)out[p]; depth[p] = (float)(minD + i); }
    }
    free(vol); free(othP); free(G); free(Pn); free(out); free(best);
    return 0;
}
