/* CPU oracle (TEST INFRASTRUCTURE): the post-filter of postProcess_ (disp_method/methods_disp.cpp:1299-1313):
 * ximgproc::createDisparityWLSFilterGeneric(false) with lambda 8000 and sigmaColor 0.5, filter(disp, view, out, Mat(), ROI)
 * with ROI = the whole frame (computeROI_my, :1315-1334).  Without confidence that filter is
 * fastGlobalSmootherFilter(view, disp, out, lambda, sigma_color) with lambda_attenuation 0.25 and 3 iterations, output of
 * the input's type (setDepthDiscontinuityRadius and setLRCthresh only act on the confidence branch).
 *
 * The arithmetic is opencv_contrib 4.1.0's modules/ximgproc/src/fgs_filter.cpp (Min et al., "Fast global image smoothing
 * based on weighted least squares", 2014), not vendored in the reference and restated from the published algorithm:
 * PARITY WITH THE REAL LIBRARY IS UNPINNED.  Conventions assumed:
 *   - weights: LUT[i] = -exp(-sqrt(i) / sigmaColor) in float, i = squared colour distance of neighbouring guide pixels
 *     (sum over the channels); Chor(y, x) joins x and x+1 (0 in the last column), Cvert(y, x) joins y and y+1.
 *   - one iteration = horizontal pass (every row) then vertical pass (every column); lambda *= attenuation after it.
 *   - a pass solves (I + lambda * A) u = f along the line by forward elimination and back substitution in float:
 *       coef_prev = lambda * C[j-1]; coef_cur = lambda * C[j];
 *       denom = (1 - coef_prev - coef_cur) - interD[j-1] * coef_prev;
 *       interD[j] = coef_cur / denom;  u[j] = (u[j] - u[j-1] * coef_prev) / denom;     then u[j] -= interD[j] * u[j+1].
 *   - CV_16S input is converted to float, filtered and converted back with round-to-nearest-even and saturation. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>

void orc_fgs_weights_lut(double sigmaColor, float* lut /* 3*255*255 + 1 */)
{
    const float s = (float)sigmaColor;
    for (int i = 0; i <= 3 * 255 * 255; i++) lut[i] = -expf(-sqrtf((float)i) / s);
}

static void fgs_line(float* u, const float* C, float* interD, int n, size_t stride, float lambda)
{
    float coef_cur = lambda * C[0];
    float denom = 1.0f - coef_cur;
    interD[0] = coef_cur / denom;
    u[0] = u[0] / denom;
    for (int j = 1; j < n; j++) {
        const float coef_prev = lambda * C[(size_t)(j - 1) * stride];
        coef_cur = lambda * C[(size_t)j * stride];
        denom = (1.0f - coef_prev - coef_cur) - interD[j - 1] * coef_prev;
        interD[j] = coef_cur / denom;
        u[(size_t)j * stride] = (u[(size_t)j * stride] - u[(size_t)(j - 1) * stride] * coef_prev) / denom;
    }
    for (int j = n - 2; j >= 0; j--) u[(size_t)j * stride] -= interD[j] * u[(size_t)(j + 1) * stride];
}

void orc_fgs_filter_f32(const uint8_t* guide, int H, int W, int cn, const float* src, double lambda_, double sigmaColor,
                        double attenuation, int iters, float* dst)
{
    float* lut = (float*)malloc(sizeof(float) * (3 * 255 * 255 + 1));
    orc_fgs_weights_lut(sigmaColor, lut);
    float* Ch = (float*)malloc(sizeof(float) * (size_t)H * W);
    float* Cv = (float*)malloc(sizeof(float) * (size_t)H * W);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const uint8_t* p = guide + ((size_t)y * W + x) * cn;
            int dh = 0, dv = 0;
            for (int c = 0; c < cn; c++) {
                if (x + 1 < W) { const int d = (int)p[c] - (int)p[cn + c]; dh += d * d; }
                if (y + 1 < H) { const int d = (int)p[c] - (int)p[(size_t)W * cn + c]; dv += d * d; }
            }
            Ch[(size_t)y * W + x] = x + 1 < W ? lut[dh] : 0.0f;
            Cv[(size_t)y * W + x] = y + 1 < H ? lut[dv] : 0.0f;
        }
    for (size_t i = 0; i < (size_t)H * W; i++) dst[i] = src[i];
    float lambda = (float)lambda_;
    for (int it = 0; it < iters; it++) {
#pragma omp parallel
        {
            float* interD = (float*)malloc(sizeof(float) * (size_t)(W > H ? W : H));
#pragma omp for
            for (int y = 0; y < H; y++) fgs_line(dst + (size_t)y * W, Ch + (size_t)y * W, interD, W, 1, lambda);
#pragma omp for
            for (int x = 0; x < W; x++) fgs_line(dst + x, Cv + x, interD, H, (size_t)W, lambda);
            free(interD);
        }
        lambda *= (float)attenuation;
    }
    free(lut); free(Ch); free(Cv);
}

void orc_fgs_filter_s16(const uint8_t* guide, int H, int W, int cn, const int16_t* src, double lambda, double sigmaColor,
                        double attenuation, int iters, int16_t* dst)
{
    float* f = (float*)malloc(sizeof(float) * (size_t)H * W * 2);
    for (size_t i = 0; i < (size_t)H * W; i++) f[i] = (float)src[i];
    orc_fgs_filter_f32(guide, H, W, cn, f, lambda, sigmaColor, attenuation, iters, f + (size_t)H * W);
    for (size_t i = 0; i < (size_t)H * W; i++) {
        const float r = nearbyintf(f[(size_t)H * W + i]);
        dst[i] = (int16_t)(r < -32768.0f ? -32768 : r > 32767.0f ? 32767 : (int)r);
    }
    free(f);
}
