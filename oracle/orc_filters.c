/*
 * CPU oracle (test infrastructure only, parity unpinned -- see orc.h):
 * OpenCV-style box filter / min-max normalisation, the guided-image-filter family used as
 * cost-slice aggregator, the steerable-filter orientation-strength map, and the asw_gifs driver.
 *
 * Reference: cameraCalibration/disp_method/gifs.cpp, steerFilter.cpp, methods_disp.cpp:1147-1297.
 */
#include "orc.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/*
 * cv::boxFilter(src, dst, CV_32F, Size(kw,kh)) on a CV_32F source: anchor (kw/2, kh/2),
 * BORDER_REFLECT_101, OpenCV's RowSum<float,double> (sliding sum along x in double) followed by
 * ColumnSum<double,float> (sliding sum down the image in double, scaled by 1/(kw*kh), cast to float).
 */
void orc_box_filter_f32(const float* src, int H, int W, int kw, int kh, int normalize, float* dst)
{
    int ax = kw / 2, ay = kh / 2;
    double* rows = (double*)malloc(sizeof(double) * (size_t)H * W);
#pragma omp parallel for
    for (int y = 0; y < H; y++) {
        const float* S = src + (long)y * W;
        double* D = rows + (long)y * W;
        double s = 0;
        for (int i = 0; i < kw; i++) s += (double)S[orc_reflect101(i - ax, W)];
        D[0] = s;
        for (int i = 0; i < W - 1; i++) {
            s += (double)S[orc_reflect101(i + kw - ax, W)] - (double)S[orc_reflect101(i - ax, W)];
            D[i + 1] = s;
        }
    }
    double scale = normalize ? 1.0 / ((double)kw * kh) : 1.0;
#pragma omp parallel for
    for (int x = 0; x < W; x++) {
        double sum = 0;
        for (int i = 0; i < kh - 1; i++) sum += rows[(long)orc_reflect101(i - ay, H) * W + x];
        for (int y = 0; y < H; y++) {
            double s0 = sum + rows[(long)orc_reflect101(y + kh - 1 - ay, H) * W + x];
            dst[(long)y * W + x] = (float)(s0 * scale);
            sum = s0 - rows[(long)orc_reflect101(y - ay, H) * W + x];
        }
    }
    free(rows);
}

/* cv::normalize(src, dst, 0, 1, NORM_MINMAX, CV_32F): scale = 1/(max-min) (0 when the range is
 * <= DBL_EPSILON), shift = -min*scale, then convertTo.  convertTo works in float for 8U and 32F
 * sources and in double for 32S sources. */
void orc_normalize_minmax_u8(const uint8_t* src, long n, float* dst)
{
    int mn = 255, mx = 0;
    for (long i = 0; i < n; i++) { if (src[i] < mn) mn = src[i]; if (src[i] > mx) mx = src[i]; }
    double scale = ((double)mx - mn > DBL_EPSILON) ? 1.0 / ((double)mx - mn) : 0.0;
    double shift = 0.0 - mn * scale;
    float a = (float)scale, b = (float)shift;
    for (long i = 0; i < n; i++) { volatile float t = (float)src[i] * a; dst[i] = t + b; }
}

void orc_normalize_minmax_s32(const int32_t* src, long n, float* dst)
{
    int mn = src[0], mx = src[0];
    for (long i = 0; i < n; i++) { if (src[i] < mn) mn = src[i]; if (src[i] > mx) mx = src[i]; }
    double scale = ((double)mx - mn > DBL_EPSILON) ? 1.0 / ((double)mx - mn) : 0.0;
    double shift = 0.0 - mn * scale;
    for (long i = 0; i < n; i++) dst[i] = (float)(src[i] * scale + shift);
}

void orc_normalize_minmax_f32(const float* src, long n, float* dst)
{
    float mn = src[0], mx = src[0];
    for (long i = 0; i < n; i++) { if (src[i] < mn) mn = src[i]; if (src[i] > mx) mx = src[i]; }
    double scale = ((double)mx - (double)mn > DBL_EPSILON) ? 1.0 / ((double)mx - (double)mn) : 0.0;
    double shift = 0.0 - (double)mn * scale;
    float a = (float)scale, b = (float)shift;
    for (long i = 0; i < n; i++) { volatile float t = src[i] * a; dst[i] = t + b; }
}

static float* fnew(long n) { return (float*)malloc(sizeof(float) * (size_t)n); }

/* element-wise float helpers; every Mat expression in the reference rounds to float per op */
static void mul_f(const float* a, const float* b, long n, float* o) { for (long i = 0; i < n; i++) o[i] = a[i] * b[i]; }
static void submul_f(const float* c, const float* a, const float* b, long n, float* o)
{ for (long i = 0; i < n; i++) { volatile float t = a[i] * b[i]; o[i] = c[i] - t; } }

/* shared front half: means, variance, covariance (gifs.cpp:58-96 and twins) */
typedef struct { float *meanG, *meanP, *varG, *covGP; } gif_stats;
static gif_stats gif_front(const float* G, const float* P, int H, int W, int r)
{
    long N = (long)H * W;
    gif_stats s;
    s.meanG = fnew(N); s.meanP = fnew(N); s.varG = fnew(N); s.covGP = fnew(N);
    float* t = fnew(N); float* c = fnew(N);
    orc_box_filter_f32(G, H, W, r, r, 1, s.meanG);
    orc_box_filter_f32(P, H, W, r, r, 1, s.meanP);
    mul_f(G, P, N, t);
    orc_box_filter_f32(t, H, W, r, r, 1, c);             /* corrGuidP */
    submul_f(c, s.meanG, s.meanP, N, s.covGP);           /* covGuidP = corrGuidP - meanG.mul(meanP) */
    mul_f(G, G, N, t);
    orc_box_filter_f32(t, H, W, r, r, 1, c);             /* corrGuid */
    submul_f(c, s.meanG, s.meanG, N, s.varG);            /* varGuid = corrGuid - meanG.mul(meanG) */
    free(t); free(c);
    return s;
}
static void gif_free(gif_stats* s) { free(s->meanG); free(s->meanP); free(s->varG); free(s->covGP); }

/*
 * getGuidedFilter, gifs.cpp:47-135 (single-channel guide; twin at
 * methods/method_StereoMatching.cpp:3612-3672).  Inputs are the already min-max-normalised guide
 * and slice (gifs.cpp:55-56); `r` is the full box window size.
 */
void orc_guided_filter(const float* G, const float* P, int H, int W, int r, double eps, float* out)
{
    long N = (long)H * W;
    gif_stats s = gif_front(G, P, H, W, r);
    float* a = fnew(N); float* b = fnew(N);
    float epsf = (float)eps;
    for (long i = 0; i < N; i++) {
        volatile float den = s.varG[i] + epsf;            /* varGuid + ones*eps  (:127) */
        a[i] = s.covGP[i] / den;
        volatile float t = a[i] * s.meanG[i];
        b[i] = s.meanP[i] - t;                            /* (:128) */
    }
    float* ma = fnew(N); float* mb = fnew(N);
    orc_box_filter_f32(a, H, W, r, r, 1, ma);             /* (:130-131) */
    orc_box_filter_f32(b, H, W, r, r, 1, mb);
    for (long i = 0; i < N; i++) { volatile float t = ma[i] * G[i]; out[i] = t + mb[i]; }
    free(a); free(b); free(ma); free(mb); gif_free(&s);
}

/* getGuidedFilter_egif, gifs.cpp:227-273: eps' = mean(varG)*eps + 1e-6 (global mean, :252-253). */
void orc_guided_filter_egif(const float* G, const float* P, int H, int W, int r, double eps, float* out, float* beta)
{
    long N = (long)H * W;
    gif_stats s = gif_front(G, P, H, W, r);
    double sum = 0;
    for (long i = 0; i < N; i++) sum += (double)s.varG[i];
    double mean2Var = sum / ((double)W * H);
    float add = (float)(mean2Var * eps + 0.001 * 0.001);
    float* a = fnew(N); float* b = fnew(N);
    for (long i = 0; i < N; i++) {
        volatile float den = s.varG[i] + add;
        a[i] = s.covGP[i] / den;
        volatile float t = a[i] * s.meanG[i];
        b[i] = s.meanP[i] - t;
    }
    float* ma = fnew(N); float* mb = fnew(N);
    orc_box_filter_f32(a, H, W, r, r, 1, ma);
    orc_box_filter_f32(b, H, W, r, r, 1, mb);
    for (long i = 0; i < N; i++) {
        if (beta) { volatile float om = 1.0f - ma[i]; beta[i] = ma[i] / om; }   /* (:261-268), gamma = 1 */
        volatile float t = ma[i] * G[i];
        out[i] = t + mb[i];
    }
    free(a); free(b); free(ma); free(mb); gif_free(&s);
}

/* 3x3 GaussianBlur(sigma) on float, BORDER_REFLECT_101, separable float kernel */
static void gauss3(const float* src, int H, int W, double sigma, float* dst)
{
    double k[3], ks = 0;
    for (int i = 0; i < 3; i++) { double x = i - 1; k[i] = exp(-(x * x) / (2 * sigma * sigma)); ks += k[i]; }
    float kf[3];
    for (int i = 0; i < 3; i++) kf[i] = (float)(k[i] / ks);
    float* t = fnew((long)H * W);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float s = 0;
            for (int i = 0; i < 3; i++) s += kf[i] * src[(long)y * W + orc_reflect101(x + i - 1, W)];
            t[(long)y * W + x] = s;
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float s = 0;
            for (int i = 0; i < 3; i++) s += kf[i] * t[(long)orc_reflect101(y + i - 1, H) * W + x];
            dst[(long)y * W + x] = s;
        }
    free(t);
}

/* edgeAwareWeight, gifs.cpp:347-374.  With the hard-coded r = 1 the 1x1 box "variance" is
 * G*G - G*G = 0, so gamma is the constant (eps_w * mean(1/eps_w)) blurred: 1 up to float rounding. */
static void edge_aware_weight(const float* G, int H, int W, float* gamma)
{
    long N = (long)H * W;
    float mn = G[0], mx = G[0];
    for (long i = 0; i < N; i++) { if (G[i] < mn) mn = G[i]; if (G[i] > mx) mx = G[i]; }
    double L = (double)mx - (double)mn;
    double eps = (0.001 * L) * (0.001 * L);
    float* varG1 = fnew(N);
    double sum = 0;
    for (long i = 0; i < N; i++) {
        volatile float gg = G[i] * G[i];
        volatile float var = gg - gg;                     /* meanGG - meanG.mul(meanG) with 1x1 boxes */
        varG1[i] = var + (float)eps;
        sum += (double)(1.0f / varG1[i]);
    }
    double val_ = sum / ((double)W * H);
    float* g0 = fnew(N);
    for (long i = 0; i < N; i++) g0[i] = varG1[i] * (float)val_;
    gauss3(g0, H, W, 2.0, gamma);
    free(varG1); free(g0);
}

/* getGuidedFilter_wgif, gifs.cpp:376-409 */
void orc_guided_filter_wgif(const float* G, const float* P, int H, int W, int r, double eps, float* out)
{
    long N = (long)H * W;
    float* gamma = fnew(N);
    edge_aware_weight(G, H, W, gamma);
    gif_stats s = gif_front(G, P, H, W, r);
    float* a = fnew(N); float* b = fnew(N);
    float epsf = (float)eps;
    for (long i = 0; i < N; i++) {
        volatile float e = epsf / gamma[i];
        volatile float den = s.varG[i] + e;
        a[i] = s.covGP[i] / den;
        volatile float t = a[i] * s.meanG[i];
        b[i] = s.meanP[i] - t;
    }
    float* ma = fnew(N); float* mb = fnew(N);
    orc_box_filter_f32(a, H, W, r, r, 1, ma);
    orc_box_filter_f32(b, H, W, r, r, 1, mb);
    for (long i = 0; i < N; i++) { volatile float t = ma[i] * G[i]; out[i] = t + mb[i]; }
    free(a); free(b); free(ma); free(mb); free(gamma); gif_free(&s);
}

/*
 * getGuidedFilter_skwgif2, gifs.cpp:537-617.  a,b with gamma from the r2-window variance
 * (global sum of 1/(var2+namuda), :564-565), then the steer-weighted mean of a and b over
 * Rect(Point(max(0,i-r2/2), max(0,j-r2/2)), Point(min(W-1,i+r2/2), min(H-1,j+r2/2))) -- a
 * half-open rectangle, i.e. columns i-r2/2 .. i+r2/2-1 clipped (:587-611).
 */
void orc_guided_filter_skwgif2(const float* G, const float* P, const double* steerW, int H, int W,
                               int r, int r2, double eps, double namuda, float* out)
{
    long N = (long)H * W;
    gif_stats s = gif_front(G, P, H, W, r);
    float* meanG2 = fnew(N); float* corrG2 = fnew(N); float* t = fnew(N);
    orc_box_filter_f32(G, H, W, r2, r2, 1, meanG2);
    mul_f(G, G, N, t);
    orc_box_filter_f32(t, H, W, r2, r2, 1, corrG2);
    float* temp = fnew(N);
    double sum = 0;
    for (long i = 0; i < N; i++) {
        volatile float m2 = meanG2[i] * meanG2[i];
        volatile float v2 = corrG2[i] - m2;
        temp[i] = v2 + (float)namuda;
        sum += (double)(1.0f / temp[i]);
    }
    float gscale = (float)(sum / ((double)H * W));
    float* a = fnew(N); float* b = fnew(N);
    float epsf = (float)eps;
    for (long i = 0; i < N; i++) {
        volatile float gamma = temp[i] * gscale;
        volatile float e = epsf / gamma;
        volatile float den = s.varG[i] + e;
        a[i] = s.covGP[i] / den;
        volatile float tt = a[i] * s.meanG[i];
        b[i] = s.meanP[i] - tt;
    }
    float* sw = fnew(N);
    for (long i = 0; i < N; i++) sw[i] = (float)steerW[i];
    int hr = r2 / 2;
#pragma omp parallel for
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            int x0 = i - hr > 0 ? i - hr : 0, y0 = j - hr > 0 ? j - hr : 0;
            int x1 = i + hr < W - 1 ? i + hr : W - 1, y1 = j + hr < H - 1 ? j + hr : H - 1;
            double sa = 0, sb = 0, swt = 0;
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) {
                    float w = sw[(long)y * W + x];
                    sa += (double)(w * a[(long)y * W + x]);
                    sb += (double)(w * b[(long)y * W + x]);
                    swt += (double)w;
                }
            float ma = (float)(sa * 1.0 / swt), mb = (float)(sb * 1.0 / swt);
            volatile float tt = ma * G[(long)j * W + i];
            out[(long)j * W + i] = tt + mb;
        }
    free(meanG2); free(corrG2); free(t); free(temp); free(a); free(b); free(sw); gif_free(&s);
}

/*
 * cv::bilateralFilter(src, dst, d, sigmaColor, sigmaSpace) on a CV_32FC1 image, restated from the
 * published OpenCV 4.1.0 imgproc algorithm (bilateralFilter_32f and the scalar loop of
 * BilateralFilter_32f_Invoker; the library itself is not part of the reference tree): radius = d/2
 * (at least 1; cvRound(1.5*sigmaSpace) when d <= 0), BORDER_REFLECT_101, circular footprint,
 * float space weights, colour weights from a 4096-bin table over [0, max-min] with linear
 * interpolation, float accumulation in footprint order, and a plain copy when the image is flat.
 */
void orc_bilateral_f32(const float* src, int H, int W, int d, double sigma_color, double sigma_space, float* dst)
{
    long N = (long)H * W;
    if (sigma_color <= 0) sigma_color = 1;
    if (sigma_space <= 0) sigma_space = 1;
    double gcc = -0.5 / (sigma_color * sigma_color), gsc = -0.5 / (sigma_space * sigma_space);
    int radius = d <= 0 ? (int)lrint(sigma_space * 1.5) : d / 2;
    if (radius < 1) radius = 1;
    d = radius * 2 + 1;
    float mn = src[0], mx = src[0];
    for (long i = 0; i < N; i++) { if (src[i] < mn) mn = src[i]; if (src[i] > mx) mx = src[i]; }
    double minV = mn, maxV = mx;
    if (fabs(minV - maxV) < FLT_EPSILON) { memcpy(dst, src, sizeof(float) * (size_t)N); return; }
    int tw = W + 2 * radius, th = H + 2 * radius;
    float* temp = fnew((long)tw * th);
    for (int y = 0; y < th; y++)
        for (int x = 0; x < tw; x++)
            temp[(long)y * tw + x] = src[(long)orc_reflect101(y - radius, H) * W + orc_reflect101(x - radius, W)];
    const int kExpNumBins = 1 << 12;
    float len = (float)(maxV - minV);
    float scale_index = kExpNumBins / len;
    float* expLUT = fnew(kExpNumBins + 2);
    float lastExpVal = 1.f;
    for (int i = 0; i < kExpNumBins + 2; i++) {
        if (lastExpVal > 0.f) {
            double val = i / scale_index;
            expLUT[i] = (float)exp(val * val * gcc);
            lastExpVal = expLUT[i];
        } else expLUT[i] = 0.f;
    }
    float* sw = fnew((long)d * d);
    int* so = (int*)malloc(sizeof(int) * (size_t)d * d);
    int maxk = 0;
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            double r = sqrt((double)i * i + (double)j * j);
            if (r > radius) continue;
            sw[maxk] = (float)exp(r * r * gsc);
            so[maxk++] = i * tw + j;
        }
#pragma omp parallel for
    for (int y = 0; y < H; y++) {
        const float* sp = temp + (long)(y + radius) * tw + radius;
        for (int x = 0; x < W; x++) {
            float sum = 0, wsum = 0;
            float val0 = sp[x];
            for (int k = 0; k < maxk; k++) {
                float val = sp[x + so[k]];
                float alpha = (float)(fabsf(val - val0) * scale_index);
                int idx = (int)floorf(alpha);
                alpha -= idx;
                volatile float dl = expLUT[idx + 1] - expLUT[idx];
                volatile float t = alpha * dl;
                volatile float cw = expLUT[idx] + t;
                volatile float w = sw[k] * cw;
                volatile float vw = val * w;
                sum += vw;
                wsum += w;
            }
            dst[(long)y * W + x] = sum / wsum;
        }
    }
    free(temp); free(expLUT); free(sw); free(so);
}

/*
 * getGuidedFilter_ours_gif / _ours_gif2 / _ours_gif3, gifs.cpp:620-669, 672-725, 728-804: the
 * coefficients a, b of the WGIF-like (gamma from the r2-window variance), plain and EGIF-like
 * guided filter, then bilateralFilter(a), bilateralFilter(b) with sigmaColor 2, sigmaSpace 3 and
 * diameter r2 (variants 1, 2) or r (variant 3), and a_bil*G + b_bil.  beta (variant 3, :791-800)
 * = a_bil / (1 - a_bil).
 */
void orc_guided_filter_ours(const float* G, const float* P, int H, int W, int r, int r2, double eps, double namuda,
                            int variant, float* out, float* beta)
{
    long N = (long)H * W;
    gif_stats s = gif_front(G, P, H, W, r);
    float* a = fnew(N); float* b = fnew(N); float* den = fnew(N);
    float epsf = (float)eps;
    if (variant == 1) {
        float* meanG2 = fnew(N); float* corrG2 = fnew(N); float* t = fnew(N); float* temp = fnew(N);
        orc_box_filter_f32(G, H, W, r2, r2, 1, meanG2);
        mul_f(G, G, N, t);
        orc_box_filter_f32(t, H, W, r2, r2, 1, corrG2);
        double sum = 0;
        for (long i = 0; i < N; i++) {
            volatile float m2 = meanG2[i] * meanG2[i];
            volatile float v2 = corrG2[i] - m2;
            temp[i] = v2 + (float)namuda;                      /* (:646) */
            sum += (double)(1.0f / temp[i]);
        }
        float gscale = (float)(sum / ((double)H * W));         /* (:647) */
        for (long i = 0; i < N; i++) {
            volatile float gamma = temp[i] * gscale;
            volatile float e = epsf / gamma;
            den[i] = s.varG[i] + e;                            /* (:652) */
        }
        free(meanG2); free(corrG2); free(t); free(temp);
    } else if (variant == 2) {
        for (long i = 0; i < N; i++) den[i] = s.varG[i] + epsf;   /* (:706) */
    } else {
        double sum = 0;
        for (long i = 0; i < N; i++) sum += (double)s.varG[i];
        double mean2Var = sum / ((double)W * H);               /* (:753) */
        float add = (float)(mean2Var * eps + 0.001 * 0.001);
        for (long i = 0; i < N; i++) den[i] = s.varG[i] + add; /* (:754) */
    }
    for (long i = 0; i < N; i++) {
        a[i] = s.covGP[i] / den[i];
        volatile float t = a[i] * s.meanG[i];
        b[i] = s.meanP[i] - t;
    }
    float* ma = fnew(N); float* mb = fnew(N);
    int d = variant == 3 ? r : r2;
    orc_bilateral_f32(a, H, W, d, 2, 3, ma);
    orc_bilateral_f32(b, H, W, d, 2, 3, mb);
    for (long i = 0; i < N; i++) {
        if (beta) { volatile float om = 1.0f - ma[i]; beta[i] = ma[i] / om; }
        volatile float t = ma[i] * G[i];
        out[i] = t + mb[i];
    }
    free(a); free(b); free(den); free(ma); free(mb); gif_free(&s);
}

/* basis functions sampled at integer taps, steerFilter.cpp:3-87 */
static double G2a(int x, int y) { double xs = x * x, ys = y * y; ys = exp(-(xs + ys)); return 0.9213 * ((2 * xs) - 1) * ys; }
static double G2b(int x, int y) { double xs = x * x, ys = y * y; ys = exp(-(xs + ys)); return 1.843 * x * y * ys; }
static double H2a(int x, int y) { double xs = x * x, ys = y * y; ys = exp(-(xs + ys)); return 0.9780 * (-2.254 * x + x * xs) * ys; }
static double H2b(int x, int y) { double xs = x * x, ys = y * y; ys = exp(-(xs + ys)); return 0.9780 * (-0.7515 + xs) * y * ys; }
static double H2c(int x, int y) { double xs = x * x, ys = y * y; ys = exp(-(xs + ys)); return 0.9780 * (-0.7515 + y * y) * x * ys; }
static double G2c(int x, int y) { double xs = x * x, ys = y * y; xs = exp(-(xs + ys)); return 0.9213 * ((2 * ys) - 1) * xs; }
static double H2d(int x, int y) { double xs = pow(x, 2), ys = pow(y, 2); xs = exp(-(xs + ys)); return 0.9780 * (-2.254 * y + y * ys) * xs; }

/*
 * steerFilter, steerFilter.cpp:239-446: pad 6 px BORDER_REFLECT, seven dense 11x11 correlations
 * in double (:298-340), the C2/C3 quadratic forms (:374-395), orientation strength
 * sqrt(C2^2+C3^2) (:403), division by the global maximum times 255 (:400-414), crop back to the
 * source size (:418) and clamp to [1e-6, 120] (:430-443).  The 32-angle `steered` stack
 * (:349-368) never reaches the output and is not restated.
 */
void orc_steer_filter(const uint8_t* src, int H, int W, double* dst)
{
    int w = W + 12, h = H + 12;
    double* img = (double*)malloc(sizeof(double) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            img[(long)y * w + x] = (double)src[(long)orc_reflect(y - 6, H) * W + orc_reflect(x - 6, W)];
    double K[7][11][11];
    for (int y = -5; y <= 5; y++)
        for (int x = -5; x <= 5; x++) {
            K[0][y + 5][x + 5] = G2a(x, y);
            K[1][y + 5][x + 5] = G2b(x, y);
            K[2][y + 5][x + 5] = H2a(x, y);
            K[3][y + 5][x + 5] = H2b(x, y);
            K[4][y + 5][x + 5] = G2c(x, y);
            K[5][y + 5][x + 5] = H2c(x, y);
            K[6][y + 5][x + 5] = H2d(x, y);
        }
    double* O = (double*)calloc((size_t)w * h, sizeof(double));
#pragma omp parallel for
    for (int y = 6; y < h - 6; y++)
        for (int x = 6; x < w - 6; x++) {
            double s[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int k = -5; k <= 5; k++)
                for (int j = -5; j <= 5; j++) {
                    double v = img[(long)(y + k) * w + x + j];
                    for (int f = 0; f < 7; f++) s[f] = s[f] + K[f][k + 5][j + 5] * v;
                }
            double g2a = s[0], g2b = s[1], h2a = s[2], h2b = s[3], g2c = s[4], h2c = s[5], h2d = s[6];
            double C2 = 0.5 * (pow(g2a, 2) - pow(g2c, 2)) + 0.46875 * (pow(h2a, 2) - pow(h2d, 2))
                      + 0.28125 * (pow(h2b, 2) - pow(h2c, 2)) + 0.1875 * (h2a * h2c - h2b * h2d);
            double C3 = -(g2a * g2b) - (g2b * g2c) - 0.9375 * (h2c * h2d - h2a * h2b)
                      - 1.6875 * h2b * h2c - 0.1875 * h2a * h2d;
            O[(long)y * w + x] = sqrt(pow(C2, 2) + pow(C3, 2));
        }
    double mx = 0;
    for (int y = 6; y < h - 6; y++)
        for (int x = 6; x < w - 6; x++)
            if (O[(long)y * w + x] > mx) mx = O[(long)y * w + x];
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            double v = ((O[(long)(y + 6) * w + x + 6] * 1.0 / mx) * 255);
            if (v > 120) v = 120;
            if (v < 1e-6) v = 1e-6;
            dst[(long)y * W + x] = v;
        }
    free(img); free(O);
}

/*
 * asw_gifs, methods_disp.cpp:1147-1297, from the cost slices onward (the dense-SIFT cost source
 * at :1162 is outside the hot path; callers pass the D cost slices).  Per slice: the selected
 * guided filter on (guide, slice), then the strict-'<' WTA of :1278-1295.  Output CV_32F.
 */
int orc_asw_gifs(const uint8_t* guide, const float* costs, int H, int W, int D, int minD,
                 int winSize, double eps, int gifType, int r2, double namuda, float* disp, float* filtered)
{
    if (winSize % 2 == 0) return -1;
    long N = (long)H * W;
    float* G = fnew(N); float* P = fnew(N); float* o = fnew(N);
    double* best = (double*)malloc(sizeof(double) * N);
    double* steer = NULL;
    orc_normalize_minmax_u8(guide, N, G);
    if (gifType == ORC_SKWGIF) {
        steer = (double*)malloc(sizeof(double) * N);
        orc_steer_filter(guide, H, W, steer);
    }
    for (long i = 0; i < N; i++) { best[i] = DBL_MAX; disp[i] = 0.f; }
    for (int k = 0; k < D; k++) {
        orc_normalize_minmax_f32(costs + (long)k * N, N, P);
        switch (gifType) {
        case ORC_GIF: orc_guided_filter(G, P, H, W, winSize, eps, o); break;
        case ORC_WGIF: orc_guided_filter_wgif(G, P, H, W, winSize, eps, o); break;
        case ORC_EGIF: orc_guided_filter_egif(G, P, H, W, winSize, eps, o, NULL); break;
        case ORC_SKWGIF: orc_guided_filter_skwgif2(G, P, steer, H, W, winSize, r2, eps, namuda, o); break;
        case ORC_OURS_GIF: case ORC_OURS_GIF2: case ORC_OURS_GIF3:
            orc_guided_filter_ours(G, P, H, W, winSize, r2, eps, namuda, gifType - ORC_OURS_GIF + 1, o, NULL); break;
        default: return -2;
        }
        if (filtered) memcpy(filtered + (long)k * N, o, sizeof(float) * N);
        for (long i = 0; i < N; i++)
            if ((double)o[i] < best[i]) { best[i] = (double)o[i]; disp[i] = (float)(k + minD); }
    }
    free(G); free(P); free(o); free(best); free(steer);
    return 0;
}
