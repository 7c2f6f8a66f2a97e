/* TEST INFRASTRUCTURE ONLY: CPU restatement of stereomatch_1::computeNCC (reference
 * methods/method_StereoMatching.cpp:871-1106), used by tests/ to check the HIP path.  Parity unpinned: the reference
 * ships no output of this function; the arithmetic below follows its source line by line.
 *
 * getInputImgNCC (:871-904): the support window of pixel (x, y) is the winSize x winSize block of the image padded by
 * winSize/2 with BORDER_REFLECT, converted to float, minus the float mean at (x, y), where the mean is
 * boxFilter(src, CV_32F, Size(winSize, winSize)) -- normalised, BORDER_REFLECT_101 (the library's default), i.e.
 * (float)(integer window sum * (1.0 / winSize^2)).  The two borders differ: that is what the source does.
 * computeNCC: the views are matched through a REFLECT pad of max_offset columns (right view padded on the left for the
 * left map, left view padded on the right for the right map), and the "correlation" is
 *     sum(l .* r) / (sum(l .* l) * sum(r .* r))
 * -- products rounded to float, sums in double, the denominator a PRODUCT of the two energies (no square root).
 * Map overload (:915-1015): offsets minDisparity .. minDisparity + numDisparity - 2 (the last one is never tried),
 * the left map keeps the offset of the strictly smallest value.  The right map compares with "> DBL_MAX", never
 * assigns and returns uninitialised memory in the reference: here it is all zeros.  Pixels where no offset was
 * accepted (all values NaN, or numDisparity == 1) are zero as well.
 * Volume overload (:1017-1106): every offset minDisparity .. minDisparity + numDisparity - 1, float slices, each
 * normalised to [0, 1] with cv::normalize(NORM_MINMAX). */
#include "orc.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int refl(int p, int n)          /* BORDER_REFLECT: fedcba|abcdefgh|hgfedcb */
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - p - 1;
    return p;
}
static int refl101(int p, int n)       /* BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba */
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - p - 2;
    return p;
}

typedef struct { int H, W; const uint8_t* img; float* mean; double* energy; } NccView;

static float win_at(const NccView* v, int x, int y, int i, int j, int h)
{
    return (float)v->img[(size_t)refl(y + j - h, v->H) * v->W + refl(x + i - h, v->W)] - v->mean[(size_t)y * v->W + x];
}

static void ncc_prepare(NccView* v, int win)
{
    const int h = win / 2, H = v->H, W = v->W;
    const double scale = 1.0 / ((double)win * win);
    v->mean = (float*)malloc(sizeof(float) * (size_t)H * W);
    v->energy = (double*)malloc(sizeof(double) * (size_t)H * W);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            long s = 0;
            for (int j = -h; j <= h; j++)
                for (int i = -h; i <= h; i++) s += v->img[(size_t)refl101(y + j, H) * W + refl101(x + i, W)];
            v->mean[(size_t)y * W + x] = (float)((double)s * scale);
        }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            double e = 0.0;
            for (int j = 0; j < win; j++)
                for (int i = 0; i < win; i++) { const float a = win_at(v, x, y, i, j, h); const float p = a * a; e += (double)p; }
            v->energy[(size_t)y * W + x] = e;
        }
}

static double ncc_value(const NccView* L, int xl, const NccView* R, int xr, int y, int win)
{
    const int h = win / 2;
    double c = 0.0;
    for (int j = 0; j < win; j++)
        for (int i = 0; i < win; i++) { const float p = win_at(L, xl, y, i, j, h) * win_at(R, xr, y, i, j, h); c += (double)p; }
    return c / (L->energy[(size_t)y * L->W + xl] * R->energy[(size_t)y * R->W + xr]);
}

/* builds the two views (one of them padded by max_offset reflected columns); returns the padded buffer to free */
static uint8_t* ncc_views(const uint8_t* L, const uint8_t* R, int H, int W, int is_left, int max_offset, NccView* vl, NccView* vr)
{
    const int Wp = W + max_offset;
    uint8_t* pad = (uint8_t*)malloc((size_t)H * Wp);
    const uint8_t* src = is_left ? R : L;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < Wp; x++) pad[(size_t)y * Wp + x] = src[(size_t)y * W + refl(is_left ? x - max_offset : x, W)];
    vl->H = vr->H = H;
    if (is_left) { vl->img = L; vl->W = W; vr->img = pad; vr->W = Wp; }
    else { vl->img = pad; vl->W = Wp; vr->img = R; vr->W = W; }
    return pad;
}

int orc_ncc_disparity(const uint8_t* L, const uint8_t* R, int H, int W, int is_left, int win, int minD, int numD, float* out)
{
    if (win % 2 == 0 || win < 1 || numD < 1 || minD < 0) return 1;
    memset(out, 0, sizeof(float) * (size_t)H * W);
    if (!is_left) return 0;
    const int max_offset = minD + numD - 1;
    NccView vl, vr;
    uint8_t* pad = ncc_views(L, R, H, W, 1, max_offset, &vl, &vr);
    ncc_prepare(&vl, win); ncc_prepare(&vr, win);
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            double best = DBL_MAX;
            for (int off = minD; off < max_offset; off++) {
                const double c = ncc_value(&vl, x, &vr, x + max_offset - off, y, win);
                if (c < best) { best = c; out[(size_t)y * W + x] = (float)off; }
            }
        }
    free(vl.mean); free(vl.energy); free(vr.mean); free(vr.energy); free(pad);
    return 0;
}

/* cost: numD slices of H*W floats, slice k = offset minD + k */
int orc_ncc_volume(const uint8_t* L, const uint8_t* R, int H, int W, int is_left, int win, int minD, int numD, float* cost)
{
    if (win % 2 == 0 || win < 1 || numD < 1 || minD < 0) return 1;
    const int max_offset = minD + numD - 1;
    const long N = (long)H * W;
    NccView vl, vr;
    uint8_t* pad = ncc_views(L, R, H, W, is_left, max_offset, &vl, &vr);
    ncc_prepare(&vl, win); ncc_prepare(&vr, win);
    float* cur = (float*)malloc(sizeof(float) * (size_t)N);
    for (int off = minD; off <= max_offset; off++) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++)
                cur[(size_t)y * W + x] = (float)(is_left ? ncc_value(&vl, x, &vr, x + max_offset - off, y, win)
                                                         : ncc_value(&vl, x + off, &vr, x, y, win));
        orc_normalize_minmax_f32(cur, N, cost + (long)(off - minD) * N);
    }
    free(cur); free(vl.mean); free(vl.energy); free(vr.mean); free(vr.energy); free(pad);
    return 0;
}
