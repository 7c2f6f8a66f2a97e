/*
 * CPU oracle (test infrastructure only, parity unpinned -- see orc.h): the fixed-window matchers of
 * namespace stereomatch_1, cameraCalibration/methods/method_StereoMatching.cpp:32-869
 * (computeAD / TAD / SD, computeSAD / STAD / SSD and their _inteOpti / _BFOpti variants).
 * Left and right CV_8UC1, `numDisparity` is a count, output CV_8U holding (char)(16 * index).
 *
 *  - AD, TAD, SD (:32-72, :311-351, :596-637): columns i >= numDisparity - minDisparity only; the left
 *    pixel is read with at<char> and stored into a CV_8U row, so values above 127 become 0; candidate
 *    k is the right pixel at column i - numDisparity + k; SD squares with 8-bit saturation.
 *  - SAD, STAD, SSD (:83-141, :364-422, :649-708): interior pixels only; candidate k sits at column
 *    i - minDisparity - k and is evaluated (full winSize x winSize window) only when that column is
 *    >= winSize/2, otherwise its cost stays 0; SSD squares each absolute difference with 8-bit saturation.
 *  - _inteOpti (:155-221, :437-504, :722-789): absolute differences of the overlapping columns, 0 to the
 *    left of minDisparity + k, summed through an integral image whose four-corner lookup covers rows
 *    j-h .. j+h-1 and columns i-h .. i+h-1; interior pixels only.  (The reference's table is float;
 *    sums are exact integers here, identical while every table entry stays below 2^24.)
 *  - _BFOpti (:235-300, :519-586, :803-869): the same difference image goes through
 *    sqrBoxFilter(normalize = false), i.e. it is squared (again, for SSD) and summed over the centred
 *    window with BORDER_REFLECT_101; SAD_BFOpti fills every pixel, the other two interior pixels only.
 *  - "T" variants only clamp the winning INDEX: 16 * min(index, threshold_T).
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

static inline int iabs(int v) { return v < 0 ? -v : v; }

int orc_legacy_disparity(const uint8_t* L, const uint8_t* R, int H, int W, int method, int winSize, int T, int minD, int numD,
                         uint8_t* out)
{
    if (!L || !R || !out || numD <= 0 || minD < 0) return -1;
    const int pixel = method == ORC_LEGACY_AD || method == ORC_LEGACY_TAD || method == ORC_LEGACY_SD;
    const int clampT = method == ORC_LEGACY_TAD || method == ORC_LEGACY_STAD || method == ORC_LEGACY_STAD_INTE || method == ORC_LEGACY_STAD_BF;
    const int square = method == ORC_LEGACY_SD || method == ORC_LEGACY_SSD || method == ORC_LEGACY_SSD_INTE || method == ORC_LEGACY_SSD_BF;
    const int inte = method == ORC_LEGACY_SAD_INTE || method == ORC_LEGACY_STAD_INTE || method == ORC_LEGACY_SSD_INTE;
    const int bf = method == ORC_LEGACY_SAD_BF || method == ORC_LEGACY_STAD_BF || method == ORC_LEGACY_SSD_BF;
    memset(out, 0, (size_t)H * W);
    if (pixel) {
        if (minD != 0 || numD > W) return -2;     /* the reference's ROI would leave the image */
#pragma omp parallel for
        for (int j = 0; j < H; j++)
            for (int i = numD; i < W; i++) {
                int lv = L[(long)j * W + i];
                if (lv > 127) lv = 0;
                int best = 0, bc = 1 << 30;
                for (int k = 0; k < numD; k++) {
                    int c = iabs(lv - R[(long)j * W + i - numD + k]);
                    if (square) { c = c * c; if (c > 255) c = 255; }
                    if (c < bc) { bc = c; best = k; }
                }
                if (clampT && best > T) best = T;
                out[(long)j * W + i] = (uint8_t)(16 * best);
            }
        return 0;
    }
    if (winSize <= 0 || winSize % 2 == 0) return -3;
    if ((inte || bf) && minD + numD - 1 >= W) return -2;
    const int h = winSize / 2;
    const int all = method == ORC_LEGACY_SAD_BF;
#pragma omp parallel for
    for (int j = all ? 0 : h; j < (all ? H : H - h); j++)
        for (int i = all ? 0 : h; i < (all ? W : W - h); i++) {
            int best = 0;
            double bc = 1e300;
            for (int k = 0; k < numD; k++) {
                const int d = minD + k;
                double c = 0;
                if (!inte && !bf) {
                    if (i - d >= h)
                        for (int y = j - h; y <= j + h; y++)
                            for (int x = i - h; x <= i + h; x++) {
                                int v = iabs(L[(long)y * W + x] - R[(long)y * W + x - d]);
                                if (square) { v = v * v; if (v > 255) v = 255; }
                                c += v;
                            }
                } else {
                    const int y0 = j - h, y1 = inte ? j + h - 1 : j + h, x0 = i - h, x1 = inte ? i + h - 1 : i + h;
                    for (int y = y0; y <= y1; y++)
                        for (int x = x0; x <= x1; x++) {
                            const int yy = bf ? orc_reflect101(y, H) : y, xx = bf ? orc_reflect101(x, W) : x;
                            int v = xx >= d ? iabs(L[(long)yy * W + xx] - R[(long)yy * W + xx - d]) : 0;
                            if (square) { v = v * v; if (v > 255) v = 255; }
                            c += bf ? (double)v * v : (double)v;
                        }
                }
                if (c < bc) { bc = c; best = k; }
            }
            if (clampT && best > T) best = T;
            out[(long)j * W + i] = (uint8_t)(16 * best);
        }
    return 0;
}

/*
 * The _inteOpti matchers with the reference's own table arithmetic (method_StereoMatching.cpp:155-221, 437-504,
 * 722-789): per candidate the difference image (0 left of column minDisparity + k; SSD: pow(.,2) saturated to 8 bits),
 * integral(..., CV_32F) filled in the library's scalar order -- per row a float running sum s,
 * sum[y+1][x+1] = sum[y][x+1] + s -- then at interior pixels
 * I(j+h,i+h) - I(j-h,i+h) - I(j+h,i-h) + I(j-h,i-h) evaluated left to right in float, first minimum (minMaxLoc),
 * out = (char)(16 * index) (index clamped to T by STAD).  Equal to orc_legacy_disparity while every table entry stays
 * below 2^24; above, the float entries round and the winner follows them.
 */
int orc_legacy_inte_f32(const uint8_t* L, const uint8_t* R, int H, int W, int method, int winSize, int T, int minD, int numD,
                        uint8_t* out)
{
    if (!L || !R || !out || numD <= 0 || minD < 0) return -1;
    if (method != ORC_LEGACY_SAD_INTE && method != ORC_LEGACY_STAD_INTE && method != ORC_LEGACY_SSD_INTE) return -4;
    if (winSize <= 0 || winSize % 2 == 0) return -3;
    if (minD + numD - 1 >= W) return -2;
    const int clampT = method == ORC_LEGACY_STAD_INTE, square = method == ORC_LEGACY_SSD_INTE;
    const int h = winSize / 2, IW = W + 1;
    const long N = (long)H * W;
    memset(out, 0, (size_t)N);
    float* I = (float*)calloc((size_t)IW * (H + 1), sizeof(float));
    float* best = (float*)malloc(sizeof(float) * N);
    int* bestK = (int*)calloc((size_t)N, sizeof(int));
    for (int k = 0; k < numD; k++) {
        const int d = minD + k;
        for (int y = 0; y < H; y++) {
            float s = 0.f;
            for (int x = 0; x < W; x++) {
                int v = x >= d ? iabs(L[(long)y * W + x] - R[(long)y * W + x - d]) : 0;
                if (square) { v = v * v; if (v > 255) v = 255; }
                s += (float)v;
                I[(long)(y + 1) * IW + x + 1] = I[(long)y * IW + x + 1] + s;
            }
        }
#pragma omp parallel for
        for (int j = h; j < H - h; j++)
            for (int i = h; i < W - h; i++) {
                volatile float c = I[(long)(j + h) * IW + i + h];
                c = c - I[(long)(j - h) * IW + i + h];
                c = c - I[(long)(j + h) * IW + i - h];
                c = c + I[(long)(j - h) * IW + i - h];
                const long p = (long)j * W + i;
                if (k == 0 || c < best[p]) { best[p] = c; bestK[p] = k; }
            }
    }
    for (int j = h; j < H - h; j++)
        for (int i = h; i < W - h; i++) {
            int b = bestK[(long)j * W + i];
            if (clampT && b > T) b = T;
            out[(long)j * W + i] = (uint8_t)(16 * b);
        }
    free(I); free(best); free(bestK);
    return 0;
}
