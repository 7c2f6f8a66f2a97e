/* CPU oracle (TEST INFRASTRUCTURE): cv::StereoBM as the reference calls it in getDisparity_BM
 * (methods/methods.cpp:1825-1881: StereoBM::create(144, 35), minDisparity 186, textureThreshold 10,
 * disp12MaxDiff -1, preFilterCap 31, uniquenessRatio 25, speckleRange 32, speckleWindowSize 0) and at
 * methods/methods.cpp:1205-1213.
 *
 * The arithmetic is OpenCV 4.1.0's modules/calib3d/src/stereobm.cpp (not vendored in the reference), restated from
 * the published algorithm: PARITY WITH THE REAL LIBRARY IS UNPINNED, the reference holds no output of it.
 * Conventions assumed:
 *   - PREFILTER_XSOBEL: rows are produced in pairs; v = clamp(d(y-1) + 2 d(y) + d(y+1), -cap, cap) + cap with
 *     d(y) = src(x+1, y) - src(x-1, y) and rows mirrored without repeating the edge; the first and last column and,
 *     when the height is odd, the last row hold the value `cap`.
 *   - the library computes every column with incremental window sums over clamped column indices, then overwrites
 *     everything outside getValidDisparityROI() = [maxD + w/2, W - w/2) x [w/2, H - w/2) with FILTERED
 *     = (minDisparity - 1) * 16.  Inside that rectangle no index is clamped and the incremental sums are exact
 *     integer window sums, so this file evaluates the window sums directly there.
 *   - candidate index d runs over descending disparities (disparity = numDisparities - 1 - d + minDisparity); the first
 *     minimum in d wins, i.e. ties go to the LARGER disparity.
 *   - texture: sum over the window of |Lp - cap| < textureThreshold -> FILTERED (tested before uniqueness).
 *   - uniqueness: any d outside [mind-1, mind+1] with sad[d] <= minsad + minsad * ratio / 100 -> FILTERED.
 *   - sub-pixel: sad[-1] = sad[1], sad[D] = sad[D-2]; p = sad[mind+1], n = sad[mind-1];
 *     q = p + n - 2 sad[mind] + |p - n|;  result = (disparity * 256 + (q ? (p - n) * 256 / q : 0) + 15) >> 4.
 *   - disp12MaxDiff >= 0: validateDisparity() with the winning window sum as the cost; speckle filter with
 *     maxDiff = speckleRange (unscaled, as StereoBM passes it). */
#include "orc.h"
#include <limits.h>
#include <stdlib.h>
#include <string.h>

void orc_bm_prefilter_xsobel(const uint8_t* src, int H, int W, int cap, uint8_t* dst)
{
    for (int y = 0; y < H; y++) {
        uint8_t* d = dst + (size_t)y * W;
        if ((H & 1) && y == H - 1) { memset(d, cap, W); continue; }
        const int ya = y > 0 ? y - 1 : (H > 1 ? y + 1 : y), yb = y < H - 1 ? y + 1 : (H > 1 ? y - 1 : y);
        const uint8_t *r0 = src + (size_t)ya * W, *r1 = src + (size_t)y * W, *r2 = src + (size_t)yb * W;
        for (int x = 0; x < W; x++) {
            if (x == 0 || x == W - 1) { d[x] = (uint8_t)cap; continue; }
            int v = (r0[x + 1] - r0[x - 1]) + 2 * (r1[x + 1] - r1[x - 1]) + (r2[x + 1] - r2[x - 1]);
            v = v < -cap ? 0 : v > cap ? 2 * cap : v + cap;
            d[x] = (uint8_t)v;
        }
    }
}

/* validateDisparity of calib3d (stereosgbm.cpp), int cost */
static void bm_validate(int16_t* disp, const int* cost, int H, int W, int minD, int ndisp, int maxDiff)
{
    const int maxD = minD + ndisp, minX1 = maxD > 0 ? maxD : 0, maxX1 = W + (minD < 0 ? minD : 0);
    const int INVALID = (minD - 1) * 16;
    int* d2 = (int*)malloc(sizeof(int) * 2 * (size_t)W);
    int* c2 = d2 + W;
    maxDiff *= 16;
    for (int y = 0; y < H; y++) {
        int16_t* dp = disp + (size_t)y * W;
        const int* cp = cost + (size_t)y * W;
        for (int x = 0; x < W; x++) { d2[x] = INVALID; c2[x] = INT_MAX; }
        for (int x = minX1; x < maxX1; x++) {
            const int d = dp[x], c = cp[x];
            if (d == INVALID) continue;
            const int x2 = x - ((d + 8) >> 4);
            if (x2 < 0 || x2 >= W) continue;
            if (c2[x2] > c) { c2[x2] = c; d2[x2] = d; }
        }
        for (int x = minX1; x < maxX1; x++) {
            const int d = dp[x];
            if (d == INVALID) continue;
            const int d0 = d >> 4, d1 = (d + 15) >> 4, x0 = x - d0, x1 = x - d1;
            if ((0 <= x0 && x0 < W && d2[x0] > INVALID && abs(d2[x0] - d) > maxDiff) &&
                (0 <= x1 && x1 < W && d2[x1] > INVALID && abs(d2[x1] - d) > maxDiff))
                dp[x] = (int16_t)INVALID;
        }
    }
    free(d2);
}

int orc_stereobm(const uint8_t* L, const uint8_t* R, int H, int W, const orc_bm_params* p, int16_t* disp)
{
    const int ndisp = p->numDisparities, mind = p->minDisparity, wsz = p->blockSize, w = wsz / 2, cap = p->preFilterCap;
    if (ndisp <= 0 || ndisp % 16 || wsz < 5 || wsz > 255 || !(wsz & 1) || wsz >= (W < H ? W : H) || cap < 1 || cap > 63 ||
        p->textureThreshold < 0 || p->uniquenessRatio < 0 || mind < 0 /* not restated */)
        return -1;
    const int FILTERED = (mind - 1) * 16;
    for (size_t i = 0; i < (size_t)H * W; i++) disp[i] = (int16_t)FILTERED;
    const int maxD = mind + ndisp - 1;
    const int lofs = maxD > 0 ? maxD : 0, rofs = maxD < 0 ? -maxD : 0;
    if (lofs >= W || rofs >= W || W - rofs - ndisp + 1 < 1) return 0;
    const int xmin = (maxD > 0 ? maxD : 0) + w, xmax = W - w, ymin = w, ymax = H - w;
    if (xmax <= xmin || ymax <= ymin) return 0;
    uint8_t* Lp = (uint8_t*)malloc((size_t)H * W);
    uint8_t* Rp = (uint8_t*)malloc((size_t)H * W);
    orc_bm_prefilter_xsobel(L, H, W, cap, Lp);
    orc_bm_prefilter_xsobel(R, H, W, cap, Rp);
    int* cost = p->disp12MaxDiff >= 0 ? (int*)calloc((size_t)H * W, sizeof(int)) : NULL;
#pragma omp parallel
    {
        int* sad = (int*)malloc(sizeof(int) * (ndisp + 2));
#pragma omp for schedule(dynamic, 4)
        for (int y = ymin; y < ymax; y++) {
            for (int x = xmin; x < xmax; x++) {
                int tsum = 0;
                for (int j = -w; j <= w; j++)
                    for (int i = -w; i <= w; i++) tsum += abs((int)Lp[(size_t)(y + j) * W + x + i] - cap);
                if (tsum < p->textureThreshold) continue;
                int minsad = INT_MAX, md = -1;
                for (int d = 0; d < ndisp; d++) {
                    const int dv = ndisp - 1 - d + mind;
                    int s = 0;
                    for (int j = -w; j <= w; j++) {
                        const uint8_t* lp = Lp + (size_t)(y + j) * W + x;
                        const uint8_t* rp = Rp + (size_t)(y + j) * W + x - dv;
                        for (int i = -w; i <= w; i++) s += abs((int)lp[i] - (int)rp[i]);
                    }
                    sad[d + 1] = s;
                    if (s < minsad) { minsad = s; md = d; }
                }
                if (p->uniquenessRatio > 0) {
                    const int thresh = minsad + minsad * p->uniquenessRatio / 100;
                    int d;
                    for (d = 0; d < ndisp; d++)
                        if ((d < md - 1 || d > md + 1) && sad[d + 1] <= thresh) break;
                    if (d < ndisp) continue;
                }
                sad[0] = sad[2];
                sad[ndisp + 1] = sad[ndisp - 1];
                const int pp = sad[md + 2], nn = sad[md], q = pp + nn - 2 * sad[md + 1] + abs(pp - nn);
                disp[(size_t)y * W + x] = (int16_t)(((ndisp - md - 1 + mind) * 256 + (q != 0 ? (pp - nn) * 256 / q : 0) + 15) >> 4);
                if (cost) cost[(size_t)y * W + x] = sad[md + 1];
            }
        }
        free(sad);
    }
    if (cost) {
        /* the library validates the rows of the valid rectangle only, before the columns outside it are overwritten
         * (they already hold FILTERED here) */
        bm_validate(disp + (size_t)ymin * W, cost + (size_t)ymin * W, ymax - ymin, W, mind, ndisp, p->disp12MaxDiff);
        free(cost);
    }
    if (p->speckleRange >= 0 && p->speckleWindowSize > 0)
        orc_filter_speckles_s16(disp, H, W, FILTERED, p->speckleWindowSize, p->speckleRange);
    free(Lp);
    free(Rp);
    return 0;
}
