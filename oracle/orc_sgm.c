/*
 * CPU oracle (test infrastructure only, parity unpinned -- see orc.h): semi-global matching.
 *
 * The reference never implements SGM itself; it calls cv::StereoSGBM
 * (cameraCalibration/methods/methods.cpp:1883-1921, :1216-1219, :1565-1568).  OpenCV 4.1.0's
 * modules/calib3d/src/stereosgbm.cpp is not vendored, so its recursion and disparity selection
 * are restated here from its published algorithm:
 *
 *   L_r(p,d) = C(p,d) + min( L_r(p-r,d), L_r(p-r,d-1)+P1, L_r(p-r,d+1)+P1, min_k L_r(p-r,k)+P2 )
 *              - ( min_k L_r(p-r,k) + P2 )
 *   with L_r == 0 for a predecessor outside the image, L_r(.,-1) = L_r(.,D) = +inf,
 *   S = sum over paths, first-minimum WTA, the uniqueness test
 *   `S[d]*(100-u) < minS*100 && |d-best| > 1  =>  invalid`, the x16 parabola refinement
 *   `d*16 + ((S[d-1]-S[d+1])*16 + den)/(2*den)`, den = max(S[d-1]+S[d+1]-2S[d], 1), and the
 *   disp2 / disp12MaxDiff left-right check (pixels visited right to left, so on equal cost the
 *   larger x keeps the disp2 slot).  Parameter defaults as in StereoSGBM: P1<=0 -> 2,
 *   P2<=0 -> 5, P2 = max(P2, P1+1), uniquenessRatio<0 -> 10, disp12MaxDiff<=0 -> 1.
 *
 * The matching cost C is the reference-owned one: the Hamming distance between the census codes
 * of the left image and of the left-padded (BORDER_REFLECT) right image, exactly as the
 * left-disparity branch of censusStereo builds it (methods_disp.cpp:859-982), optionally
 * box-summed over costWin x costWin (BORDER_REFLECT_101).  Because the right image is padded,
 * every column has all D candidates (OpenCV instead skips columns x < maxD).
 *
 * Internally the path costs are kept biased by +P2 (L' = L + P2 >= 0) so they fit uint16; the
 * bias npaths*P2 is removed wherever S enters a ratio test.
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>
#include <limits.h>

static const int DIRS[8][2] = {
    {+1, 0}, {-1, 0}, {0, +1}, {0, -1}, {+1, +1}, {-1, +1}, {+1, -1}, {-1, -1}
};

static int path_list(int paths, int* out)
{
    if (paths == 8) { for (int i = 0; i < 8; i++) out[i] = i; return 8; }
    if (paths == 5) { int l[5] = {0, 1, 2, 4, 5}; memcpy(out, l, sizeof l); return 5; }
    if (paths == 4) { for (int i = 0; i < 4; i++) out[i] = i; return 4; }
    return 0;
}

/* one scan line: start (x0,y0), step (dx,dy), until leaving the image */
static void scan_line(const uint16_t* C, uint16_t* S, int H, int W, int D, int P1, int P2,
                      int x0, int y0, int dx, int dy, int* prev, int* cur)
{
    int minPrev = 0;
    for (int k = 0; k < D + 2; k++) prev[k] = 0;      /* predecessor outside the image: L == 0 */
    prev[0] = prev[D + 1] = INT_MAX / 2;
    for (int x = x0, y = y0; x >= 0 && x < W && y >= 0 && y < H; x += dx, y += dy) {
        const uint16_t* c = C + ((long)y * W + x) * D;
        uint16_t* s = S + ((long)y * W + x) * D;
        int mn = INT_MAX;
        /* biased form: L' = C + min(L'p[d], L'p[d-1]+P1, L'p[d+1]+P1, minL'p + P2) - minL'p,
         * where L'p = Lp + P2; for the out-of-image predecessor Lp = 0, i.e. L' = C + P2 - ... = C
         * after the -(0+P2) of the unbiased form, so seed with L'p = 0, minL'p = 0 and no +P2 term
         * smaller than 0: min(0, P1, P1, P2) = 0. */
        for (int k = 0; k < D; k++) {
            int a = prev[k + 1], b = prev[k] + P1, cc = prev[k + 2] + P1, e = minPrev + P2;
            int m = a < b ? a : b;
            if (cc < m) m = cc;
            if (e < m) m = e;
            int L = (int)c[k] + m - minPrev;
            cur[k + 1] = L;
            if (L < mn) mn = L;
            s[k] = (uint16_t)(s[k] + L);
        }
        cur[0] = cur[D + 1] = INT_MAX / 2;
        memcpy(prev, cur, sizeof(int) * (D + 2));
        minPrev = mn;
    }
}

int orc_sgm(const uint8_t* L, const uint8_t* R, int H, int W, const orc_sgm_params* p,
            int16_t* disp, uint16_t* Svol)
{
    int minD = p->minDisparity, D = p->numDisparities;
    if (!L || !R || D <= 0 || minD < 0) return -1;
    int P1 = p->P1 > 0 ? p->P1 : 2;
    int P2 = p->P2 > 0 ? p->P2 : 5;
    if (P2 < P1 + 1) P2 = P1 + 1;
    int uniq = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    int maxDiff = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    int plist[8];
    int np = path_list(p->paths, plist);
    if (!np) return -2;
    int cw = p->costWin > 0 ? p->costWin : 1;
    if (cw % 2 == 0) return -3;
    int maxD = minD + D - 1, Wp = W + maxD;
    if ((long)np * (8L * cw * cw + P2) > 65535) return -4;

    /* census of left and padded right, methods_disp.cpp:861-878 */
    uint8_t* pad = (uint8_t*)malloc((size_t)H * Wp);
    uint8_t* cL = (uint8_t*)malloc((size_t)H * W);
    uint8_t* cR = (uint8_t*)malloc((size_t)H * Wp);
    orc_pad_reflect_lr(R, H, W, maxD, 0, pad);
    if (orc_census(L, H, W, p->censusMethod, cw, 1, cL) || orc_census(pad, H, Wp, p->censusMethod, cw, 1, cR)) {
        free(pad); free(cL); free(cR); return -5;
    }
    long N = (long)H * W;
    uint16_t* C = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)N * D);
    int own = 0;
    uint16_t* S = Svol;
    if (!S) { S = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)N * D); own = 1; }
    memset(S, 0, sizeof(uint16_t) * (size_t)N * D);
    /* cost volume, d-fastest: countHummingDist on Rect(maxD - d, 0, W, H), methods_disp.cpp:969-975 */
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            uint8_t a = cL[(long)y * W + x];
            for (int k = 0; k < D; k++) {
                int d = minD + k;
                C[((long)y * W + x) * D + k] = (uint16_t)__builtin_popcount(a ^ cR[(long)y * Wp + x + maxD - d]);
            }
        }
    if (cw > 1) {
        uint32_t* a = (uint32_t*)malloc(sizeof(uint32_t) * N);
        uint32_t* b = (uint32_t*)malloc(sizeof(uint32_t) * N);
        for (int k = 0; k < D; k++) {
            for (long i = 0; i < N; i++) a[i] = C[i * D + k];
            orc_box_sum_u32(a, H, W, cw, cw, b);
            for (long i = 0; i < N; i++) C[i * D + k] = (uint16_t)b[i];
        }
        free(a); free(b);
    }
    free(pad); free(cL); free(cR);

    for (int pi = 0; pi < np; pi++) {
        int dx = DIRS[plist[pi]][0], dy = DIRS[plist[pi]][1];
        /* every line starts on the border the path enters through */
        int nstart = 0;
        int* sx = (int*)malloc(sizeof(int) * (W + H));
        int* sy = (int*)malloc(sizeof(int) * (W + H));
        if (dy == 0) { for (int y = 0; y < H; y++) { sx[nstart] = dx > 0 ? 0 : W - 1; sy[nstart++] = y; } }
        else {
            int ys = dy > 0 ? 0 : H - 1;
            for (int x = 0; x < W; x++) { sx[nstart] = x; sy[nstart++] = ys; }
            if (dx != 0) {
                int xs = dx > 0 ? 0 : W - 1;
                for (int y = 0; y < H; y++) { if (y == ys) continue; sx[nstart] = xs; sy[nstart++] = y; }
            }
        }
#pragma omp parallel
        {
            int* prev = (int*)malloc(sizeof(int) * (D + 2));
            int* cur = (int*)malloc(sizeof(int) * (D + 2));
#pragma omp for schedule(dynamic, 8)
            for (int i = 0; i < nstart; i++)
                scan_line(C, S, H, W, D, P1, P2, sx[i], sy[i], dx, dy, prev, cur);
            free(prev); free(cur);
        }
        free(sx); free(sy);
    }
    free(C);

    const int bias = np * P2;
    const int INVALID = (minD - 1) * 16;
#pragma omp parallel
    {
        int* disp2 = (int*)malloc(sizeof(int) * W);
        int* disp2cost = (int*)malloc(sizeof(int) * W);
#pragma omp for
        for (int y = 0; y < H; y++) {
            int16_t* d1 = disp + (long)y * W;
            for (int x = 0; x < W; x++) { d1[x] = (int16_t)INVALID; disp2[x] = INVALID; disp2cost[x] = INT_MAX; }
            for (int x = W - 1; x >= 0; x--) {
                const uint16_t* Sp = S + ((long)y * W + x) * D;
                int minS = INT_MAX, best = -1;
                for (int k = 0; k < D; k++) {
                    int v = (int)Sp[k] - bias;
                    if (v < minS) { minS = v; best = k; }
                }
                int k;
                for (k = 0; k < D; k++)
                    if (((int)Sp[k] - bias) * (100 - uniq) < minS * 100 && abs(best - k) > 1) break;
                if (k < D) continue;
                int d = best;
                int x2 = x - d - minD;
                if (x2 >= 0 && x2 < W && disp2cost[x2] > minS) { disp2cost[x2] = minS; disp2[x2] = d + minD; }
                if (p->subpixel && 0 < d && d < D - 1) {
                    int den = (int)Sp[d - 1] + (int)Sp[d + 1] - 2 * (int)Sp[d];
                    if (den < 1) den = 1;
                    d = d * 16 + (((int)Sp[d - 1] - (int)Sp[d + 1]) * 16 + den) / (den * 2);
                } else d *= 16;
                d1[x] = (int16_t)(d + minD * 16);
            }
            for (int x = 0; x < W; x++) {
                int v = d1[x];
                if (v == INVALID) continue;
                int _d = v >> 4, d_ = (v + 15) >> 4;
                int _x = x - _d, x_ = x - d_;
                if (0 <= _x && _x < W && disp2[_x] >= minD && abs(disp2[_x] - _d) > maxDiff &&
                    0 <= x_ && x_ < W && disp2[x_] >= minD && abs(disp2[x_] - d_) > maxDiff)
                    d1[x] = (int16_t)INVALID;
            }
        }
        free(disp2); free(disp2cost);
    }
    if (own) free(S);
    return 0;
}
