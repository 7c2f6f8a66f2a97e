/*
 * CPU oracle (test infrastructure only, PARITY UNPINNED -- see orc.h): cv::StereoSGBM::compute as the
 * reference calls it (cameraCalibration/methods/methods.cpp:1883-1921 `create(0, 48, 7)`;
 * :1216-1219 and :1565-1568 `create(0,64,7,490,1960,1,63,10,100,32,MODE_SGBM)`).
 *
 * OpenCV 4.1.0 (modules/calib3d/src/stereosgbm.cpp, imgproc medianBlur, calib3d filterSpeckles) is a
 * third-party dependency that is not part of the reference tree and is not installed here.  This
 * file restates its published algorithm; nothing in the reference pins the numbers, so parity
 * with the real library is unverified:
 *
 *  - images: 8-bit, 1 or 3 interleaved channels; per channel an x-Sobel response
 *    (2*(p[x+1]-p[x-1]) on the row plus the rows above and below, rows replicated at the top and
 *    bottom) clipped to +-ftzero and offset by ftzero, ftzero = max(preFilterCap, 15) | 1, and the
 *    raw value; columns 0 and W-1 of every plane hold ftzero;
 *  - pixel cost: Birchfield-Tomasi between left x and right x-d on each plane (half-sample
 *    interpolation against both neighbours, min of the two one-sided costs), raw planes >> 2,
 *    summed over the planes, for the columns x in [minD+D, W) only;
 *  - block cost: blockSize x blockSize sum with the window clamped to the computed columns and
 *    to the image rows, 16-bit wrap-around arithmetic; the first computed column keeps the block
 *    cost of row 0 for every row (the library's sliding update starts at the second column;
 *    in MODE_HH the library leaves that column unwritten, this restatement uses the same rule);
 *  - aggregation over the computed columns: L_r = C + min(L_r(p-r,d), L_r(p-r,d-1)+P1,
 *    L_r(p-r,d+1)+P1, min_k L_r(p-r,k)+P2) - (min_k L_r(p-r,k)+P2), zero outside; MODE_SGBM:
 *    S = sat16(sat16(L_-> + L_\ + L_| + L_/) + L_<-); MODE_HH: the second group is the four
 *    opposite directions;
 *  - first-minimum WTA, uniqueness, disp2 election right-to-left, x16 parabola, disp12MaxDiff,
 *    columns left of minD+D invalid ((minD-1)*16);
 *  - medianBlur(3) with replicated borders, then filterSpeckles(newVal=(minD-1)*16,
 *    speckleWindowSize, 16*speckleRange) when speckleWindowSize > 0: 4-connected regions of
 *    pixels != newVal whose neighbouring values differ by <= maxDiff, regions of at most
 *    speckleWindowSize pixels are set to newVal.
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>
#include <limits.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int16_t sat16(int v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

/* planes [2*cn][H][W]: clipped x-Sobel of channel c, then raw channel c */
static void bt_planes(const uint8_t* img, int H, int W, int cn, int ftzero, uint8_t* pl)
{
    long N = (long)H * W;
#pragma omp parallel for
    for (int y = 0; y < H; y++) {
        const uint8_t* r0 = img + (long)(y > 0 ? y - 1 : y) * W * cn;
        const uint8_t* r1 = img + (long)y * W * cn;
        const uint8_t* r2 = img + (long)(y < H - 1 ? y + 1 : y) * W * cn;
        for (int c = 0; c < cn; c++)
            for (int x = 0; x < W; x++) {
                int s = ftzero, raw = ftzero;
                if (x > 0 && x < W - 1) {
                    int v = (r1[(x + 1) * cn + c] - r1[(x - 1) * cn + c]) * 2 + r0[(x + 1) * cn + c] - r0[(x - 1) * cn + c] +
                            r2[(x + 1) * cn + c] - r2[(x - 1) * cn + c];
                    s = imin(imax(v, -ftzero), ftzero) + ftzero;
                    raw = r1[x * cn + c];
                }
                pl[(long)c * N + (long)y * W + x] = (uint8_t)s;
                pl[(long)(cn + c) * N + (long)y * W + x] = (uint8_t)raw;
            }
    }
}

/* Birchfield-Tomasi cost of one plane pair at (x, x2) of one row */
static inline int bt_one(const uint8_t* a, const uint8_t* b, int W, int x, int x2)
{
    int u = a[x];
    int ul = x > 0 ? (u + a[x - 1]) / 2 : u, ur = x < W - 1 ? (u + a[x + 1]) / 2 : u;
    int u0 = imin(imin(ul, ur), u), u1 = imax(imax(ul, ur), u);
    int v = b[x2];
    int vl = x2 > 0 ? (v + b[x2 - 1]) / 2 : v, vr = x2 < W - 1 ? (v + b[x2 + 1]) / 2 : v;
    int v0 = imin(imin(vl, vr), v), v1 = imax(imax(vl, vr), v);
    int c0 = imax(imax(0, u - v1), v0 - u);
    int c1 = imax(imax(0, v - u1), u0 - v);
    return imin(c0, c1);
}

static const int DIRS[8][2] = {{+1, 0}, {+1, +1}, {0, +1}, {-1, +1}, {-1, 0}, {-1, -1}, {0, -1}, {+1, -1}};

/* one scan line over the cropped volume (W1 columns), accumulating L into acc (int) */
static void scan_line(const int16_t* C, int* acc, int H, int W1, int D, int P1, int P2, int x0, int y0, int dx, int dy,
                      int* prev, int* cur)
{
    int minPrev = 0;
    for (int k = 0; k < D + 2; k++) prev[k] = 0;
    prev[0] = prev[D + 1] = SHRT_MAX;
    for (int x = x0, y = y0; x >= 0 && x < W1 && y >= 0 && y < H; x += dx, y += dy) {
        const int16_t* c = C + ((long)y * W1 + x) * D;
        int* s = acc + ((long)y * W1 + x) * D;
        int delta = minPrev + P2, mn = SHRT_MAX;
        for (int k = 0; k < D; k++) {
            int L = c[k] + imin(prev[k + 1], imin(prev[k] + P1, imin(prev[k + 2] + P1, delta))) - delta;
            L = (int16_t)L;                       /* path costs are stored as 16-bit values */
            cur[k + 1] = L;
            mn = imin(mn, L);
            s[k] += L;
        }
        cur[0] = cur[D + 1] = SHRT_MAX;
        memcpy(prev, cur, sizeof(int) * (D + 2));
        minPrev = mn;
    }
}

static void run_direction(const int16_t* C, int* acc, int H, int W1, int D, int P1, int P2, int dx, int dy)
{
    int nstart = 0;
    int* sx = (int*)malloc(sizeof(int) * (W1 + H));
    int* sy = (int*)malloc(sizeof(int) * (W1 + H));
    if (dy == 0) { for (int y = 0; y < H; y++) { sx[nstart] = dx > 0 ? 0 : W1 - 1; sy[nstart++] = y; } }
    else {
        int ys = dy > 0 ? 0 : H - 1;
        for (int x = 0; x < W1; x++) { sx[nstart] = x; sy[nstart++] = ys; }
        if (dx != 0) {
            int xs = dx > 0 ? 0 : W1 - 1;
            for (int y = 0; y < H; y++) { if (y == ys) continue; sx[nstart] = xs; sy[nstart++] = y; }
        }
    }
#pragma omp parallel
    {
        int* prev = (int*)malloc(sizeof(int) * (D + 2));
        int* cur = (int*)malloc(sizeof(int) * (D + 2));
#pragma omp for schedule(dynamic, 8)
        for (int i = 0; i < nstart; i++) scan_line(C, acc, H, W1, D, P1, P2, sx[i], sy[i], dx, dy, prev, cur);
        free(prev); free(cur);
    }
    free(sx); free(sy);
}

/* cv::medianBlur(src, dst, 3) on CV_16SC1 (replicated borders) */
void orc_median3_s16(const int16_t* src, int H, int W, int16_t* dst)
{
#pragma omp parallel for
    for (int y = 0; y < H; y++) {
        const int16_t* r[3] = {src + (long)(y > 0 ? y - 1 : 0) * W, src + (long)y * W, src + (long)(y < H - 1 ? y + 1 : H - 1) * W};
        for (int x = 0; x < W; x++) {
            int xs[3] = {x > 0 ? x - 1 : 0, x, x < W - 1 ? x + 1 : W - 1};
            int16_t v[9];
            int n = 0;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) v[n++] = r[i][xs[j]];
            for (int i = 1; i < 9; i++) {            /* insertion sort */
                int16_t t = v[i];
                int j = i - 1;
                while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; j--; }
                v[j + 1] = t;
            }
            dst[(long)y * W + x] = v[4];
        }
    }
}

/* cv::filterSpeckles(img, newVal, maxSpeckleSize, maxDiff) on CV_16SC1, in place */
void orc_filter_speckles_s16(int16_t* img, int H, int W, int newVal, int maxSpeckleSize, int maxDiff)
{
    long N = (long)H * W;
    int* label = (int*)calloc((size_t)N, sizeof(int));
    long* stack = (long*)malloc(sizeof(long) * (size_t)N);
    uint8_t* small = (uint8_t*)calloc((size_t)N + 1, 1);
    int cur = 0;
    for (long i = 0; i < N; i++) {
        if (img[i] == newVal) continue;
        if (label[i]) { if (small[label[i]]) img[i] = (int16_t)newVal; continue; }
        cur++;
        long top = 0, count = 0;
        stack[top++] = i; label[i] = cur;
        while (top) {
            long p = stack[--top];
            count++;
            int y = (int)(p / W), x = (int)(p % W), dp = img[p];
            long nb[4] = {y < H - 1 ? p + W : -1, y > 0 ? p - W : -1, x < W - 1 ? p + 1 : -1, x > 0 ? p - 1 : -1};
            for (int k = 0; k < 4; k++) {
                long q = nb[k];
                if (q < 0 || label[q] || img[q] == newVal || abs(dp - img[q]) > maxDiff) continue;
                label[q] = cur; stack[top++] = q;
            }
        }
        if (count <= maxSpeckleSize) { small[cur] = 1; img[i] = (int16_t)newVal; }
    }
    free(label); free(stack); free(small);
}

int orc_sgbm_bt(const uint8_t* L, const uint8_t* R, int H, int W, int cn, const orc_sgbm_bt_params* p, int16_t* disp,
                int16_t* Cvol /* nullable: H*(W-minD-D)*D block costs */, int16_t* raw /* nullable: map before the median */)
{
    int minD = p->minDisparity, D = p->numDisparities;
    if (!L || !R || D <= 0 || D % 16 || minD < 0 || (cn != 1 && cn != 3)) return -1;
    if (p->mode != ORC_SGBM_MODE_SGBM && p->mode != ORC_SGBM_MODE_HH) return -2;
    int maxD = minD + D;
    int bs = p->blockSize > 0 ? p->blockSize : 5;
    int ftzero = imax(p->preFilterCap, 15) | 1;
    int uniq = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    int maxDiff12 = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    int P1 = p->P1 > 0 ? p->P1 : 2, P2 = imax(p->P2 > 0 ? p->P2 : 5, P1 + 1);
    int minX1 = maxD, maxX1 = W, W1 = maxX1 - minX1;
    int INVALID = (minD - 1) * 16;
    long N = (long)H * W;
    int16_t* d0 = (int16_t*)malloc(sizeof(int16_t) * (size_t)N);
    for (long i = 0; i < N; i++) d0[i] = (int16_t)INVALID;
    if (W1 <= 0) {      /* no column has all its candidates: the map is all invalid, and stays so through the filters */
        memcpy(disp, d0, sizeof(int16_t) * (size_t)N);
        if (raw) memcpy(raw, d0, sizeof(int16_t) * (size_t)N);
        free(d0);
        return 0;
    }
    int SW2 = bs / 2, SH2 = bs / 2;

    uint8_t* pl1 = (uint8_t*)malloc((size_t)N * 2 * cn);
    uint8_t* pl2 = (uint8_t*)malloc((size_t)N * 2 * cn);
    bt_planes(L, H, W, cn, ftzero, pl1);
    bt_planes(R, H, W, cn, ftzero, pl2);
    long V = (long)H * W1 * D;
    int16_t* pix = (int16_t*)malloc(sizeof(int16_t) * (size_t)V);
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x1 = 0; x1 < W1; x1++) {
            int x = x1 + minX1;
            for (int k = 0; k < D; k++) {
                int x2 = x - (minD + k), cost = 0;
                for (int c = 0; c < 2 * cn; c++) {
                    const uint8_t* a = pl1 + (long)c * N + (long)y * W;
                    const uint8_t* b = pl2 + (long)c * N + (long)y * W;
                    cost = (int16_t)(cost + (bt_one(a, b, W, x, x2) >> (c < cn ? 0 : 2)));
                }
                pix[((long)y * W1 + x1) * D + k] = (int16_t)cost;
            }
        }
    free(pl1); free(pl2);
    /* horizontal window sums (columns clamped to the computed range), then vertical (rows clamped) */
    int16_t* hs = (int16_t*)malloc(sizeof(int16_t) * (size_t)V);
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x1 = 0; x1 < W1; x1++)
            for (int k = 0; k < D; k++) {
                int s = 0;
                for (int i = -SW2; i <= SW2; i++) s += pix[((long)y * W1 + imin(imax(x1 + i, 0), W1 - 1)) * D + k];
                hs[((long)y * W1 + x1) * D + k] = (int16_t)s;
            }
    int16_t* C = (int16_t*)malloc(sizeof(int16_t) * (size_t)V);
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x1 = 0; x1 < W1; x1++)
            for (int k = 0; k < D; k++) {
                int yy = x1 == 0 ? 0 : y;          /* the first computed column keeps row 0's block cost */
                int s = 0;
                for (int i = -SH2; i <= SH2; i++) s += hs[((long)imin(imax(yy + i, 0), H - 1) * W1 + x1) * D + k];
                C[((long)y * W1 + x1) * D + k] = (int16_t)s;
            }
    free(pix); free(hs);
    if (Cvol) memcpy(Cvol, C, sizeof(int16_t) * (size_t)V);

    int16_t* S = (int16_t*)calloc((size_t)V, sizeof(int16_t));
    int* acc = (int*)malloc(sizeof(int) * (size_t)V);
    int groups = 2;
    for (int g = 0; g < groups; g++) {
        memset(acc, 0, sizeof(int) * (size_t)V);
        int first = g * 4, n = (g == 1 && p->mode == ORC_SGBM_MODE_SGBM) ? 1 : 4;
        for (int i = 0; i < n; i++) run_direction(C, acc, H, W1, D, P1, P2, DIRS[first + i][0], DIRS[first + i][1]);
#pragma omp parallel for
        for (long i = 0; i < V; i++) S[i] = sat16(S[i] + acc[i]);
    }
    free(acc); free(C);

#pragma omp parallel
    {
        int* disp2 = (int*)malloc(sizeof(int) * W);
        int* disp2cost = (int*)malloc(sizeof(int) * W);
#pragma omp for
        for (int y = 0; y < H; y++) {
            int16_t* d1 = d0 + (long)y * W;
            for (int x = 0; x < W; x++) { disp2[x] = INVALID; disp2cost[x] = SHRT_MAX; }
            for (int x1 = W1 - 1; x1 >= 0; x1--) {
                const int16_t* Sp = S + ((long)y * W1 + x1) * D;
                int minS = SHRT_MAX, best = -1, k;
                for (k = 0; k < D; k++)
                    if (Sp[k] < minS) { minS = Sp[k]; best = k; }
                for (k = 0; k < D; k++)
                    if (Sp[k] * (100 - uniq) < minS * 100 && abs(best - k) > 1) break;
                if (k < D) continue;
                int d = best;
                int x2 = x1 + minX1 - d - minD;
                if (disp2cost[x2] > minS) { disp2cost[x2] = minS; disp2[x2] = d + minD; }
                if (0 < d && d < D - 1) {
                    int den = imax(Sp[d - 1] + Sp[d + 1] - 2 * Sp[d], 1);
                    d = d * 16 + ((Sp[d - 1] - Sp[d + 1]) * 16 + den) / (den * 2);
                } else d *= 16;
                d1[x1 + minX1] = (int16_t)(d + minD * 16);
            }
            for (int x = minX1; x < maxX1; x++) {
                int v = d1[x];
                if (v == INVALID) continue;
                int _d = v >> 4, d_ = (v + 15) >> 4;
                int _x = x - _d, x_ = x - d_;
                if (0 <= _x && _x < W && disp2[_x] >= minD && abs(disp2[_x] - _d) > maxDiff12 &&
                    0 <= x_ && x_ < W && disp2[x_] >= minD && abs(disp2[x_] - d_) > maxDiff12)
                    d1[x] = (int16_t)INVALID;
            }
        }
        free(disp2); free(disp2cost);
    }
    free(S);
    if (raw) memcpy(raw, d0, sizeof(int16_t) * (size_t)N);
    orc_median3_s16(d0, H, W, disp);
    free(d0);
    if (p->speckleWindowSize > 0) orc_filter_speckles_s16(disp, H, W, INVALID, p->speckleWindowSize, 16 * p->speckleRange);
    return 0;
}
