/*
 * CPU oracle (test infrastructure only, parity unpinned -- see orc.h):
 * borders, census family, Hamming distance, census stereo driver, SAD drivers.
 *
 * Reference: cameraCalibration/disp_method/methods_disp.cpp (cited per function).
 */
#include "orc.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* cv::borderInterpolate(BORDER_REFLECT): fedcba|abcdef|fedcba, iterated for pads wider than n. */
int orc_reflect(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p - 1;
        else p = 2 * n - p - 1;
    }
    return p;
}

/* cv::borderInterpolate(BORDER_REFLECT_101): gfedcb|abcdef|edcba. */
int orc_reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * n - p - 2;
    }
    return p;
}

/* copyMakeBorder(src, dst, 0, 0, left, right, BORDER_REFLECT) -- methods_disp.cpp:224,268,862,1005 */
void orc_pad_reflect_lr(const uint8_t* src, int H, int W, int left, int right, uint8_t* dst)
{
    int Wp = W + left + right;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < Wp; x++)
            dst[(long)y * Wp + x] = src[(long)y * W + orc_reflect(x - left, W)];
}

/* cvtColor(COLOR_BGR2GRAY) on 8U as called by checkImg, methods_disp.cpp:138-141.
 * OpenCV fixed point: (B*1868 + G*9617 + R*4899 + 2^13) >> 14. */
void orc_bgr2gray(const uint8_t* bgr, int H, int W, uint8_t* gray)
{
    for (long i = 0; i < (long)H * W; i++) {
        int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        gray[i] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
    }
}

static inline uint8_t census3x3(const uint8_t* s, int W, int y, int x)
{
    /* bit order methods_disp.cpp:436-443: 7=TL 6=T 5=TR 4=R 3=BR 2=B 1=BL 0=L, test nbr >= centre */
    uint8_t c = s[(long)y * W + x];
    uint8_t v = 0;
    v |= (uint8_t)((s[(long)(y - 1) * W + x - 1] >= c) << 7);
    v |= (uint8_t)((s[(long)(y - 1) * W + x] >= c) << 6);
    v |= (uint8_t)((s[(long)(y - 1) * W + x + 1] >= c) << 5);
    v |= (uint8_t)((s[(long)y * W + x + 1] >= c) << 4);
    v |= (uint8_t)((s[(long)(y + 1) * W + x + 1] >= c) << 3);
    v |= (uint8_t)((s[(long)(y + 1) * W + x] >= c) << 2);
    v |= (uint8_t)((s[(long)(y + 1) * W + x - 1] >= c) << 1);
    v |= (uint8_t)((s[(long)y * W + x - 1] >= c) << 0);
    return v;
}

/* countCensusImg, methods_disp.cpp:415-447: 1-px frame stays 0. */
void orc_census_basic(const uint8_t* src, int H, int W, uint8_t* dst)
{
    memset(dst, 0, (size_t)H * W);
#pragma omp parallel for
    for (int y = 1; y < H - 1; y++)
        for (int x = 1; x < W - 1; x++)
            dst[(long)y * W + x] = census3x3(src, W, y, x);
}

/* RITable of countCensusImg_rotationInv, methods_disp.cpp:519-533.  The inner loop compares
 * against i>>1 seven times and never rotates, so the table is min(i, i>>1) == i>>1. */
void orc_ri_table(uint8_t* t)
{
    for (int i = 0; i < 256; i++) {
        int val = i;
        for (int j = 0; j < 7; j++) {
            int temp = i >> 1;
            if (val > temp) val = temp;
        }
        t[i] = (uint8_t)val;
    }
}

void orc_census_rotation_inv(const uint8_t* src, int H, int W, uint8_t* dst)
{
    uint8_t t[256];
    orc_ri_table(t);
    memset(dst, 0, (size_t)H * W);
#pragma omp parallel for
    for (int y = 1; y < H - 1; y++)
        for (int x = 1; x < W - 1; x++)
            dst[(long)y * W + x] = t[census3x3(src, W, y, x)];
}

/* hopCount, methods_disp.cpp:562-580: `while (k)` fills a[7]..a[1] from the low 7 bits and
 * leaves a[0] = 0, so bit 7 of the code is ignored. */
int orc_hop_count(uint8_t i)
{
    uint8_t a[8] = {0};
    int cnt = 0, k = 7;
    while (k) {
        a[k] = i & 1;
        i = (uint8_t)(i >> 1);
        k--;
    }
    for (int j = 0; j < 8; j++)
        if (a[j] != a[j + 1 == 8 ? 0 : j + 1]) cnt++;
    return cnt;
}

/* UPTable of countCensusImg_uniform, methods_disp.cpp:599-609 (uchar counter, wraps). */
void orc_uniform_table(uint8_t* t)
{
    memset(t, 0, 256);
    uint8_t temp = 1;
    for (int i = 0; i < 256; i++)
        if (orc_hop_count((uint8_t)i) <= 2) {
            t[i] = temp;
            temp++;
        }
}

void orc_census_uniform(const uint8_t* src, int H, int W, uint8_t* dst)
{
    uint8_t t[256];
    orc_uniform_table(t);
    memset(dst, 0, (size_t)H * W);
#pragma omp parallel for
    for (int y = 1; y < H - 1; y++)
        for (int x = 1; x < W - 1; x++)
            dst[(long)y * W + x] = t[census3x3(src, W, y, x)];
}

/* cell-mean image of countCensusImg_multiScale, methods_disp.cpp:639-667 */
static void cell_mean(const uint8_t* src, int H, int W, int scale, uint8_t* cell)
{
    int cellSize = scale / 3;
    int offset = cellSize / 2;
    memset(cell, 0, (size_t)H * W);
    if (cellSize <= 0) return;
#pragma omp parallel for
    for (int i = offset; i < H - offset; i++)
        for (int j = offset; j < W - offset; j++) {
            int temp = 0;
            for (int m = -offset; m < offset + 1; m++)
                for (int n = -offset; n < offset + 1; n++)
                    temp += src[(long)(i + n) * W + (j + m)];
            temp /= (cellSize * cellSize);
            cell[(long)i * W + j] = (uint8_t)temp;
        }
}

void orc_census_multiscale(const uint8_t* src, int H, int W, int scale, uint8_t* dst)
{
    uint8_t* cell = (uint8_t*)malloc((size_t)H * W);
    cell_mean(src, H, W, scale, cell);
    orc_census_basic(cell, H, W, dst);
    free(cell);
}

static int cmp_desc_f(const void* a, const void* b)
{
    float x = *(const float*)a, y = *(const float*)b;
    return (x < y) - (x > y);
}

/* countCensusImg_multiScale2, methods_disp.cpp:677-714.  calcHist with 256 uniform bins over
 * [0,255) counts values 0..254 in their own bin and drops 255. */
void orc_census_multiscale2(const uint8_t* src, int H, int W, int scale, uint8_t* dst)
{
    orc_census_multiscale(src, H, W, scale, dst);
    float hist[256], sorted[256];
    memset(hist, 0, sizeof hist);
    for (long i = 0; i < (long)H * W; i++)
        if (dst[i] < 255) hist[dst[i]] += 1.0f;
    memcpy(sorted, hist, sizeof hist);
    qsort(sorted, 256, sizeof(float), cmp_desc_f);
    uint8_t table[256];
    memset(table, 64, 256);
    for (int i = 0; i < 63; i++)
        for (int j = 0; j < 256; j++)
            if (hist[j] == sorted[i]) table[j] = (uint8_t)i;
    for (long i = 0; i < (long)H * W; i++) dst[i] = table[dst[i]];
}

/* countCensusImg_circle, methods_disp.cpp:456-500, restated deterministically: the reference
 * accumulates `dst(i-r, j-r) += bit << n` from concurrent OpenMP iterations over n (a data
 * race); here every pixel sums its bits in one thread. */
void orc_census_circle(const uint8_t* src, int H, int W, int radius, int npts, uint8_t* dst)
{
    memset(dst, 0, (size_t)H * W);
    for (int n = 0; n < npts; n++) {
        float x = (float)(radius * cos(2.0 * 3.1415926535897932384626433832795 * n / (float)npts));
        float y = (float)(-radius * sin(2.0 * 3.1415926535897932384626433832795 * n / (float)npts));
        int fx = (int)floor(x), fy = (int)floor(y);
        int cx = (int)ceil(x), cy = (int)ceil(y);
        float tx = x - fx, ty = y - fy;
        float w1 = (1 - tx) * (1 - ty), w2 = tx * (1 - ty), w3 = (1 - tx) * ty, w4 = tx * ty;
        for (int i = radius; i < H - radius; i++)
            for (int j = radius; j < W - radius; j++) {
                volatile float t = w1 * src[(long)(i + fy) * W + j + fx];
                t = t + w2 * src[(long)(i + fy) * W + j + cx];
                t = t + w3 * src[(long)(i + cy) * W + j + fx];
                t = t + w4 * src[(long)(i + cy) * W + j + cx];
                float c = (float)src[(long)i * W + j];
                int bit = (t > c) || (fabsf(t - c) < FLT_EPSILON);
                dst[(long)(i - radius) * W + (j - radius)] =
                    (uint8_t)(dst[(long)(i - radius) * W + (j - radius)] + (bit << n));
            }
    }
}

/* Dispatch used by the stereo driver.  `isPaddedView` selects the per-branch literals of
 * censusStereo (methods_disp.cpp:885,927 vs :1028,1070): in the left-disparity branch the
 * circle radius is 5 and the multi-scale window 16; in the right-disparity branch both are
 * param.winSize.  (isPaddedView: 1 = left-disparity branch.) */
int orc_census(const uint8_t* src, int H, int W, int method, int winSize, int leftBranch, uint8_t* dst)
{
    switch (method) {
    case ORC_BASIC_CENSUS: orc_census_basic(src, H, W, dst); return 0;
    case ORC_CIRCLE_CENSUS: orc_census_circle(src, H, W, leftBranch ? 5 : winSize, 8, dst); return 0;
    case ORC_ROTATION_INVARIANT_CENSUS: orc_census_rotation_inv(src, H, W, dst); return 0;
    case ORC_UNIFORM_CENSUS: orc_census_uniform(src, H, W, dst); return 0;
    case ORC_MULTISCALE_CENSUS: orc_census_multiscale(src, H, W, leftBranch ? 16 : winSize, dst); return 0;
    case ORC_STATISTIC_MULTISCALE_CENSUS: orc_census_multiscale2(src, H, W, winSize, dst); return 0;
    default: return -1; /* CENSUS_2017 (methods_disp.cpp:722-800) mutates shared state and shifts past bit 31 */
    }
}

/* countHummingDist, methods_disp.cpp:808-838: nibble-table popcount of the XOR. */
void orc_hamming(const uint8_t* a, const uint8_t* b, long n, int32_t* dst)
{
    static const int countTable[16] = {0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4};
    for (long i = 0; i < n; i++) {
        int num = a[i] ^ b[i];
        int count = 0;
        while (num) {
            count += countTable[num & 0xF];
            num >>= 4;
        }
        dst[i] = count;
    }
}

/* integer box sum with BORDER_REFLECT_101 and centred anchor (anchor = k/2) */
void orc_box_sum_u32(const uint32_t* src, int H, int W, int kw, int kh, uint32_t* dst)
{
    int ax = kw / 2, ay = kh / 2;
    uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)H * W);
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            uint32_t s = 0;
            for (int k = 0; k < kw; k++) s += src[(long)y * W + orc_reflect101(x - ax + k, W)];
            tmp[(long)y * W + x] = s;
        }
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            uint32_t s = 0;
            for (int k = 0; k < kh; k++) s += tmp[(long)orc_reflect101(y - ay + k, H) * W + x];
            dst[(long)y * W + x] = s;
        }
    free(tmp);
}

/*
 * censusStereo, methods_disp.cpp:845-1145.
 * Left-disparity branch (:859-1001): right image padded on the left by maxDisparity
 * (BORDER_REFLECT), census on left and on the PADDED right, slice d compares censusLeft with
 * censusRight(Rect(maxD - d, 0, W, H)).  Right-disparity branch (:1002-1144): left image padded on
 * the right, slice d compares censusRight with censusLeft(Rect(d, 0, W, H)).
 * Each slice is aggregated (agg) and the first minimum over d wins (minMaxLoc, :995-997).
 * Output CV_64F, value k + minDisparity.
 */
int orc_census_stereo(const uint8_t* L, const uint8_t* R, int H, int W, int winSize, int minD, int maxD,
                      int isDispLeft, int method, int agg, double* disp, float* costVolume)
{
    if (!L || !R || winSize % 2 == 0 || maxD < minD || minD < 0) return -1;
    int D = maxD - minD + 1;
    int Wp = W + maxD;
    uint8_t* pad = (uint8_t*)malloc((size_t)H * Wp);
    uint8_t* cRef = (uint8_t*)malloc((size_t)H * W);
    uint8_t* cPad = (uint8_t*)malloc((size_t)H * Wp);
    const uint8_t* guide = isDispLeft ? L : R;
    int rc;
    if (isDispLeft) {
        orc_pad_reflect_lr(R, H, W, maxD, 0, pad);
        rc = orc_census(L, H, W, method, winSize, 1, cRef);
        rc |= orc_census(pad, H, Wp, method, winSize, 1, cPad);
    } else {
        orc_pad_reflect_lr(L, H, W, 0, maxD, pad);
        rc = orc_census(pad, H, Wp, method, winSize, 0, cPad);
        rc |= orc_census(R, H, W, method, winSize, 0, cRef);
    }
    if (rc) { free(pad); free(cRef); free(cPad); return -2; }

    long N = (long)H * W;
    float* best = (float*)malloc(sizeof(float) * N);
    int32_t* dist = (int32_t*)malloc(sizeof(int32_t) * N);
    uint32_t* u0 = (uint32_t*)malloc(sizeof(uint32_t) * N);
    uint32_t* u1 = (uint32_t*)malloc(sizeof(uint32_t) * N);
    float* slice = (float*)malloc(sizeof(float) * N);
    float* guideN = NULL; float* pN = NULL;
    if (agg == ORC_AGG_GUIDED) {
        guideN = (float*)malloc(sizeof(float) * N);
        pN = (float*)malloc(sizeof(float) * N);
        orc_normalize_minmax_u8(guide, N, guideN);
    }
    for (int k = 0; k < D; k++) {
        int d = minD + k;
        int off = isDispLeft ? (maxD - d) : d;
#pragma omp parallel for
        for (int y = 0; y < H; y++)
            orc_hamming(cRef + (long)y * W, cPad + (long)y * Wp + off, W, dist + (long)y * W);
        switch (agg) {
        case ORC_AGG_NONE:
            for (long i = 0; i < N; i++) slice[i] = (float)dist[i];
            break;
        case ORC_AGG_BOX:
        case ORC_AGG_SQRBOX:
            for (long i = 0; i < N; i++) u0[i] = (uint32_t)(agg == ORC_AGG_BOX ? dist[i] : dist[i] * dist[i]);
            orc_box_sum_u32(u0, H, W, winSize, winSize, u1);
            for (long i = 0; i < N; i++) slice[i] = (float)u1[i];
            break;
        case ORC_AGG_GUIDED:
            /* the left-disparity branch narrows the slice to CV_8U first (:975) and so normalises
             * in float; the right-disparity branch passes the CV_32S slice (:1120), which OpenCV
             * normalises in double */
            if (isDispLeft) {
                uint8_t* d8 = (uint8_t*)u0;
                for (long i = 0; i < N; i++) d8[i] = (uint8_t)dist[i];
                orc_normalize_minmax_u8(d8, N, pN);
            } else orc_normalize_minmax_s32(dist, N, pN);
            orc_guided_filter(guideN, pN, H, W, winSize, 1e-6, slice);
            break;
        default:
            return -3;
        }
        if (costVolume) memcpy(costVolume + (long)k * N, slice, sizeof(float) * N);
        /* minMaxLoc keeps the first minimum: strict '<' while scanning d upward */
        if (k == 0) {
            for (long i = 0; i < N; i++) { best[i] = slice[i]; disp[i] = (double)minD; }
        } else {
            for (long i = 0; i < N; i++)
                if (slice[i] < best[i]) { best[i] = slice[i]; disp[i] = (double)d; }
        }
    }
    free(pad); free(cRef); free(cPad); free(best); free(dist); free(u0); free(u1); free(slice);
    free(guideN); free(pN);
    return 0;
}

/*
 * computeSAD_inteOpti, methods_disp.cpp:207-309.  Per d: absdiff slice, integral image, 4-corner
 * lookup with getMatVal (:159-171; out-of-range index -> 0), first-min WTA, result stored through
 * `.at<char>` (wraps modulo 256).
 * satMode ORC_SAT_F32: the table is float, filled in OpenCV's scalar order (row running sum `s`
 * in float, sum[y+1][x+1] = sum[y][x+1] + s) and the 4 corners combine left to right in float.
 * satMode ORC_SAT_INT: same footprint and zero-outside rule on exact 64-bit prefix sums.
 */
int orc_sad_inte(const uint8_t* L, const uint8_t* R, int H, int W, int winSize, int minD, int maxD,
                 int isDispLeft, int satMode, uint8_t* disp)
{
    if (!L || !R || winSize % 2 == 0 || maxD < minD || minD < 0) return -1;
    int D = maxD - minD + 1, Wp = W + maxD, h = winSize / 2;
    long N = (long)H * W;
    uint8_t* pad = (uint8_t*)malloc((size_t)H * Wp);
    const uint8_t* ref = isDispLeft ? L : R;
    if (isDispLeft) orc_pad_reflect_lr(R, H, W, maxD, 0, pad);
    else orc_pad_reflect_lr(L, H, W, 0, maxD, pad);
    int IW = W + 1, IH = H + 1;
    float* If = (float*)calloc((size_t)IW * IH, sizeof(float));
    int64_t* Ii = (int64_t*)calloc((size_t)IW * IH, sizeof(int64_t));
    float* bestF = (float*)malloc(sizeof(float) * N);
    int64_t* bestI = (int64_t*)malloc(sizeof(int64_t) * N);
    for (int k = 0; k < D; k++) {
        int d = minD + k;
        int off = isDispLeft ? (maxD - d) : d;
        for (int y = 0; y < H; y++) {
            float sf = 0.f; int64_t si = 0;
            for (int x = 0; x < W; x++) {
                int a = ref[(long)y * W + x], b = pad[(long)y * Wp + off + x];
                int v = a > b ? a - b : b - a;
                sf += (float)v; si += v;
                If[(long)(y + 1) * IW + x + 1] = If[(long)y * IW + x + 1] + sf;
                Ii[(long)(y + 1) * IW + x + 1] = Ii[(long)y * IW + x + 1] + si;
            }
        }
#define GETF(xx, yy) (((xx) >= 0 && (xx) < IW && (yy) >= 0 && (yy) < IH) ? If[(long)(yy) * IW + (xx)] : 0.f)
#define GETI(xx, yy) (((xx) >= 0 && (xx) < IW && (yy) >= 0 && (yy) < IH) ? Ii[(long)(yy) * IW + (xx)] : (int64_t)0)
#pragma omp parallel for
        for (int j = 0; j < H; j++)
            for (int i = 0; i < W; i++) {
                long p = (long)j * W + i;
                if (satMode == ORC_SAT_F32) {
                    volatile float c = GETF(i + h, j + h);
                    c = c - GETF(i + h, j - h);
                    c = c - GETF(i - h, j + h);
                    c = c + GETF(i - h, j - h);
                    if (k == 0 || c < bestF[p]) { bestF[p] = c; disp[p] = (uint8_t)(d & 0xFF); }
                } else {
                    int64_t c = GETI(i + h, j + h) - GETI(i + h, j - h) - GETI(i - h, j + h) + GETI(i - h, j - h);
                    if (k == 0 || c < bestI[p]) { bestI[p] = c; disp[p] = (uint8_t)(d & 0xFF); }
                }
            }
#undef GETF
#undef GETI
    }
    free(pad); free(If); free(Ii); free(bestF); free(bestI);
    return 0;
}

/*
 * computeSAD_BFOpti, methods_disp.cpp:311-408.  Per d: absdiff slice, then
 * sqrBoxFilter(win x win, normalize=false, BORDER_REFLECT_101) -- i.e. the window sum of SQUARED
 * absolute differences -- read back as float (:357,:399), first-min WTA, `.at<char>` store.
 * The sums are exact integers (<= 255^2 * win^2 < 2^32); the float cast rounds above 2^24.
 */
int orc_sad_box(const uint8_t* L, const uint8_t* R, int H, int W, int winSize, int minD, int maxD,
                int isDispLeft, uint8_t* disp)
{
    if (!L || !R || winSize % 2 == 0 || maxD < minD || minD < 0) return -1;
    int D = maxD - minD + 1, Wp = W + maxD;
    long N = (long)H * W;
    uint8_t* pad = (uint8_t*)malloc((size_t)H * Wp);
    const uint8_t* ref = isDispLeft ? L : R;
    if (isDispLeft) orc_pad_reflect_lr(R, H, W, maxD, 0, pad);
    else orc_pad_reflect_lr(L, H, W, 0, maxD, pad);
    uint32_t* sq = (uint32_t*)malloc(sizeof(uint32_t) * N);
    uint32_t* bs = (uint32_t*)malloc(sizeof(uint32_t) * N);
    float* best = (float*)malloc(sizeof(float) * N);
    for (int k = 0; k < D; k++) {
        int d = minD + k;
        int off = isDispLeft ? (maxD - d) : d;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                int a = ref[(long)y * W + x], b = pad[(long)y * Wp + off + x];
                int v = a > b ? a - b : b - a;
                sq[(long)y * W + x] = (uint32_t)(v * v);
            }
        orc_box_sum_u32(sq, H, W, winSize, winSize, bs);
        for (long p = 0; p < N; p++) {
            float c = (float)bs[p];
            if (k == 0 || c < best[p]) { best[p] = c; disp[p] = (uint8_t)(d & 0xFF); }
        }
    }
    free(pad); free(sq); free(bs); free(best);
    return 0;
}

/* PatchMatch::postProcess validity test, cameraCalibration/methods/patchmatch.cpp:817-828.
 * The matched column is clamped to [0, cols] for the left view and -- as written -- to
 * [0, rows] for the right view; an index equal to cols reads one past the row end in the
 * reference, which this restatement clamps to cols-1. */
void orc_lr_check(const float* dL, const float* dR, int H, int W, float thresh, uint8_t* validL, uint8_t* validR)
{
#pragma omp parallel for
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float a = dL[(long)y * W + x];
            int xr = (int)fmaxf(0.f, fminf((float)W, (float)x - a));
            if (xr > W - 1) xr = W - 1;
            validL[(long)y * W + x] = (uint8_t)(fabsf(a - dR[(long)y * W + xr]) <= thresh);
            float b = dR[(long)y * W + x];
            int xl = (int)fmaxf(0.f, fminf((float)H, (float)x + b));
            if (xl > W - 1) xl = W - 1;
            validR[(long)y * W + x] = (uint8_t)(fabsf(b - dL[(long)y * W + xl]) <= thresh);
        }
}
