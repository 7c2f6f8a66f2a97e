// cv::StereoBM replacement (the reference's getDisparity_BM, methods/methods.cpp:1825-1881, and the call site at
// methods/methods.cpp:1205-1213): x-Sobel prefilter, window SAD over all candidate disparities, texture and
// uniqueness tests, sub-pixel fit, optional left-right validation and speckle filter.
//
// The library computes every column with incremental sums over clamped indices and then overwrites everything
// outside [maxD + w/2, W - w/2) x [w/2, H - w/2) with FILTERED; inside that rectangle no index is clamped, so the
// kernels evaluate exact window sums there and only there.
//
//   bm_xsobel_kernel   prefiltered planes of both images
//   bm_hsum_kernel     horizontal window sums per (row, candidate, valid column) as u16, plus the texture term
//   bm_vsum_wta_kernel a workgroup marches down a band of rows for 64 valid columns: wave k keeps the vertical running
//                      sums of candidates 32k..32k+31 in registers; per row the waves exchange their minima through
//                      LDS, test uniqueness on their own candidates and wave 0 writes the pixel
//   bm_validate_kernel validateDisparity(): per row, election of the cheapest match per right-image column by a
//                      packed 64-bit atomicMin (cost, then x), then the disp12MaxDiff test
#include "common.h"
#include <limits.h>

namespace {

__global__ __launch_bounds__(256) void bm_xsobel_kernel(const u8* src, size_t sp, int H, int W, int cap, u8* dst, size_t dp)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    int v = cap;
    const bool flatRow = (H & 1) && y == H - 1;           // rows come in pairs; an odd last row is left flat
    if (!flatRow && x > 0 && x < W - 1) {
        const int ya = y > 0 ? y - 1 : (H > 1 ? y + 1 : y), yb = y < H - 1 ? y + 1 : (H > 1 ? y - 1 : y);
        const u8 *r0 = src + (size_t)ya * sp, *r1 = src + (size_t)y * sp, *r2 = src + (size_t)yb * sp;
        const int g = (r0[x + 1] - r0[x - 1]) + 2 * (r1[x + 1] - r1[x - 1]) + (r2[x + 1] - r2[x - 1]);
        v = g < -cap ? 0 : g > cap ? 2 * cap : g + cap;
    }
    dst[(size_t)y * dp + x] = (u8)v;
}

__global__ __launch_bounds__(256) void bm_fill_kernel(int16_t* p, size_t n, int16_t v)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

struct BmGeom {
    int H, W, ndisp, mind, w, cap;
    int xmin, Wv;            // first valid column and number of valid columns
    int ymin, Hv;            // first valid row and number of valid rows
    int band;                // output rows per workgroup of the vertical pass
};

// hs[y][d][xv] (d = 0..ndisp-1 candidates in the library's order: disparity = ndisp-1-d+mind; d = ndisp: texture).
// The two row segments sit in LDS as bytes; a window is summed four taps at a time with v_sad_u8 on words assembled from
// aligned LDS dwords by v_alignbyte (the byte offset of a lane's window is not a multiple of four).
#define BM_LWORDS 12      // window words kept in registers (windows up to 47 columns)
__global__ __launch_bounds__(256) void bm_hsum_kernel(const u8* Lp, const u8* Rp, size_t pitch, BmGeom g, u16* hs)
{
    extern __shared__ u32 ldsw[];
    u8* lds = reinterpret_cast<u8*>(ldsw);
    const int w = g.w, wsz = 2 * w + 1;
    const int nL = 256 + 2 * w, nR = nL + g.ndisp - 1;
    const int offR = (nL + 8 + 3) & ~3;                    // 8 bytes of slack: the last word of a window may start a dword early
    u8* Ls = lds;
    u8* Rs = lds + offR;
    const int y = blockIdx.y, xv0 = blockIdx.x * 256;
    const int X0 = g.xmin + xv0;                           // image column of this block's first output
    const int maxD = g.mind + g.ndisp - 1;
    // left columns [X0 - w, X0 + 255 + w]; right columns start maxD further left (candidate d = 0 is the largest disparity)
    for (int i = threadIdx.x; i < nL + 8; i += 256) {
        const int X = X0 - w + i;
        Ls[i] = (i < nL && X < g.W) ? Lp[(size_t)y * pitch + X] : 0;
    }
    for (int i = threadIdx.x; i < nR + 8; i += 256) {
        const int X = X0 - w - maxD + i;
        Rs[i] = (i < nR && X >= 0 && X < g.W) ? Rp[(size_t)y * pitch + X] : 0;
    }
    __syncthreads();
    const int xv = xv0 + threadIdx.x;
    if (xv >= g.Wv) return;
    u16* out = hs + ((size_t)y * (g.ndisp + 1)) * g.Wv + xv;
    const u32* Lw = ldsw;
    const u32* Rw = ldsw + (offR >> 2);
    const int nW = (wsz + 3) >> 2, rem = wsz - 4 * (nW - 1);
    const u32 tailMask = rem == 4 ? 0xFFFFFFFFu : ((1u << (8 * rem)) - 1u);
    const int t = threadIdx.x, wl = t >> 2, shl = t & 3;
    auto word = [](u32 hi, u32 lo, int sh) -> u32 {          // bytes sh .. sh+3 of the pair (lo, hi); the shift is a per-lane operand
        return __builtin_amdgcn_alignbyte(hi, lo, (u32)sh);
    };
    if (nW <= BM_LWORDS) {
        // the left window does not depend on the candidate: its words are assembled once and stay in registers
        u32 lv[BM_LWORDS];
        {
            u32 lprev = Lw[wl];
#pragma unroll
            for (int k = 0; k < BM_LWORDS; k++) {
                const u32 lnext = k < nW ? Lw[wl + k + 1] : 0u;
                lv[k] = k < nW ? word(lnext, lprev, shl) : 0u;
                if (k == nW - 1) lv[k] &= tailMask;
                lprev = lnext;
            }
        }
        for (int d = 0; d < g.ndisp; d++) {
            const int ob = t + d, wr = ob >> 2, shr = ob & 3;   // right column = left column - (ndisp-1-d+mind)
            u32 rprev = Rw[wr], s = 0;
#pragma unroll
            for (int k = 0; k < BM_LWORDS; k++) {
                if (k < nW) {
                    const u32 rnext = Rw[wr + k + 1];
                    u32 rv = word(rnext, rprev, shr);
                    if (k == nW - 1) rv &= tailMask;
                    s = __builtin_amdgcn_sad_u8(lv[k], rv, s);
                    rprev = rnext;
                }
            }
            out[(size_t)d * g.Wv] = (u16)s;
        }
    } else {
        for (int d = 0; d < g.ndisp; d++) {
            const int ob = t + d, wr = ob >> 2, shr = ob & 3;   // right column = left column - (ndisp-1-d+mind)
            u32 lprev = Lw[wl], rprev = Rw[wr], s = 0;
            for (int k = 0; k < nW; k++) {
                const u32 lnext = Lw[wl + k + 1], rnext = Rw[wr + k + 1];
                u32 lv = word(lnext, lprev, shl), rv = word(rnext, rprev, shr);
                if (k == nW - 1) { lv &= tailMask; rv &= tailMask; }
                s = __builtin_amdgcn_sad_u8(lv, rv, s);
                lprev = lnext; rprev = rnext;
            }
            out[(size_t)d * g.Wv] = (u16)s;
        }
    }
    u32 lprev = Lw[wl], tx = 0;
    const u32 capw = (u32)g.cap * 0x01010101u;
    for (int k = 0; k < nW; k++) {
        const u32 lnext = Lw[wl + k + 1];
        u32 lv = word(lnext, lprev, shl), cv = capw;
        if (k == nW - 1) { lv &= tailMask; cv &= tailMask; }
        tx = __builtin_amdgcn_sad_u8(lv, cv, tx);
        lprev = lnext;
    }
    out[(size_t)g.ndisp * g.Wv] = (u16)tx;
}

constexpr int BM_DPT = 32;      // candidates per thread

struct BmSelect { int textureThreshold, uniquenessRatio; };

__global__ void bm_vsum_wta_kernel(const u16* hs, BmGeom g, BmSelect sel, int16_t* disp, int* cost)
{
    // exchange slots, double-buffered by row parity
    extern __shared__ int sm[];
    const int NW = blockDim.x / 64;
    int* s_min = sm;                            // [2][NW][64]
    int* s_arg = s_min + 2 * NW * 64;           // [2][NW][64]
    int* s_pn = s_arg + 2 * NW * 64;            // [2][2][64]   sad[mind+1], sad[mind-1]
    int* s_bad = s_pn + 2 * 2 * 64;             // [2][NW][64]  a candidate of this wave breaks uniqueness
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int xv = blockIdx.x * 64 + lane;
    const bool colOk = xv < g.Wv;
    const int xvc = colOk ? xv : g.Wv - 1;
    const int d0 = wv * BM_DPT;
    const int w = g.w, ndisp = g.ndisp;
    const int yv0 = blockIdx.y * g.band, yv1 = min(yv0 + g.band, g.Hv);
    const size_t rowStride = (size_t)(ndisp + 1) * g.Wv;
    // a row's base is uniform and the element offsets are 32-bit (one row of window sums is far below 2^31 elements):
    // no 64-bit address per candidate is kept in registers
    auto rowbase = [&](int y) { return hs + (size_t)y * rowStride; };
    auto cand_off = [&](int i) { return (unsigned)(min(d0 + i, ndisp - 1) * g.Wv + xvc); };
    // Every load below is unconditional (candidates beyond ndisp repeat the last one and are never looked at): a load under
    // its own condition is waited for alone, one memory round trip per candidate and row.
    int sums[BM_DPT], tsum = 0;
#pragma unroll
    for (int i = 0; i < BM_DPT; i++) sums[i] = 0;
    // window of the band's first row: image rows [ymin + yv0 - w, ymin + yv0 + w]
    const int yTop = g.ymin + yv0 - w;
    for (int j = 0; j <= 2 * w; j++) {
#pragma unroll
        for (int i = 0; i < BM_DPT; i++) sums[i] += rowbase(yTop + j)[cand_off(i)];
        tsum += rowbase(yTop + j)[(unsigned)(ndisp * g.Wv + xvc)];
    }
    for (int yv = yv0; yv < yv1; yv++) {
        const int par = yv & 1;
        const int Y = g.ymin + yv;
        // this wave's first minimum
        int lmin = INT_MAX, larg = 0;
#pragma unroll
        for (int i = 0; i < BM_DPT; i++)
            if (d0 + i < ndisp && sums[i] < lmin) { lmin = sums[i]; larg = d0 + i; }
        s_min[(par * NW + wv) * 64 + lane] = lmin;
        s_arg[(par * NW + wv) * 64 + lane] = larg;
        __syncthreads();
        int minsad = INT_MAX, mind = 0;
        for (int k = 0; k < NW; k++) {
            const int v = s_min[(par * NW + k) * 64 + lane];
            if (v < minsad) { minsad = v; mind = s_arg[(par * NW + k) * 64 + lane]; }
        }
        // uniqueness on this wave's candidates; the neighbours of the minimum for the sub-pixel fit
        const int thresh = minsad + minsad * sel.uniquenessRatio / 100;
        const int dP = mind == ndisp - 1 ? ndisp - 2 : mind + 1, dN = mind == 0 ? 1 : mind - 1;
        bool bad = false;
#pragma unroll
        for (int i = 0; i < BM_DPT; i++) {
            const int d = d0 + i;
            if (d < ndisp) {
                if ((d < mind - 1 || d > mind + 1) && sums[i] <= thresh) bad = true;
                if (d == dP) s_pn[(par * 2 + 0) * 64 + lane] = sums[i];
                if (d == dN) s_pn[(par * 2 + 1) * 64 + lane] = sums[i];
            }
        }
        s_bad[(par * NW + wv) * 64 + lane] = bad;
        __syncthreads();
        if (wv == 0 && colOk) {
            const int FILTERED = (g.mind - 1) * 16;
            int out = FILTERED;
            bool anyBad = false;
            if (sel.uniquenessRatio > 0)
                for (int k = 0; k < NW; k++) anyBad |= s_bad[(par * NW + k) * 64 + lane] != 0;
            if (tsum >= sel.textureThreshold && !anyBad) {
                const int p = s_pn[(par * 2 + 0) * 64 + lane], n = s_pn[(par * 2 + 1) * 64 + lane];
                const int q = p + n - 2 * minsad + abs(p - n);
                out = ((ndisp - mind - 1 + g.mind) * 256 + (q != 0 ? (p - n) * 256 / q : 0) + 15) >> 4;
                if (cost) cost[(size_t)Y * g.W + g.xmin + xv] = minsad;
            }
            disp[(size_t)Y * g.W + g.xmin + xv] = (int16_t)out;
        }
        // slide the window one row down
        if (yv + 1 < yv1) {
            const u16* add = rowbase(Y + w + 1);
            const u16* sub = rowbase(Y - w);
            int va[BM_DPT], vs[BM_DPT];
#pragma unroll
            for (int i = 0; i < BM_DPT; i++) { va[i] = add[cand_off(i)]; vs[i] = sub[cand_off(i)]; }
            const unsigned ot = (unsigned)(ndisp * g.Wv + xvc);
            tsum += (int)add[ot] - (int)sub[ot];
#pragma unroll
            for (int i = 0; i < BM_DPT; i++) sums[i] += va[i] - vs[i];
        }
    }
}

// validateDisparity (calib3d) on rows [y0, y0 + rows): one workgroup per row
__global__ __launch_bounds__(256) void bm_validate_kernel(int16_t* disp, const int* cost, int y0, int W, int minD, int ndisp, int maxDiff)
{
    extern __shared__ unsigned long long key[];          // [W] (cost << 32) | x of the cheapest pixel that lands on this column
    const int y = y0 + blockIdx.x;
    int16_t* dp = disp + (size_t)y * W;
    const int* cp = cost + (size_t)y * W;
    const int maxD = minD + ndisp, minX1 = max(maxD, 0), maxX1 = W + min(minD, 0);
    const int INVALID = (minD - 1) * 16;
    maxDiff *= 16;
    for (int x = threadIdx.x; x < W; x += 256) key[x] = ~0ull;
    __syncthreads();
    for (int x = minX1 + threadIdx.x; x < maxX1; x += 256) {
        const int d = dp[x];
        if (d == INVALID) continue;
        const int x2 = x - ((d + 8) >> 4);
        if (x2 < 0 || x2 >= W) continue;
        atomicMin(&key[x2], ((unsigned long long)(unsigned)cp[x] << 32) | (unsigned)x);    // smallest cost, then smallest x
    }
    __syncthreads();
    // the library keeps the elected disparities in a buffer of their own: freeze them before any pixel is rejected
    for (int x = threadIdx.x; x < W; x += 256) {
        const unsigned long long k = key[x];
        key[x] = (unsigned)(k == ~0ull ? INVALID : (int)dp[(unsigned)(k & 0xFFFFFFFFu)]);
    }
    __syncthreads();
    for (int x = minX1 + threadIdx.x; x < maxX1; x += 256) {
        const int d = dp[x];
        if (d == INVALID) continue;
        const int d0 = d >> 4, d1 = (d + 15) >> 4, x0 = x - d0, x1 = x - d1;
        const int a = (0 <= x0 && x0 < W) ? (int)(unsigned)key[x0] : INVALID, b = (0 <= x1 && x1 < W) ? (int)(unsigned)key[x1] : INVALID;
        if ((0 <= x0 && x0 < W && a > INVALID && abs(a - d) > maxDiff) && (0 <= x1 && x1 < W && b > INVALID && abs(b - d) > maxDiff))
            dp[x] = (int16_t)INVALID;
    }
}

}  // namespace

static bool bm_geom(int H, int W, const mdisp_stereobm_params* p, BmGeom& g)
{
    g.H = H; g.W = W; g.ndisp = p->numDisparities; g.mind = p->minDisparity; g.w = p->blockSize / 2; g.cap = p->preFilterCap; g.band = 32;
    const int maxD = g.mind + g.ndisp - 1;
    const int lofs = maxD > 0 ? maxD : 0, rofs = maxD < 0 ? -maxD : 0;
    g.xmin = lofs + g.w; g.Wv = W - g.w - g.xmin;
    g.ymin = g.w; g.Hv = H - 2 * g.w;
    if (lofs >= W || rofs >= W || W - rofs - g.ndisp + 1 < 1) return false;
    return g.Wv > 0 && g.Hv > 0;
}

int md_stereobm_check(int H, int W, const mdisp_stereobm_params* p)
{
    MD_REQUIRE(p->numDisparities > 0 && p->numDisparities % 16 == 0, MDISP_E_BADARG, "numDisparities must be positive and divisible by 16");
    MD_REQUIRE(p->blockSize >= 5 && p->blockSize <= 255 && (p->blockSize & 1) && p->blockSize < (W < H ? W : H), MDISP_E_BADARG,
               "SADWindowSize must be odd, be within 5..255 and be not larger than image width or height");
    MD_REQUIRE(p->preFilterCap >= 1 && p->preFilterCap <= 63, MDISP_E_BADARG, "preFilterCap must be within 1..63");
    MD_REQUIRE(p->textureThreshold >= 0 && p->uniquenessRatio >= 0, MDISP_E_BADARG, "texture threshold and uniqueness ratio must be non-negative");
    MD_REQUIRE(p->minDisparity >= 0, MDISP_E_UNSUPPORTED, "negative minDisparity is not built");
    MD_REQUIRE(p->numDisparities <= 16 * BM_DPT, MDISP_E_RANGE, "numDisparities above %d is not built", 16 * BM_DPT);
    MD_REQUIRE((long)p->minDisparity * 16 > -32768 && (long)(p->minDisparity + p->numDisparities) * 16 < 32767, MDISP_E_RANGE,
               "disparities do not fit CV_16S");
    return MDISP_OK;
}

size_t md_stereobm_bytes(int H, int W, const mdisp_stereobm_params* p)
{
    BmGeom g;
    const size_t pitch = md_align(W, 256);
    size_t n = 2 * md_align(pitch * H, 256);
    if (bm_geom(H, W, p, g)) n += md_align((size_t)H * (g.ndisp + 1) * g.Wv * sizeof(u16), 256);
    if (p->disp12MaxDiff >= 0) n += md_align((size_t)H * W * sizeof(int), 256);
    if (p->speckleRange >= 0 && p->speckleWindowSize > 0) n += md_align(md_speckle_scratch_bytes(H, W), 256);
    return n;
}

// dL / dR: 8-bit single-channel images; dDisp: H*W int16, contiguous
int md_stereobm_run(hipStream_t s, const u8* dL, const u8* dR, size_t pitchIn, int H, int W, const mdisp_stereobm_params* p,
                    int16_t* dDisp, void* ws)
{
    MD_TRY(md_stereobm_check(H, W, p));
    BmGeom g;
    const bool any = bm_geom(H, W, p, g);
    const int FILTERED = (p->minDisparity - 1) * 16;
    const size_t n = (size_t)H * W;
    bm_fill_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(dDisp, n, (int16_t)FILTERED);
    MD_HIP(hipGetLastError());
    if (!any) return MDISP_OK;
    const size_t pitch = md_align(W, 256);
    u8* base = (u8*)ws;
    size_t off = 0;
    auto take = [&](size_t b) { u8* q = base + off; off += md_align(b, 256); return q; };
    u8* Lp = take(pitch * H);
    u8* Rp = take(pitch * H);
    u16* hs = (u16*)take((size_t)H * (g.ndisp + 1) * g.Wv * sizeof(u16));
    int* cost = p->disp12MaxDiff >= 0 ? (int*)take(n * sizeof(int)) : nullptr;
    void* spk = (p->speckleRange >= 0 && p->speckleWindowSize > 0) ? take(md_speckle_scratch_bytes(H, W)) : nullptr;
    dim3 gp((W + 255) / 256, H);
    bm_xsobel_kernel<<<gp, 256, 0, s>>>(dL, pitchIn, H, W, g.cap, Lp, pitch);
    bm_xsobel_kernel<<<gp, 256, 0, s>>>(dR, pitchIn, H, W, g.cap, Rp, pitch);
    const int nL = 256 + 2 * g.w;
    const size_t ldsH = (((nL + 8 + 3) & ~3) + nL + g.ndisp + 8 + 3) & ~(size_t)3;
    bm_hsum_kernel<<<dim3((g.Wv + 255) / 256, H), 256, ldsH, s>>>(Lp, Rp, pitch, g, hs);
    const int NW = (g.ndisp + BM_DPT - 1) / BM_DPT;
    const size_t ldsV = (size_t)(3 * 2 * NW * 64 + 2 * 2 * 64) * sizeof(int);
    BmSelect sel{p->textureThreshold, p->uniquenessRatio};
    if (cost) MD_HIP(hipMemsetAsync(cost, 0, n * sizeof(int), s));
    // Rows per workgroup of the vertical pass: a band first sums the 2w + 1 rows above its first output row, so taller bands
    // read less, but there must be enough workgroups for the device (16 waves per CU at this kernel's register count).
    // Measured (1920x1080 x 256, block 21: 16 / 32 / 64 / 128 rows = 1.26 / 0.87 / 0.57 / 1.23 ms; 1280x720 x 144, block 35:
    // 32 best; 640x480 x 64: 16 best): the tallest of 64, 32, 16 that still gives 0.4 workgroups per place.
    {
        int dev = 0, cus = 0;
        MD_HIP(hipGetDevice(&dev));
        MD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const long places = (long)std::max(cus, 1) * std::max(16 / NW, 1);
        g.band = 16;
        for (int b = 64; b > 16; b /= 2)
            if ((long)((g.Wv + 63) / 64) * ((g.Hv + b - 1) / b) * 10 >= places * 4) { g.band = b; break; }
    }
    bm_vsum_wta_kernel<<<dim3((g.Wv + 63) / 64, (g.Hv + g.band - 1) / g.band), NW * 64, ldsV, s>>>(hs, g, sel, dDisp, cost);
    MD_HIP(hipGetLastError());
    if (cost) {
        MD_REQUIRE((size_t)W * 8 <= 64 * 1024, MDISP_E_RANGE, "disp12MaxDiff >= 0 is built for widths up to 8192");
        bm_validate_kernel<<<g.Hv, 256, (size_t)W * 8, s>>>(dDisp, cost, g.ymin, W, p->minDisparity, g.ndisp, p->disp12MaxDiff);
        MD_HIP(hipGetLastError());
    }
    if (spk) MD_TRY(md_filter_speckles_s16(s, dDisp, H, W, FILTERED, p->speckleWindowSize, p->speckleRange, spk));
    return MDISP_OK;
}
