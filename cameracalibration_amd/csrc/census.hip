// Census family, Hamming distance, reflect padding and BGR->gray for gfx950.
//
// Replaces countCensusImg / _rotationInv / _uniform / _multiScale / _circle and countHummingDist
// of cameraCalibration/disp_method/methods_disp.cpp:415-714, 808-838, plus the
// copyMakeBorder(BORDER_REFLECT) calls at :224,268,862,1005 and checkImg's cvtColor (:138-141).
#include "common.h"
#include <math.h>
#include <stdint.h>

// cv::borderInterpolate(BORDER_REFLECT): fedcba|abcdef|fedcba
__device__ __forceinline__ int reflect_idx(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - p - 1;
    return p;
}

// 16 bytes per lane along x; rows over blockIdx.y
__global__ __launch_bounds__(256) void pad_reflect_lr_kernel(const u8* __restrict__ src, size_t sp, int H, int W,
                                                              int left, int Wp, u8* __restrict__ dst, size_t dp)
{
    int y = blockIdx.y;
    int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x4 >= Wp) return;
    const u8* srow = src + (size_t)y * sp;
    u32 v = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int x = x4 + i;
        u32 b = x < Wp ? srow[reflect_idx(x - left, W)] : 0;
        v |= b << (8 * i);
    }
    // dst pitch is a multiple of 4 and covers the rounded-up width
    *reinterpret_cast<u32*>(dst + (size_t)y * dp + x4) = v;
}

struct CensusLut { u8 t[256]; };

// 3x3 census with the reference's bit order (7=TL 6=T 5=TR 4=R 3=BR 2=B 1=BL 0=L, nbr >= centre),
// 1-px frame left 0, optional 256-entry LUT applied to the code (rotation-invariant / uniform).
// Tile: 256 x 16 pixels per 256-thread workgroup, staged in LDS with a 1-px halo; each thread
// produces 4 horizontally adjacent codes per row and stores them as one dword.
#define CT_W 256
#define CT_H 16
#define CT_LW (CT_W + 8)   // 4 bytes of halo on each side keeps rows dword-aligned
template <bool USE_LUT>
__global__ __launch_bounds__(256) void census3x3_kernel(const u8* __restrict__ src, size_t sp, int H, int W,
                                                        u8* __restrict__ dst, size_t dp, CensusLut lut)
{
    __shared__ u32 tile[(CT_H + 2) * (CT_LW / 4)];
    __shared__ u8 slut[256];
    const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CT_H;
    const int t = threadIdx.x;
    if (USE_LUT) slut[t] = lut.t[t];
    constexpr int DW = CT_LW / 4;   // dwords per LDS row
    for (int i = t; i < (CT_H + 2) * DW; i += 256) {
        int r = i / DW, c = i % DW;
        int y = y0 - 1 + r, x = x0 - 4 + 4 * c;
        u32 v = 0;
        if (y >= 0 && y < H && x >= 0 && x < W)   // rows are padded to a multiple of 4 bytes
            v = *reinterpret_cast<const u32*>(src + (size_t)y * sp + x);
        tile[i] = v;
    }
    __syncthreads();
    const int tx = t & 63, ty = t >> 6;
    const u8* tb = reinterpret_cast<const u8*>(tile);
#pragma unroll
    for (int rr = 0; rr < CT_H / 4; rr++) {
        int ly = ty + 4 * rr;          // tile row
        int y = y0 + ly;
        if (y >= H) break;
        int xg = x0 + 4 * tx;
        if (xg >= W) continue;
        const u8* r0 = tb + (ly + 0) * CT_LW + 4 + 4 * tx;   // row y-1, column xg
        const u8* r1 = r0 + CT_LW;
        const u8* r2 = r1 + CT_LW;
        u32 out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int x = xg + i;
            u32 code = 0;
            if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) {
                u32 c = r1[i];
                code |= (u32)(r0[i - 1] >= c) << 7;
                code |= (u32)(r0[i] >= c) << 6;
                code |= (u32)(r0[i + 1] >= c) << 5;
                code |= (u32)(r1[i + 1] >= c) << 4;
                code |= (u32)(r2[i + 1] >= c) << 3;
                code |= (u32)(r2[i] >= c) << 2;
                code |= (u32)(r2[i - 1] >= c) << 1;
                code |= (u32)(r1[i - 1] >= c) << 0;
                if (USE_LUT) code = slut[code];
            }
            out |= code << (8 * i);
        }
        *reinterpret_cast<u32*>(dst + (size_t)y * dp + xg) = out;
    }
}

// The census planes of a batch of stereo pairs in one launch: blockIdx.z = 2 * pair + view.  View 0 is the left image as
// it is; view 1 is the right image with `left` reflected columns in front of it (what pad_reflect_lr_kernel followed by
// census3x3_kernel would give), read straight from the unpadded source.
struct CensusPairs {
    const u8* srcL; const u8* srcR; size_t sp, pairStride;
    int H, W, left;
    u8* dstL; size_t dpL, strideL;
    u8* dstR; size_t dpR, strideR;
    int alignedL;           // the left rows can be read as dwords
};
template <bool USE_LUT>
__global__ __launch_bounds__(256) void census3x3_pairs_kernel(CensusPairs a, CensusLut lut)
{
    __shared__ u32 tile[(CT_H + 2) * (CT_LW / 4)];
    __shared__ u8 slut[256];
    const int view = blockIdx.z & 1, pair = blockIdx.z >> 1;
    const int H = a.H, W = view ? a.W + a.left : a.W, left = view ? a.left : 0;
    const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CT_H;
    if (x0 >= W) return;
    const u8* src = (view ? a.srcR : a.srcL) + (size_t)pair * a.pairStride;
    u8* dst = view ? a.dstR + (size_t)pair * a.strideR : a.dstL + (size_t)pair * a.strideL;
    const size_t dp = view ? a.dpR : a.dpL;
    const int t = threadIdx.x;
    if (USE_LUT) slut[t] = lut.t[t];
    constexpr int DW = CT_LW / 4;
    const bool dwords = !view && a.alignedL;
    for (int i = t; i < (CT_H + 2) * DW; i += 256) {
        const int r = i / DW, c = i % DW;
        const int y = y0 - 1 + r, x = x0 - 4 + 4 * c;
        u32 v = 0;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const u8* row = src + (size_t)y * a.sp;
            if (dwords) v = *reinterpret_cast<const u32*>(row + x);
            else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (x + k < W) v |= (u32)row[reflect_idx(x + k - left, a.W)] << (8 * k);
            }
        }
        tile[i] = v;
    }
    __syncthreads();
    const int tx = t & 63, ty = t >> 6;
    const u8* tb = reinterpret_cast<const u8*>(tile);
#pragma unroll
    for (int rr = 0; rr < CT_H / 4; rr++) {
        const int ly = ty + 4 * rr, y = y0 + ly;
        if (y >= H) break;
        const int xg = x0 + 4 * tx;
        if (xg >= W) continue;
        const u8* r0 = tb + (ly + 0) * CT_LW + 4 + 4 * tx;
        const u8* r1 = r0 + CT_LW;
        const u8* r2 = r1 + CT_LW;
        u32 out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = xg + i;
            u32 code = 0;
            if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) {
                const u32 c = r1[i];
                code |= (u32)(r0[i - 1] >= c) << 7;
                code |= (u32)(r0[i] >= c) << 6;
                code |= (u32)(r0[i + 1] >= c) << 5;
                code |= (u32)(r1[i + 1] >= c) << 4;
                code |= (u32)(r2[i + 1] >= c) << 3;
                code |= (u32)(r2[i] >= c) << 2;
                code |= (u32)(r2[i - 1] >= c) << 1;
                code |= (u32)(r1[i - 1] >= c) << 0;
                if (USE_LUT) code = slut[code];
            }
            out |= code << (8 * i);
        }
        *reinterpret_cast<u32*>(dst + (size_t)y * dp + xg) = out;
    }
}

// cell-mean image of countCensusImg_multiScale (methods_disp.cpp:649-667): mean over a
// (2*offset+1)^2 block divided (integer) by cellSize^2, zero within `offset` of the border.
__global__ __launch_bounds__(256) void cell_mean_kernel(const u8* __restrict__ src, size_t sp, int H, int W,
                                                        int cellSize, u8* __restrict__ dst, size_t dp)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    int offset = cellSize / 2;
    u32 v = 0;
    if (cellSize > 0 && y >= offset && y < H - offset && x >= offset && x < W - offset) {
        int s = 0;
        for (int n = -offset; n <= offset; n++)
            for (int m = -offset; m <= offset; m++) s += src[(size_t)(y + n) * sp + x + m];
        v = (u32)(s / (cellSize * cellSize)) & 0xFF;
    }
    dst[(size_t)y * dp + x] = (u8)v;
}

struct CircleTaps { int fx[8], fy[8], cx[8], cy[8]; float w1[8], w2[8], w3[8], w4[8]; };

// countCensusImg_circle (methods_disp.cpp:456-500) restated deterministically: every output pixel
// sums its own sample bits (the reference accumulates them from racing OpenMP threads) and, as in
// the reference, the code of pixel (i,j) lands at (i-radius, j-radius).  The float expression is
// evaluated without fused multiply-add, left to right, as the reference's compiler would.
__global__ __launch_bounds__(256) void census_circle_kernel(const u8* __restrict__ src, size_t sp, int H, int W,
                                                            int radius, int npts, CircleTaps taps,
                                                            u8* __restrict__ dst, size_t dp)
{
    int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;   // output coordinates
    if (ox >= W) return;
    int i = oy + radius, j = ox + radius;
    u32 code = 0;
    if (i < H - radius && j < W - radius) {
        float c = (float)src[(size_t)i * sp + j];
        for (int n = 0; n < npts; n++) {
            float a = (float)src[(size_t)(i + taps.fy[n]) * sp + j + taps.fx[n]];
            float b = (float)src[(size_t)(i + taps.fy[n]) * sp + j + taps.cx[n]];
            float cc = (float)src[(size_t)(i + taps.cy[n]) * sp + j + taps.fx[n]];
            float d = (float)src[(size_t)(i + taps.cy[n]) * sp + j + taps.cx[n]];
            float tv = __fmul_rn(taps.w1[n], a);
            tv = __fadd_rn(tv, __fmul_rn(taps.w2[n], b));
            tv = __fadd_rn(tv, __fmul_rn(taps.w3[n], cc));
            tv = __fadd_rn(tv, __fmul_rn(taps.w4[n], d));
            u32 bit = (tv > c) || (fabsf(__fsub_rn(tv, c)) < 1.1920928955078125e-07f);
            code = (code + (bit << n)) & 0xFF;
        }
    }
    dst[(size_t)oy * dp + ox] = (u8)code;
}

__global__ __launch_bounds__(256) void hamming_kernel(const u8* __restrict__ a, size_t ap, const u8* __restrict__ b,
                                                      size_t bp, int H, int W, int32_t* __restrict__ dst, size_t dp)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    dst[(size_t)y * (dp / 4) + x] = __popc((u32)(a[(size_t)y * ap + x] ^ b[(size_t)y * bp + x]));
}

// OpenCV 8-bit BGR2GRAY: (B*1868 + G*9617 + R*4899 + 2^13) >> 14; rgbOrder: COLOR_RGB2GRAY, which reads the first
// channel as red (what stereomatch_1::computeNCC applies to its BGR input)
__global__ __launch_bounds__(256) void bgr2gray_kernel(const u8* __restrict__ bgr, size_t sp, int H, int W,
                                                       u8* __restrict__ gray, size_t dp, int rgbOrder)
{
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const u8* p = bgr + (size_t)y * sp + 3 * x;
    const int b = rgbOrder ? p[2] : p[0], r = rgbOrder ? p[0] : p[2];
    gray[(size_t)y * dp + x] = (u8)((b * 1868 + p[1] * 9617 + r * 4899 + (1 << 13)) >> 14);
}

// ---------------------------------------------------------------------------------------------

int md_pad_reflect_lr(hipStream_t s, const u8* src, size_t spitch, int H, int W, int left, int right, u8* dst, size_t dpitch)
{
    int Wp = W + left + right;
    dim3 grid((Wp + 1023) / 1024, H);
    pad_reflect_lr_kernel<<<grid, 256, 0, s>>>(src, spitch, H, W, left, Wp, dst, dpitch);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// RITable of countCensusImg_rotationInv (methods_disp.cpp:519-533): the loop never rotates, the
// table is min(i, i>>1) = i>>1.
static void ri_table(u8* t)
{
    for (int i = 0; i < 256; i++) {
        int val = i;
        for (int j = 0; j < 7; j++) { int temp = i >> 1; if (val > temp) val = temp; }
        t[i] = (u8)val;
    }
}

// hopCount (methods_disp.cpp:562-580): transitions of the circular pattern 0,b6,b5,...,b0.
static int hop_count(unsigned i)
{
    int a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 7; k >= 1; k--) { a[k] = i & 1; i >>= 1; }
    int cnt = 0;
    for (int j = 0; j < 8; j++) cnt += a[j] != a[(j + 1) & 7];
    return cnt;
}

// UPTable of countCensusImg_uniform (methods_disp.cpp:599-609)
static void uniform_table(u8* t)
{
    memset(t, 0, 256);
    u8 temp = 1;
    for (int i = 0; i < 256; i++)
        if (hop_count(i) <= 2) t[i] = temp++;
}

static int launch_census3x3(hipStream_t s, const u8* src, size_t sp, int H, int W, u8* dst, size_t dp, const u8* lut)
{
    dim3 grid((W + CT_W - 1) / CT_W, (H + CT_H - 1) / CT_H);
    CensusLut l;
    if (lut) {
        memcpy(l.t, lut, 256);
        census3x3_kernel<true><<<grid, 256, 0, s>>>(src, sp, H, W, dst, dp, l);
    } else {
        memset(l.t, 0, 256);
        census3x3_kernel<false><<<grid, 256, 0, s>>>(src, sp, H, W, dst, dp, l);
    }
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// src and dst rows must be padded to a multiple of 4 bytes (pitch % 4 == 0, pitch >= roundup4(W)).
// ---- STATISTIC_MULTISCALE_CENSUS (countCensusImg_multiScale2, methods_disp.cpp:677-714) ----
// histogram of the multi-scale codes (calcHist with 256 bins over [0,255): code 255 is dropped),
// rank table, remap.  All on the device, no host round trip.
__global__ __launch_bounds__(256) void code_hist_kernel(const u8* __restrict__ img, size_t pitch, int H, int W, u32* __restrict__ hist)
{
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (int y = blockIdx.y; y < H; y += gridDim.y)
        for (int x = blockIdx.x * 256 + threadIdx.x; x < W; x += gridDim.x * 256) {
            const int v = img[(size_t)y * pitch + x];
            if (v < 255) atomicAdd(&h[v], 1u);
        }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// The reference sorts the counts in decreasing order and, for i = 0..62, gives every bin whose
// count equals the i-th sorted count the code i (later i overwrite earlier ones); all other bins
// get 64.  With g = number of bins with a larger count and e = number with an equal count, the
// sorted positions holding this bin's count are g .. g+e-1, so the code is min(g+e-1, 62) when
// g <= 62, else 64.
__global__ __launch_bounds__(256) void rank_table_kernel(const u32* __restrict__ hist, u8* __restrict__ table)
{
    __shared__ u32 h[256];
    h[threadIdx.x] = hist[threadIdx.x];
    __syncthreads();
    const u32 mine = h[threadIdx.x];
    int g = 0, e = 0;
    for (int j = 0; j < 256; j++) { g += h[j] > mine; e += h[j] == mine; }
    table[threadIdx.x] = (u8)(g <= 62 ? min(g + e - 1, 62) : 64);
}
__global__ __launch_bounds__(256) void apply_table_kernel(u8* __restrict__ img, size_t pitch, int H, int W, const u8* __restrict__ table)
{
    __shared__ u8 t[256];
    t[threadIdx.x] = table[threadIdx.x];
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < W) img[(size_t)y * pitch + x] = t[img[(size_t)y * pitch + x]];
}

int md_census(hipStream_t s, const u8* src, size_t spitch, int H, int W, int method, int arg, u8* dst, size_t dpitch,
              u8* scratch)
{
    u8 lut[256];
    switch (method) {
    case MDISP_BASIC_CENSUS:
        return launch_census3x3(s, src, spitch, H, W, dst, dpitch, nullptr);
    case MDISP_ROTATION_INVARIANT_CENSUS:
        ri_table(lut);
        return launch_census3x3(s, src, spitch, H, W, dst, dpitch, lut);
    case MDISP_UNIFORM_CENSUS:
        uniform_table(lut);
        return launch_census3x3(s, src, spitch, H, W, dst, dpitch, lut);
    case MDISP_MULTISCALE_CENSUS: {
        MD_REQUIRE(scratch, MDISP_E_BADARG, "multi-scale census needs a scratch image");
        dim3 grid((W + 255) / 256, H);
        cell_mean_kernel<<<grid, 256, 0, s>>>(src, spitch, H, W, arg / 3, scratch, dpitch);
        MD_HIP(hipGetLastError());
        return launch_census3x3(s, scratch, dpitch, H, W, dst, dpitch, nullptr);
    }
    case MDISP_CIRCLE_CENSUS: {
        int radius = arg, npts = 8;
        MD_REQUIRE(radius >= 1, MDISP_E_BADARG, "circle census radius must be >= 1");
        CircleTaps tp;
        for (int n = 0; n < npts; n++) {
            // methods_disp.cpp:473-487, evaluated on the host in the reference's precision
            float x = (float)(radius * cos(2.0 * 3.1415926535897932384626433832795 * n / (float)npts));
            float y = (float)(-radius * sin(2.0 * 3.1415926535897932384626433832795 * n / (float)npts));
            int fx = (int)floor(x), fy = (int)floor(y), cx = (int)ceil(x), cy = (int)ceil(y);
            float tx = x - fx, ty = y - fy;
            tp.fx[n] = fx; tp.fy[n] = fy; tp.cx[n] = cx; tp.cy[n] = cy;
            tp.w1[n] = (1 - tx) * (1 - ty); tp.w2[n] = tx * (1 - ty); tp.w3[n] = (1 - tx) * ty; tp.w4[n] = tx * ty;
        }
        dim3 grid((W + 255) / 256, H);
        census_circle_kernel<<<grid, 256, 0, s>>>(src, spitch, H, W, radius, npts, tp, dst, dpitch);
        MD_HIP(hipGetLastError());
        return MDISP_OK;
    }
    case MDISP_STATISTIC_MULTISCALE_CENSUS: {
        MD_REQUIRE(scratch, MDISP_E_BADARG, "multi-scale census needs a scratch image");
        dim3 grid((W + 255) / 256, H);
        cell_mean_kernel<<<grid, 256, 0, s>>>(src, spitch, H, W, arg / 3, scratch, dpitch);
        MD_HIP(hipGetLastError());
        int rc = launch_census3x3(s, scratch, dpitch, H, W, dst, dpitch, nullptr);
        if (rc) return rc;
        // histogram and table live behind the scratch image (MD_CENSUS_SCRATCH_TAIL bytes)
        u32* hist = reinterpret_cast<u32*>(scratch + md_align((size_t)H * dpitch, 256));
        u8* table = reinterpret_cast<u8*>(hist + 256);
        MD_HIP(hipMemsetAsync(hist, 0, 256 * sizeof(u32), s));
        code_hist_kernel<<<dim3(4, 64), 256, 0, s>>>(dst, dpitch, H, W, hist);
        rank_table_kernel<<<1, 256, 0, s>>>(hist, table);
        apply_table_kernel<<<grid, 256, 0, s>>>(dst, dpitch, H, W, table);
        MD_HIP(hipGetLastError());
        return MDISP_OK;
    }
    case MDISP_CENSUS_2017:
        mdisp_set_error("CENSUS_2017 shifts past bit 31 and mutates shared state in the reference; not built");
        return MDISP_E_UNSUPPORTED;
    default:
        mdisp_set_error("unknown census method %d", method);
        return MDISP_E_BADARG;
    }
}

int md_hamming(hipStream_t s, const u8* a, size_t ap, const u8* b, size_t bp, int H, int W, int32_t* dst, size_t dp)
{
    dim3 grid((W + 255) / 256, H);
    hamming_kernel<<<grid, 256, 0, s>>>(a, ap, b, bp, H, W, dst, dp);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

int md_bgr2gray(hipStream_t s, const u8* bgr, size_t spitch, int H, int W, u8* gray, size_t dpitch, int rgbOrder)
{
    dim3 grid((W + 255) / 256, H);
    bgr2gray_kernel<<<grid, 256, 0, s>>>(bgr, spitch, H, W, gray, dpitch, rgbOrder);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// Census planes of `batch` stereo pairs (pair i at +i*pairStride bytes of dL / dR, rows `pitch` bytes apart) in one
// launch: cL receives the census of the left views (W columns), cR that of the right views with `left` reflected
// columns in front (W + left columns).  3x3 census methods only (basic, rotation-invariant, uniform).  The destination
// pitches must be multiples of 4 that cover the widths rounded up to 4.
int md_census_pairs(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, size_t pairStride, int batch, int H, int W, int left, int method,
                    u8* cL, size_t pitchL, size_t strideL, u8* cR, size_t pitchR, size_t strideR)
{
    MD_REQUIRE(method == MDISP_BASIC_CENSUS || method == MDISP_ROTATION_INVARIANT_CENSUS || method == MDISP_UNIFORM_CENSUS, MDISP_E_UNSUPPORTED,
               "batched census: 3x3 methods only");
    MD_REQUIRE(pitchL % 4 == 0 && pitchR % 4 == 0 && pitchL >= md_align((size_t)W, 4) && pitchR >= md_align((size_t)(W + left), 4), MDISP_E_BADARG,
               "batched census: destination pitch");
    CensusPairs a;
    a.srcL = dL; a.srcR = dR; a.sp = pitch; a.pairStride = pairStride; a.H = H; a.W = W; a.left = left;
    a.dstL = cL; a.dpL = pitchL; a.strideL = strideL; a.dstR = cR; a.dpR = pitchR; a.strideR = strideR;
    a.alignedL = pitch % 4 == 0 && pitch >= md_align((size_t)W, 4) && pairStride % 4 == 0 && ((uintptr_t)dL & 3) == 0;
    CensusLut l;
    memset(l.t, 0, 256);
    const dim3 grid((W + left + CT_W - 1) / CT_W, (H + CT_H - 1) / CT_H, 2 * batch);
    MD_REQUIRE(2 * batch <= 65535, MDISP_E_RANGE, "batched census: too many pairs for one launch");
    if (method == MDISP_BASIC_CENSUS) census3x3_pairs_kernel<false><<<grid, 256, 0, s>>>(a, l);
    else {
        if (method == MDISP_ROTATION_INVARIANT_CENSUS) ri_table(l.t); else uniform_table(l.t);
        census3x3_pairs_kernel<true><<<grid, 256, 0, s>>>(a, l);
    }
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}
