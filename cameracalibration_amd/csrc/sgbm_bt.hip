// cv::StereoSGBM::compute on gfx950 with the library's own matching cost: x-Sobel prefilter +
// Birchfield-Tomasi pixel cost + block sum, MODE_SGBM (5 paths) / MODE_HH (8 paths) aggregation with
// 16-bit saturating sums, winner-take-all + uniqueness + x16 parabola + disp12MaxDiff, then
// medianBlur(3) and filterSpeckles.  Replaces the call sites of
// cameraCalibration/methods/methods.cpp:1883-1921 (`StereoSGBM::create(0, 48, 7)`), :1216-1219 and
// :1565-1568 (`create(0,64,7,490,1960,1,63,10,100,32,MODE_SGBM)`).  OpenCV 4.1.0's stereosgbm.cpp
// is not part of the reference tree; its published algorithm is restated here (conventions listed
// at the kernels).
#include "common.h"
#include <stdlib.h>

namespace {

struct BtGeom {
    int H, W, cn, D, minD, W1, minX1, ftzero, bs;
    size_t N;         // H*W, elements per plane
};

// planes [2*cn][H][W]: clipped x-Sobel of channel c (rows replicated at the top / bottom), then the
// raw channel; columns 0 and W-1 of every plane hold ftzero
__global__ __launch_bounds__(256) void bt_planes_kernel(const u8* __restrict__ img, size_t pitch, BtGeom g, u8* __restrict__ pl)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.W) return;
    const int cn = g.cn;
    const u8* r1 = img + (size_t)y * pitch;
    const u8* r0 = img + (size_t)(y > 0 ? y - 1 : y) * pitch;
    const u8* r2 = img + (size_t)(y < g.H - 1 ? y + 1 : y) * pitch;
    for (int c = 0; c < cn; c++) {
        int s = g.ftzero, raw = g.ftzero;
        if (x > 0 && x < g.W - 1) {
            const int a = (x + 1) * cn + c, b = (x - 1) * cn + c;
            const int v = ((int)r1[a] - (int)r1[b]) * 2 + (int)r0[a] - (int)r0[b] + (int)r2[a] - (int)r2[b];
            s = min(max(v, -g.ftzero), g.ftzero) + g.ftzero;
            raw = r1[x * cn + c];
        }
        pl[(size_t)c * g.N + (size_t)y * g.W + x] = (u8)s;
        pl[(size_t)(cn + c) * g.N + (size_t)y * g.W + x] = (u8)raw;
    }
}

__device__ __forceinline__ int bt_one(const u8* __restrict__ a, const u8* __restrict__ b, int W, int x, int x2)
{
    const int u = a[x];
    const int ul = x > 0 ? (u + a[x - 1]) / 2 : u, ur = x < W - 1 ? (u + a[x + 1]) / 2 : u;
    const int u0 = min(min(ul, ur), u), u1 = max(max(ul, ur), u);
    const int v = b[x2];
    const int vl = x2 > 0 ? (v + b[x2 - 1]) / 2 : v, vr = x2 < W - 1 ? (v + b[x2 + 1]) / 2 : v;
    const int v0 = min(min(vl, vr), v), v1 = max(max(vl, vr), v);
    const int c0 = max(max(0, u - v1), v0 - u);
    const int c1 = max(max(0, v - u1), u0 - v);
    return min(c0, c1);
}

// the same on rows whose edge pixels are repeated once on either side
__device__ __forceinline__ int bt_one_padded(const u8* a, const u8* b, int x, int x2)
{
    const int u = a[x], ul = (u + a[x - 1]) / 2, ur = (u + a[x + 1]) / 2;
    const int u0 = min(min(ul, ur), u), u1 = max(max(ul, ur), u);
    const int v = b[x2], vl = (v + b[x2 - 1]) / 2, vr = (v + b[x2 + 1]) / 2;
    const int v0 = min(min(vl, vr), v), v1 = max(max(vl, vr), v);
    const int c0 = max(max(0, u - v1), v0 - u);
    const int c1 = max(max(0, v - u1), u0 - v);
    return min(c0, c1);
}

// pixel cost of the computed columns x in [minX1, W): pix[y][x1][k], disparity minD + k; the raw
// planes count a quarter
__global__ __launch_bounds__(256) void bt_cost_kernel(const u8* __restrict__ p1, const u8* __restrict__ p2, BtGeom g,
                                                      int16_t* __restrict__ pix)
{
    const int e = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (e >= g.W1 * g.D) return;
    const int x1 = e / g.D, k = e - x1 * g.D;
    const int x = x1 + g.minX1, x2 = x - (g.minD + k);
    int cost = 0;
    for (int c = 0; c < 2 * g.cn; c++) {
        const u8* a = p1 + (size_t)c * g.N + (size_t)y * g.W;
        const u8* b = p2 + (size_t)c * g.N + (size_t)y * g.W;
        cost += bt_one(a, b, g.W, x, x2) >> (c < g.cn ? 0 : 2);
    }
    pix[((size_t)y * g.W1 + x1) * g.D + k] = (int16_t)cost;
}

// horizontal (VERT = false) or vertical window sum with the window clamped to the computed
// columns / the image rows; 16-bit wrap-around like the library's CostType arithmetic.  The
// vertical pass gives the first computed column the block cost of row 0 in every row.
template <bool VERT>
__global__ __launch_bounds__(256) void bt_window_kernel(const int16_t* __restrict__ in, BtGeom g, int16_t* __restrict__ out)
{
    const int e = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (e >= g.W1 * g.D) return;
    const int x1 = e / g.D, k = e - x1 * g.D, h = g.bs / 2;
    int s = 0;
    if (VERT) {
        const int yy = x1 == 0 ? 0 : y;
        for (int i = -h; i <= h; i++) s += in[((size_t)min(max(yy + i, 0), g.H - 1) * g.W1 + x1) * g.D + k];
    } else {
        for (int i = -h; i <= h; i++) s += in[((size_t)y * g.W1 + min(max(x1 + i, 0), g.W1 - 1)) * g.D + k];
    }
    out[((size_t)y * g.W1 + x1) * g.D + k] = (int16_t)s;
}

// Which accesses of the cost / path volumes are marked non-temporal (streamed once, far larger than the caches), as measured at
// 1920 x 1080 x 256: 1 | 2 the scan-line kernels' loads and stores (3.94 -> 3.85 ms; either alone is slower), 8 the stores of the
// horizontal sums (-> 3.82); not 4, the vertical sum (it reads every row twice, bs rows apart: 3.96), 16 the selection (no change).
#ifndef BT_NT
#define BT_NT 11
#endif
typedef u32 bt_u32x2 __attribute__((ext_vector_type(2)));
typedef u32 bt_u32x4 __attribute__((ext_vector_type(4)));
template <int BIT = 1, class T> __device__ __forceinline__ T bt_ld(const T* p)
{
    if constexpr ((BT_NT & BIT) == 0) return *p;
    else {
        T r;
        if constexpr (sizeof(T) == 2) { const unsigned short v = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(p)); __builtin_memcpy(&r, &v, 2); }
        else if constexpr (sizeof(T) == 4) { const u32 v = __builtin_nontemporal_load(reinterpret_cast<const u32*>(p)); __builtin_memcpy(&r, &v, 4); }
        else if constexpr (sizeof(T) == 8) { const bt_u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const bt_u32x2*>(p)); __builtin_memcpy(&r, &v, 8); }
        else if constexpr (sizeof(T) == 16) { const bt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const bt_u32x4*>(p)); __builtin_memcpy(&r, &v, 16); }
        else { static_assert(sizeof(T) == 32, "size"); const bt_u32x4 v0 = __builtin_nontemporal_load(reinterpret_cast<const bt_u32x4*>(p)); const bt_u32x4 v1 = __builtin_nontemporal_load(reinterpret_cast<const bt_u32x4*>(p) + 1); __builtin_memcpy(&r, &v0, 16); __builtin_memcpy(reinterpret_cast<char*>(&r) + 16, &v1, 16); }
        return r;
    }
}
template <int BIT = 2, class T> __device__ __forceinline__ void bt_st(T* p, const T& val)
{
    if constexpr ((BT_NT & BIT) == 0) *p = val;
    else {
        if constexpr (sizeof(T) == 2) { unsigned short v; __builtin_memcpy(&v, &val, 2); __builtin_nontemporal_store(v, reinterpret_cast<unsigned short*>(p)); }
        else if constexpr (sizeof(T) == 4) { u32 v; __builtin_memcpy(&v, &val, 4); __builtin_nontemporal_store(v, reinterpret_cast<u32*>(p)); }
        else if constexpr (sizeof(T) == 8) { bt_u32x2 v; __builtin_memcpy(&v, &val, 8); __builtin_nontemporal_store(v, reinterpret_cast<bt_u32x2*>(p)); }
        else { static_assert(sizeof(T) == 16, "size"); bt_u32x4 v; __builtin_memcpy(&v, &val, 16); __builtin_nontemporal_store(v, reinterpret_cast<bt_u32x4*>(p)); }
    }
}
// Pixel cost and horizontal window sum in one pass: a workgroup holds the plane rows of image row y in LDS, lane k
// marches along the computed columns with a running sum (entering column - leaving column, both with the window
// clamped to the computed columns), so the pixel costs never reach memory.  Same values as bt_cost_kernel followed by
// bt_window_kernel<false>.
#define BT_MULTI_MAX_UNITS ((size_t)96 << 20)   // cost-volume elements up to which a group's directions share one launch
#define BT_SEG 128     // columns per marching segment of bt_cost_hsum_kernel
#define BT_VBAND 120   // rows per marching band of bt_vsum_kernel
template <int CN>      // channels at compile time (the plane loop of a pixel cost unrolls); 0: any
__global__ __launch_bounds__(256) void bt_cost_hsum_kernel(const u8* __restrict__ p1, const u8* __restrict__ p2, BtGeom g,
                                                           int16_t* __restrict__ out)
{
    // Per plane and pixel of image row y one dword: the value u, and the minimum u0 and maximum u1 of u and its two
    // half-way neighbours (u + u(x-1)) / 2, (u + u(x+1)) / 2 -- at the image edge the pixel itself, as the library defines
    // them.  Two tables with their own origins: the left planes [np][TSL] hold the columns [xl0, xl1] of the segment's
    // windows (at most BT_SEG + bs - 1 of them), the right planes [np][TSR] the columns those reach through this
    // workgroup's 256 candidates, [xl0 - minD - kHi, xl1 - minD - kLo] clipped to the image (at most 255 more), so
    // neither extent depends on minDisparity or on which block of candidates the workgroup owns.  A pixel cost is then
    // two table reads per plane (the left one the same for every lane) and a dozen integer operations.
    extern __shared__ u32 btTable[];
    const int cn = CN ? CN : g.cn, np = 2 * cn, W = g.W, y = blockIdx.y;
    const int h = g.bs / 2, W1 = g.W1;
    const int xa = blockIdx.z * BT_SEG, xb = min(xa + BT_SEG, W1);
    const int k = blockIdx.x * 256 + threadIdx.x, TSL = BT_SEG + g.bs, TSR = BT_SEG + g.bs + 255;
    u32* const tabL = btTable;
    u32* const tabR = btTable + np * TSL;
    int llo, rlo;
    {
        const int kLo = blockIdx.x * 256, kHi = min(kLo + 255, g.D - 1);
        const int xl0 = max(xa - h, 0) + g.minX1, xl1 = min(xb - 1 + h, W1 - 1) + g.minX1;      // left columns of the segment's windows
        const int r0 = max(xl0 - (g.minD + kHi), 0), r1 = min(xl1 - (g.minD + kLo), W - 1);
        const int nl = xl1 - xl0 + 1, nr = r1 - r0 + 1;
        llo = xl0; rlo = r0;
        auto entry = [&](const u8* row, int x) -> u32 {
            const int u = row[x], ul = (u + row[max(x - 1, 0)]) / 2, ur = (u + row[min(x + 1, W - 1)]) / 2;
            return (u32)u | ((u32)min(min(ul, ur), u) << 8) | ((u32)max(max(ul, ur), u) << 16);
        };
        for (int i = threadIdx.x; i < np * nl; i += 256) {
            const int c = i / nl, x = xl0 + i - c * nl;
            tabL[c * TSL + x - xl0] = entry(p1 + (size_t)c * g.N + (size_t)y * W, x);
        }
        for (int i = threadIdx.x; i < np * nr; i += 256) {
            const int c = i / nr, x = r0 + i - c * nr;
            tabR[c * TSR + x - r0] = entry(p2 + (size_t)c * g.N + (size_t)y * W, x);
        }
    }
    __syncthreads();
    if (k >= g.D) return;
    auto pix = [&](int x1) -> int {
        const int x = x1 + g.minX1, x2 = x - (g.minD + k);
        int cost = 0;
#pragma unroll
        for (int c = 0; c < np; c++) {
            const u32 l = tabL[c * TSL + x - llo], r = tabR[c * TSR + x2 - rlo];
            const int u = l & 0xFF, u0 = (l >> 8) & 0xFF, u1 = (l >> 16) & 0xFF;
            const int v = r & 0xFF, v0 = (r >> 8) & 0xFF, v1 = (r >> 16) & 0xFF;
            const int c0 = max(max(0, u - v1), v0 - u);
            const int c1 = max(max(0, v - u1), u0 - v);
            cost += min(c0, c1) >> (c < cn ? 0 : 2);
        }
        return cost;
    };
    // the row is cut into segments of BT_SEG columns (blockIdx.z), each starting from a freshly summed window.
    // Every pixel cost is evaluated once: a lane keeps the costs of the window's columns in its own ring of bs + 1
    // slots in LDS ([slot][lane], conflict-free), so the column that leaves the window is read back, not recomputed.
    int16_t* ring = reinterpret_cast<int16_t*>(btTable + np * (TSL + TSR)) + threadIdx.x;
    const int R = g.bs + 1;
    int slotLeave = (xa - h) % R;
    if (slotLeave < 0) slotLeave += R;
    int slotEnter = slotLeave;
    int s = 0;
    for (int i = -h; i <= h; i++) {
        const int v = pix(min(max(xa + i, 0), W1 - 1));
        s += v;
        ring[slotEnter * 256] = (int16_t)v;
        slotEnter = slotEnter + 1 == R ? 0 : slotEnter + 1;
    }
    int16_t* o = out + (size_t)y * W1 * g.D + k;
    bt_st<8>(o + (size_t)xa * g.D, (int16_t)s);
    for (int x1 = xa + 1; x1 < xb; x1++) {
        const int enter = pix(min(x1 + h, W1 - 1));
        s += enter - (int)ring[slotLeave * 256];
        ring[slotEnter * 256] = (int16_t)enter;
        slotEnter = slotEnter + 1 == R ? 0 : slotEnter + 1;
        slotLeave = slotLeave + 1 == R ? 0 : slotLeave + 1;
        bt_st<8>(o + (size_t)x1 * g.D, (int16_t)s);
    }
}

// Vertical window sum as a running sum down the image (window clamped to the image rows); the first computed
// column keeps the block cost of row 0 in every row.  Same values as bt_window_kernel<true>.
// A lane owns four neighbouring elements of a row (D is a multiple of 16, so the row length is a multiple of four and the
// four share their column): 8-byte loads and stores.
__global__ __launch_bounds__(256) void bt_vsum_kernel(const int16_t* __restrict__ in, BtGeom g, int16_t* __restrict__ out)
{
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4, rowN = (size_t)g.W1 * g.D;
    if (e >= rowN) return;
    const int h = g.bs / 2, H = g.H;
    const bool frozen = e < (size_t)g.D;                 // x1 == 0: row 0's sum everywhere
    const int ya = blockIdx.y * BT_VBAND, yb = min(ya + BT_VBAND, H), yc = frozen ? 0 : ya;
    auto load4 = [&](int y, int (&v)[4]) {
        const uint2 w = bt_ld<4>(reinterpret_cast<const uint2*>(in + (size_t)y * rowN + e));
        v[0] = (int16_t)(w.x & 0xFFFFu); v[1] = (int16_t)(w.x >> 16); v[2] = (int16_t)(w.y & 0xFFFFu); v[3] = (int16_t)(w.y >> 16);
    };
    auto store4 = [&](int y, const int (&v)[4]) {
        uint2 w;
        w.x = ((u32)v[0] & 0xFFFFu) | ((u32)v[1] << 16); w.y = ((u32)v[2] & 0xFFFFu) | ((u32)v[3] << 16);
        bt_st<4>(reinterpret_cast<uint2*>(out + (size_t)y * rowN + e), w);
    };
    int s[4] = {0, 0, 0, 0};
    for (int i = -h; i <= h; i++) {
        int v[4];
        load4(min(max(yc + i, 0), H - 1), v);
#pragma unroll
        for (int q = 0; q < 4; q++) s[q] += v[q];
    }
    store4(ya, s);
    // unconditional loads (the frozen column drops the difference), several rows in flight
#pragma unroll 4
    for (int y = ya + 1; y < yb; y++) {
        int a[4], b[4];
        load4(min(y + h, H - 1), a);
        load4(max(y - h - 1, 0), b);
#pragma unroll
        for (int q = 0; q < 4; q++) s[q] += frozen ? 0 : a[q] - b[q];
        store4(y, s);
    }
}

// ---------------------------------------------------------------------------------------------
// One wave per scan line of the cropped W1 x H domain, lane l owns disparities l*ND .. l*ND+ND-1.
// L_r = C + min(L_r(p-r,d), L_r(p-r,d-1)+P1, L_r(p-r,d+1)+P1, minL_r(p-r)+P2) - (minL_r(p-r)+P2),
// zero outside the domain, stored as 16-bit values; the direction's costs are added into `acc`.
#define BT_BIG 0x3FFF0000
template <int ND> struct CVec;
template <> struct CVec<1> { typedef int16_t type; };
template <> struct CVec<2> { typedef u32 type; };
template <> struct CVec<4> { typedef uint2 type; };
template <> struct CVec<8> { typedef uint4 type; };
// ND accumulator values as one register block (what a vector load returns)
struct alignas(16) bt_int8 { uint4 a, b; };
template <class T, int ND> struct AccVec { typedef typename CVec<ND>::type type; };      // int16
template <> struct AccVec<int, 1> { typedef int type; };
template <> struct AccVec<int, 2> { typedef uint2 type; };
template <> struct AccVec<int, 4> { typedef uint4 type; };
template <> struct AccVec<int, 8> { typedef bt_int8 type; };

struct PathArgs {
    const int16_t* C; void* acc; int H, W1, D, P1, P2; int16_t* S; int firstGroup;
    int mdx[4], mdy[4], mfirst[5];        // MULTI: the directions of one launch and the first line index of each
    const void* accIn;                    // the group's sum so far (read); `acc` receives the new sum
    void* opp;                            // OPP: the path values of the direction opposite to the launch's own
    // SEL: the last direction of the last group selects the pixel's disparity instead of storing S
    u32* key; uint2* rec; int W, minX1, minD, q;      // q = 100 - uniquenessRatio
};

// cross-lane moves on the DPP path (no LDS crossbar round trip in the serial recursion)
__device__ __forceinline__ int wave_shr1(int v, int fill)      // lane i <- lane i-1, lane 0 <- fill
{ return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xF, 0xF, false); }
__device__ __forceinline__ int wave_shl1(int v, int fill)      // lane i <- lane i+1, lane 63 <- fill
{ return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xF, 0xF, false); }
__device__ __forceinline__ int wave_min_dpp(int v)
{
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true));    // row_half_mirror
    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true));    // row_mirror: every lane holds its row's minimum
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return min(min(r0, r1), min(r2, r3));
}

// FIRST: the group's first direction (nothing accumulated yet).  LAST: the group's last direction adds the group's sum
// into S with 16-bit saturation (S = saturate(S + acc + L); the first group starts S at zero) instead of storing it.
// AccT: the group's running sum of up to three directions is kept as int16 when the cost bound allows it, else int32.
// MULTI: one launch runs the lines of all directions of a group side by side and adds their values into a zeroed
// int32 `acc` with atomics (integer sums do not depend on the order) -- for small problems, where a single direction
// has too few lines to fill the GPU and the serial recursion of each line sets the time.
// OPP (large problems, 16-bit sums): a horizontal direction has one line per image row, about one wave per SIMD, and takes
// the time of its serial chain whatever else runs.  With FIRST, the launch therefore runs the opposite direction as well, on
// as many more waves (lines nlines .. 2 * nlines - 1), and keeps its path values in `opp`.  With LAST, `opp` is added
// after the group's own sum: S = saturate(saturate(S + acc + L) + opp), the sum the next group would have formed.
// SEL (with LAST, for the vertical and diagonal directions: thousands of lines, bound by memory and not by the chain of a
// line): the wave that completes a pixel's sums S holds all of them in its lanes, so it selects on the spot what
// bt_select_kernel would read S back for -- first-minimum winner, uniqueness test, the candidate for the right view's map
// (a 32-bit atomic minimum per pixel, order-independent) and the two neighbours of the minimum for the parabola -- and
// writes an 8-byte record per pixel; S is never stored (2 bytes per candidate written and read back saved).
// bt_finish_kernel turns the records into the map.
// PK (16-bit group sums, at least two candidates per lane): the recursion itself runs on packed pairs of 16-bit values --
// the costs and the group sums are packed pairs in memory already, so nothing is unpacked or packed on the way.  Legal
// whenever the 16-bit group sums are (3 * max(Cmax, P2) <= 32767): every path value lies in [-P2, Cmax], minL + P2 in
// [0, Cmax + P2], so no 16-bit operation wraps; only "SHRT_MAX + P1" of the candidates beyond the range must not, and that
// addition saturates.  Only the direction that completes S widens its values (the library's saturation points).
typedef short bt_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 pk_min_s16(u32 a, u32 b)
{ return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(bt_s2, a), __builtin_bit_cast(bt_s2, b))); }
__device__ __forceinline__ u32 pk_add_s16(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_bit_cast(bt_s2, a) + __builtin_bit_cast(bt_s2, b)); }
__device__ __forceinline__ u32 pk_sub_s16(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_bit_cast(bt_s2, a) - __builtin_bit_cast(bt_s2, b)); }
__device__ __forceinline__ u32 pk_add_sat_s16(u32 a, u32 b)
{ return __builtin_bit_cast(u32, __builtin_elementwise_add_sat(__builtin_bit_cast(bt_s2, a), __builtin_bit_cast(bt_s2, b))); }

template <int ND, bool FIRST, bool LAST = false, class AccT = int, bool MULTI = false, bool OPP = false, bool SEL = false, bool PK = false>
__global__ __launch_bounds__(256) void bt_path_kernel(PathArgs a, int dx, int dy, int nlines)
{
    static_assert(!PK || (ND >= 2 && sizeof(AccT) == 2 && !MULTI), "packed recursion: 16-bit group sums, pairs of candidates per lane");
    constexpr int NP = ND >= 2 ? ND / 2 : 1;
    constexpr u32 SENT2 = 0x7FFF7FFFu;
    AccT* accp = static_cast<AccT*>(a.acc);
    const AccT* const accInp = static_cast<const AccT*>(a.accIn);
    // every element of S is read (several steps ahead) and later written by the same lane and by nobody else
    const int16_t* __restrict__ Sin = a.S;
    int16_t* __restrict__ Sout = a.S;
    const int lane = threadIdx.x & 63;
    // the line is the same for every lane of a wave: kept in a scalar register, so that the loop bounds are uniform
    int line = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    // OPP without LAST: the opposite direction runs beside the launch's own, on as many more waves; it is always the first
    // to write its buffer, whether the launch's own direction is the group's first or not
    constexpr bool BESIDE = OPP && !LAST;
    if (line >= (BESIDE ? 2 * nlines : nlines)) return;
    const bool oppHalf = BESIDE && line >= nlines;         // (wave-uniform: `line` lives in a scalar register)
    if (oppHalf) { line -= nlines; dx = -dx; dy = -dy; accp = static_cast<AccT*>(a.opp); }
    const bool first = FIRST || oppHalf;
    if (MULTI) {
        int i = 0;
        while (i < 3 && line >= a.mfirst[i + 1]) i++;
        dx = a.mdx[i]; dy = a.mdy[i]; line -= a.mfirst[i];
    }
    const int H = a.H, W = a.W1, D = a.D;
    int x0, y0;
    if (dy == 0) { y0 = line; x0 = dx > 0 ? 0 : W - 1; }
    else {
        const int ys = dy > 0 ? 0 : H - 1;
        if (line < W) { x0 = line; y0 = ys; }
        else { const int i = line - W; y0 = dy > 0 ? 1 + i : H - 2 - i; x0 = dx > 0 ? 0 : W - 1; }
    }
    const int nx = dx > 0 ? W - x0 : (dx < 0 ? x0 + 1 : 0x7FFFFFFF);
    const int ny = dy > 0 ? H - y0 : (dy < 0 ? y0 + 1 : 0x7FFFFFFF);
    const int n = min(nx, ny);
    const int k0 = lane * ND;
    const bool mine = k0 < D;                 // D is a multiple of 16 >= ND, so a lane is all in or all out
    int Lp[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) Lp[j] = mine ? 0 : 32767;    // L_r(., -1) = L_r(., D) = SHRT_MAX
    u32 Pp[NP];                               // PK: the same as packed pairs [candidate 2i | candidate 2i + 1]
#pragma unroll
    for (int i = 0; i < NP; i++) Pp[i] = mine ? 0u : SENT2;
    const u32 P1P1 = (u32)a.P1 | ((u32)a.P1 << 16);
    int minPrev = 0;
    const int P1 = a.P1, P2 = a.P2;
    typedef typename CVec<ND>::type V;
    // operands of step t + PF are requested when step t retires, so PF cost rows (and accumulator rows)
    // are in flight per wave while the serial recursion runs
    // The requests are unconditional (lanes beyond D ask for the first lane's values, steps beyond the end for the
    // last pixel again) and what they return stays packed until it is consumed: a load under a per-lane condition, or
    // one whose result is widened for the ring, is waited for right behind the load, which would put a memory round trip
    // into every step.
    constexpr int PF = ND <= 2 ? 16 : 8;
    typedef typename AccVec<AccT, ND>::type AV;
    struct In { V c; V sOld; AV old; AV opp; };
    In ring[PF];
    const int kl = mine ? k0 : 0;
    auto fetch = [&](int t, In& in) {
        const int tf = min(t, n - 1);
        const size_t o = ((size_t)(y0 + tf * dy) * W + (x0 + tf * dx)) * D + kl;
        in.c = bt_ld(reinterpret_cast<const V*>(a.C + o));
        if (!first) in.old = bt_ld(reinterpret_cast<const AV*>(accInp + o));
        if (LAST && !a.firstGroup) in.sOld = bt_ld(reinterpret_cast<const V*>(Sin + o));
        static_assert(!SEL || (LAST && !MULTI), "selection belongs to the direction that completes S");
        if (OPP && LAST) in.opp = bt_ld(reinterpret_cast<const AV*>(static_cast<const AccT*>(a.opp) + o));
    };
#pragma unroll
    for (int u = 0; u < PF; u++) fetch(u, ring[u]);
    auto step = [&](int t, const In& in) {
        int16_t c[ND];
        int old[ND];
        __builtin_memcpy(c, &in.c, sizeof(V));
        if (first) {
#pragma unroll
            for (int j = 0; j < ND; j++) old[j] = 0;
        } else {
            AccT oldv[ND];
            __builtin_memcpy(oldv, &in.old, sizeof(AV));
#pragma unroll
            for (int j = 0; j < ND; j++) old[j] = (int)oldv[j];
        }
        const int delta = minPrev + P2;
        int Ln[ND], mn = BT_BIG;
        [[maybe_unused]] u32 Lw[NP];
        if constexpr (PK) {
            u32 cw[NP];
            __builtin_memcpy(cw, &in.c, sizeof(V));
            // n[i] = [candidate 2i - 1 | candidate 2i]: the lower neighbours of pair i and the upper neighbours of pair i - 1
            const u32 left = (u32)wave_shr1((int)Pp[NP - 1], (int)SENT2), right = (u32)wave_shl1((int)Pp[0], (int)SENT2);
            u32 nb[NP + 1];
            nb[0] = __builtin_amdgcn_alignbit(Pp[0], left, 16);
#pragma unroll
            for (int i = 1; i < NP; i++) nb[i] = __builtin_amdgcn_alignbit(Pp[i], Pp[i - 1], 16);
            nb[NP] = __builtin_amdgcn_alignbit(right, Pp[NP - 1], 16);
            const u32 d2 = (u32)delta | ((u32)delta << 16);          // 0 <= delta <= 32767
            u32 mn2 = SENT2;
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const u32 t = pk_add_sat_s16(pk_min_s16(nb[i], nb[i + 1]), P1P1);
                const u32 m = pk_min_s16(pk_min_s16(Pp[i], t), d2);
                u32 L = pk_add_s16(pk_sub_s16(m, d2), cw[i]);
                if (!mine) L = SENT2;
                Lw[i] = L;
                mn2 = pk_min_s16(mn2, L);
            }
            mn = min((int)(int16_t)(mn2 & 0xFFFFu), (int)mn2 >> 16);
            if constexpr (LAST) {
#pragma unroll
                for (int i = 0; i < NP; i++) { Ln[2 * i] = (int)(int16_t)(Lw[i] & 0xFFFFu); Ln[2 * i + 1] = (int)Lw[i] >> 16; }
            }
        } else {
            const int left = wave_shr1(Lp[ND - 1], 32767);
            const int right = wave_shl1(Lp[0], 32767);
#pragma unroll
            for (int j = 0; j < ND; j++) {
                const int lo = j > 0 ? Lp[j - 1] : left, hi = j < ND - 1 ? Lp[j + 1] : right;
                int L = (int)c[j] + min(min(Lp[j], lo + P1), min(hi + P1, delta)) - delta;
                L = (int)(int16_t)L;
                if (!mine) L = BT_BIG;
                Ln[j] = L;
                mn = min(mn, L);
            }
        }
        mn = wave_min_dpp(mn);
        if (mine) {
            const size_t o = ((size_t)(y0 + t * dy) * W + (x0 + t * dx)) * D + k0;
            if (LAST) {
                int16_t sv[ND];
                if (!a.firstGroup) __builtin_memcpy(sv, &in.sOld, sizeof(V));
#pragma unroll
                for (int j = 0; j < ND; j++) {
                    const int v = (a.firstGroup ? 0 : (int)sv[j]) + old[j] + Ln[j];
                    sv[j] = (int16_t)min(max(v, -32768), 32767);
                }
                if (OPP) {
                    AccT ov[ND];
                    __builtin_memcpy(ov, &in.opp, sizeof(AV));
#pragma unroll
                    for (int j = 0; j < ND; j++) sv[j] = (int16_t)min(max((int)sv[j] + (int)ov[j], -32768), 32767);
                }
                if constexpr (!SEL) {
                    V packed;
                    __builtin_memcpy(&packed, sv, sizeof(V));
                    bt_st(reinterpret_cast<V*>(Sout + o), packed);
                }
            } else if (MULTI) {
                int* ap = static_cast<int*>(a.acc) + o;
#pragma unroll
                for (int j = 0; j < ND; j++) atomicAdd(ap + j, Ln[j]);
            } else if constexpr (PK) {
                u32 aw[NP];
                if (!first) __builtin_memcpy(aw, &in.old, sizeof(AV));
#pragma unroll
                for (int i = 0; i < NP; i++) aw[i] = first ? Lw[i] : pk_add_s16(aw[i], Lw[i]);
                AV packedAcc;
                __builtin_memcpy(&packedAcc, aw, sizeof(AV));
                bt_st(reinterpret_cast<AV*>(accp + o), packedAcc);
            } else {
                AccT av[ND];
#pragma unroll
                for (int j = 0; j < ND; j++) av[j] = (AccT)(old[j] + Ln[j]);
                if constexpr (sizeof(AV) <= 16) {
                    AV packedAcc;
                    __builtin_memcpy(&packedAcc, av, sizeof(AV));
                    bt_st(reinterpret_cast<AV*>(accp + o), packedAcc);
                } else {
                    AccT* ap = accp + o;
#pragma unroll
                    for (int j = 0; j < ND; j++) ap[j] = av[j];
                }
            }
        }
        if constexpr (SEL) {
            // the pixel's sums once more, in every lane (lanes beyond D: excluded below)
            int sv[ND];
            {
                int16_t so[ND];
                if (!a.firstGroup) __builtin_memcpy(so, &in.sOld, sizeof(V));
                AccT ov[ND];
                if (OPP) __builtin_memcpy(ov, &in.opp, sizeof(AV));
#pragma unroll
                for (int j = 0; j < ND; j++) {
                    int v = (a.firstGroup ? 0 : (int)so[j]) + old[j] + Ln[j];
                    v = min(max(v, -32768), 32767);
                    if (OPP) v = min(max(v + (int)ov[j], -32768), 32767);
                    sv[j] = v;
                }
            }
            u32 best = 0xFFFFFFFFu;
            if (mine) {
#pragma unroll
                for (int j = 0; j < ND; j++) best = min(best, ((u32)(sv[j] + 32768) << 16) | (u32)(k0 + j));
            }
            best = (u32)wave_min_dpp((int)(best ^ 0x80000000u)) ^ 0x80000000u;
            const int bk = (int)(best & 0xFFFF), minS = (int)(best >> 16) - 32768;
            const int lim = minS * 100;
            bool bad = false;
            if (mine) {
#pragma unroll
                for (int j = 0; j < ND; j++) bad |= (__mul24(sv[j], a.q) < lim) && ((u32)(k0 + j - bk + 1) > 2u);
            }
            const bool rejected = __any(bad);
            // S(best - 1), S(best + 1): every lane lines up the neighbours of its own candidates (the edge ones come from the
            // neighbouring lanes through the cross-lane network), and the lane that owns the winner writes the pixel's record
            const int e = bk - k0;
            const int left = wave_shr1(sv[ND - 1], 0), right = wave_shl1(sv[0], 0);
            int sm = left, spl = ND > 1 ? sv[1] : right;
#pragma unroll
            for (int j = 1; j < ND; j++) {
                sm = e == j ? sv[j - 1] : sm;
                spl = e == j ? (j + 1 < ND ? sv[j + 1] : right) : spl;
            }
            if ((u32)e < (u32)ND) {
                const int yy = y0 + t * dy, x = x0 + t * dx + a.minX1;
                uint2 rec;
                rec.x = rejected ? 0xFFFFFFFFu : best;
                rec.y = ((u32)sm & 0xFFFFu) | ((u32)spl << 16);
                a.rec[(size_t)yy * a.W + x] = rec;
                if (!rejected) atomicMin(&a.key[(size_t)yy * a.W + (x - bk - a.minD)], ((best >> 16) << 16) | (u32)(0xFFFF - x));
            }
        }
        if constexpr (PK) {
#pragma unroll
            for (int i = 0; i < NP; i++) Pp[i] = Lw[i];
        } else {
#pragma unroll
            for (int j = 0; j < ND; j++) Lp[j] = mine ? Ln[j] : 32767;
        }
        minPrev = mn;
    };
    // whole groups of PF steps run without an exit in the middle (an early exit makes the compiler rotate the ring through
    // copies, which wait for the requests just issued); the last steps of a line use what the ring already holds
    int t = 0;
    for (; t + PF <= n; t += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const In cur = ring[u];
            fetch(t + u + PF, ring[u]);
            step(t + u, cur);
        }
    }
#pragma unroll
    for (int u = 0; u < PF; u++)
        if (t + u < n) step(t + u, ring[u]);
}

// The vertical window sum and the downward path in one pass: a wave owns a column and walks down the image with the running
// window sum of bt_vsum_kernel; the block cost of a pixel is complete the moment it is formed, so the wave runs the
// recursion of the direction (0, +1) on it at once and leaves both the block cost (for the other directions) and the
// group's first sum.  Saves that direction's launch: one read of the block cost and one of the group sum.
template <int ND, class AccT>
__global__ __launch_bounds__(256) void bt_vsum_down_kernel(const int16_t* __restrict__ in, BtGeom g, int16_t* __restrict__ out,
                                                           AccT* __restrict__ acc, int P1, int P2)
{
    const int lane = threadIdx.x & 63;
    const int x1 = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (x1 >= g.W1) return;
    const int D = g.D, H = g.H, h = g.bs / 2;
    const int k0 = lane * ND;
    const bool mine = k0 < D;
    const size_t rowN = (size_t)g.W1 * D, col = (size_t)x1 * D + (mine ? k0 : 0);
    const bool frozen = x1 == 0;                          // row 0's sum in every row (bt_vsum_kernel)
    typedef typename CVec<ND>::type V;
    typedef typename AccVec<AccT, ND>::type AV;
    auto ld = [&](int y) { return bt_ld<4>(reinterpret_cast<const V*>(in + (size_t)y * rowN + col)); };
    int s[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) s[j] = 0;
    for (int i = -h; i <= h; i++) {
        const V v = ld(min(max(i, 0), H - 1));
        int16_t e[ND];
        __builtin_memcpy(e, &v, sizeof(V));
#pragma unroll
        for (int j = 0; j < ND; j++) s[j] += (int)e[j];
    }
    int Lp[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) Lp[j] = mine ? 0 : 32767;
    int minPrev = 0;
    auto step = [&](int y) {
        int16_t c[ND];
#pragma unroll
        for (int j = 0; j < ND; j++) c[j] = (int16_t)s[j];            // the 16-bit wrap of the stored block cost
        const int left = wave_shr1(Lp[ND - 1], 32767), right = wave_shl1(Lp[0], 32767);
        const int delta = minPrev + P2;
        int Ln[ND], mn = BT_BIG;
#pragma unroll
        for (int j = 0; j < ND; j++) {
            const int lo = j > 0 ? Lp[j - 1] : left, hi = j < ND - 1 ? Lp[j + 1] : right;
            int L = (int)c[j] + min(min(Lp[j], lo + P1), min(hi + P1, delta)) - delta;
            L = (int)(int16_t)L;
            if (!mine) L = BT_BIG;
            Ln[j] = L;
            mn = min(mn, L);
        }
        mn = wave_min_dpp(mn);
        if (mine) {
            const size_t o = (size_t)y * rowN + col;
            V pc;
            __builtin_memcpy(&pc, c, sizeof(V));
            bt_st<4>(reinterpret_cast<V*>(out + o), pc);
            AccT av[ND];
#pragma unroll
            for (int j = 0; j < ND; j++) av[j] = (AccT)Ln[j];
            if constexpr (sizeof(AV) <= 16) {
                AV pa;
                __builtin_memcpy(&pa, av, sizeof(AV));
                bt_st(reinterpret_cast<AV*>(acc + o), pa);
            } else {
#pragma unroll
                for (int j = 0; j < ND; j++) acc[o + j] = av[j];
            }
        }
#pragma unroll
        for (int j = 0; j < ND; j++) Lp[j] = mine ? Ln[j] : 32767;
        minPrev = mn;
    };
    // the rows entering and leaving the window of step y (y >= 1), requested PF steps ahead; requests beyond the last
    // row repeat it (whole groups of steps without an exit, as in bt_path_kernel)
    constexpr int PF = ND <= 2 ? 8 : 4;
    struct In { V a, b; };
    In ring[PF];
    auto fetch = [&](int y, In& in) {
        const int yy = min(y, H - 1);
        in.a = ld(min(yy + h, H - 1));
        in.b = ld(max(yy - h - 1, 0));
    };
    auto advance = [&](const In& in) {
        int16_t ea[ND], eb[ND];
        __builtin_memcpy(ea, &in.a, sizeof(V));
        __builtin_memcpy(eb, &in.b, sizeof(V));
#pragma unroll
        for (int j = 0; j < ND; j++) s[j] += frozen ? 0 : (int)ea[j] - (int)eb[j];
    };
#pragma unroll
    for (int u = 0; u < PF; u++) fetch(1 + u, ring[u]);
    step(0);
    int y = 1;
    for (; y + PF <= H; y += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const In cur = ring[u];
            fetch(y + u + PF, ring[u]);
            advance(cur);
            step(y + u);
        }
    }
#pragma unroll
    for (int u = 0; u < PF; u++)
        if (y + u < H) { advance(ring[u]); step(y + u); }
}

// S = saturate_cast<short>(S + acc)   (first group: S starts at zero)
__global__ __launch_bounds__(256) void bt_saturate_kernel(const int* __restrict__ acc, size_t n, int first, int16_t* __restrict__ S)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int v = (first ? 0 : (int)S[i]) + acc[i];
    S[i] = (int16_t)min(max(v, -32768), 32767);
}

struct SelArgs {
    const int16_t* S; int H, W, W1, D, minD, minX1, uniq, maxDiff;
    int16_t* disp;
};
// One workgroup per image row.  First the waves walk the row's pixels (one pixel per wave and step): first-minimum WTA,
// uniqueness, disp2 election (minimum S, ties to the larger x: the library walks right to left with a strict '>').  Only what
// needs the whole wave stays in that serial loop; the x16 parabola of every pixel (an integer division) is done afterwards by
// one lane per pixel, reading the two neighbours of the minimum back from S (the row was just read, so from the cache).  Then
// the disp12MaxDiff check.  Columns left of minX1 stay invalid.
#define BT_SEL_THREADS 1024      // sixteen waves per image row
#ifndef BT_SEL_AHEAD
#define BT_SEL_AHEAD 2
#endif
template <int ND>
__global__ __launch_bounds__(BT_SEL_THREADS) void bt_select_kernel(SelArgs a)
{
    constexpr int NWAVES = BT_SEL_THREADS / 64;
    extern __shared__ u32 lds[];
    u32* key = lds;                  // [W] (S + 32768) << 16 | (0xFFFF - x)
    u32* bst = lds + a.W;            // [W] winner of the pixel, (S + 32768) << 16 | k; later its disparity x16
    const int y = blockIdx.x, W = a.W, D = a.D;
    const int INVALID = (a.minD - 1) * 16;
    for (int i = threadIdx.x; i < W; i += BT_SEL_THREADS) { key[i] = 0xFFFFFFFFu; bst[i] = 0xFFFFFFFFu; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // lane l holds the ND consecutive sums l*ND .. l*ND+ND-1 of a pixel (one vector load); D is a multiple of 16 >= ND
    typedef typename CVec<ND>::type V;
    const int k0 = lane * ND;
    const bool mine = k0 < D;
    const int q = 100 - a.uniq;      // the caller keeps |q| below 2^15: 24-bit multiplies below
    auto decide = [&](int x1, const V& cur) {
        int16_t sv[ND];
        __builtin_memcpy(sv, &cur, sizeof(V));
        u32 best = 0xFFFFFFFFu;
        if (mine) {
#pragma unroll
            for (int j = 0; j < ND; j++) best = min(best, ((u32)((int)sv[j] + 32768) << 16) | (u32)(k0 + j));
        }
        best = (u32)wave_min_dpp((int)(best ^ 0x80000000u)) ^ 0x80000000u;      // unsigned minimum through the signed reduction
        const int bk = (int)(best & 0xFFFF), minS = (int)(best >> 16) - 32768;
        const int lim = minS * 100;
        bool bad = false;
        if (mine) {
#pragma unroll
            for (int j = 0; j < ND; j++) bad |= (__mul24((int)sv[j], q) < lim) && ((u32)(k0 + j - bk + 1) > 2u);
        }
        if (__any(bad)) return;
        if (lane == 0) {
            const int x = x1 + a.minX1, x2 = x - bk - a.minD;
            atomicMin(&key[x2], ((best >> 16) << 16) | (u32)(0xFFFF - x));
            bst[x] = best;
        }
    };
    // the sums of the wave's next BT_SEL_AHEAD pixels are in flight while one pixel is decided (requests are unconditional:
    // lanes beyond D ask for the first lane's values, steps beyond the row's end for its last pixel again)
    if (a.W1 > 0) {
        const int16_t* const srow = a.S + (size_t)y * a.W1 * D + (mine ? k0 : 0);
        auto request = [&](int i) { return bt_ld<16>(reinterpret_cast<const V*>(srow + (size_t)min(wv + i * NWAVES, a.W1 - 1) * D)); };
        const int n = wv < a.W1 ? (a.W1 - wv + NWAVES - 1) / NWAVES : 0;      // pixels of this wave: x1 = wv + i * NWAVES
        V ring[BT_SEL_AHEAD];
#pragma unroll
        for (int u = 0; u < BT_SEL_AHEAD; u++) ring[u] = request(u);
        // whole groups of BT_SEL_AHEAD pixels without an exit in the middle, so the ring keeps its registers
        int i = 0;
        for (; i + BT_SEL_AHEAD <= n; i += BT_SEL_AHEAD) {
#pragma unroll
            for (int u = 0; u < BT_SEL_AHEAD; u++) {
                const V cur = ring[u];
                ring[u] = request(i + u + BT_SEL_AHEAD);
                decide(wv + (i + u) * NWAVES, cur);
            }
        }
#pragma unroll
        for (int u = 0; u < BT_SEL_AHEAD; u++)
            if (i + u < n) decide(wv + (i + u) * NWAVES, ring[u]);
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += BT_SEL_THREADS) {
        const u32 b = bst[x];
        int v = INVALID;
        if (b != 0xFFFFFFFFu) {
            const int bk = (int)(b & 0xFFFF), s0 = (int)(b >> 16) - 32768;
            int d = bk * 16;
            if (bk > 0 && bk < D - 1) {
                const int16_t* sp = a.S + ((size_t)y * a.W1 + (x - a.minX1)) * D + bk;
                const int sm = sp[-1], spl = sp[1];
                const int den = max(sm + spl - 2 * s0, 1);
                d += ((sm - spl) * 16 + den) / (den * 2);
            }
            v = d + a.minD * 16;
        }
        bst[x] = (u32)(int)(int16_t)v;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += BT_SEL_THREADS) {
        int v = (int)bst[x];
        if (v != INVALID && x >= a.minX1) {
            auto inconsistent = [&](int xx, int dd) -> bool {
                if (xx < 0 || xx >= W) return false;
                const u32 kk = key[xx];
                const int d2 = (kk == 0xFFFFFFFFu) ? INVALID : (0xFFFF - (int)(kk & 0xFFFF)) - xx;
                return d2 >= a.minD && abs(d2 - dd) > a.maxDiff;
            };
            const int _d = v >> 4, d_ = (v + 15) >> 4;
            if (inconsistent(x - _d, _d) && inconsistent(x - d_, d_)) v = INVALID;
        }
        a.disp[(size_t)y * W + x] = (int16_t)v;
    }
}

// The map from the records of a selecting last direction (SEL): the x16 parabola through the winner's neighbours, then the
// disp12MaxDiff check against the right view's candidates (bt_select_kernel's last two passes).  One workgroup per image row.
__global__ __launch_bounds__(256) void bt_finish_kernel(SelArgs a, const uint2* __restrict__ rec, const u32* __restrict__ key)
{
    extern __shared__ u32 lds[];
    int* dsp = reinterpret_cast<int*>(lds);      // [W] disparity x16 of the row's pixels
    const int y = blockIdx.x, W = a.W, D = a.D;
    const int INVALID = (a.minD - 1) * 16;
    for (int x = threadIdx.x; x < W; x += 256) {
        int v = INVALID;
        if (x >= a.minX1) {
            const uint2 r = rec[(size_t)y * W + x];
            if (r.x != 0xFFFFFFFFu) {
                const int bk = (int)(r.x & 0xFFFF), s0 = (int)(r.x >> 16) - 32768;
                int d = bk * 16;
                if (bk > 0 && bk < D - 1) {
                    const int sm = (int)(int16_t)(r.y & 0xFFFFu), spl = (int)(int16_t)(r.y >> 16);
                    const int den = max(sm + spl - 2 * s0, 1);
                    d += ((sm - spl) * 16 + den) / (den * 2);
                }
                v = (int)(int16_t)(d + a.minD * 16);
            }
        }
        dsp[x] = v;
    }
    __syncthreads();
    const u32* krow = key + (size_t)y * W;
    for (int x = threadIdx.x; x < W; x += 256) {
        int v = dsp[x];
        if (v != INVALID && x >= a.minX1) {
            auto inconsistent = [&](int xx, int dd) -> bool {
                if (xx < 0 || xx >= W) return false;
                const u32 kk = krow[xx];
                const int d2 = (kk == 0xFFFFFFFFu) ? INVALID : (0xFFFF - (int)(kk & 0xFFFF)) - xx;
                return d2 >= a.minD && abs(d2 - dd) > a.maxDiff;
            };
            const int _d = v >> 4, d_ = (v + 15) >> 4;
            if (inconsistent(x - _d, _d) && inconsistent(x - d_, d_)) v = INVALID;
        }
        a.disp[(size_t)y * W + x] = (int16_t)v;
    }
}

// cv::medianBlur(src, dst, 3) on CV_16SC1: replicated borders
__global__ __launch_bounds__(256) void median3_s16_kernel(const int16_t* __restrict__ src, int H, int W, int16_t* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int xs[3] = {x > 0 ? x - 1 : 0, x, x < W - 1 ? x + 1 : W - 1};
    const int ys[3] = {y > 0 ? y - 1 : 0, y, y < H - 1 ? y + 1 : H - 1};
    int v[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) v[i * 3 + j] = src[(size_t)ys[i] * W + xs[j]];
    // median of nine by a fixed exchange network
#define MD_SORT2(a, b) { const int lo__ = min(v[a], v[b]), hi__ = max(v[a], v[b]); v[a] = lo__; v[b] = hi__; }
    MD_SORT2(1, 2) MD_SORT2(4, 5) MD_SORT2(7, 8) MD_SORT2(0, 1) MD_SORT2(3, 4) MD_SORT2(6, 7)
    MD_SORT2(1, 2) MD_SORT2(4, 5) MD_SORT2(7, 8) MD_SORT2(0, 3) MD_SORT2(5, 8) MD_SORT2(4, 7)
    MD_SORT2(3, 6) MD_SORT2(1, 4) MD_SORT2(2, 5) MD_SORT2(4, 7) MD_SORT2(4, 2) MD_SORT2(6, 4)
    MD_SORT2(4, 2)
#undef MD_SORT2
    dst[(size_t)y * W + x] = (int16_t)v[4];
}

// ---- filterSpeckles: 4-connected regions of pixels != newVal whose neighbouring values differ by
// at most maxDiff; regions of at most maxSize pixels become newVal.  Union-find over pixel indices.
__device__ __forceinline__ int uf_find(const int* parent, int a)
{
    int p = __hip_atomic_load(parent + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != a) { a = p; p = __hip_atomic_load(parent + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return a;
}
__device__ __forceinline__ void uf_union(int* parent, int a, int b)
{
    for (int it = 0; it < (1 << 24); it++) {       // bounded; each round strictly lowers a root
        a = uf_find(parent, a); b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(parent + b, a);
        if (old == b) return;
        b = old;
    }
}
// Horizontal runs inside a wave's 64 columns are linked without atomics: every pixel starts out pointing at the first
// pixel of its run (ballot of the run breaks + count-leading-zeros).
__global__ __launch_bounds__(256) void speckle_init_kernel(const int16_t* __restrict__ img, int W, int newVal, int maxDiff, int* parent, int* count)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, lane = threadIdx.x & 63;
    const bool valid = x < W;
    const int i = y * W + x;
    const int v = valid ? (int)img[i] : newVal, l = (valid && x > 0) ? (int)img[i - 1] : newVal;
    const bool joined = lane != 0 && v != newVal && l != newVal && abs(v - l) <= maxDiff;
    const unsigned long long breaks = __ballot(!joined);                       // bit 0 is always set
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1);
    const int start = 63 - __clzll((long long)(breaks & upto));
    if (valid) { parent[i] = i - (lane - start); count[i] = 0; }
}
// The remaining links: across wave boundaries of a row, and down to the next row.  A vertical link is skipped when the
// pixel on the left and the one below it are linked to this pair and to each other: that neighbour (or one further
// left, by induction) makes the same union.
__global__ __launch_bounds__(256) void speckle_link_kernel(const int16_t* __restrict__ img, int H, int W, int newVal, int maxDiff, int* parent)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, lane = threadIdx.x & 63;
    if (x >= W) return;
    const int i = y * W + x, v = img[i];
    if (v == newVal) return;
    const int l = x > 0 ? (int)img[i - 1] : newVal;
    const bool joinedL = l != newVal && abs(v - l) <= maxDiff;
    if (lane == 0 && joinedL) uf_union(parent, i - 1, i);
    if (y + 1 < H) {
        const int q = img[i + W];
        if (q != newVal && abs(v - q) <= maxDiff) {
            bool redundant = false;
            if (lane != 0 && joinedL) {
                const int lq = img[i - 1 + W];
                redundant = lq != newVal && abs(q - lq) <= maxDiff && abs(l - lq) <= maxDiff;
            }
            if (!redundant) uf_union(parent, i, i + W);
        }
    }
}
// Only "at most maxSize pixels or more" is needed, so a run stops adding once its region's count is beyond maxSize: a large
// region is thousands of runs whose atomics would otherwise queue on one word.
__global__ __launch_bounds__(256) void speckle_count_kernel(const int16_t* __restrict__ img, int n, int newVal, int maxSize, int* parent, int* count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    int r = -1;
    if (i < n && img[i] != newVal) {
        r = uf_find(parent, i);
        parent[i] = r;                   // every write moves a node closer to its root, so concurrent finds stay valid
    }
    // consecutive pixels mostly share a root: one atomic per run of equal roots inside the wave
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(r, 1, 64);
    const bool head = lane == 0 || prev != r;
    const unsigned long long heads = __ballot(head);
    if (head && r >= 0) {
        const unsigned long long after = lane == 63 ? 0ull : (heads >> (lane + 1));
        const int len = after ? __ffsll((long long)after) : 64 - lane;
        if (__hip_atomic_load(count + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= maxSize) atomicAdd(count + r, len);
    }
}
__global__ __launch_bounds__(256) void speckle_apply_kernel(int16_t* __restrict__ img, int n, int newVal, int maxSize, const int* __restrict__ parent,
                                                            const int* __restrict__ count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || img[i] == newVal) return;
    if (count[uf_find(parent, i)] <= maxSize) img[i] = (int16_t)newVal;
}

static inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

int md_median3_s16(hipStream_t s, const int16_t* src, int H, int W, int16_t* dst)
{
    median3_s16_kernel<<<dim3((W + 255) / 256, H), 256, 0, s>>>(src, H, W, dst);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

size_t md_speckle_scratch_bytes(int H, int W) { return md_align((size_t)H * W * 4, 256) * 2; }

int md_filter_speckles_s16(hipStream_t s, int16_t* img, int H, int W, int newVal, int maxSize, int maxDiff, void* scratch)
{
    const int n = H * W;
    int* parent = (int*)scratch;
    int* count = (int*)((u8*)scratch + md_align((size_t)n * 4, 256));
    speckle_init_kernel<<<dim3((W + 255) / 256, H), 256, 0, s>>>(img, W, newVal, maxDiff, parent, count);
    speckle_link_kernel<<<dim3((W + 255) / 256, H), 256, 0, s>>>(img, H, W, newVal, maxDiff, parent);
    speckle_count_kernel<<<nblk(n), 256, 0, s>>>(img, n, newVal, maxSize, parent, count);
    speckle_apply_kernel<<<nblk(n), 256, 0, s>>>(img, n, newVal, maxSize, parent, count);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

size_t md_stereosgbm_bytes(int H, int W, int cn, const mdisp_stereosgbm_params* p)
{
    const int D = p->numDisparities, W1 = W - p->minDisparity - D;
    const size_t N = (size_t)H * W, V = W1 > 0 ? (size_t)H * W1 * D : 0;
    return md_align(N * 2 * cn, 256) * 2 + md_align(V * 2, 256) * 3 + md_align(V * 4, 256) + md_align(N * 2, 256) * 2 +
           md_speckle_scratch_bytes(H, W) + md_align(N * 4, 256) + md_align(N * 8, 256) + 4096;
}

static int validate(int H, int W, int cn, const mdisp_stereosgbm_params* p)
{
    MD_REQUIRE(p, MDISP_E_BADARG, "null StereoSGBM parameters");
    MD_REQUIRE(cn == 1 || cn == 3, MDISP_E_BADARG, "StereoSGBM takes 1- or 3-channel 8-bit images");
    MD_REQUIRE(p->numDisparities > 0 && p->numDisparities % 16 == 0 && p->numDisparities <= 512, MDISP_E_BADARG,
               "numDisparities must be a positive multiple of 16 (at most 512)");
    MD_REQUIRE(p->minDisparity >= 0, MDISP_E_UNSUPPORTED, "negative minDisparity is not supported");
    MD_REQUIRE(p->mode == MDISP_SGBM_MODE_SGBM || p->mode == MDISP_SGBM_MODE_HH, MDISP_E_UNSUPPORTED,
               "StereoSGBM mode %d is not built (MODE_SGBM and MODE_HH are)", p->mode);
    MD_REQUIRE((p->blockSize <= 0 || p->blockSize % 2 == 1) && p->blockSize <= 31, MDISP_E_BADARG, "blockSize must be odd (at most 31)");
    MD_REQUIRE((size_t)H * W < ((size_t)1 << 30), MDISP_E_RANGE, "image too large");
    // S * (100 - uniquenessRatio) is a 32-bit product in the library (S is a 16-bit sum); it is evaluated with 24-bit operands here
    MD_REQUIRE(p->uniquenessRatio <= 32767, MDISP_E_RANGE, "uniquenessRatio above 32767 is not supported");
    return MDISP_OK;
}

// dL / dR: interleaved 8-bit images on the device; dDisp: H*W int16; ws: md_stereosgbm_bytes()
int md_stereosgbm_run(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, int H, int W, int cn,
                      const mdisp_stereosgbm_params* p, int16_t* dDisp, void* ws, int16_t* dCost, int16_t* dRaw)
{
    MD_TRY(validate(H, W, cn, p));
    BtGeom g;
    g.H = H; g.W = W; g.cn = cn; g.D = p->numDisparities; g.minD = p->minDisparity;
    g.minX1 = g.minD + g.D; g.W1 = W - g.minX1; g.N = (size_t)H * W;
    g.ftzero = (p->preFilterCap > 15 ? p->preFilterCap : 15) | 1;
    g.bs = p->blockSize > 0 ? p->blockSize : 5;
    const int P1 = p->P1 > 0 ? p->P1 : 2;
    int P2 = p->P2 > 0 ? p->P2 : 5;
    if (P2 < P1 + 1) P2 = P1 + 1;
    const int INVALID = (g.minD - 1) * 16;
    const size_t N = g.N, V = g.W1 > 0 ? (size_t)H * g.W1 * g.D : 0;
    u8* base = (u8*)ws;
    auto take = [&](size_t bytes) { u8* r = base; base += md_align(bytes, 256); return r; };
    u8* pl1 = take(N * 2 * cn); u8* pl2 = take(N * 2 * cn);
    int16_t* v0 = (int16_t*)take(V * 2); int16_t* v1 = (int16_t*)take(V * 2); int16_t* S = (int16_t*)take(V * 2);
    int* acc = (int*)take(V * 4);
    int16_t* raw = (int16_t*)take(N * 2); int16_t* med = (int16_t*)take(N * 2);
    void* spk = take(md_speckle_scratch_bytes(H, W));
    u32* selKey = (u32*)take(N * 4); uint2* selRec = (uint2*)take(N * 8);
    if (g.W1 <= 0) {
        // every column is left of minDisparity + numDisparities: the library returns an all-invalid map
        // (the median and speckle stages keep it constant)
        SelArgs sa{};
        sa.S = nullptr; sa.H = H; sa.W = W; sa.W1 = 0; sa.D = g.D; sa.minD = g.minD; sa.minX1 = W; sa.uniq = 0; sa.maxDiff = 1; sa.disp = dDisp;
        bt_select_kernel<1><<<H, BT_SEL_THREADS, (size_t)W * 8 + 16, s>>>(sa);
        MD_HIP(hipGetLastError());
        if (dRaw) MD_HIP(hipMemcpyAsync(dRaw, dDisp, N * 2, hipMemcpyDeviceToDevice, s));
        return MDISP_OK;
    }
    MD_REQUIRE((size_t)W * 8 + 16 <= 64 * 1024, MDISP_E_RANGE, "image wider than 8190 columns");
    const int nd = g.D <= 64 ? 1 : g.D <= 128 ? 2 : g.D <= 256 ? 4 : 8;
    // a path value lies in [-P2, Cmax]; three of them fit int16 when the block cost cannot exceed 10922 (then it cannot
    // have wrapped either): block cost <= bs^2 * cn * (2 * ftzero + 255 / 4)
    const long cmax = (long)g.bs * g.bs * cn * (2 * g.ftzero + 63);
    const bool acc16 = 3 * std::max<long>(cmax, P2) <= 32767;
    const int seqTune = md_tune(MDISP_TUNE_SGBM_BT_SEQUENTIAL);
    const bool largeSchedule = !(V <= BT_MULTI_MAX_UNITS && seqTune <= 0);      // one launch per direction
    // large problems: the vertical window sum also runs the downward path (MDISP_TUNE_SGBM_BT_SEQUENTIAL = 3 keeps them apart)
    const size_t rowLdsEarly = (size_t)2 * cn * ((BT_SEG + g.bs) + (BT_SEG + g.bs + 255)) * sizeof(u32) + (size_t)(g.bs + 1) * 256 * sizeof(int16_t);
    const bool fuseDown = largeSchedule && seqTune != 3 && rowLdsEarly <= 60 * 1024;
    const dim3 gi((W + 255) / 256, H), gv(nblk((size_t)g.W1 * g.D), H);
    bt_planes_kernel<<<gi, 256, 0, s>>>(dL, pitch, g, pl1);
    bt_planes_kernel<<<gi, 256, 0, s>>>(dR, pitch, g, pl2);
    // plane rows + the per-lane ring of pixel costs (bs + 1 slots of 256 int16)
    const size_t rowLds = (size_t)2 * cn * ((BT_SEG + g.bs) + (BT_SEG + g.bs + 255)) * sizeof(u32) + (size_t)(g.bs + 1) * 256 * sizeof(int16_t);
    if (rowLds <= 60 * 1024) {
        const dim3 gh((g.D + 255) / 256, H, (g.W1 + BT_SEG - 1) / BT_SEG);
        if (cn == 1) bt_cost_hsum_kernel<1><<<gh, 256, rowLds, s>>>(pl1, pl2, g, v1);
        else if (cn == 3) bt_cost_hsum_kernel<3><<<gh, 256, rowLds, s>>>(pl1, pl2, g, v1);
        else bt_cost_hsum_kernel<0><<<gh, 256, rowLds, s>>>(pl1, pl2, g, v1);
        if (!fuseDown) bt_vsum_kernel<<<dim3(nblk((size_t)g.W1 * g.D / 4), (H + BT_VBAND - 1) / BT_VBAND), 256, 0, s>>>(v1, g, v0);      // v0 = block cost C
    } else {
        bt_cost_kernel<<<gv, 256, 0, s>>>(pl1, pl2, g, v0);
        bt_window_kernel<false><<<gv, 256, 0, s>>>(v0, g, v1);
        bt_window_kernel<true><<<gv, 256, 0, s>>>(v1, g, v0);
    }
    MD_HIP(hipGetLastError());
    if (dCost && !fuseDown) MD_HIP(hipMemcpyAsync(dCost, v0, V * 2, hipMemcpyDeviceToDevice, s));
    static const int DIRS[8][2] = {{+1, 0}, {+1, +1}, {0, +1}, {-1, +1}, {-1, 0}, {-1, -1}, {0, -1}, {+1, -1}};
    PathArgs pa{};
    pa.C = v0; pa.acc = acc; pa.accIn = acc; pa.opp = v1; pa.H = H; pa.W1 = g.W1; pa.D = g.D; pa.P1 = P1; pa.P2 = P2; pa.S = S; pa.firstGroup = 1;
    if (fuseDown) {
        const dim3 gf((g.W1 + 3) / 4);
#define BT_FUSED(NDV)                                                                                          \
        do {                                                                                                   \
            if (acc16) bt_vsum_down_kernel<NDV, int16_t><<<gf, 256, 0, s>>>(v1, g, v0, (int16_t*)acc, P1, P2);   \
            else bt_vsum_down_kernel<NDV, int><<<gf, 256, 0, s>>>(v1, g, v0, acc, P1, P2);                       \
        } while (0)
        if (nd == 1) BT_FUSED(1); else if (nd == 2) BT_FUSED(2); else if (nd == 4) BT_FUSED(4); else BT_FUSED(8);
#undef BT_FUSED
        MD_HIP(hipGetLastError());
        if (dCost) MD_HIP(hipMemcpyAsync(dCost, v0, V * 2, hipMemcpyDeviceToDevice, s));
    }
    const int uniq = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    pa.key = selKey; pa.rec = selRec; pa.W = W; pa.minX1 = g.minX1; pa.minD = g.minD; pa.q = 100 - uniq;
    bool selected = false;        // the last direction has selected: records instead of S
    // (MDISP_TUNE_SGBM_BT_SEQUENTIAL = 2 keeps the separate selection kernel on this schedule: the tests run both)
    const bool selInLast = md_tune(MDISP_TUNE_SGBM_BT_SEQUENTIAL) != 2;
    for (int grp = 0; grp < 2; grp++) {
        const int ndir = (grp == 1 && p->mode == MDISP_SGBM_MODE_SGBM) ? 1 : 4;
        pa.firstGroup = grp == 0;
        if (ndir == 4 && V <= BT_MULTI_MAX_UNITS && md_tune(MDISP_TUNE_SGBM_BT_SEQUENTIAL) <= 0) {
            // small problem: the four directions of the group in one launch
            int total = 0;
            for (int i = 0; i < 4; i++) {
                pa.mdx[i] = DIRS[grp * 4 + i][0]; pa.mdy[i] = DIRS[grp * 4 + i][1];
                pa.mfirst[i] = total;
                total += pa.mdy[i] == 0 ? H : (pa.mdx[i] == 0 ? g.W1 : g.W1 + H - 1);
            }
            pa.mfirst[4] = total;
            MD_HIP(hipMemsetAsync(acc, 0, V * 4, s));
            const dim3 gm((total + 3) / 4);
            if (nd == 1) bt_path_kernel<1, true, false, int, true><<<gm, 256, 0, s>>>(pa, 0, 0, total);
            else if (nd == 2) bt_path_kernel<2, true, false, int, true><<<gm, 256, 0, s>>>(pa, 0, 0, total);
            else if (nd == 4) bt_path_kernel<4, true, false, int, true><<<gm, 256, 0, s>>>(pa, 0, 0, total);
            else bt_path_kernel<8, true, false, int, true><<<gm, 256, 0, s>>>(pa, 0, 0, total);
            bt_saturate_kernel<<<nblk(V), 256, 0, s>>>(acc, V, grp == 0, S);
            continue;
        }
        if (ndir == 1 && V <= BT_MULTI_MAX_UNITS && md_tune(MDISP_TUNE_SGBM_BT_SEQUENTIAL) <= 0) {
            // small problem, single direction: its serial chain is shorter without the read-modify-write of S
            const int dx = DIRS[grp * 4][0], dy = DIRS[grp * 4][1];
            const int nlines = dy == 0 ? H : (dx == 0 ? g.W1 : g.W1 + H - 1);
            const dim3 gp((nlines + 3) / 4);
            if (nd == 1) bt_path_kernel<1, true, false, int><<<gp, 256, 0, s>>>(pa, dx, dy, nlines);
            else if (nd == 2) bt_path_kernel<2, true, false, int><<<gp, 256, 0, s>>>(pa, dx, dy, nlines);
            else if (nd == 4) bt_path_kernel<4, true, false, int><<<gp, 256, 0, s>>>(pa, dx, dy, nlines);
            else bt_path_kernel<8, true, false, int><<<gp, 256, 0, s>>>(pa, dx, dy, nlines);
            bt_saturate_kernel<<<nblk(V), 256, 0, s>>>(acc, V, grp == 0, S);
            continue;
        }
        // 16-bit sums: the first group's -> launch runs <- beside it (the first direction of the second group) and keeps its
        // values in the buffer the horizontal window sums no longer need
        const bool opp = acc16;
        for (int i = 0; i < ndir; i++) {
            const int dx = DIRS[grp * 4 + i][0], dy = DIRS[grp * 4 + i][1];
            const int nlines = dy == 0 ? H : (dx == 0 ? g.W1 : g.W1 + H - 1);
            dim3 gp((nlines + 3) / 4);
            if (opp && grp == 1 && i == 0) continue;                     // <- ran beside ->
            const bool fused0 = fuseDown && grp == 0;                    // the downward path ran with the vertical window sum: it was the group's first
            if (fused0 && i == 2) continue;
            const bool oppFirst = opp && grp == 0 && i == 0;
            const bool oppLast = opp && grp == 0 && i == ndir - 1 && p->mode == MDISP_SGBM_MODE_SGBM;      // MODE_SGBM: <- is all of the second group
            pa.accIn = (opp && grp == 1 && i == 1) ? (const void*)v1 : (const void*)acc;
            if (oppFirst) gp = dim3((2 * nlines + 3) / 4);
            // the direction that completes S (MODE_SGBM with <- folded in: the first group's last; MODE_HH: the second group's)
            const bool sel = selInLast && ndir == 4 && i == ndir - 1 && (oppLast || (grp == 1 && p->mode == MDISP_SGBM_MODE_HH));
            if (sel) { MD_HIP(hipMemsetAsync(selKey, 0xFF, N * 4, s)); selected = true; }
#define BT_PATH(NDV, ACC, PKV)                                                                                 \
            do {                                                                                               \
                if (oppFirst && fused0) bt_path_kernel<NDV, false, false, ACC, false, true, false, PKV><<<gp, 256, 0, s>>>(pa, dx, dy, nlines); \
                else if (oppFirst) bt_path_kernel<NDV, true, false, ACC, false, true, false, PKV><<<gp, 256, 0, s>>>(pa, dx, dy, nlines);          \
                else if (oppLast && sel) bt_path_kernel<NDV, false, true, ACC, false, true, true, PKV><<<gp, 256, 0, s>>>(pa, dx, dy, nlines); \
                else if (sel) bt_path_kernel<NDV, false, true, ACC, false, false, true, PKV><<<gp, 256, 0, s>>>(pa, dx, dy, nlines);   \
                else if (oppLast) bt_path_kernel<NDV, false, true, ACC, false, true, false, PKV><<<gp, 256, 0, s>>>(pa, dx, dy, nlines);      \
                else if (ndir == 1) bt_path_kernel<NDV, true, true, ACC, false, false, false, PKV><<<gp, 256, 0, s>>>(pa, dx, dy, nlines);        \
                else if (i == 0 && !fused0) bt_path_kernel<NDV, true, false, ACC, false, false, false, PKV><<<gp, 256, 0, s>>>(pa, dx, dy, nlines);     \
                else if (i == ndir - 1) bt_path_kernel<NDV, false, true, ACC, false, false, false, PKV><<<gp, 256, 0, s>>>(pa, dx, dy, nlines); \
                else bt_path_kernel<NDV, false, false, ACC, false, false, false, PKV><<<gp, 256, 0, s>>>(pa, dx, dy, nlines);                \
            } while (0)
            // 16-bit group sums: the recursion on packed pairs (two candidates per lane at least)
            if (acc16) { if (nd == 1) BT_PATH(1, int16_t, false); else if (nd == 2) BT_PATH(2, int16_t, true); else if (nd == 4) BT_PATH(4, int16_t, true); else BT_PATH(8, int16_t, true); }
            else { if (nd == 1) BT_PATH(1, int, false); else if (nd == 2) BT_PATH(2, int, false); else if (nd == 4) BT_PATH(4, int, false); else BT_PATH(8, int, false); }
#undef BT_PATH
        }
    }
    MD_HIP(hipGetLastError());
    SelArgs sa;
    sa.S = S; sa.H = H; sa.W = W; sa.W1 = g.W1; sa.D = g.D; sa.minD = g.minD; sa.minX1 = g.minX1;
    sa.uniq = uniq;
    sa.maxDiff = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    sa.disp = raw;
    if (selected) bt_finish_kernel<<<H, 256, (size_t)W * 4, s>>>(sa, selRec, selKey);
    else if (nd == 1) bt_select_kernel<1><<<H, BT_SEL_THREADS, (size_t)W * 8 + 16, s>>>(sa);
    else if (nd == 2) bt_select_kernel<2><<<H, BT_SEL_THREADS, (size_t)W * 8 + 16, s>>>(sa);
    else if (nd == 4) bt_select_kernel<4><<<H, BT_SEL_THREADS, (size_t)W * 8 + 16, s>>>(sa);
    else bt_select_kernel<8><<<H, BT_SEL_THREADS, (size_t)W * 8 + 16, s>>>(sa);
    MD_HIP(hipGetLastError());
    if (dRaw) MD_HIP(hipMemcpyAsync(dRaw, raw, N * 2, hipMemcpyDeviceToDevice, s));
    const bool speckle = p->speckleWindowSize > 0;
    MD_TRY(md_median3_s16(s, raw, H, W, speckle ? med : dDisp));
    if (speckle) {
        MD_TRY(md_filter_speckles_s16(s, med, H, W, INVALID, p->speckleWindowSize, 16 * p->speckleRange, spk));
        MD_HIP(hipMemcpyAsync(dDisp, med, N * 2, hipMemcpyDeviceToDevice, s));
    }
    return MDISP_OK;
}
