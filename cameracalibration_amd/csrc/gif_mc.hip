// Guided filter with a multi-channel guide and the matcher built on it, for gfx950.
//
// Replaces stereomatch_1::getCostSAD_d, getGuidedFilter (3- / 6-channel guide, with multiChl_to_oneChl_mul) and
// computeAdaptiveWeight_GuidedF of cameraCalibration/methods/method_StereoMatching.cpp:3288-3349, 3573-3672, 3685-3781,
// and getGuidedFilter with a 3- / 6-channel guide of disp_method/gifs.cpp:8-135.
//
// The guide of computeAdaptiveWeight_GuidedF is the six channels of the two views aligned at the slice's candidate, so
// nothing of the filter is shared between slices: per slice 1 + 3 * cn box filters for the coefficients and cn + 1 for
// their means.  Box filters run on tiles: a workgroup fills an LDS tile (with its halo) from a functor that forms the
// quantity on the fly from the 8-bit views, sums rows then columns in double (cv::boxFilter's accumulator type; the
// sums of these 24-bit values are exact, so their order does not matter) and keeps the means of its 64 x 16 pixels in
// registers.  One kernel leaves the cn + 1 coefficient planes of a slice in memory, a second one filters those and
// applies them; slices run side by side in the grid's z.  Every Mat expression of the reference rounds to float per
// operation; so does every operation here (no contraction).
#include "common.h"
#include "box_march.h"
#include <float.h>

namespace {

#define MC_TW 64
#define MC_TH 16
#define MC_MAXCN 6

__device__ __forceinline__ int mc_r101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - p - 2;
    return p;
}
__device__ __forceinline__ int mc_reflect(int p, int n)       // BORDER_REFLECT: fedcba|abcdef|fedcba
{
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
    return p;
}

// The guide of slice k: the channels of image a at column x + aOff0 + k * aOffStep, then those of image b likewise
struct McGuide {
    const u8* a; size_t ap; int acn, aOff0, aOffStep;
    const u8* b; size_t bp; int bcn, bOff0, bOffStep;     // bcn = 0: no second image
    int cn;
};
__device__ __forceinline__ float mc_guide(const McGuide& g, int c, int x, int y, int k, float sa, float sb)
{
    const int v = c < g.acn ? g.a[(size_t)y * g.ap + (size_t)(x + g.aOff0 + k * g.aOffStep) * g.acn + c]
                            : g.b[(size_t)y * g.bp + (size_t)(x + g.bOff0 + k * g.bOffStep) * g.bcn + (c - g.acn)];
    return __fadd_rn(__fmul_rn((float)v, sa), sb);          // convertTo(CV_32F, scale, shift)
}

// Box mean of f over the r x r window (anchor r / 2, BORDER_REFLECT_101) at the calling thread's four pixels
// (x0 + (t & 63), y0 + 4 * (t >> 6) + j).  tile: (MC_TH + r - 1) x (MC_TW + r - 1) floats, hs: (MC_TH + r - 1) x MC_TW doubles.
template <class F>
__device__ __forceinline__ void mc_tile_box(F f, int x0, int y0, int H, int W, int r, double scale, float* tile, double* hs,
                                            float (&res)[4])
{
    const int ax = r / 2, LW = MC_TW + r - 1, LH = MC_TH + r - 1, t = threadIdx.x;
    __syncthreads();                                    // the previous quantity's reads of tile / hs are done
    for (int e = t; e < LW * LH; e += 256) {
        const int yy = e / LW, xx = e - yy * LW;
        tile[e] = f(mc_r101(x0 - ax + xx, W), mc_r101(y0 - ax + yy, H));
    }
    __syncthreads();
    for (int item = t; item < LH * 4; item += 256) {
        const int row = item >> 2, seg = (item & 3) * 16;
        const float* p = tile + row * LW + seg;
        double s = 0;
        for (int i = 0; i < r; i++) s += (double)p[i];
        double* o = hs + row * MC_TW + seg;
        o[0] = s;
        for (int j = 1; j < 16; j++) { s += (double)p[j - 1 + r] - (double)p[j - 1]; o[j] = s; }
    }
    __syncthreads();
    const int x = t & 63, yg = (t >> 6) * 4;
    double s = 0;
    for (int i = 0; i < r; i++) s += hs[(yg + i) * MC_TW + x];
    res[0] = (float)(s * scale);
#pragma unroll
    for (int j = 1; j < 4; j++) {
        s += hs[(yg + j - 1 + r) * MC_TW + x] - hs[(yg + j - 1) * MC_TW + x];
        res[j] = (float)(s * scale);
    }
}

// getCostSAD_d for the slices k0 .. k0 + n - 1: |ref(x) - oth(x + off0 + k * offStep)| on gray views, box mean
struct McCost { const u8* ref; size_t rp; const u8* oth; size_t op; int off0, offStep; };
__global__ __launch_bounds__(256) void mc_sad_cost_kernel(McCost c, int H, int W, int r, int k0, float* __restrict__ out)
{
    extern __shared__ double mcLds[];
    const int LH = MC_TH + r - 1;
    double* hs = mcLds;
    float* tile = reinterpret_cast<float*>(mcLds + LH * MC_TW);
    const int x0 = blockIdx.x * MC_TW, y0 = blockIdx.y * MC_TH, k = k0 + blockIdx.z;
    const int off = c.off0 + k * c.offStep;
    float res[4];
    mc_tile_box([&](int x, int y) -> float {
        const int a = c.ref[(size_t)y * c.rp + x], b = c.oth[(size_t)y * c.op + x + off];
        return (float)(a > b ? a - b : b - a);
    }, x0, y0, H, W, r, 1.0 / ((double)r * r), tile, hs, res);
    const int x = x0 + (threadIdx.x & 63), yb = y0 + (threadIdx.x >> 6) * 4;
    if (x >= W) return;
    for (int j = 0; j < 4; j++)
        if (yb + j < H) out[((size_t)blockIdx.z * H + yb + j) * W + x] = res[j];
}

// order-preserving map float -> uint32 (atomicMin / Max on floats of either sign)
__device__ __forceinline__ u32 mc_f2ord(float f) { u32 u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float mc_ord2f(u32 o) { u32 u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; return __uint_as_float(u); }

__global__ void mc_mm_init_kernel(u32* mm, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { mm[2 * i] = 0xFFFFFFFFu; mm[2 * i + 1] = 0u; }
}
__global__ __launch_bounds__(256) void mc_minmax_kernel(const float* __restrict__ src, size_t N, u32* mm)
{
    const float* s = src + (size_t)blockIdx.y * N;
    u32 lo = 0xFFFFFFFFu, hi = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (size_t)gridDim.x * 256) {
        const u32 o = mc_f2ord(s[i]);
        lo = min(lo, o); hi = max(hi, o);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { lo = min(lo, (u32)__shfl_xor((int)lo, m, 64)); hi = max(hi, (u32)__shfl_xor((int)hi, m, 64)); }
    // one pair of atomics per workgroup: hundreds of slices x hundreds of waves on the same two words queue up otherwise
    __shared__ u32 slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&mm[2 * blockIdx.y], min(min(slo[0], slo[1]), min(slo[2], slo[3])));
        atomicMax(&mm[2 * blockIdx.y + 1], max(max(shi[0], shi[1]), max(shi[2], shi[3])));
    }
}

// per-column minimum / maximum of an interleaved 8-bit image over all rows and channels
__global__ __launch_bounds__(256) void mc_col_minmax_kernel(const u8* __restrict__ img, size_t pitch, int H, int Wimg, int cn,
                                                            u8* __restrict__ cmin, u8* __restrict__ cmax)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= Wimg) return;
    int lo = 255, hi = 0;
    for (int y = 0; y < H; y++)
        for (int c = 0; c < cn; c++) { const int v = img[(size_t)y * pitch + (size_t)x * cn + c]; lo = min(lo, v); hi = max(hi, v); }
    cmin[x] = (u8)lo; cmax[x] = (u8)hi;
}

// cv::normalize(0, 1, NORM_MINMAX, CV_32F) coefficients: scale = 1 / (max - min) (0 when the range is <= DBL_EPSILON),
// shift = -min * scale, both then rounded to float by convertTo
__device__ __forceinline__ float2 mc_norm_coeff(double smin, double smax)
{
    const double scale = (smax - smin > DBL_EPSILON) ? 1.0 / (smax - smin) : 0.0;
    return make_float2((float)scale, (float)(0.0 - smin * scale));
}
// one workgroup per slice: the guide's minimum / maximum over both images' column ranges, and the slice's own
__global__ __launch_bounds__(256) void mc_coeff_kernel(McGuide g, int W, int k0, const u8* __restrict__ aMin, const u8* __restrict__ aMax,
                                                       const u8* __restrict__ bMin, const u8* __restrict__ bMax,
                                                       const u32* __restrict__ pmm, float2* __restrict__ gco, float2* __restrict__ pco)
{
    __shared__ int slo[4], shi[4];
    const int z = blockIdx.x, k = k0 + z;
    int lo = 255, hi = 0;
    const int ao = g.aOff0 + k * g.aOffStep, bo = g.bOff0 + k * g.bOffStep;
    for (int x = threadIdx.x; x < W; x += 256) {
        lo = min(lo, (int)aMin[x + ao]); hi = max(hi, (int)aMax[x + ao]);
        if (g.bcn) { lo = min(lo, (int)bMin[x + bo]); hi = max(hi, (int)bMax[x + bo]); }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { lo = min(lo, __shfl_xor(lo, m, 64)); hi = max(hi, __shfl_xor(hi, m, 64)); }
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = min(min(slo[0], slo[1]), min(slo[2], slo[3])); hi = max(max(shi[0], shi[1]), max(shi[2], shi[3]));
        gco[z] = mc_norm_coeff((double)lo, (double)hi);
        pco[z] = mc_norm_coeff((double)mc_ord2f(pmm[2 * z]), (double)mc_ord2f(pmm[2 * z + 1]));
    }
}

// Coefficients of slice z: means, covariance and variance per guide channel, a_c and b (gifs.cpp:58-128).
// variant 0: a = cov / (var + eps); variant 1: a = cov / (var + (var / var) * eps) with cv::divide's 0 for a zero divisor.
struct McAbArgs {
    McGuide g; int H, W, r, k0, variant; float epsf;
    const float* P;            // raw slices [n][H][W]
    const float2* gco; const float2* pco;
    float* A;                  // [n][cn][H][W]
    float* B;                  // [n][H][W]
};
__global__ __launch_bounds__(256) void mc_ab_kernel(McAbArgs a)
{
    extern __shared__ double mcLds[];
    const int H = a.H, W = a.W, r = a.r, LH = MC_TH + r - 1, cn = a.g.cn;
    double* hs = mcLds;
    float* tile = reinterpret_cast<float*>(mcLds + LH * MC_TW);
    const int x0 = blockIdx.x * MC_TW, y0 = blockIdx.y * MC_TH, z = blockIdx.z, k = a.k0 + z;
    const size_t N = (size_t)H * W;
    const float2 gc = a.gco[z], pc = a.pco[z];
    const float* P = a.P + (size_t)z * N;
    const double scale = 1.0 / ((double)r * r);
    auto pv = [&](int x, int y) -> float { return __fadd_rn(__fmul_rn(P[(size_t)y * W + x], pc.x), pc.y); };
    float meanP[4], dot[4] = {0.f, 0.f, 0.f, 0.f};
    mc_tile_box(pv, x0, y0, H, W, r, scale, tile, hs, meanP);
    const int x = x0 + (threadIdx.x & 63), yb = y0 + (threadIdx.x >> 6) * 4;
    for (int c = 0; c < cn; c++) {
        float mg[4], cgp[4], cgg[4];
        mc_tile_box([&](int xx, int yy) -> float { return mc_guide(a.g, c, xx, yy, k, gc.x, gc.y); }, x0, y0, H, W, r, scale, tile, hs, mg);
        mc_tile_box([&](int xx, int yy) -> float { return __fmul_rn(mc_guide(a.g, c, xx, yy, k, gc.x, gc.y), pv(xx, yy)); }, x0, y0, H, W, r,
                    scale, tile, hs, cgp);
        mc_tile_box([&](int xx, int yy) -> float { const float v = mc_guide(a.g, c, xx, yy, k, gc.x, gc.y); return __fmul_rn(v, v); }, x0, y0, H,
                    W, r, scale, tile, hs, cgg);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float cov = __fsub_rn(cgp[j], __fmul_rn(mg[j], meanP[j]));
            const float var = __fsub_rn(cgg[j], __fmul_rn(mg[j], mg[j]));
            float den;
            if (a.variant == 0) den = __fadd_rn(var, a.epsf);
            else den = __fadd_rn(var, var != 0.f ? __fdiv_rn(__fmul_rn(var, a.epsf), var) : 0.f);
            const float av = den != 0.f ? __fdiv_rn(cov, den) : 0.f;
            dot[j] = __fadd_rn(dot[j], __fmul_rn(av, mg[j]));
            if (x < W && yb + j < H) a.A[((size_t)z * cn + c) * N + (size_t)(yb + j) * W + x] = av;
        }
    }
    if (x < W)
        for (int j = 0; j < 4; j++)
            if (yb + j < H) a.B[(size_t)z * N + (size_t)(yb + j) * W + x] = __fsub_rn(meanP[j], dot[j]);
}

// filtered slice z = dot(box(a), G) + box(b)    (gifs.cpp:130-134)
struct McOutArgs { McGuide g; int H, W, r, k0; const float2* gco; const float* A; const float* B; float* V; };
__global__ __launch_bounds__(256) void mc_out_kernel(McOutArgs a)
{
    extern __shared__ double mcLds[];
    const int H = a.H, W = a.W, r = a.r, LH = MC_TH + r - 1, cn = a.g.cn;
    double* hs = mcLds;
    float* tile = reinterpret_cast<float*>(mcLds + LH * MC_TW);
    const int x0 = blockIdx.x * MC_TW, y0 = blockIdx.y * MC_TH, z = blockIdx.z, k = a.k0 + z;
    const size_t N = (size_t)H * W;
    const float2 gc = a.gco[z];
    const double scale = 1.0 / ((double)r * r);
    const int x = x0 + (threadIdx.x & 63), yb = y0 + (threadIdx.x >> 6) * 4;
    float dot[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < cn; c++) {
        const float* Ap = a.A + ((size_t)z * cn + c) * N;
        float ma[4];
        mc_tile_box([&](int xx, int yy) -> float { return Ap[(size_t)yy * W + xx]; }, x0, y0, H, W, r, scale, tile, hs, ma);
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (x < W && yb + j < H) dot[j] = __fadd_rn(dot[j], __fmul_rn(ma[j], mc_guide(a.g, c, x, yb + j, k, gc.x, gc.y)));
    }
    const float* Bp = a.B + (size_t)z * N;
    float mb[4];
    mc_tile_box([&](int xx, int yy) -> float { return Bp[(size_t)yy * W + xx]; }, x0, y0, H, W, r, scale, tile, hs, mb);
    if (x < W)
        for (int j = 0; j < 4; j++)
            if (yb + j < H) a.V[(size_t)z * N + (size_t)(yb + j) * W + x] = __fadd_rn(dot[j], mb[j]);
}

// first minimum over the n slices of a batch against the running best (:3763-3779: strict '<' starting from DBL_MAX, which
// every finite float and -inf pass and +inf / NaN do not -- the same as starting from +inf in float).  A pixel no slice
// wins keeps the value of the reference's uninitialised map: 0 here.
__global__ __launch_bounds__(256) void mc_wta_kernel(const float* __restrict__ V, size_t N, int n, int first, int d0,
                                                     float* __restrict__ best, float* __restrict__ disp)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float b = first ? __uint_as_float(0x7F800000u) : best[i], d = first ? 0.f : disp[i];
    for (int k = 0; k < n; k++) {
        const float v = V[(size_t)k * N + i];
        if (v < b) { b = v; d = (float)(d0 + k); }
    }
    best[i] = b;
    disp[i] = d;
}

// copyMakeBorder(src, dst, 0, 0, left, right, BORDER_REFLECT) on an interleaved image
__global__ __launch_bounds__(256) void mc_pad_reflect_kernel(const u8* __restrict__ src, size_t sp, int W, int cn, int left, int Wp,
                                                             u8* __restrict__ dst, size_t dp)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= Wp) return;
    const int sx = mc_reflect(x - left, W);
    for (int c = 0; c < cn; c++) dst[(size_t)y * dp + (size_t)x * cn + c] = src[(size_t)y * sp + (size_t)sx * cn + c];
}

// ---- the same filter on the marching box kernels (box_march.h): every input row is read once and the window sums are
// running sums, so a box filter costs a few operations per pixel whatever the window; the price is that the quantities
// of a slice go through memory between passes (two box filters per pass).  Per slice and guide channel c: a pass over
// (G_c, G_c^2) leaves mean and variance, a pass over (P, G_c P) leaves a_c and adds a_c mean(G_c) to the running dot product
// (the last channel's pass writes b = mean(P) - dot); then the planes of a two at a time, and b, are box-filtered and
// accumulated into the output in the reference's order.  Same float operations per pixel as the tile kernels.
__device__ __forceinline__ int mc_guide_raw(const McGuide& g, int c, int x, int y, int k)
{
    return c < g.acn ? g.a[(size_t)y * g.ap + (size_t)(x + g.aOff0 + k * g.aOffStep) * g.acn + c]
                     : g.b[(size_t)y * g.bp + (size_t)(x + g.bOff0 + k * g.bOffStep) * g.bcn + (c - g.acn)];
}
__device__ __forceinline__ float mc_norm(int v, const float2& co) { return __fadd_rn(__fmul_rn((float)v, co.x), co.y); }

// A channel of the view that does not move with the candidate (offset step 0) has the same mean and variance in every
// slice whose guide normalisation equals the first slice's (the usual case: the guide's extremes are 0 and 255 everywhere):
// those slices are skipped here and read the first slice's planes.
__device__ __forceinline__ bool mc_same_as_first(const float2* gco, int k)
{ return k > 0 && gco[k].x == gco[0].x && gco[k].y == gco[0].y; }
struct McLoadGG {
    McGuide g; int c, k0; const float2* gco; float2 gc; bool fixedView;
    struct Raw { int v; };
    __device__ bool skip(int k) const { return fixedView && mc_same_as_first(gco, k); }
    __device__ void begin(int k) { gc = gco[k]; }
    __device__ Raw fetch(int x, int y, int k) const { Raw r; r.v = mc_guide_raw(g, c, x, y, k0 + k); return r; }
    __device__ void finish(const Raw& r, float& q0, float& q1) const { q0 = mc_norm(r.v, gc); q1 = __fmul_rn(q0, q0); }
};
struct McStoreMV {
    float* MG; float* VAR; size_t N; int W; double scale;
    __device__ void operator()(int x, int y, int k, double s0, double s1) const
    {
        const size_t o = (size_t)k * N + (size_t)y * W + x;
        const float mg = (float)(s0 * scale), cgg = (float)(s1 * scale);
        MG[o] = mg; VAR[o] = __fsub_rn(cgg, __fmul_rn(mg, mg));
    }
};
struct McLoadPGP {
    McGuide g; int c, k0; const float* P; size_t N; int W; const float2* gco; const float2* pco; float2 gc, pc;
    struct Raw { float p; int v; };
    __device__ bool skip(int) const { return false; }
    __device__ void begin(int k) { gc = gco[k]; pc = pco[k]; }
    __device__ Raw fetch(int x, int y, int k) const { Raw r; r.p = P[(size_t)k * N + (size_t)y * W + x]; r.v = mc_guide_raw(g, c, x, y, k0 + k); return r; }
    __device__ void finish(const Raw& r, float& q0, float& q1) const
    {
        q0 = __fadd_rn(__fmul_rn(r.p, pc.x), pc.y);
        q1 = __fmul_rn(mc_norm(r.v, gc), q0);
    }
};
struct McStoreA {
    float* A; float* DOT; float* B; const float* MG; const float* VAR; size_t N; int W, cn, c, variant; float epsf; double scale;
    const float2* gco; bool fixedView;
    __device__ void operator()(int x, int y, int k, double s0, double s1) const
    {
        const size_t i = (size_t)y * W + x, o = (size_t)k * N + i;
        const size_t om = fixedView && mc_same_as_first(gco, k) ? i : o;          // the first slice's planes (McLoadGG::skip)
        const float mp = (float)(s0 * scale), cgp = (float)(s1 * scale), mg = MG[om], var = VAR[om];
        const float cov = __fsub_rn(cgp, __fmul_rn(mg, mp));
        float den;
        if (variant == 0) den = __fadd_rn(var, epsf);
        else den = __fadd_rn(var, var != 0.f ? __fdiv_rn(__fmul_rn(var, epsf), var) : 0.f);
        const float av = den != 0.f ? __fdiv_rn(cov, den) : 0.f;
        A[((size_t)k * cn + c) * N + i] = av;
        const float dot = __fadd_rn(c == 0 ? 0.f : DOT[o], __fmul_rn(av, mg));
        if (c == cn - 1) B[o] = __fsub_rn(mp, dot);
        else DOT[o] = dot;
    }
};
// two planes [k][plane][N] (or one: second = nullptr)
struct McLoadPlanes {
    const float* P0; const float* P1; size_t kstride, N; int W;
    struct Raw { float a, b; };
    __device__ bool skip(int) const { return false; }
    __device__ void begin(int) {}
    __device__ Raw fetch(int x, int y, int k) const
    {
        const size_t o = (size_t)k * kstride + (size_t)y * W + x;
        Raw r; r.a = P0[o]; r.b = P1 ? P1[o] : 0.f;
        return r;
    }
    __device__ void finish(const Raw& r, float& q0, float& q1) const { q0 = r.a; q1 = r.b; }
};
// V (+)= mean(a_c) G_c (+ mean(a_c+1) G_c+1), or V += mean(b)
struct McStoreDot {
    McGuide g; int k0, c, nq; const float2* gco; float* V; size_t N; int W; double scale; bool isB;
    __device__ void operator()(int x, int y, int k, double s0, double s1) const
    {
        const size_t o = (size_t)k * N + (size_t)y * W + x;
        const float m0 = (float)(s0 * scale), m1 = (float)(s1 * scale);
        if (isB) { V[o] = __fadd_rn(V[o], m0); return; }
        const float2 gc = gco[k];
        float acc = c == 0 ? 0.f : V[o];
        acc = __fadd_rn(acc, __fmul_rn(m0, mc_norm(mc_guide_raw(g, c, x, y, k0 + k), gc)));
        if (nq == 2) acc = __fadd_rn(acc, __fmul_rn(m1, mc_norm(mc_guide_raw(g, c + 1, x, y, k0 + k), gc)));
        V[o] = acc;
    }
};
// getCostSAD_d slices: (|ref - oth|, nothing) -> mean
struct McLoadCost {
    McCost c; int k0;
    struct Raw { int a, b; };
    __device__ bool skip(int) const { return false; }
    __device__ void begin(int) {}
    __device__ Raw fetch(int x, int y, int k) const
    {
        Raw r; r.a = c.ref[(size_t)y * c.rp + x]; r.b = c.oth[(size_t)y * c.op + x + c.off0 + (k0 + k) * c.offStep];
        return r;
    }
    __device__ void finish(const Raw& r, float& q0, float& q1) const { q0 = (float)(r.a > r.b ? r.a - r.b : r.b - r.a); q1 = 0.f; }
};
struct McStoreMean {
    float* out; size_t N; int W; double scale;
    __device__ void operator()(int x, int y, int k, double s0, double) const { out[(size_t)k * N + (size_t)y * W + x] = (float)(s0 * scale); }
};

size_t mc_lds_bytes(int r) { return (size_t)(MC_TH + r - 1) * MC_TW * sizeof(double) + (size_t)(MC_TH + r - 1) * (MC_TW + r - 1) * sizeof(float); }

}  // namespace

// ---- host drivers ---------------------------------------------------------------------------------------------------

// the tile kernels (kept behind MDISP_TUNE_GIF_MC_TILES: the tests run both) hold a tile and its halo in LDS
static bool mc_use_tiles() { return md_tune(MDISP_TUNE_GIF_MC_TILES) > 0; }
int md_mc_check_window(int r)
{
    MD_REQUIRE(r > 0, MDISP_E_BADARG, "window must be positive");
    if (mc_use_tiles()) MD_REQUIRE(mc_lds_bytes(r) <= 64 * 1024, MDISP_E_UNSUPPORTED, "window %d needs more than 64 KB of LDS per tile (at most 47)", r);
    else MD_REQUIRE(march_supported(r), MDISP_E_UNSUPPORTED, "window %d is wider than the %d columns a marching workgroup holds", r, MW);
    return MDISP_OK;
}

// slices per batch and scratch bytes of md_mc_filter_wta
static int mc_batch(int H, int W, int cn, int n)
{
    const size_t per = (size_t)H * W * 4 * (cn + 6);      // raw slice, a planes, b, filtered; mean, variance, running dot product
    size_t nb = ((size_t)8 << 30) / per;      // a marching launch needs hundreds of slices x strips to fill the device: up to 8 GB of planes
    if (nb < 1) nb = 1;
    if (nb > (size_t)n) nb = n;
    return (int)nb;
}
size_t md_mc_scratch_bytes(int H, int W, int cn, int n, int Wa, int Wb)
{
    const int nb = mc_batch(H, W, cn, n);
    const size_t N = (size_t)H * W;
    return md_align(N * 4 * nb, 256) * 6 + md_align(N * 4 * nb * cn, 256) + md_align(N * 4, 256) +
           md_align((size_t)nb * 8, 256) * 3 + md_align(Wa, 256) * 2 + md_align(Wb > 0 ? Wb : 1, 256) * 2 + 4096;
}

struct McGuideHost {
    const u8* a; size_t ap; int acn, aW, aOff0, aOffStep;
    const u8* b; size_t bp; int bcn, bW, bOff0, bOffStep;
};

// Filters n slices with the guided filter and takes the first minimum; the slices are either given (dSlices, [n][H][W]
// raw, normalised per slice here) or formed as getCostSAD_d slices from the gray views in `cost`.
// dFiltered (optional): receives the n filtered slices.  dDisp (optional): the map, value d0 + index.
int md_mc_filter_wta(hipStream_t s, const McGuideHost& gh, int H, int W, int r, double eps, int variant, int n, int d0,
                     const float* dSlices, const McCost* cost, int costWin, float* dFiltered, float* dDisp, void* scratch)
{
    MD_TRY(md_mc_check_window(r));
    if (cost) MD_TRY(md_mc_check_window(costWin));
    const int cn = gh.acn + gh.bcn;
    MD_REQUIRE(cn >= 1 && cn <= MC_MAXCN, MDISP_E_UNSUPPORTED, "guide with %d channels", cn);
    const size_t N = (size_t)H * W;
    const int nb = mc_batch(H, W, cn, n);
    u8* base = (u8*)scratch;
    auto take = [&](size_t bytes) { u8* p = base; base += md_align(bytes, 256); return p; };
    float* Praw = (float*)take(N * 4 * nb);
    float* Bp = (float*)take(N * 4 * nb);
    float* V = (float*)take(N * 4 * nb);
    float* A = (float*)take(N * 4 * nb * cn);
    float* best = (float*)take(N * 4);
    float* MG = (float*)take(N * 4 * nb);
    float* VAR = (float*)take(N * 4 * nb);
    float* DOT = (float*)take(N * 4 * nb);
    u32* pmm = (u32*)take((size_t)nb * 8);
    float2* gco = (float2*)take((size_t)nb * 8);
    float2* pco = (float2*)take((size_t)nb * 8);
    u8* aMin = take(gh.aW); u8* aMax = take(gh.aW);
    u8* bMin = take(gh.bW > 0 ? gh.bW : 1); u8* bMax = take(gh.bW > 0 ? gh.bW : 1);
    McGuide g{};
    g.a = gh.a; g.ap = gh.ap; g.acn = gh.acn; g.aOff0 = gh.aOff0; g.aOffStep = gh.aOffStep;
    g.b = gh.b; g.bp = gh.bp; g.bcn = gh.bcn; g.bOff0 = gh.bOff0; g.bOffStep = gh.bOffStep; g.cn = cn;
    mc_col_minmax_kernel<<<(gh.aW + 255) / 256, 256, 0, s>>>(gh.a, gh.ap, H, gh.aW, gh.acn, aMin, aMax);
    if (gh.bcn) mc_col_minmax_kernel<<<(gh.bW + 255) / 256, 256, 0, s>>>(gh.b, gh.bp, H, gh.bW, gh.bcn, bMin, bMax);
    const dim3 tiles((W + MC_TW - 1) / MC_TW, (H + MC_TH - 1) / MC_TH);
    const size_t lds = mc_lds_bytes(r);
    const bool useTiles = mc_use_tiles();
    const double scale = 1.0 / ((double)r * r);
    for (int k0 = 0; k0 < n; k0 += nb) {
        const int m = n - k0 < nb ? n - k0 : nb;
        const dim3 grid(tiles.x, tiles.y, m);
        const float* P = Praw;
        if (cost && useTiles) mc_sad_cost_kernel<<<grid, 256, mc_lds_bytes(costWin), s>>>(*cost, H, W, costWin, k0, Praw);
        else if (cost) {
            McLoadCost lc{*cost, k0};
            McStoreMean sm{Praw, N, W, 1.0 / ((double)costWin * costWin)};
            MD_TRY(dual_box_march(s, m, H, W, costWin, 0, 0, H, lc, sm));
        }
        else P = dSlices + (size_t)k0 * N;
        mc_mm_init_kernel<<<(m + 63) / 64, 64, 0, s>>>(pmm, m);
        int mb = (int)((N + 256 * 8 - 1) / (256 * 8));
        const int mbMax = m >= 64 ? 16 : (m >= 8 ? 64 : 512);      // enough workgroups over all slices; few atomics per slice
        if (mb > mbMax) mb = mbMax;
        mc_minmax_kernel<<<dim3(mb, m), 256, 0, s>>>(P, N, pmm);
        mc_coeff_kernel<<<m, 256, 0, s>>>(g, W, k0, aMin, aMax, bMin, bMax, pmm, gco, pco);
        float* const Vout = dFiltered ? dFiltered + (size_t)k0 * N : V;
        if (useTiles) {
            McAbArgs ab{};
            ab.g = g; ab.H = H; ab.W = W; ab.r = r; ab.k0 = k0; ab.variant = variant; ab.epsf = (float)eps;
            ab.P = P; ab.gco = gco; ab.pco = pco; ab.A = A; ab.B = Bp;
            mc_ab_kernel<<<grid, 256, lds, s>>>(ab);
            McOutArgs oa{};
            oa.g = g; oa.H = H; oa.W = W; oa.r = r; oa.k0 = k0; oa.gco = gco; oa.A = A; oa.B = Bp;
            oa.V = Vout;
            mc_out_kernel<<<grid, 256, lds, s>>>(oa);
        } else {
            for (int c = 0; c < cn; c++) {
                const bool fixedView = c < g.acn ? g.aOffStep == 0 : g.bOffStep == 0;
                McLoadGG lgg{g, c, k0, gco, {}, fixedView};
                McStoreMV smv{MG, VAR, N, W, scale};
                MD_TRY(dual_box_march(s, m, H, W, r, 0, 0, H, lgg, smv));
                McLoadPGP lp{g, c, k0, P, N, W, gco, pco, {}, {}};
                McStoreA sa{A, DOT, Bp, MG, VAR, N, W, cn, c, variant, (float)eps, scale, gco, fixedView};
                MD_TRY(dual_box_march(s, m, H, W, r, 0, 0, H, lp, sa));
            }
            for (int c = 0; c < cn; c += 2) {
                const int nq = c + 1 < cn ? 2 : 1;
                McLoadPlanes la{A + (size_t)c * N, nq == 2 ? A + (size_t)(c + 1) * N : nullptr, (size_t)cn * N, N, W};
                McStoreDot sd{g, k0, c, nq, gco, Vout, N, W, scale, false};
                MD_TRY(dual_box_march(s, m, H, W, r, 0, 0, H, la, sd));
            }
            McLoadPlanes lb{Bp, nullptr, N, N, W};
            McStoreDot sb{g, k0, 0, 1, gco, Vout, N, W, scale, true};
            MD_TRY(dual_box_march(s, m, H, W, r, 0, 0, H, lb, sb));
        }
        if (dDisp) mc_wta_kernel<<<(unsigned)((N + 255) / 256), 256, 0, s>>>(Vout, N, m, k0 == 0, d0 + k0, best, dDisp);
    }
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// computeAdaptiveWeight_GuidedF on interleaved 3-channel device images (pitch in bytes); dDisp: H*W floats.
size_t md_asw_guidedf_bytes(int H, int W, int cn, int minD, int numD)
{
    const int Wp = W + minD + numD - 1;
    return md_align((size_t)H * md_align((size_t)Wp * cn, 256), 256) + md_align((size_t)H * md_align(W, 256), 256) +
           md_align((size_t)H * md_align(Wp, 256), 256) + md_mc_scratch_bytes(H, W, 2 * cn, numD, W, Wp);
}

int md_asw_guidedf(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, int H, int W, int cn, int isLeft, double eps, int win,
                   int minD, int numD, float* dDisp, void* ws)
{
    // a gray pair makes a 2-channel guide, for which the reference's multiChl_to_oneChl_mul returns an empty image and the
    // expression that uses it throws (method_StereoMatching.cpp:3578-3609, 3665)
    MD_REQUIRE(cn == 3, MDISP_E_UNSUPPORTED, "computeAdaptiveWeight_GuidedF needs 3-channel images (the reference fails on others)");
    MD_REQUIRE(numD > 0 && minD >= 0, MDISP_E_BADARG, "bad disparity range");
    MD_REQUIRE(win > 0 && win % 2 == 1, MDISP_E_BADARG, "winsize must be odd");
    const int maxOff = minD + numD - 1, Wp = W + maxOff;
    const size_t pP = md_align((size_t)Wp * cn, 256), pG = md_align(W, 256), pGp = md_align(Wp, 256);
    u8* base = (u8*)ws;
    auto take = [&](size_t bytes) { u8* p = base; base += md_align(bytes, 256); return p; };
    u8* othP = take((size_t)H * pP); u8* refG = take((size_t)H * pG); u8* othG = take((size_t)H * pGp);
    const u8* ref = isLeft ? dL : dR;
    const u8* oth = isLeft ? dR : dL;
    mc_pad_reflect_kernel<<<dim3((Wp + 255) / 256, H), 256, 0, s>>>(oth, pitch, W, cn, isLeft ? maxOff : 0, Wp, othP, pP);
    MD_TRY(md_bgr2gray(s, ref, pitch, H, W, refG, pG));
    MD_TRY(md_bgr2gray(s, othP, pP, H, Wp, othG, pGp));
    // slice i, candidate d = minD + i: left view -> right_border at column offset maxOff - d (:3323, :3727); right view ->
    // left_border at offset d (:3338, :3743).  The guide lists the left view's channels first in both branches.
    McCost c{};
    c.ref = refG; c.rp = pG; c.oth = othG; c.op = pGp;
    c.off0 = isLeft ? maxOff - minD : minD; c.offStep = isLeft ? -1 : 1;
    McGuideHost g{};
    if (isLeft) {
        g.a = ref; g.ap = pitch; g.acn = cn; g.aW = W; g.aOff0 = 0; g.aOffStep = 0;
        g.b = othP; g.bp = pP; g.bcn = cn; g.bW = Wp; g.bOff0 = c.off0; g.bOffStep = -1;
    } else {
        g.a = othP; g.ap = pP; g.acn = cn; g.aW = Wp; g.aOff0 = c.off0; g.aOffStep = 1;
        g.b = ref; g.bp = pitch; g.bcn = cn; g.bW = W; g.bOff0 = 0; g.bOffStep = 0;
    }
    return md_mc_filter_wta(s, g, H, W, win, eps, 1, numD, minD, nullptr, &c, win, nullptr, dDisp, base);
}

// getCostSAD_d on gray device views of which one is wider: out H x W floats
int md_cost_sad_d(hipStream_t s, const u8* ref, size_t rp, const u8* oth, size_t op, int H, int W, int off, int win, float* out)
{
    MD_TRY(md_mc_check_window(win));
    McCost c{};
    c.ref = ref; c.rp = rp; c.oth = oth; c.op = op; c.off0 = off; c.offStep = 0;
    if (mc_use_tiles()) mc_sad_cost_kernel<<<dim3((W + MC_TW - 1) / MC_TW, (H + MC_TH - 1) / MC_TH, 1), 256, mc_lds_bytes(win), s>>>(c, H, W, win, 0, out);
    else {
        McLoadCost lc{c, 0};
        McStoreMean sm{out, (size_t)H * W, W, 1.0 / ((double)win * win)};
        MD_TRY(dual_box_march(s, 1, H, W, win, 0, 0, H, lc, sm));
    }
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// getGuidedFilter with an interleaved cn-channel 8-bit guide and one float slice
int md_guided_filter_mc(hipStream_t s, const u8* guide, size_t gp, int cn, const float* dP, int H, int W, int r, double eps, int variant,
                        float* dOut, void* scratch)
{
    McGuideHost g{};
    g.a = guide; g.ap = gp; g.acn = cn; g.aW = W;
    return md_mc_filter_wta(s, g, H, W, r, eps, variant, 1, 0, dP, nullptr, 0, dOut, nullptr, scratch);
}

// ---- computeSimilarity (colour + gradient cost) and the two matchers that filter other cost volumes ---------------------

namespace {
// filter2D(img, CV_32F, [-3 0 3; -10 0 10; -3 0 3]) of an interleaved 3-channel image, BORDER_REFLECT_101 (integer sums: exact)
__global__ __launch_bounds__(256) void mc_sobel3_kernel(const u8* __restrict__ img, size_t pitch, int H, int W, float* __restrict__ out)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int xl = mc_r101(x - 1, W), xr = mc_r101(x + 1, W);
    const u8* r0 = img + (size_t)mc_r101(y - 1, H) * pitch;
    const u8* r1 = img + (size_t)y * pitch;
    const u8* r2 = img + (size_t)mc_r101(y + 1, H) * pitch;
    for (int c = 0; c < 3; c++) {
        const int s = 3 * ((int)r0[xr * 3 + c] - (int)r0[xl * 3 + c]) + 10 * ((int)r1[xr * 3 + c] - (int)r1[xl * 3 + c]) +
                      3 * ((int)r2[xr * 3 + c] - (int)r2[xl * 3 + c]);
        out[((size_t)y * W + x) * 3 + c] = (float)s;
    }
}

// one slice value per thread (method_StereoMatching.cpp:1145-1179, evaluated as the library evaluates those expressions)
struct McSimArgs {
    const u8* L; size_t lp; const u8* Rb; size_t rp;       // left view, right view padded on the left by maxOff (REFLECT)
    const float* sL; const float* sR;                      // their gradients: [H][W][3], [H][Wp][3]
    int H, W, Wp, maxOff, minD;
    float third, betaC, tG, wC, wG; double thresC;
    float* out; size_t sliceStride; int outPitch, border;  // slice k: out + k * sliceStride, rows of outPitch floats, `border` cells on every side
};
__global__ __launch_bounds__(256) void mc_similarity_kernel(McSimArgs a)
{
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y, k = blockIdx.z;
    const int OW = a.W + 2 * a.border;
    if (ox >= OW) return;
    // the padded overload borders every slice with BORDER_REFLECT (:1342-1347): the value of the reflected pixel
    const int x = mc_reflect(ox - a.border, a.W), y = mc_reflect(oy - a.border, a.H);
    const int off = a.maxOff - (a.minD + k);
    int c[3]; float g[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const int p = a.L[(size_t)y * a.lp + x * 3 + ch], q = a.Rb[(size_t)y * a.rp + (size_t)(off + x) * 3 + ch];
        c[ch] = p > q ? p - q : q - p;
        g[ch] = fabsf(__fsub_rn(a.sL[((size_t)y * a.W + x) * 3 + ch], a.sR[((size_t)y * a.Wp + off + x) * 3 + ch]));
    }
    const int m1 = min(c[0] + c[1], 255);
    const int colour = min(__float2int_rn(__fadd_rn(__fmul_rn((float)m1, a.third), __fmul_rn((float)c[2], a.third))), 255);
    float cc = 0.f;
    if ((double)colour > a.thresC) cc = (float)min(max(__float2int_rn(__fadd_rn(__fmul_rn(255.0f, a.betaC), (float)colour)), 0), 255);
    const float grad = __fadd_rn(__fmul_rn(__fadd_rn(g[0], g[1]), a.third), __fmul_rn(g[2], a.third));
    const float cg = grad > a.tG ? __fadd_rn(__fmul_rn(254.0f, a.tG), grad) : __fmul_rn(255.0f, a.tG);
    a.out[(size_t)k * a.sliceStride + (size_t)oy * a.outPitch + ox] = __fadd_rn(__fmul_rn(cc, a.wC), __fmul_rn(cg, a.wG));
}
}  // namespace

size_t md_similarity_bytes(int H, int W, int minD, int numD)
{
    const int Wp = W + minD + numD - 1;
    return md_align((size_t)H * md_align((size_t)Wp * 3, 256), 256) + md_align((size_t)H * W * 12, 256) + md_align((size_t)H * Wp * 12, 256);
}
// dL / dR: interleaved 3-channel views (pitch bytes).  dCost: numD slices of (H + 2 border) x (W + 2 border) floats.
int md_compute_similarity(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, int H, int W, double regularity, double thresC, double thresG,
                          int minD, int numD, int border, float* dCost, void* ws)
{
    MD_REQUIRE(numD > 0 && minD >= 0 && border >= 0, MDISP_E_BADARG, "bad disparity range");
    const int maxOff = minD + numD - 1, Wp = W + maxOff;
    const size_t pP = md_align((size_t)Wp * 3, 256);
    u8* base = (u8*)ws;
    auto take = [&](size_t bytes) { u8* p = base; base += md_align(bytes, 256); return p; };
    u8* Rb = take((size_t)H * pP);
    float* sL = (float*)take((size_t)H * W * 12);
    float* sR = (float*)take((size_t)H * Wp * 12);
    mc_pad_reflect_kernel<<<dim3((Wp + 255) / 256, H), 256, 0, s>>>(dR, pitch, W, 3, maxOff, Wp, Rb, pP);
    mc_sobel3_kernel<<<dim3((W + 255) / 256, H), 256, 0, s>>>(dL, pitch, H, W, sL);
    mc_sobel3_kernel<<<dim3((Wp + 255) / 256, H), 256, 0, s>>>(Rb, pP, H, Wp, sR);
    McSimArgs a{};
    a.L = dL; a.lp = pitch; a.Rb = Rb; a.rp = pP; a.sL = sL; a.sR = sR; a.H = H; a.W = W; a.Wp = Wp; a.maxOff = maxOff; a.minD = minD;
    a.third = (float)(1.0 / 3.0); a.betaC = (float)(thresC / 255.0); a.tG = (float)thresG;
    a.wC = (float)(1.0 - regularity); a.wG = (float)regularity; a.thresC = thresC;
    a.out = dCost; a.border = border; a.outPitch = W + 2 * border; a.sliceStride = (size_t)(H + 2 * border) * (W + 2 * border);
    mc_similarity_kernel<<<dim3((W + 2 * border + 255) / 256, H + 2 * border, numD), 256, 0, s>>>(a);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// computeAdaptiveWeight_GuidedF_2 (which = 2) / _3 (which = 3) on interleaved 3-channel device views
size_t md_asw_guidedf23_bytes(int H, int W, int which, int isLeft, int minD, int numD)
{
    const int Wp = W + minD + numD - 1;
    const size_t N = (size_t)H * W;
    size_t b = md_align(N * 4 * numD, 256);
    if (which == 2) b += md_similarity_bytes(H, W, minD, numD) + md_mc_scratch_bytes(H, W, 3, numD, W, 0);
    else b += md_align((size_t)H * md_align(W, 256), 256) * 2 + md_ncc_bytes(H, W, minD, numD) + md_align((size_t)H * md_align((size_t)Wp * 3, 256), 256) +
              md_mc_scratch_bytes(H, W, isLeft ? 6 : 3, numD, isLeft ? W : W, isLeft ? Wp : 0);
    return b + 4096;
}
int md_asw_guidedf23(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, int H, int W, int which, int isLeft, double eps, int win,
                     int minD, int numD, float* dDisp, void* ws)
{
    MD_REQUIRE(which == 2 || which == 3, MDISP_E_BADARG, "unknown matcher");
    MD_REQUIRE(numD > 0 && minD >= 0, MDISP_E_BADARG, "bad disparity range");
    MD_REQUIRE(win > 0 && win % 2 == 1, MDISP_E_BADARG, "winsize must be odd");
    // computeSimilarity runs for colour input and DISPARITY_LEFT only: its other branches multiply a CV_32F matrix by a
    // CV_8U mask, which the library refuses (method_StereoMatching.cpp:1218, 1257, 1292)
    MD_REQUIRE(which != 2 || isLeft, MDISP_E_UNSUPPORTED, "computeAdaptiveWeight_GuidedF_2: computeSimilarity fails for DISPARITY_RIGHT in the reference");
    const size_t N = (size_t)H * W;
    const int maxOff = minD + numD - 1, Wp = W + maxOff;
    u8* base = (u8*)ws;
    auto take = [&](size_t bytes) { u8* p = base; base += md_align(bytes, 256); return p; };
    float* vol = (float*)take(N * 4 * numD);
    McGuideHost g{};
    if (which == 2) {
        void* sim = take(md_similarity_bytes(H, W, minD, numD));
        MD_TRY(md_compute_similarity(s, dL, dR, pitch, H, W, 0.4, 10.0, 50.0, minD, numD, 0, vol, sim));        // :3808
        g.a = dL; g.ap = pitch; g.acn = 3; g.aW = W;                                                             // :3822
    } else {
        const size_t pG = md_align(W, 256), pP = md_align((size_t)Wp * 3, 256);
        u8* gl = take((size_t)H * pG); u8* gr = take((size_t)H * pG);
        void* nccWs = take(md_ncc_bytes(H, W, minD, numD));
        u8* othP = take((size_t)H * pP);
        MD_TRY(md_bgr2gray(s, dL, pitch, H, W, gl, pG, 1));          // computeNCC converts with COLOR_RGB2GRAY (:934-942)
        MD_TRY(md_bgr2gray(s, dR, pitch, H, W, gr, pG, 1));
        MD_TRY(md_ncc(s, gl, gr, pG, H, W, isLeft, win, minD, numD, nccWs, nullptr, vol));                       // :3895
        if (isLeft) {      // the six-channel guide of computeAdaptiveWeight_GuidedF (:3902-3909)
            mc_pad_reflect_kernel<<<dim3((Wp + 255) / 256, H), 256, 0, s>>>(dR, pitch, W, 3, maxOff, Wp, othP, pP);
            g.a = dL; g.ap = pitch; g.acn = 3; g.aW = W;
            g.b = othP; g.bp = pP; g.bcn = 3; g.bW = Wp; g.bOff0 = maxOff - minD; g.bOffStep = -1;
        } else { g.a = dR; g.ap = pitch; g.acn = 3; g.aW = W; }      // getGuidedFilter(rightImg, ...) (:3924)
    }
    return md_mc_filter_wta(s, g, H, W, win, eps, 1, numD, minD, vol, nullptr, 0, nullptr, dDisp, base);
}

// ---- C ABI ----------------------------------------------------------------------------------------------------------

namespace {
__global__ __launch_bounds__(256) void mc_u8_to_f32_kernel(const u8* __restrict__ src, size_t sp, int H, int W, float* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < W) dst[(size_t)y * W + x] = (float)src[(size_t)y * sp + x];
}
struct McCarver {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off = md_align(off + bytes, 256); return o; }
};
int mc_check_u8(const mdisp_image* im, const char* what)
{
    MD_REQUIRE(im && im->data && im->rows > 0 && im->cols > 0, MDISP_E_BADARG, "%s: empty image", what);
    MD_REQUIRE(im->depth == MDISP_8U, MDISP_E_UNSUPPORTED, "%s: only CV_8U images are supported (got depth %d)", what, im->depth);
    MD_REQUIRE(im->step_bytes >= (size_t)im->cols * im->channels, MDISP_E_BADARG, "%s: step smaller than a row", what);
    return MDISP_OK;
}
int mc_check_f32_out(const mdisp_image* im, int rows, int cols, const char* what)
{
    MD_REQUIRE(im && im->data && im->rows == rows && im->cols == cols && im->depth == MDISP_32F && im->channels == 1 &&
               im->step_bytes >= (size_t)cols * 4, MDISP_E_BADARG, "%s: expected a %dx%d CV_32FC1 image", what, cols, rows);
    return MDISP_OK;
}
}  // namespace

extern "C" int mdisp_adaptive_weight_guidedf(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, int is_left, double eps,
                                             int winSize, int minD, int numD, mdisp_image* out)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    MD_TRY(mc_check_u8(L, "left"));
    MD_TRY(mc_check_u8(R, "right"));
    MD_REQUIRE(L->rows == R->rows && L->cols == R->cols && L->channels == R->channels, MDISP_E_BADARG, "left/right size mismatch");
    MD_REQUIRE(L->channels == 3, MDISP_E_UNSUPPORTED,
               "computeAdaptiveWeight_GuidedF needs 3-channel images (its 2-channel guide of a gray pair fails in the reference)");
    MD_REQUIRE(numD > 0 && minD >= 0, MDISP_E_BADARG, "bad disparity range");
    MD_REQUIRE(winSize > 0 && winSize % 2 == 1, MDISP_E_BADARG, "winsize must be odd");
    MD_TRY(md_mc_check_window(winSize));
    const int H = L->rows, W = L->cols, cn = 3;
    MD_TRY(mc_check_f32_out(out, H, W, "out"));
    MD_HIP(hipSetDevice(c->device));
    const size_t p3 = md_align((size_t)W * cn, 256);
    McCarver cv;
    const size_t o_l = cv.take(p3 * H), o_r = cv.take(p3 * H), o_d = cv.take((size_t)H * W * 4), o_ws = cv.take(md_asw_guidedf_bytes(H, W, cn, minD, numD));
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    MD_TRY(md_h2d_2d(c, base + o_l, p3, L->data, L->step_bytes, (size_t)W * cn, H));
    MD_TRY(md_h2d_2d(c, base + o_r, p3, R->data, R->step_bytes, (size_t)W * cn, H));
    MD_TRY(md_asw_guidedf(c->stream, base + o_l, base + o_r, p3, H, W, cn, is_left != 0, eps, winSize, minD, numD, (float*)(base + o_d), base + o_ws));
    MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, base + o_d, (size_t)W * 4, (size_t)W * 4, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_cost_sad_d(mdisp_ctx* c, const mdisp_image* left, const mdisp_image* right, int disparity, int is_left, int winSize,
                                mdisp_image* out)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    MD_TRY(mc_check_u8(left, "left"));
    MD_TRY(mc_check_u8(right, "right"));
    MD_REQUIRE((left->channels == 1 || left->channels == 3) && (right->channels == 1 || right->channels == 3), MDISP_E_UNSUPPORTED,
               "only 1- or 3-channel images are supported");
    MD_REQUIRE(left->rows == right->rows, MDISP_E_BADARG, "left/right height mismatch");
    MD_REQUIRE(winSize > 0 && winSize % 2 == 1, MDISP_E_BADARG, "winsize must be odd");
    const mdisp_image* ref = is_left ? left : right;
    const mdisp_image* oth = is_left ? right : left;
    const int H = ref->rows, W = ref->cols, Wo = oth->cols;
    // the reference returns an empty image unless the other view was bordered (:3318-3321, :3333-3336)
    MD_REQUIRE(Wo > W, MDISP_E_BADARG, "the other view must be the bordered (wider) one");
    const int off = is_left ? Wo - W - disparity : disparity;
    MD_REQUIRE(off >= 0 && off + W <= Wo, MDISP_E_BADARG, "disparity %d leaves the bordered view", disparity);
    MD_TRY(mc_check_f32_out(out, H, W, "out"));
    MD_HIP(hipSetDevice(c->device));
    const size_t pr = md_align(W, 256), po = md_align(Wo, 256), p3 = md_align((size_t)Wo * 3, 256);
    McCarver cv;
    const size_t o_r = cv.take(pr * H), o_o = cv.take(po * H), o_st = cv.take(p3 * H), o_d = cv.take((size_t)H * W * 4);
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    auto up = [&](const mdisp_image* im, u8* d, size_t dp) -> int {
        if (im->channels == 1) return md_h2d_2d(c, d, dp, im->data, im->step_bytes, im->cols, im->rows);
        MD_TRY(md_h2d_2d(c, base + o_st, p3, im->data, im->step_bytes, (size_t)im->cols * 3, im->rows));
        return md_bgr2gray(c->stream, base + o_st, p3, im->rows, im->cols, d, dp);
    };
    MD_TRY(up(ref, base + o_r, pr));
    MD_TRY(up(oth, base + o_o, po));
    MD_TRY(md_cost_sad_d(c->stream, base + o_r, pr, base + o_o, po, H, W, off, winSize, (float*)(base + o_d)));
    MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, base + o_d, (size_t)W * 4, (size_t)W * 4, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_guided_filter_multi(mdisp_ctx* c, const mdisp_image* guide, const mdisp_image* P, int r, double eps, int variant,
                                         mdisp_image* out)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    MD_TRY(mc_check_u8(guide, "guide"));
    MD_REQUIRE(P && P->data && P->channels == 1 && (P->depth == MDISP_8U || P->depth == MDISP_32F), MDISP_E_UNSUPPORTED,
               "P: a single-channel CV_8U or CV_32F image is expected");
    MD_REQUIRE(guide->rows == P->rows && guide->cols == P->cols, MDISP_E_BADARG, "guide / input size mismatch");
    // gifs.cpp:101-125 builds its regulariser for 1, 3 or 6 channels only; stereomatch_1's multiChl_to_oneChl_mul handles 3 and 6
    MD_REQUIRE(guide->channels == 3 || guide->channels == 6, MDISP_E_UNSUPPORTED,
               "guide with %d channels (3 or 6; single-channel guides go through mdisp_guided_filter)", guide->channels);
    MD_REQUIRE(variant == MDISP_GIF_MC_GIFS || variant == MDISP_GIF_MC_STEREOMATCH, MDISP_E_BADARG, "unknown variant %d", variant);
    MD_TRY(md_mc_check_window(r));
    const int H = guide->rows, W = guide->cols, cn = guide->channels;
    MD_TRY(mc_check_f32_out(out, H, W, "out"));
    MD_HIP(hipSetDevice(c->device));
    const size_t pg = md_align((size_t)W * cn, 256), pp = md_align(W, 256);
    McCarver cv;
    const size_t o_g = cv.take(pg * H), o_p8 = cv.take(pp * H), o_p = cv.take((size_t)H * W * 4), o_d = cv.take((size_t)H * W * 4),
                 o_ws = cv.take(md_mc_scratch_bytes(H, W, cn, 1, W, 0));
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    MD_TRY(md_h2d_2d(c, base + o_g, pg, guide->data, guide->step_bytes, (size_t)W * cn, H));
    if (P->depth == MDISP_8U) {
        MD_TRY(md_h2d_2d(c, base + o_p8, pp, P->data, P->step_bytes, W, H));
        mc_u8_to_f32_kernel<<<dim3((W + 255) / 256, H), 256, 0, c->stream>>>(base + o_p8, pp, H, W, (float*)(base + o_p));
    } else {
        MD_TRY(md_h2d_2d(c, base + o_p, (size_t)W * 4, P->data, P->step_bytes, (size_t)W * 4, H));
    }
    MD_TRY(md_guided_filter_mc(c->stream, base + o_g, pg, cn, (const float*)(base + o_p), H, W, r, eps, variant, (float*)(base + o_d), base + o_ws));
    MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, base + o_d, (size_t)W * 4, (size_t)W * 4, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_compute_similarity(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, int is_left, double regularity, double thresC,
                                        double thresG, int winSize, int minD, int numD, float* cost)
{
    MD_REQUIRE(c && cost, MDISP_E_BADARG, "null argument");
    MD_TRY(mc_check_u8(L, "left"));
    MD_TRY(mc_check_u8(R, "right"));
    MD_REQUIRE(L->rows == R->rows && L->cols == R->cols, MDISP_E_BADARG, "left/right size mismatch");
    MD_REQUIRE(winSize == 0 || winSize % 2 == 1, MDISP_E_BADARG, "winsize must be odd");
    MD_REQUIRE(numD > 0 && minD >= 0, MDISP_E_BADARG, "bad disparity range");
    // every other branch of the reference multiplies a CV_32F matrix by a CV_8U mask and fails in the library
    MD_REQUIRE(L->channels == 3 && R->channels == 3 && is_left, MDISP_E_UNSUPPORTED,
               "computeSimilarity runs in the reference for 3-channel views and DISPARITY_LEFT only");
    const int H = L->rows, W = L->cols, border = winSize / 2;
    MD_HIP(hipSetDevice(c->device));
    const size_t p3 = md_align((size_t)W * 3, 256), outN = (size_t)(H + 2 * border) * (W + 2 * border) * numD;
    McCarver cv;
    const size_t o_l = cv.take(p3 * H), o_r = cv.take(p3 * H), o_c = cv.take(outN * 4), o_ws = cv.take(md_similarity_bytes(H, W, minD, numD));
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    MD_TRY(md_h2d_2d(c, base + o_l, p3, L->data, L->step_bytes, (size_t)W * 3, H));
    MD_TRY(md_h2d_2d(c, base + o_r, p3, R->data, R->step_bytes, (size_t)W * 3, H));
    MD_TRY(md_compute_similarity(c->stream, base + o_l, base + o_r, p3, H, W, regularity, thresC, thresG, minD, numD, border, (float*)(base + o_c),
                                 base + o_ws));
    MD_HIP(hipMemcpyAsync(cost, base + o_c, outN * 4, hipMemcpyDeviceToHost, c->stream));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}

extern "C" int mdisp_adaptive_weight_guidedf_n(mdisp_ctx* c, const mdisp_image* L, const mdisp_image* R, int which, int is_left, double eps,
                                               int winSize, int minD, int numD, mdisp_image* out)
{
    if (which == 1) return mdisp_adaptive_weight_guidedf(c, L, R, is_left, eps, winSize, minD, numD, out);
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    MD_REQUIRE(which == 2 || which == 3, MDISP_E_BADARG, "which: 1, 2 or 3");
    MD_TRY(mc_check_u8(L, "left"));
    MD_TRY(mc_check_u8(R, "right"));
    MD_REQUIRE(L->rows == R->rows && L->cols == R->cols && L->channels == R->channels, MDISP_E_BADARG, "left/right size mismatch");
    MD_REQUIRE(L->channels == 3, MDISP_E_UNSUPPORTED, "the guided-filter matchers need 3-channel views (their guides fail on others in the reference)");
    MD_REQUIRE(numD > 0 && minD >= 0, MDISP_E_BADARG, "bad disparity range");
    MD_REQUIRE(winSize > 0 && winSize % 2 == 1, MDISP_E_BADARG, "winsize must be odd");
    MD_REQUIRE(which != 2 || is_left, MDISP_E_UNSUPPORTED, "computeAdaptiveWeight_GuidedF_2: computeSimilarity fails for DISPARITY_RIGHT in the reference");
    MD_TRY(md_mc_check_window(winSize));
    const int H = L->rows, W = L->cols;
    MD_TRY(mc_check_f32_out(out, H, W, "out"));
    MD_HIP(hipSetDevice(c->device));
    const size_t p3 = md_align((size_t)W * 3, 256);
    McCarver cv;
    const size_t o_l = cv.take(p3 * H), o_r = cv.take(p3 * H), o_d = cv.take((size_t)H * W * 4),
                 o_ws = cv.take(md_asw_guidedf23_bytes(H, W, which, is_left != 0, minD, numD));
    u8* base = (u8*)mdisp_ws(c, cv.off);
    if (!base) return MDISP_E_NOMEM;
    MD_TRY(md_h2d_2d(c, base + o_l, p3, L->data, L->step_bytes, (size_t)W * 3, H));
    MD_TRY(md_h2d_2d(c, base + o_r, p3, R->data, R->step_bytes, (size_t)W * 3, H));
    MD_TRY(md_asw_guidedf23(c->stream, base + o_l, base + o_r, p3, H, W, which, is_left != 0, eps, winSize, minD, numD, (float*)(base + o_d), base + o_ws));
    MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, base + o_d, (size_t)W * 4, (size_t)W * 4, H));
    MD_HIP(hipStreamSynchronize(c->stream));
    return MDISP_OK;
}
