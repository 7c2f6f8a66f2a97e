// Guided-image-filter family used as cost-slice aggregator, the steerable-filter weight map, and
// the asw_gifs / censusStereo(GUIDED) drivers, for gfx950.
//
// Replaces cameraCalibration/disp_method/gifs.cpp:47-135 (getGuidedFilter), :227-273 (_egif),
// :347-409 (edgeAwareWeight, _wgif), :537-617 (_skwgif2), disp_method/steerFilter.cpp:239-446
// (steerFilter) and the per-slice loop + WTA of disp_method/methods_disp.cpp:1181-1215,1278-1295
// and :969-1000.
//
// Numerics follow the reference's rounding points: images are float, every cv::Mat expression
// rounds to float per operation (no fused multiply-add), cv::boxFilter accumulates in double and
// rounds the scaled sum to float.  Slices are processed in batches of planar [n][H][W] arrays; all
// guide-only terms (means, variance, regularisation map, steer weights) are computed once per
// guide image, not once per disparity.
#include "common.h"
#include "box_march.h"
#include <math.h>
#include <float.h>
#include <vector>


// ---------------------------------------------------------------------------------------------
// min / max and normalisation (cv::normalize(src, dst, 0, 1, NORM_MINMAX, CV_32F))

// order-preserving map float -> uint32 so atomicMin/Max work on floats of either sign
__device__ __forceinline__ u32 f2ord(float f) { u32 u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(u32 o) { u32 u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; return __uint_as_float(u); }

// mm[2*k] = min, mm[2*k+1] = max, both in the ordered-uint domain; init to (0xFFFFFFFF, 0)
__global__ void mm_init_kernel(u32* mm, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { mm[2 * i] = 0xFFFFFFFFu; mm[2 * i + 1] = 0u; }
}

__device__ __forceinline__ void block_minmax_commit(u32 lo, u32 hi, u32* mm)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        lo = min(lo, (u32)__shfl_xor((int)lo, m, 64));
        hi = max(hi, (u32)__shfl_xor((int)hi, m, 64));
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }
}

template <typename T>
__global__ __launch_bounds__(256) void minmax_kernel(const T* __restrict__ src, size_t n_per_slice, u32* mm)
{
    const int k = blockIdx.y;
    const T* s = src + (size_t)k * n_per_slice;
    u32 lo = 0xFFFFFFFFu, hi = 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_per_slice; i += (size_t)gridDim.x * 256) {
        u32 o = f2ord((float)s[i]);
        lo = min(lo, o); hi = max(hi, o);
    }
    block_minmax_commit(lo, hi, mm + 2 * k);
}

struct CostSrc {
    const u8* ref; size_t rp; const u8* pad; size_t pp;
    int W, H, off0, offStep, kind;   // kind: MD_COST_HAMMING or MD_COST_ABSDIFF
};
__device__ __forceinline__ int raw_cost(const CostSrc& c, int x, int y, int k)
{
    const int r = c.ref[(size_t)y * c.rp + x], p = c.pad[(size_t)y * c.pp + x + c.off0 + k * c.offStep];
    if (c.kind == MD_COST_HAMMING) return __popc((u32)(r ^ p));
    return r > p ? r - p : p - r;
}

// min/max of integer cost slices k0 .. k0+n-1 computed on the fly.  A workgroup owns a tile of
// 256 columns x MM_ROWS rows and walks the slices with the tile's reference pixels in registers, so
// each census / gray byte is fetched once and the shifted operand rows stay in cache.
#define MM_ROWS 16
#define MM_SLICES 4     // slices per workgroup (grid z): a tile's slices spread over several workgroups so that the launch fills the device
__global__ __launch_bounds__(256) void cost_minmax_kernel(CostSrc c, int k0, int n, u32* __restrict__ part, int yBeg, int yEnd)
{
    __shared__ u32 slo[4], shi[4];
    const int x = blockIdx.x * 256 + threadIdx.x, y0 = yBeg + blockIdx.y * MM_ROWS;
    const bool inx = x < c.W;
    const int rows = min(MM_ROWS, yEnd - y0);
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    int ref[MM_ROWS];
#pragma unroll
    for (int j = 0; j < MM_ROWS; j++) ref[j] = (inx && j < rows) ? c.ref[(size_t)(y0 + j) * c.rp + x] : 0;
    const int kEnd = min(n, (int)(blockIdx.z + 1) * MM_SLICES);
    for (int k = blockIdx.z * MM_SLICES; k < kEnd; k++) {
        int lo = 0x7FFFFFFF, hi = -1;
        if (inx) {
            const u8* p = c.pad + (size_t)y0 * c.pp + x + c.off0 + (k0 + k) * c.offStep;
#pragma unroll
            for (int j = 0; j < MM_ROWS; j++) {
                if (j < rows) {
                    const int pv = p[(size_t)j * c.pp];
                    const int v = c.kind == MD_COST_HAMMING ? __popc((u32)(ref[j] ^ pv)) : abs(ref[j] - pv);
                    lo = min(lo, v); hi = max(hi, v);
                }
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { lo = min(lo, __shfl_xor(lo, m, 64)); hi = max(hi, __shfl_xor(hi, m, 64)); }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = (u32)lo; shi[threadIdx.x >> 6] = (u32)(hi + 1); }
        __syncthreads();
        if (threadIdx.x == 0) {
            // per-workgroup partial: min as is, max + 1 so that 0 means "no pixel"
            part[((size_t)wg * n + k) * 2] = min(min(slo[0], slo[1]), min(slo[2], slo[3]));
            part[((size_t)wg * n + k) * 2 + 1] = max(max(shi[0], shi[1]), max(shi[2], shi[3]));
        }
    }
}
// combine the partials of all workgroups; costs are small non-negative integers, whose float
// images order like the integers
__global__ __launch_bounds__(256) void cost_minmax_finish_kernel(const u32* __restrict__ part, int nwg, int n, u32* mm)
{
    __shared__ u32 slo[256], shi[256];
    const int k = blockIdx.x;
    u32 lo = 0x7FFFFFFFu, hi = 0u;
    for (int w = threadIdx.x; w < nwg; w += 256) {
        lo = min(lo, part[((size_t)w * n + k) * 2]); hi = max(hi, part[((size_t)w * n + k) * 2 + 1]);
    }
    slo[threadIdx.x] = lo; shi[threadIdx.x] = hi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { slo[threadIdx.x] = min(slo[threadIdx.x], slo[threadIdx.x + o]); shi[threadIdx.x] = max(shi[threadIdx.x], shi[threadIdx.x + o]); }
        __syncthreads();
    }
    if (threadIdx.x == 0 && shi[0] > 0) { mm[2 * k] = f2ord((float)slo[0]); mm[2 * k + 1] = f2ord((float)(shi[0] - 1)); }
}

// convertTo(dst, CV_32F, scale, shift): float arithmetic for 8U / 32F sources, double for 32S
enum { NORM_FLOAT = 0, NORM_DOUBLE = 1 };
__device__ __forceinline__ float norm_apply(float v, u32 olo, u32 ohi, int mode)
{
    const double smin = (double)ord2f(olo), smax = (double)ord2f(ohi);
    const double scale = (smax - smin > DBL_EPSILON) ? 1.0 / (smax - smin) : 0.0;
    const double shift = 0.0 - smin * scale;
    if (mode == NORM_DOUBLE) return (float)((double)v * scale + shift);
    return __fadd_rn(__fmul_rn(v, (float)scale), (float)shift);
}

template <typename T>
__global__ __launch_bounds__(256) void normalize_kernel(const T* __restrict__ src, size_t n_per_slice, const u32* mm, int mode,
                                                        float* __restrict__ dst)
{
    const int k = blockIdx.y;
    const u32 lo = mm[2 * k], hi = mm[2 * k + 1];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_per_slice; i += (size_t)gridDim.x * 256)
        dst[(size_t)k * n_per_slice + i] = norm_apply((float)src[(size_t)k * n_per_slice + i], lo, hi, mode);
}

// strided 8-bit image -> planar normalised float
__global__ __launch_bounds__(256) void normalize_u8_image_kernel(const u8* __restrict__ src, size_t sp, int H, int W, const u32* mm,
                                                                 float* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    dst[(size_t)y * W + x] = norm_apply((float)src[(size_t)y * sp + x], mm[0], mm[1], NORM_FLOAT);
}
__global__ __launch_bounds__(256) void minmax_u8_image_kernel(const u8* __restrict__ src, size_t sp, int H, int W, u32* mm)
{
    u32 lo = 0xFFFFFFFFu, hi = 0u;
    for (int y = blockIdx.y; y < H; y += gridDim.y)
        for (int x = blockIdx.x * 256 + threadIdx.x; x < W; x += gridDim.x * 256) {
            u32 o = f2ord((float)src[(size_t)y * sp + x]);
            lo = min(lo, o); hi = max(hi, o);
        }
    block_minmax_commit(lo, hi, mm);
}

// ---------------------------------------------------------------------------------------------
// separable box sums with double accumulators (cv::boxFilter's RowSum<float,double> /
// ColumnSum<double,float>).  mode 0: centred window k with BORDER_REFLECT_101.  mode 1: the
// clipped half-open window of getGuidedFilter_skwgif2, [p - k/2, p + k/2) intersected with
// [0, n-1)  (gifs.cpp:591-592: Rect(Point(max(0,i-r2/2), ...), Point(min(W-1,i+r2/2), ...))).

#define BH_SEG 1024   // outputs per workgroup along x
// out[k][y][x] = sum over the window of A[k][y][.] (times B[.][y][.], rounded to float first, when B != null)
__global__ __launch_bounds__(256) void box_h_kernel(const float* __restrict__ A, size_t aStride, const float* __restrict__ B,
                                                    size_t bStride, int H, int W, int ksz, int mode, double* __restrict__ out,
                                                    int yOff = 0)
{
    extern __shared__ float seg[];   // BH_SEG + ksz values
    const int y = yOff + blockIdx.y, k = blockIdx.z, x0 = blockIdx.x * BH_SEG;
    const int ax = ksz / 2;
    const float* a = A + (size_t)k * aStride + (size_t)y * W;
    const float* b = B ? B + (size_t)k * bStride + (size_t)y * W : nullptr;
    const int nload = min(BH_SEG, W - x0) + ksz;
    for (int i = threadIdx.x; i < nload; i += 256) {
        int xs = x0 - ax + i;
        float v = 0.f;
        if (mode == 0) { xs = r101(xs, W); v = b ? __fmul_rn(a[xs], b[xs]) : a[xs]; }
        else if (xs >= 0 && xs < W - 1) v = b ? __fmul_rn(a[xs], b[xs]) : a[xs];
        seg[i] = v;
    }
    __syncthreads();
    // mode 1 window is [x-ax, x+ax): ksz-1 (even) or ksz (odd ksz: 2*ax = ksz-1) taps
    const int taps = mode == 0 ? ksz : 2 * ax;
    const int xb = threadIdx.x * 4;
    if (x0 + xb >= W) return;
    double s = 0;
    for (int i = 0; i < taps; i++) s += (double)seg[xb + i];
    double* o = out + ((size_t)k * H + y) * W + x0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (x0 + xb + j >= W) break;
        if (j > 0) s += (double)seg[xb + j - 1 + taps] - (double)seg[xb + j - 1];
        o[xb + j] = s;
    }
}

#define BV_SEG 16   // outputs per thread along y
// out[k][y][x] = (float)(sum over the window of in[k][.][x] * scale)            (div == null)
//              = (float)(sum / div[y][x])                                        (div != null)
__global__ __launch_bounds__(256) void box_v_kernel(const double* __restrict__ in, int H, int W, int ksz, int mode, double scale,
                                                    const double* __restrict__ div, float* __restrict__ out)
{
    const int x = blockIdx.x * 256 + threadIdx.x, k = blockIdx.z, y0 = blockIdx.y * BV_SEG;
    if (x >= W) return;
    const int ay = ksz / 2;
    const int taps = mode == 0 ? ksz : 2 * ay;
    const double* col = in + (size_t)k * H * W + x;
    auto tap = [&](int yy) -> double {
        if (mode == 0) return col[(size_t)r101(yy, H) * W];
        return (yy >= 0 && yy < H - 1) ? col[(size_t)yy * W] : 0.0;
    };
    double s = 0;
    for (int i = 0; i < taps; i++) s += tap(y0 - ay + i);
    for (int j = 0; j < BV_SEG; j++) {
        const int y = y0 + j;
        if (y >= H) break;
        if (j > 0) s += tap(y - 1 - ay + taps) - tap(y - 1 - ay);
        const size_t o = ((size_t)k * H + y) * W + x;
        out[o] = div ? (float)(s * 1.0 / div[(size_t)y * W + x]) : (float)(s * scale);
    }
}

// same vertical pass but keeping the double sum (used for the steer-weight normaliser)
__global__ __launch_bounds__(256) void box_v_f64_kernel(const double* __restrict__ in, int H, int W, int ksz, int mode,
                                                        double* __restrict__ out, int yBeg, int yEnd)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y0 = yBeg + blockIdx.y * BV_SEG;
    if (x >= W) return;
    const int ay = ksz / 2;
    const int taps = mode == 0 ? ksz : 2 * ay;
    const double* col = in + x;
    auto tap = [&](int yy) -> double {
        if (mode == 0) return col[(size_t)r101(yy, H) * W];
        return (yy >= 0 && yy < H - 1) ? col[(size_t)yy * W] : 0.0;
    };
    double s = 0;
    for (int i = 0; i < taps; i++) s += tap(y0 - ay + i);
    for (int j = 0; j < BV_SEG; j++) {
        const int y = y0 + j;
        if (y >= yEnd) break;
        if (j > 0) s += tap(y - 1 - ay + taps) - tap(y - 1 - ay);
        out[(size_t)y * W + x] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// element-wise stages

// varG = corrG - meanG*meanG   (gifs.cpp:80)
__global__ __launch_bounds__(256) void var_kernel(const float* __restrict__ corr, const float* __restrict__ mean, size_t N,
                                                  float* __restrict__ var)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) var[i] = __fsub_rn(corr[i], __fmul_rn(mean[i], mean[i]));
}

// den = var + add              (GIF / EGIF: add is a scalar)
__global__ __launch_bounds__(256) void den_scalar_kernel(const float* __restrict__ var, size_t N, float add, float* __restrict__ den)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) den[i] = __fadd_rn(var[i], add);
}

// SKWGIF: temp = var2 + namuda;  writes temp and 1/temp (for the global sum)   (gifs.cpp:564-565)
__global__ __launch_bounds__(256) void skw_temp_kernel(const float* __restrict__ var2, size_t N, float namuda,
                                                       float* __restrict__ temp, float* __restrict__ inv)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) { float t = __fadd_rn(var2[i], namuda); temp[i] = t; inv[i] = __fdiv_rn(1.0f, t); }
}
// den = var + eps / (temp * gscale)                                            (gifs.cpp:565,570)
__global__ __launch_bounds__(256) void skw_den_kernel(const float* __restrict__ var, const float* __restrict__ temp, size_t N,
                                                      float gscale, float epsf, float* __restrict__ den)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) den[i] = __fadd_rn(var[i], __fdiv_rn(epsf, __fmul_rn(temp[i], gscale)));
}

__global__ __launch_bounds__(256) void f64_to_f32_kernel(const double* __restrict__ a, size_t n, float* __restrict__ o)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = (float)a[i];
}

// per-block partial sums in double, combined on the host in block order (deterministic)
__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ src, size_t N, double* __restrict__ part)
{
    __shared__ double sh[256];
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (size_t)gridDim.x * 256) s += (double)src[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

// a = (corrGP - meanG*meanP) / den ;  b = meanP - a*meanG        (gifs.cpp:96,127-128)
// in: meanP, corrGP (per slice), meanG, den (per guide);  out: a overwrites corrGP's slot, b meanP's
__global__ __launch_bounds__(256) void ab_kernel(float* __restrict__ meanP_b, float* __restrict__ corr_a,
                                                 const float* __restrict__ meanG, const float* __restrict__ den, size_t N)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const size_t o = (size_t)blockIdx.y * N + i;
    const float mp = meanP_b[o], mg = meanG[i];
    const float cov = __fsub_rn(corr_a[o], __fmul_rn(mg, mp));
    const float a = __fdiv_rn(cov, den[i]);
    corr_a[o] = a;
    meanP_b[o] = __fsub_rn(mp, __fmul_rn(a, mg));
}

// w*a products for the steer-weighted mean are formed inside box_h (B = steer weights)

// out = a_bar*G + b_bar (gifs.cpp:133); optional store of the filtered slice and of
// beta = a_bar / (1 - a_bar) (gifs.cpp:261); optional first-minimum WTA over the n slices of the
// batch against the running best (strict '<': methods_disp.cpp:1287, minMaxLoc at :996).
struct OutArgs {
    const float* v;                                       // filtered slices (a_bar*G + b_bar), or null to form them here
    const float* abar; const float* bbar; const float* G; size_t N; int n;
    size_t i0, i1;                                        // pixel range [i0, i1) to process
    float* filtered; float* beta;
    float* best; float* disp32; double* disp64; int d0;   // d0 = disparity of the batch's first slice
    int first;                                            // 1: the running best is uninitialised
};
__global__ __launch_bounds__(256) void out_wta_kernel(OutArgs a)
{
    const size_t i = a.i0 + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.i1) return;
    const float g = a.v ? 0.f : a.G[i];
    float best = 0.f; int bestd = 0; bool have = false;
    if (a.best && !a.first) { best = a.best[i]; have = true; bestd = a.disp32 ? (int)a.disp32[i] : (int)a.disp64[i]; }
    for (int k = 0; k < a.n; k++) {
        const size_t o = (size_t)k * a.N + i;
        float v;
        if (a.v) v = a.v[o];
        else {
            const float ab = a.abar[o];
            v = __fadd_rn(__fmul_rn(ab, g), a.bbar[o]);
            if (a.filtered) a.filtered[o] = v;
            if (a.beta) a.beta[o] = __fdiv_rn(ab, __fsub_rn(1.0f, ab));
        }
        if (a.best) {
            if (!have || v < best) { best = v; bestd = a.d0 + k; have = true; }
        }
    }
    if (a.best) {
        a.best[i] = best;
        if (a.disp32) a.disp32[i] = (float)bestd;
        if (a.disp64) a.disp64[i] = (double)bestd;
    }
}

struct NormScale { float scale, shift; double dscale, dshift; };
__device__ __forceinline__ NormScale norm_scale(u32 olo, u32 ohi)
{
    const double smin = (double)ord2f(olo), smax = (double)ord2f(ohi);
    NormScale n;
    n.dscale = (smax - smin > DBL_EPSILON) ? 1.0 / (smax - smin) : 0.0;
    n.dshift = 0.0 - smin * n.dscale;
    n.scale = (float)n.dscale; n.shift = (float)n.dshift;
    return n;
}

// Load functors: fetch() only issues the memory reads of one element (so that the next row can be
// requested before the current one is consumed), finish() turns them into the two filter inputs.
// P = normalised integer cost computed on the fly; (P, G*P)
struct LoadCostGP {
    CostSrc c; int k0; const u32* mm; int mode; const float* G;
    NormScale ns;
    struct Raw { int r, p; float g; };
    __device__ bool skip(int) const { return false; }
    __device__ void begin(int k) { ns = norm_scale(mm[2 * k], mm[2 * k + 1]); }
    __device__ Raw fetch(int x, int y, int k) const
    {
        Raw w;
        w.r = c.ref[(size_t)y * c.rp + x]; w.p = c.pad[(size_t)y * c.pp + x + c.off0 + (k0 + k) * c.offStep];
        w.g = G[(size_t)y * c.W + x];
        return w;
    }
    __device__ void finish(const Raw& w, float& q0, float& q1) const
    {
        const float v = (float)(c.kind == MD_COST_HAMMING ? __popc((u32)(w.r ^ w.p)) : abs(w.r - w.p));
        const float P = mode == NORM_DOUBLE ? (float)((double)v * ns.dscale + ns.dshift) : __fadd_rn(__fmul_rn(v, ns.scale), ns.shift);
        q0 = P; q1 = __fmul_rn(P, w.g);
    }
    __device__ void operator()(int x, int y, int k, float& q0, float& q1) const { finish(fetch(x, y, k), q0, q1); }
};
// P = already normalised planar slices; (P, G*P)
struct LoadPlanarGP {
    const float* P; size_t N; int W; const float* G;
    struct Raw { float p, g; };
    __device__ bool skip(int) const { return false; }
    __device__ void begin(int) {}
    __device__ Raw fetch(int x, int y, int k) const
    {
        const size_t i = (size_t)y * W + x;
        Raw w; w.p = P[(size_t)k * N + i]; w.g = G[i];
        return w;
    }
    __device__ void finish(const Raw& w, float& q0, float& q1) const { q0 = w.p; q1 = __fmul_rn(w.p, w.g); }
    __device__ void operator()(int x, int y, int k, float& q0, float& q1) const { finish(fetch(x, y, k), q0, q1); }
};
// (a, b), optionally both times the steer weight
struct LoadAB {
    const float* A; const float* B; size_t N; int W; const float* w;
    struct Raw { float a, b, ww; };
    __device__ bool skip(int) const { return false; }
    __device__ void begin(int) {}
    __device__ Raw fetch(int x, int y, int k) const
    {
        const size_t i = (size_t)y * W + x, o = (size_t)k * N + i;
        Raw r; r.a = A[o]; r.b = B[o]; r.ww = w ? w[i] : 1.0f;
        return r;
    }
    __device__ void finish(const Raw& r, float& q0, float& q1) const
    {
        q0 = r.a; q1 = r.b;
        if (w) { q0 = __fmul_rn(q0, r.ww); q1 = __fmul_rn(q1, r.ww); }
    }
    __device__ void operator()(int x, int y, int k, float& q0, float& q1) const { finish(fetch(x, y, k), q0, q1); }
};
// means of P and G*P -> a = (corrGP - meanG*meanP) / den, b = meanP - a*meanG   (gifs.cpp:96,127-128)
struct StoreAB {
    float* A; float* B; size_t N; int W; const float* meanG; const float* den; double scale;
    __device__ void operator()(int x, int y, int k, double s0, double s1) const
    {
        const size_t i = (size_t)y * W + x, o = (size_t)k * N + i;
        const float mp = (float)(s0 * scale), cr = (float)(s1 * scale), mg = meanG[i];
        const float cov = __fsub_rn(cr, __fmul_rn(mg, mp));
        const float a = __fdiv_rn(cov, den[i]);
        A[o] = a; B[o] = __fsub_rn(mp, __fmul_rn(a, mg));
    }
};
// means of a and b -> filtered value a_bar*G + b_bar (gifs.cpp:133), optionally beta = a_bar/(1-a_bar) (:261)
struct StoreOut {
    float* V; size_t N; int W; const float* G; double scale; const double* div; float* beta;
    __device__ void operator()(int x, int y, int k, double s0, double s1) const
    {
        const size_t i = (size_t)y * W + x, o = (size_t)k * N + i;
        const float gv = G[i];                   // requested together with the divisor: one memory round trip, not two
        float ab, bb;
        if (div) { const double d = div[i]; ab = (float)(s0 * 1.0 / d); bb = (float)(s1 * 1.0 / d); }
        else { ab = (float)(s0 * scale); bb = (float)(s1 * scale); }
        V[o] = __fadd_rn(__fmul_rn(ab, gv), bb);
        if (beta) beta[o] = __fdiv_rn(ab, __fsub_rn(1.0f, ab));
    }
};

// ---------------------------------------------------------------------------------------------
// bilateralFilter(a), bilateralFilter(b) of the coefficient planes (gifs.cpp:661-662, 717-718,
// 785-786: sigmaColor 2, sigmaSpace 3) fused with the final a*G + b (:667, 723, 802) and the
// optional beta = a/(1-a) (:792).  Footprint = the taps within `radius` of the centre, visited in
// row-major order with float accumulation like the library's scalar loop; BORDER_REFLECT_101.
// The library interpolates the colour weight from a 4096-bin table of exp(); this kernel evaluates the exponential
// directly (difference below 1e-6 of the weight) -- on this hardware the table's two gathers and its interpolation cost
// more instructions than v_exp_f32 does.  What a tap costs is instructions, so the arithmetic is arranged for few of them:
// the tile holds the planes scaled by s = sqrt(0.5 / sigmaColor^2 * log2 e), so that (p' - c')^2 is the exponent without
// a multiply; the space weight enters as log2(w_s) in the same fused multiply-add, weight = exp2(-(p' - c')^2 + log2 w_s);
// the weighted sum is a fused multiply-add as well; the quotient is scaled back by 1 / s at the end.  Per tap and pair
// of planes: one subtraction, two fused multiply-adds, one addition (packed fp32) and two exponentials.
#define BIL_X 32
#define BIL_Y 16
struct BilArgs {
    const float* A; const float* B; size_t N;
    int H, W, R, ntaps, yA, yB;
    const int2* taps;        // {offset in the (a,b) tile, bits of log2 of the float space weight}
    const float* G; float* V; float* beta;
    float colorS;            // sqrt(0.5/sigmaColor^2 * log2(e))
};
__global__ __launch_bounds__(256) void bilateral_ab_kernel(BilArgs a)
{
    extern __shared__ float2 biltile[];
    const int R = a.R, TW = BIL_X + 2 * R, TH = BIL_Y + 2 * R;
    const int x0 = blockIdx.x * BIL_X, y0 = a.yA + blockIdx.y * BIL_Y, k = blockIdx.z;
    const float* A = a.A + (size_t)k * a.N;
    const float* B = a.B + (size_t)k * a.N;
    const float S = a.colorS;
    for (int i = threadIdx.x; i < TW * TH; i += 256) {
        const int ty = i / TW, tx = i - ty * TW;
        const size_t src = (size_t)r101(y0 + ty - R, a.H) * a.W + r101(x0 + tx - R, a.W);
        biltile[i] = make_float2(A[src] * S, B[src] * S);
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;    // pixels (tx, ty) and (tx, ty + 8) of the tile
    const int c0 = (ty + R) * TW + tx + R, c1 = c0 + 8 * TW;
    // the a and b planes of a pixel travel as one packed pair: every step below is a packed fp32 instruction
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2* tile2 = reinterpret_cast<const f2*>(biltile);
    const f2 v0 = tile2[c0], v1 = tile2[c1];
    f2 s0 = {0.0f, 0.0f}, w0 = {0.0f, 0.0f}, s1 = {0.0f, 0.0f}, w1 = {0.0f, 0.0f};
    for (int t = 0; t < a.ntaps; t++) {
        const int2 tp = a.taps[t];
        const float lw = __int_as_float(tp.y);
        const f2 lws = {lw, lw};
        const f2 p = tile2[c0 + tp.x], q = tile2[c1 + tp.x];
        const f2 d = p - v0, e = q - v1;
        const f2 m = __builtin_elementwise_fma(-d, d, lws), n = __builtin_elementwise_fma(-e, e, lws);
        const f2 w = {__builtin_amdgcn_exp2f(m.x), __builtin_amdgcn_exp2f(m.y)}, u = {__builtin_amdgcn_exp2f(n.x), __builtin_amdgcn_exp2f(n.y)};
        s0 = __builtin_elementwise_fma(p, w, s0); w0 += w;
        s1 = __builtin_elementwise_fma(q, u, s1); w1 += u;
    }
    const float inv = 1.0f / S;
    s0 *= inv; s1 *= inv;
    const float sa0 = s0.x, sb0 = s0.y, wa0 = w0.x, wb0 = w0.y, sa1 = s1.x, sb1 = s1.y, wa1 = w1.x, wb1 = w1.y;
    const int x = x0 + tx;
    if (x >= a.W) return;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int y = y0 + ty + 8 * h;
        if (y >= a.yB) break;
        const float ab = h ? __fdiv_rn(sa1, wa1) : __fdiv_rn(sa0, wa0);
        const float bb = h ? __fdiv_rn(sb1, wb1) : __fdiv_rn(sb0, wb0);
        const size_t i = (size_t)y * a.W + x, o = (size_t)k * a.N + i;
        a.V[o] = __fadd_rn(__fmul_rn(ab, a.G[i]), bb);
        if (a.beta) a.beta[o] = __fdiv_rn(ab, __fsub_rn(1.0f, ab));
    }
}

// ---------------------------------------------------------------------------------------------
// steerable-filter orientation strength (steerFilter.cpp:239-446)

struct SteerTaps { double k[7][121]; };   // G2a, G2b, H2a, H2b, G2c, H2c, H2d sampled at integer taps

__device__ __forceinline__ int reflect_b(int p, int n)   // BORDER_REFLECT
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - p - 1;
    return p;
}

#define ST_T 16
// One thread per source pixel: seven 11x11 correlations over the 6-px reflect-padded image
// (taps reach 5 px), the C2/C3 quadratic forms and sqrt(C2^2 + C3^2); the running maximum is kept
// as an ordered-bits atomicMax (the values are non-negative doubles).
__global__ __launch_bounds__(ST_T * ST_T) void steer_strength_kernel(const u8* __restrict__ src, size_t sp, int H, int W,
                                                                     const SteerTaps* __restrict__ taps,
                                                                     double* __restrict__ O, unsigned long long* __restrict__ maxbits,
                                                                     int yBeg, int yEnd)
{
    __shared__ float tile[ST_T + 10][ST_T + 10];
    __shared__ double kk[7][121];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int x0 = blockIdx.x * ST_T, y0 = yBeg + blockIdx.y * ST_T;
    for (int i = ty * ST_T + tx; i < 7 * 121; i += ST_T * ST_T) kk[i / 121][i % 121] = taps->k[i / 121][i % 121];
    for (int i = ty * ST_T + tx; i < (ST_T + 10) * (ST_T + 10); i += ST_T * ST_T) {
        const int ry = i / (ST_T + 10), rx = i % (ST_T + 10);
        tile[ry][rx] = (float)src[(size_t)reflect_b(y0 + ry - 5, H) * sp + reflect_b(x0 + rx - 5, W)];
    }
    __syncthreads();
    const int x = x0 + tx, y = y0 + ty;
    double v = 0.0;
    if (x < W && y < yEnd) {
        double s[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 11; k++)
            for (int j = 0; j < 11; j++) {
                const double p = (double)tile[ty + k][tx + j];
#pragma unroll
                for (int f = 0; f < 7; f++) s[f] = __dadd_rn(s[f], __dmul_rn(kk[f][k * 11 + j], p));
            }
        const double g2a = s[0], g2b = s[1], h2a = s[2], h2b = s[3], g2c = s[4], h2c = s[5], h2d = s[6];
        auto sq = [](double t) { return __dmul_rn(t, t); };
        double C2 = __dmul_rn(0.5, __dsub_rn(sq(g2a), sq(g2c)));
        C2 = __dadd_rn(C2, __dmul_rn(0.46875, __dsub_rn(sq(h2a), sq(h2d))));
        C2 = __dadd_rn(C2, __dmul_rn(0.28125, __dsub_rn(sq(h2b), sq(h2c))));
        C2 = __dadd_rn(C2, __dmul_rn(0.1875, __dsub_rn(__dmul_rn(h2a, h2c), __dmul_rn(h2b, h2d))));
        double C3 = __dsub_rn(-__dmul_rn(g2a, g2b), __dmul_rn(g2b, g2c));
        C3 = __dsub_rn(C3, __dmul_rn(0.9375, __dsub_rn(__dmul_rn(h2c, h2d), __dmul_rn(h2a, h2b))));
        C3 = __dsub_rn(C3, __dmul_rn(__dmul_rn(1.6875, h2b), h2c));
        C3 = __dsub_rn(C3, __dmul_rn(__dmul_rn(0.1875, h2a), h2d));
        v = sqrt(__dadd_rn(sq(C2), sq(C3)));
        O[(size_t)y * W + x] = v;
    }
    // block max -> one atomic
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        unsigned long long o = __shfl_xor(b, m, 64);
        b = o > b ? o : b;
    }
    if (((ty * ST_T + tx) & 63) == 0) atomicMax(maxbits, b);
}

// dst = clamp(O / max * 255, 1e-6, 120) with the reference's comparison order (:412, :434-441)
__global__ __launch_bounds__(256) void steer_scale_kernel(const double* __restrict__ O, size_t i0, size_t i1, const unsigned long long* maxbits,
                                                          double* __restrict__ dst, float* __restrict__ dstf)
{
    const size_t i = i0 + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= i1) return;
    const double mx = __longlong_as_double((long long)*maxbits);
    double v = __dmul_rn(__ddiv_rn(__dmul_rn(O[i], 1.0), mx), 255.0);
    if (v > 120) v = 120;
    if (v < 1e-6) v = 1e-6;
    if (dst) dst[i] = v;
    if (dstf) dstf[i] = (float)v;
}

static double G2a(int x, int y) { double xs = x * x, ys = y * y; ys = exp(-(xs + ys)); return 0.9213 * ((2 * xs) - 1) * ys; }
static double G2b(int x, int y) { double xs = x * x, ys = y * y; ys = exp(-(xs + ys)); return 1.843 * x * y * ys; }
static double H2a(int x, int y) { double xs = x * x, ys = y * y; ys = exp(-(xs + ys)); return 0.9780 * (-2.254 * x + x * xs) * ys; }
static double H2b(int x, int y) { double xs = x * x, ys = y * y; ys = exp(-(xs + ys)); return 0.9780 * (-0.7515 + xs) * y * ys; }
static double H2c(int x, int y) { double xs = x * x, ys = y * y; ys = exp(-(xs + ys)); return 0.9780 * (-0.7515 + y * y) * x * ys; }
static double G2c(int x, int y) { double xs = x * x, ys = y * y; xs = exp(-(xs + ys)); return 0.9213 * ((2 * ys) - 1) * xs; }
static double H2d(int x, int y) { double xs = pow(x, 2), ys = pow(y, 2); xs = exp(-(xs + ys)); return 0.9780 * (-2.254 * y + y * ys) * xs; }

// O, maxbits, tapsDev: device scratch (N doubles, 8 bytes, sizeof(SteerTaps)); dst / dstf may be null
// Rows [yBeg, yEnd) only (yBeg < 0: the whole image).  extMax (host, nullable): the global maximum of the
// orientation strength when the caller already knows it (row-band sharding: the maximum over all bands);
// scaleToo = false stops after the strength pass (maxbits then holds the maximum over the rows computed).
static int steer_run(hipStream_t s, const u8* src, size_t sp, int H, int W, double* O, unsigned long long* maxbits,
                     SteerTaps* tapsDev, double* dst, float* dstf, int yBeg = -1, int yEnd = -1, const double* extMax = nullptr,
                     bool scaleToo = true)
{
    if (yBeg < 0) { yBeg = 0; yEnd = H; }
    // basis kernels, steerFilter.cpp:283-293: built once, thread-safely (the reference runs its left and right
    // routines from two threads), read-only afterwards
    static const SteerTaps host = [] {
        SteerTaps t;
        for (int y = -5; y <= 5; y++)
            for (int x = -5; x <= 5; x++) {
                const int i = (y + 5) * 11 + (x + 5);
                t.k[0][i] = G2a(x, y); t.k[1][i] = G2b(x, y); t.k[2][i] = H2a(x, y); t.k[3][i] = H2b(x, y);
                t.k[4][i] = G2c(x, y); t.k[5][i] = H2c(x, y); t.k[6][i] = H2d(x, y);
            }
        return t;
    }();
    MD_HIP(hipMemcpyAsync(tapsDev, &host, sizeof(SteerTaps), hipMemcpyHostToDevice, s));
    // the strengths are non-negative doubles: their bit patterns order like the values, and starting the running
    // maximum at the known global maximum leaves it there
    if (extMax) MD_HIP(hipMemcpyAsync(maxbits, extMax, 8, hipMemcpyHostToDevice, s));
    else MD_HIP(hipMemsetAsync(maxbits, 0, 8, s));
    if (yEnd > yBeg) {
        dim3 grid((W + ST_T - 1) / ST_T, (yEnd - yBeg + ST_T - 1) / ST_T), block(ST_T, ST_T);
        steer_strength_kernel<<<grid, block, 0, s>>>(src, sp, H, W, tapsDev, O, maxbits, yBeg, yEnd);
        const size_t i0 = (size_t)yBeg * W, i1 = (size_t)yEnd * W;
        if (scaleToo) steer_scale_kernel<<<(unsigned)((i1 - i0 + 255) / 256), 256, 0, s>>>(O, i0, i1, maxbits, dst, dstf);
    }
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// ---------------------------------------------------------------------------------------------
// host orchestration

static inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

static int box_f32(hipStream_t s, const float* A, size_t aStride, const float* B, size_t bStride, int n, int H, int W, int k,
                   int mode, double* tmp, const double* div, float* out)
{
    dim3 gh((W + BH_SEG - 1) / BH_SEG, H, n);
    box_h_kernel<<<gh, 256, (BH_SEG + k + 8) * sizeof(float), s>>>(A, aStride, B, bStride, H, W, k, mode, tmp);
    dim3 gv((W + 255) / 256, (H + BV_SEG - 1) / BV_SEG, n);
    box_v_kernel<<<gv, 256, 0, s>>>(tmp, H, W, k, mode, 1.0 / ((double)k * k), div, out);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

static int device_sum(mdisp_ctx* c, const float* src, size_t N, double* partDev, double* result)
{
    const int nb = 240;
    partial_sum_kernel<<<nb, 256, 0, c->stream>>>(src, N, partDev);
    MD_HIP(hipGetLastError());
    std::vector<double> h(nb);
    MD_HIP(hipMemcpyAsync(h.data(), partDev, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MD_HIP(hipStreamSynchronize(c->stream));
    double s = 0;
    for (int i = 0; i < nb; i++) s += h[i];
    *result = s;
    return MDISP_OK;
}

// everything that depends on the guide image only
struct GuideState {
    int H, W, r, r2, type;
    float* G;        // normalised guide
    float* meanG;
    float* den;      // var + regulariser
    float* w;        // steer weights as float (SKWGIF)
    double* wsum;    // clipped-window sum of w (SKWGIF)
    int2* bilTaps; int bilR, bilN;   // OURS_GIF*: footprint of the bilateral stage (radius, tap count)
};
static bool is_ours(int type) { return type == MDISP_OURS_GIF || type == MDISP_OURS_GIF2 || type == MDISP_OURS_GIF3; }
// diameter cv::bilateralFilter receives (gifs.cpp:661, 717: r2; :785: r) -> radius
static int ours_radius(int type, int r, int r2)
{
    const int d = type == MDISP_OURS_GIF3 ? r : r2;
    const int radius = d <= 0 ? 4 /* cvRound(3 * 1.5) */ : d / 2;
    return radius < 1 ? 1 : radius;
}

struct GifScratch {   // device scratch sized for n slices
    size_t N; int n;
    float *F0, *F1, *F2;      // n*N each
    double* T;                // n*N
    float *g0, *g1, *g2;      // N each (guide-only temporaries)
    double* gd;               // N
    u32* mm;                  // 2*(n+1)
    double* part;             // 256
    unsigned long long* maxbits; SteerTaps* taps;
};

static size_t gif_scratch_bytes(size_t N, int n)
{
    return md_align(N * n * 4, 256) * 3 + md_align(N * n * 8, 256) + md_align(N * 4, 256) * 8 + md_align(N * 8, 256) * 3 +
           md_align((size_t)(2 * (n + 1)) * 4, 256) + 4096 + md_align(sizeof(SteerTaps), 256) + 4096;
}

struct Carve2 { u8* p; template <typename T> T* take(size_t count) { T* r = (T*)p; p += md_align(count * sizeof(T), 256); return r; } };

// gaussian of a constant image: getGaussianKernel(3, 2) applied separably in float
static float gauss3_const(float c)
{
    double k[3], ks = 0;
    for (int i = 0; i < 3; i++) { double x = i - 1; k[i] = exp(-(x * x) / (2 * 2.0 * 2.0)); ks += k[i]; }
    float kf[3];
    for (int i = 0; i < 3; i++) kf[i] = (float)(k[i] / ks);
    volatile float s = 0;
    for (int i = 0; i < 3; i++) { volatile float t = kf[i] * c; s = s + t; }
    const float row = s;
    s = 0;
    for (int i = 0; i < 3; i++) { volatile float t = kf[i] * row; s = s + t; }
    return s;
}

// [wA, wB): the rows of the steer weights that are valid in steerW_dev and that the caller's output rows need
// (the whole image by default)
static int guide_prepare(mdisp_ctx* c, const u8* dGuide, size_t gpitch, int H, int W, int r, int r2, int type, double eps,
                         double namuda, const double* steerW_dev, GifScratch& sc, GuideState& gs, Carve2& cv, int wA = -1, int wB = -1,
                         int yA = -1, int yB = -1)
{
    if (wA < 0) { wA = 0; wB = H; }
    if (yA < 0) { yA = 0; yB = H; }
    hipStream_t s = c->stream;
    const size_t N = (size_t)H * W;
    gs.H = H; gs.W = W; gs.r = r; gs.r2 = r2; gs.type = type;
    gs.G = cv.take<float>(N); gs.meanG = cv.take<float>(N); gs.den = cv.take<float>(N);
    gs.w = nullptr; gs.wsum = nullptr;
    u32* mm = sc.mm;
    mm_init_kernel<<<1, 64, 0, s>>>(mm, 1);
    minmax_u8_image_kernel<<<dim3(4, 64), 256, 0, s>>>(dGuide, gpitch, H, W, mm);
    normalize_u8_image_kernel<<<dim3((W + 255) / 256, H), 256, 0, s>>>(dGuide, gpitch, H, W, mm, gs.G);
    int rc;
    // meanG, corrG -> var (gifs.cpp:58-59,76-80)
    if ((rc = box_f32(s, gs.G, 0, nullptr, 0, 1, H, W, r, 0, sc.gd, nullptr, gs.meanG))) return rc;
    if ((rc = box_f32(s, gs.G, 0, gs.G, 0, 1, H, W, r, 0, sc.gd, nullptr, sc.g0))) return rc;
    var_kernel<<<nblk(N), 256, 0, s>>>(sc.g0, gs.meanG, N, sc.g1);   // g1 = varG
    const float epsf = (float)eps;
    gs.bilTaps = nullptr; gs.bilR = 0; gs.bilN = 0;
    // regulariser of the coefficient stage: OURS_GIF as SKWGIF (gifs.cpp:646-652), OURS_GIF2 as GIF (:706),
    // OURS_GIF3 as EGIF (:753-754)
    const int denType = type == MDISP_OURS_GIF ? MDISP_SKWGIF : type == MDISP_OURS_GIF2 ? MDISP_GIF : type == MDISP_OURS_GIF3 ? MDISP_EGIF : type;
    if (is_ours(type)) {
        const int R = ours_radius(type, r, r2), TW = BIL_X + 2 * R;
        std::vector<int2> taps;
        const double gsc = -0.5 / (3.0 * 3.0);
        for (int i = -R; i <= R; i++)
            for (int j = -R; j <= R; j++) {
                const double rr = sqrt((double)i * i + (double)j * j);
                if (rr > R) continue;
                const float w = (float)(rr * rr * gsc * 1.4426950408889634);      // log2 of the space weight exp(r^2 * gsc)
                int2 t; t.x = i * TW + j; memcpy(&t.y, &w, 4);
                taps.push_back(t);
            }
        gs.bilR = R; gs.bilN = (int)taps.size();
        gs.bilTaps = cv.take<int2>(taps.size());
        MD_HIP(hipMemcpyAsync(gs.bilTaps, taps.data(), taps.size() * sizeof(int2), hipMemcpyHostToDevice, s));
        MD_HIP(hipStreamSynchronize(s));     // `taps` leaves scope
    }
    if (denType == MDISP_GIF) {
        den_scalar_kernel<<<nblk(N), 256, 0, s>>>(sc.g1, N, epsf, gs.den);
    } else if (denType == MDISP_EGIF) {
        double sum;
        if ((rc = device_sum(c, sc.g1, N, sc.part, &sum))) return rc;
        const double mean2Var = sum / ((double)W * H);            // gifs.cpp:252
        den_scalar_kernel<<<nblk(N), 256, 0, s>>>(sc.g1, N, (float)(mean2Var * eps + 0.001 * 0.001), gs.den);
    } else if (denType == MDISP_WGIF) {
        // edgeAwareWeight (gifs.cpp:347-374): with r = 1 the box "variance" G*G - G*G is exactly 0,
        // so gamma is one constant, evaluated here with the same float operations
        u32 hmm[2];
        MD_HIP(hipMemcpyAsync(hmm, mm, 8, hipMemcpyDeviceToHost, s));
        MD_HIP(hipStreamSynchronize(s));
        auto ord2f_h = [](u32 o) { u32 u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; float f; memcpy(&f, &u, 4); return f; };
        const double smin = ord2f_h(hmm[0]), smax = ord2f_h(hmm[1]);
        const double scale = (smax - smin > DBL_EPSILON) ? 1.0 / (smax - smin) : 0.0, shift = 0.0 - smin * scale;
        volatile float gmin = (float)smin * (float)scale; gmin = gmin + (float)shift;
        volatile float gmax = (float)smax * (float)scale; gmax = gmax + (float)shift;
        const double L = (double)gmax - (double)gmin;
        const double epsw = (0.001 * L) * (0.001 * L);
        volatile float varG1 = 0.0f + (float)epsw;
        volatile float inv = 1.0f / varG1;
        const double val_ = ((double)inv * (double)N) / ((double)W * H);
        volatile float g0v = varG1 * (float)val_;
        const float gamma = gauss3_const(g0v);
        volatile float e = epsf / gamma;
        den_scalar_kernel<<<nblk(N), 256, 0, s>>>(sc.g1, N, e, gs.den);
    } else if (denType == MDISP_SKWGIF) {
        MD_REQUIRE(type != MDISP_SKWGIF || (steerW_dev && r2 >= 2), MDISP_E_BADARG, "SKWGIF needs the steer-filter weights and r2 >= 2");
        MD_REQUIRE(r2 >= 1, MDISP_E_BADARG, "r2 must be >= 1");
        // var over the r2 window (gifs.cpp:547,555,562)
        if ((rc = box_f32(s, gs.G, 0, nullptr, 0, 1, H, W, r2, 0, sc.gd, nullptr, sc.g2))) return rc;     // meanG2
        if ((rc = box_f32(s, gs.G, 0, gs.G, 0, 1, H, W, r2, 0, sc.gd, nullptr, sc.g0))) return rc;        // corrG2
        float* var2 = cv.take<float>(N); float* temp = cv.take<float>(N); float* inv = cv.take<float>(N);
        var_kernel<<<nblk(N), 256, 0, s>>>(sc.g0, sc.g2, N, var2);
        skw_temp_kernel<<<nblk(N), 256, 0, s>>>(var2, N, (float)namuda, temp, inv);
        double sum;
        if ((rc = device_sum(c, inv, N, sc.part, &sum))) return rc;
        const float gscale = (float)(sum / ((double)H * W));
        skw_den_kernel<<<nblk(N), 256, 0, s>>>(sc.g1, temp, N, gscale, epsf, gs.den);
        if (type == MDISP_SKWGIF) {
        // steer weights: float copy (gifs.cpp:574-577) and their clipped-window sum (:600)
        gs.w = cv.take<float>(N); gs.wsum = cv.take<double>(N);
        const size_t w0 = (size_t)wA * W, wn = (size_t)(wB - wA) * W;
        f64_to_f32_kernel<<<nblk(wn), 256, 0, s>>>(steerW_dev + w0, wn, gs.w + w0);
        dim3 gh((W + BH_SEG - 1) / BH_SEG, wB - wA, 1);
        box_h_kernel<<<gh, 256, (BH_SEG + r2 + 8) * sizeof(float), s>>>(gs.w, 0, nullptr, 0, H, W, r2, 1, sc.gd, wA);
        dim3 gv((W + 255) / 256, (yB - yA + BV_SEG - 1) / BV_SEG, 1);
        double* hs2 = cv.take<double>(N);
        box_v_f64_kernel<<<gv, 256, 0, s>>>(sc.gd, H, W, r2, 1, hs2, yA, yB);
        gs.wsum = hs2;
        }
    } else {
        mdisp_set_error("unknown GIF type %d", type);
        return MDISP_E_BADARG;
    }
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// Filter n slices and leave the filtered values a_bar*G + b_bar in sc.F0.  The slices come either
// normalised and planar in sc.F0 (cs == null) or as integer costs formed on the fly (cs != null,
// normalised with the per-slice min/max in mm).  beta (nullable): a_bar/(1 - a_bar) of each slice.
static int gif_filter_batch(mdisp_ctx* c, const GuideState& gs, GifScratch& sc, int n, const CostSrc* cs, int k0, const u32* mm,
                            int normMode, float* beta, int yA, int yB)
{
    hipStream_t s = c->stream;
    const int H = gs.H, W = gs.W, r = gs.r;
    const size_t N = (size_t)H * W;
    const bool skw = gs.type == MDISP_SKWGIF;
    int rc;
    if (!march_supported(r) || (skw && !march_supported(gs.r2))) {
        mdisp_set_error("guided-filter window above %d is not supported", MW);
        return MDISP_E_UNSUPPORTED;
    }
    const bool ours = is_ours(gs.type);
    // output rows [yA, yB): the second filter reads a and b up to one window above and below them
    const int k2 = ours ? gs.bilR : skw ? gs.r2 : r;
    const int yA1 = yA - k2 < 0 ? 0 : yA - k2, yB1 = yB + k2 > H ? H : yB + k2;
    // (P, G*P) -> means -> a -> F1, b -> F2   (gifs.cpp:60-61,71,96,127-128)
    StoreAB sab{sc.F1, sc.F2, N, W, gs.meanG, gs.den, 1.0 / ((double)r * r)};
    if (cs) {
        LoadCostGP ld{*cs, k0, mm, normMode, gs.G, {}};
        if ((rc = dual_box_march(s, n, H, W, r, 0, yA1, yB1, ld, sab))) return rc;
    } else {
        LoadPlanarGP ld{sc.F0, N, W, gs.G};
        if ((rc = dual_box_march(s, n, H, W, r, 0, yA1, yB1, ld, sab))) return rc;
    }
    if (ours) {
        BilArgs b;
        b.A = sc.F1; b.B = sc.F2; b.N = N; b.H = H; b.W = W; b.R = gs.bilR; b.ntaps = gs.bilN; b.yA = yA; b.yB = yB;
        b.taps = gs.bilTaps; b.G = gs.G; b.V = sc.F0; b.beta = beta;
        b.colorS = (float)sqrt(0.5 / (2.0 * 2.0) * 1.4426950408889634);
        const size_t lds = (size_t)(BIL_X + 2 * gs.bilR) * (BIL_Y + 2 * gs.bilR) * sizeof(float2);
        if (lds > 64 * 1024) { mdisp_set_error("bilateral radius %d is not supported", gs.bilR); return MDISP_E_UNSUPPORTED; }
        if (yB <= yA) return MDISP_OK;
        dim3 grid((W + BIL_X - 1) / BIL_X, (yB - yA + BIL_Y - 1) / BIL_Y, n);
        bilateral_ab_kernel<<<grid, 256, lds, s>>>(b);
        MD_HIP(hipGetLastError());
        return MDISP_OK;
    }
    // (a, b) -> means -> a_bar*G + b_bar -> F0; SKWGIF: steer-weighted means boxsum(w*a)/boxsum(w)
    // over the clipped r2 window (gifs.cpp:587-611)
    LoadAB lab{sc.F1, sc.F2, N, W, skw ? gs.w : nullptr};
    StoreOut so{sc.F0, N, W, gs.G, 1.0 / ((double)r * r), skw ? gs.wsum : nullptr, beta};
    if ((rc = dual_box_march(s, n, H, W, k2, skw ? 1 : 0, yA, yB, lab, so))) return rc;
    return MDISP_OK;
}

static void scratch_carve(Carve2& cv, size_t N, int n, GifScratch& sc)
{
    sc.N = N; sc.n = n;
    sc.F0 = cv.take<float>(N * n); sc.F1 = cv.take<float>(N * n); sc.F2 = cv.take<float>(N * n);
    sc.T = cv.take<double>(N * n);
    sc.g0 = cv.take<float>(N); sc.g1 = cv.take<float>(N); sc.g2 = cv.take<float>(N);
    sc.gd = cv.take<double>(N);
    sc.mm = cv.take<u32>(2 * (size_t)(n + 1));
    sc.part = cv.take<double>(256);
    sc.maxbits = cv.take<unsigned long long>(1);
    sc.taps = cv.take<SteerTaps>(1);
}

static int pick_batch(size_t N, int D)
{
    size_t budget = (size_t)16 << 30;                // bytes of per-slice scratch (20 B per pixel per slice): enough slices per launch to fill the GPU even when one band of a large image is processed
    size_t n = budget / (N * 20);
    if (n < 1) n = 1;
    if (n > (size_t)D) n = D;
    return (int)n;
}

static size_t guide_extra_bytes(size_t N) { return md_align(N * 4, 256) * 8 + md_align(N * 8, 256) * 2 + 32768 /* bilateral taps */; }

static int check_img(const mdisp_image* im, const char* what)
{
    MD_REQUIRE(im && im->data && im->rows > 0 && im->cols > 0, MDISP_E_BADARG, "%s: empty image", what);
    return MDISP_OK;
}

extern "C" int mdisp_steer_filter(mdisp_ctx* c, const mdisp_image* src, mdisp_image* dst)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_img(src, "src")) || (rc = check_img(dst, "dst"))) return rc;
    MD_REQUIRE(src->depth == MDISP_8U && src->channels == 1, MDISP_E_BADARG, "steerFilter needs a CV_8UC1 image");
    MD_REQUIRE(dst->depth == MDISP_64F && dst->channels == 1 && dst->rows == src->rows && dst->cols == src->cols,
               MDISP_E_BADARG, "dst must be CV_64FC1 of the source size");
    MD_HIP(hipSetDevice(c->device));
    const int H = src->rows, W = src->cols;
    const size_t N = (size_t)H * W, pitch = md_align(W, 256);
    size_t need = pitch * H + md_align(N * 8, 256) * 2 + md_align(sizeof(SteerTaps), 256) + 1024;
    u8* base = (u8*)mdisp_ws(c, need);
    if (!base) return MDISP_E_NOMEM;
    Carve2 cv{base};
    u8* dS = cv.take<u8>(pitch * H);
    double* O = cv.take<double>(N); double* out = cv.take<double>(N);
    SteerTaps* taps = cv.take<SteerTaps>(1);
    unsigned long long* mb = cv.take<unsigned long long>(1);
    hipStream_t s = c->stream;
    MD_TRY(md_h2d_2d(c, dS, pitch, src->data, src->step_bytes, W, H));
    if ((rc = steer_run(s, dS, pitch, H, W, O, mb, taps, out, nullptr))) return rc;
    MD_TRY(md_d2h_2d(c, dst->data, dst->step_bytes, out, (size_t)W * 8, (size_t)W * 8, H));
    MD_HIP(hipStreamSynchronize(s));
    return MDISP_OK;
}

extern "C" int mdisp_guided_filter(mdisp_ctx* c, const mdisp_image* guide, const mdisp_image* P, int r, double eps, int gif_type,
                                   const mdisp_image* steerW, int r2, double namuda, mdisp_image* out, mdisp_image* beta)
{
    MD_REQUIRE(c, MDISP_E_BADARG, "null context");
    int rc;
    if ((rc = check_img(guide, "guidedImg")) || (rc = check_img(P, "inputP")) || (rc = check_img(out, "out"))) return rc;
    MD_REQUIRE(guide->depth == MDISP_8U && guide->channels == 1, MDISP_E_UNSUPPORTED, "the guide must be CV_8UC1");
    MD_REQUIRE(P->channels == 1 && P->rows == guide->rows && P->cols == guide->cols, MDISP_E_BADARG, "guide / input size mismatch");
    MD_REQUIRE(P->depth == MDISP_8U || P->depth == MDISP_32S || P->depth == MDISP_32F, MDISP_E_UNSUPPORTED, "inputP must be 8U, 32S or 32F");
    MD_REQUIRE(out->depth == MDISP_32F && out->rows == guide->rows && out->cols == guide->cols && out->channels == 1, MDISP_E_BADARG,
               "out must be CV_32FC1 of the guide size");
    MD_REQUIRE(r >= 1, MDISP_E_BADARG, "window must be >= 1");
    if (beta) MD_REQUIRE(beta->depth == MDISP_32F && beta->rows == guide->rows && beta->cols == guide->cols, MDISP_E_BADARG, "bad beta image");
    if (gif_type == MDISP_SKWGIF)
        MD_REQUIRE(steerW && steerW->data && steerW->depth == MDISP_64F && steerW->rows == guide->rows && steerW->cols == guide->cols,
                   MDISP_E_BADARG, "SKWGIF needs a CV_64F steer map of the guide size");
    MD_HIP(hipSetDevice(c->device));
    const int H = guide->rows, W = guide->cols;
    const size_t N = (size_t)H * W, pitch = md_align(W, 256);
    size_t need = gif_scratch_bytes(N, 1) + guide_extra_bytes(N) + pitch * H + md_align(N * 8, 256) * 2 + md_align(N * 4, 256) * 3 + 4096;
    u8* base = (u8*)mdisp_ws(c, need);
    if (!base) return MDISP_E_NOMEM;
    Carve2 cv{base};
    GifScratch sc; scratch_carve(cv, N, 1, sc);
    u8* dG = cv.take<u8>(pitch * H);
    double* dSteer = cv.take<double>(N);
    u8* rawP = (u8*)cv.take<double>(N);      // raw input slice, any supported depth
    float* dOut = cv.take<float>(N); float* dBeta = cv.take<float>(N);
    hipStream_t s = c->stream;
    MD_TRY(md_h2d_2d(c, dG, pitch, guide->data, guide->step_bytes, W, H));
    if (gif_type == MDISP_SKWGIF)
        MD_TRY(md_h2d_2d(c, dSteer, (size_t)W * 8, steerW->data, steerW->step_bytes, (size_t)W * 8, H));
    const size_t esz = P->depth == MDISP_8U ? 1 : 4;
    MD_TRY(md_h2d_2d(c, rawP, (size_t)W * esz, P->data, P->step_bytes, (size_t)W * esz, H));
    GuideState gs;
    if ((rc = guide_prepare(c, dG, pitch, H, W, r, r2, gif_type, eps, namuda, dSteer, sc, gs, cv))) return rc;
    // normalise the slice (gifs.cpp:56)
    u32* mm = sc.mm + 2;
    mm_init_kernel<<<1, 64, 0, s>>>(mm, 1);
    dim3 g1(240, 1);
    if (P->depth == MDISP_8U) {
        minmax_kernel<u8><<<g1, 256, 0, s>>>(rawP, N, mm);
        normalize_kernel<u8><<<g1, 256, 0, s>>>(rawP, N, mm, NORM_FLOAT, sc.F0);
    } else if (P->depth == MDISP_32S) {
        minmax_kernel<int><<<g1, 256, 0, s>>>((const int*)rawP, N, mm);
        normalize_kernel<int><<<g1, 256, 0, s>>>((const int*)rawP, N, mm, NORM_DOUBLE, sc.F0);
    } else {
        minmax_kernel<float><<<g1, 256, 0, s>>>((const float*)rawP, N, mm);
        normalize_kernel<float><<<g1, 256, 0, s>>>((const float*)rawP, N, mm, NORM_FLOAT, sc.F0);
    }
    if ((rc = gif_filter_batch(c, gs, sc, 1, nullptr, 0, nullptr, NORM_FLOAT, beta ? dBeta : nullptr, 0, H))) return rc;
    dOut = sc.F0;
    MD_TRY(md_d2h_2d(c, out->data, out->step_bytes, dOut, (size_t)W * 4, (size_t)W * 4, H));
    if (beta) MD_TRY(md_d2h_2d(c, beta->data, beta->step_bytes, dBeta, (size_t)W * 4, (size_t)W * 4, H));
    MD_HIP(hipStreamSynchronize(s));
    return MDISP_OK;
}

// Shared driver: guide + cost slices (computed on the device or uploaded) -> filtered slices -> WTA.
// normMode: how the raw slice is normalised (float for 8U / 32F data, double for 32S data).
size_t md_gif_scratch_bytes(int H, int W, int D)
{
    const size_t N = (size_t)H * W;
    const int n = pick_batch(N, D);
    return gif_scratch_bytes(N, n) + guide_extra_bytes(N) + md_align(N * 8, 256) * 3 + md_align(N * 4, 256) * 2 + 16384 +
           md_align((size_t)D * 8, 256);
}

static inline u32 f2ord_host(float f) { u32 u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static inline float ord2f_host(u32 o) { u32 u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; float f; memcpy(&f, &u, 4); return f; }

// One band's share of the statistics that span the whole image: per-slice min / max of the integer cost over rows
// [yA, yB) (minmax[2k], minmax[2k+1]; +FLT_MAX / -FLT_MAX for an empty band) and, for SKWGIF, the maximum of the
// steer filter's orientation strength over the same rows.  Synchronises the stream (results go to host memory).
int md_gif_band_stats(mdisp_ctx* c, void* scratch, const u8* dGuide, size_t gpitch, int H, int W, int D, int gif_type,
                      const u8* ref, size_t rp, const u8* pad, size_t pp, int off0, int offStep, int costKind, int yA, int yB,
                      float* minmax, double* steerMax)
{
    hipStream_t s = c->stream;
    const size_t N = (size_t)H * W;
    const int n = pick_batch(N, D);
    Carve2 cv{(u8*)scratch};
    GifScratch sc; scratch_carve(cv, N, n, sc);
    (void)cv.take<float>(N);
    double* O = cv.take<double>(N);
    for (int k = 0; k < D; k++) { minmax[2 * k] = FLT_MAX; minmax[2 * k + 1] = -FLT_MAX; }
    if (steerMax) *steerMax = 0.0;
    if (yB <= yA) return MDISP_OK;
    int rc;
    if (gif_type == MDISP_SKWGIF && steerMax) {
        if ((rc = steer_run(s, dGuide, gpitch, H, W, O, sc.maxbits, sc.taps, nullptr, nullptr, yA, yB, nullptr, false))) return rc;
        MD_HIP(hipMemcpyAsync(steerMax, sc.maxbits, 8, hipMemcpyDeviceToHost, s));
    }
    CostSrc cs;
    cs.ref = ref; cs.rp = rp; cs.pad = pad; cs.pp = pp; cs.W = W; cs.H = H; cs.off0 = off0; cs.offStep = offStep; cs.kind = costKind;
    std::vector<u32> h(2 * (size_t)n);
    for (int k0 = 0; k0 < D; k0 += n) {
        const int nb = D - k0 < n ? D - k0 : n;
        mm_init_kernel<<<(nb + 63) / 64, 64, 0, s>>>(sc.mm + 2, nb);
        const dim3 gm((W + 255) / 256, (yB - yA + MM_ROWS - 1) / MM_ROWS, (nb + MM_SLICES - 1) / MM_SLICES);
        u32* part = reinterpret_cast<u32*>(sc.T);
        cost_minmax_kernel<<<gm, 256, 0, s>>>(cs, k0, nb, part, yA, yB);
        cost_minmax_finish_kernel<<<nb, 256, 0, s>>>(part, (int)(gm.x * gm.y), nb, sc.mm + 2);
        MD_HIP(hipGetLastError());
        MD_HIP(hipMemcpyAsync(h.data(), sc.mm + 2, (size_t)nb * 8, hipMemcpyDeviceToHost, s));
        MD_HIP(hipStreamSynchronize(s));
        for (int k = 0; k < nb; k++) { minmax[2 * (k0 + k)] = ord2f_host(h[2 * k]); minmax[2 * (k0 + k) + 1] = ord2f_host(h[2 * k + 1]); }
    }
    MD_HIP(hipStreamSynchronize(s));
    return MDISP_OK;
}

// extMinMax (host, 2*D floats, nullable) / extSteerMax (host, nullable): image-wide statistics the caller already
// has (merged md_gif_band_stats results); when given, the per-slice cost range is not recomputed and the steer
// filter runs on the rows the band needs only.
int md_gif_stereo(mdisp_ctx* c, void* scratch, const u8* dGuide, size_t gpitch, int H, int W, int D, int minD, int r, double eps,
                  int gif_type, int r2, double namuda, const u8* ref, size_t rp, const u8* pad, size_t pp, int off0, int offStep,
                  int costKind, const float* hostCosts, int normMode, float* dDisp32, double* dDisp64, int yA, int yB,
                  const float* extMinMax, const double* extSteerMax)
{
    hipStream_t s = c->stream;
    if (yA < 0) { yA = 0; yB = H; }
    const size_t N = (size_t)H * W;
    const int n = pick_batch(N, D);
    Carve2 cv{(u8*)scratch};
    GifScratch sc; scratch_carve(cv, N, n, sc);
    float* best = cv.take<float>(N);
    double* O = cv.take<double>(N); double* steerD = cv.take<double>(N);
    int rc;
    int wA = 0, wB = H;
    if (gif_type == MDISP_SKWGIF) {
        // asw_gifs computes the weights from the gray guide once (methods_disp.cpp:1176-1180)
        if (extSteerMax) {
            wA = yA - r2 < 0 ? 0 : yA - r2; wB = yB + r2 > H ? H : yB + r2;
            if ((rc = steer_run(s, dGuide, gpitch, H, W, O, sc.maxbits, sc.taps, steerD, nullptr, wA, wB, extSteerMax))) return rc;
        } else if ((rc = steer_run(s, dGuide, gpitch, H, W, O, sc.maxbits, sc.taps, steerD, nullptr))) return rc;
    }
    GuideState gs;
    if ((rc = guide_prepare(c, dGuide, gpitch, H, W, r, r2, gif_type, eps, namuda, steerD, sc, gs, cv, wA, wB, yA, yB))) return rc;
    u32* extDev = nullptr;       // the caller's per-slice cost ranges in the kernels' ordered-integer form
    if (extMinMax && !hostCosts) {
        std::vector<u32> extOrd(2 * (size_t)D);
        for (int k = 0; k < 2 * D; k++) extOrd[k] = f2ord_host(extMinMax[k]);
        extDev = cv.take<u32>(2 * (size_t)D);
        MD_HIP(hipMemcpyAsync(extDev, extOrd.data(), extOrd.size() * 4, hipMemcpyHostToDevice, s));
        MD_HIP(hipStreamSynchronize(s));      // extOrd leaves scope; nothing heavy has been enqueued yet
    }
    CostSrc cs;
    cs.ref = ref; cs.rp = rp; cs.pad = pad; cs.pp = pp; cs.W = W; cs.H = H; cs.off0 = off0; cs.offStep = offStep; cs.kind = costKind;
    for (int k0 = 0; k0 < D; k0 += n) {
        const int nb = D - k0 < n ? D - k0 : n;
        mm_init_kernel<<<(nb + 63) / 64, 64, 0, s>>>(sc.mm + 2, nb);
        dim3 g(120, nb);
        if (hostCosts) {
            // slices arrive as float (computeDisp_Feature output); staged through the double scratch
            float* raw = (float*)sc.T;
            MD_HIP(hipMemcpyAsync(raw, hostCosts + (size_t)k0 * N, (size_t)nb * N * 4, hipMemcpyHostToDevice, s));
            minmax_kernel<float><<<g, 256, 0, s>>>(raw, N, sc.mm + 2);
            normalize_kernel<float><<<g, 256, 0, s>>>(raw, N, sc.mm + 2, NORM_FLOAT, sc.F0);
        } else if (extDev) {
            MD_HIP(hipMemcpyAsync(sc.mm + 2, extDev + 2 * (size_t)k0, (size_t)nb * 8, hipMemcpyDeviceToDevice, s));
        } else {
            // per-workgroup partials go through the (otherwise idle) double scratch
            const dim3 gm((W + 255) / 256, (H + MM_ROWS - 1) / MM_ROWS, (nb + MM_SLICES - 1) / MM_SLICES);
            u32* part = reinterpret_cast<u32*>(sc.T);
            cost_minmax_kernel<<<gm, 256, 0, s>>>(cs, k0, nb, part, 0, H);
            cost_minmax_finish_kernel<<<nb, 256, 0, s>>>(part, (int)(gm.x * gm.y), nb, sc.mm + 2);
        }
        if ((rc = gif_filter_batch(c, gs, sc, nb, hostCosts ? nullptr : &cs, k0, sc.mm + 2, normMode, nullptr, yA, yB))) return rc;
        OutArgs oa;
        oa.v = sc.F0; oa.abar = nullptr; oa.bbar = nullptr; oa.G = gs.G; oa.N = N; oa.n = nb; oa.filtered = nullptr; oa.beta = nullptr;
        oa.best = best; oa.disp32 = dDisp32; oa.disp64 = dDisp64; oa.d0 = minD + k0; oa.first = k0 == 0;
        oa.i0 = (size_t)yA * W; oa.i1 = (size_t)yB * W;
        out_wta_kernel<<<nblk(oa.i1 - oa.i0), 256, 0, s>>>(oa);
        MD_HIP(hipGetLastError());
    }
    return MDISP_OK;
}
