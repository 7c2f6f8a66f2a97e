// Marching box filter of two quantities at once (shared by the guided-filter kernels of filters.hip and gif_mc.hip).
#pragma once
#include "common.h"

// cv::BORDER_REFLECT_101 index
__device__ __forceinline__ int r101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - p - 2;
    return p;
}

// ---------------------------------------------------------------------------------------------
// Fused separable box filter of TWO quantities at once, marching down the image: a workgroup owns
// MW columns of a row band; every input row is loaded once into LDS (with its horizontal halo),
// the horizontal window sums come from an exact double prefix sum of the row, and every thread
// keeps the vertical running sum of its column in registers (add the entering row, subtract the
// leaving one -- cv::boxFilter's ColumnSum).  The window sums never travel
// through HBM, and the element-wise stage that follows the filter in the guided filter (a, b from
// the means; a_bar*G + b_bar from the second pair of means) runs in the store functor.
// Window: starts at p - ksz/2; mode 0: ksz taps, BORDER_REFLECT_101; mode 1: 2*(ksz/2) taps, zero
// outside [0, n-1) (the clipped half-open window of getGuidedFilter_skwgif2, gifs.cpp:591-592).
// inclusive prefix sum of a double over the 64 lanes of a wave on the DPP path (no LDS crossbar round trips):
// shifts by 1, 2, 4, 8 inside each row of 16 lanes, then the row totals are broadcast into the following rows
template <int CTRL, int ROWMASK, bool BOUND>
__device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROWMASK, 0xF, BOUND);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROWMASK, 0xF, BOUND);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_scan_f64(double s)
{
    s += dpp_f64<0x111, 0xF, true>(s);      // row_shr:1
    s += dpp_f64<0x112, 0xF, true>(s);      // row_shr:2
    s += dpp_f64<0x114, 0xF, true>(s);      // row_shr:4
    s += dpp_f64<0x118, 0xF, true>(s);      // row_shr:8
    s += dpp_f64<0x142, 0xA, false>(s);     // row_bcast:15 into rows 1 and 3
    s += dpp_f64<0x143, 0xC, false>(s);     // row_bcast:31 into rows 2 and 3
    return s;
}

#define MW 128
struct MarchGeom { int H, W, ksz, mode, band; int yA, yB; };   // output rows [yA, yB) of the image

template <class Load, class Store>
__global__ __launch_bounds__(MW) void dual_box_march_kernel(MarchGeom g, Load ld, Store st)
{
    // LDS: per input step the element-wise difference (entering row - leaving row) of both
    // quantities and its prefix sums, all in double.  No ring of past rows is kept: the row that
    // leaves the window is fetched again (its bytes are still in cache), which leaves LDS almost
    // empty and lets many workgroups share a CU -- this loop is latency-bound, not bandwidth-bound.
    __shared__ double dd0[2 * MW], dd1[2 * MW];
    __shared__ double pre0[2 * MW + 2], pre1[2 * MW + 2];
    __shared__ double wtot[2][2];
    const int H = g.H, W = g.W;
    const int a = g.ksz / 2;
    const int taps = g.mode == 0 ? g.ksz : 2 * a;
    const int rowW = MW + taps - 1;                       // <= 2 * MW: two elements per thread
    const int x0 = blockIdx.x * MW, y0 = g.yA + blockIdx.y * g.band, y1 = min(g.yB, y0 + g.band), k = blockIdx.z;
    const int tx = threadIdx.x, x = x0 + tx, wave = tx >> 6, lane = tx & 63;
    ld.begin(k);
    if (ld.skip(k)) return;      // (the same for the whole workgroup) nothing to do for this slice
    // the two row elements this thread loads: tx and tx + MW; source column and validity fixed per thread
    int xsrc[2]; bool xok[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int i = tx + MW * e, xs = x0 - a + i;
        xok[e] = i < rowW && (g.mode == 0 || (xs >= 0 && xs < W - 1));
        xsrc[e] = g.mode == 0 ? r101(xs, W) : min(max(xs, 0), W - 1);
    }
    auto row_source = [&](int j, int& ys) -> bool {       // input step j -> source row, false if it contributes zeros
        const int yy = y0 - a + j;
        if (g.mode == 0) { ys = r101(yy, H); return true; }
        ys = min(max(yy, 0), H - 1);
        return yy >= 0 && yy < H - 1;
    };
    typename Load::Raw rawNew[2], rawOld[2];
    auto request = [&](int j, typename Load::Raw (&raw)[2]) {
        int ys;
        if (j < 0 || !row_source(j, ys)) return;
#pragma unroll
        for (int e = 0; e < 2; e++)
            if (xok[e]) raw[e] = ld.fetch(xsrc[e], ys, k);
    };
    auto value = [&](int j, const typename Load::Raw (&raw)[2], int e, double& v0, double& v1) {
        int ys;
        v0 = 0.0; v1 = 0.0;
        if (j >= 0 && xok[e] && row_source(j, ys)) {
            float q0, q1;
            ld.finish(raw[e], q0, q1);
            v0 = (double)q0; v1 = (double)q1;
        }
    };
    double c0 = 0.0, c1 = 0.0;
    const int nin = (y1 - y0) + taps - 1;
    request(0, rawNew);
    request(-taps, rawOld);
    for (int j = 0; j < nin; j++) {
        __syncthreads();                       // the previous step is done with the LDS buffers
#pragma unroll
        for (int e = 0; e < 2; e++) {
            double n0, n1, o0, o1;
            value(j, rawNew, e, n0, n1);
            value(j - taps, rawOld, e, o0, o1);
            dd0[tx + MW * e] = n0 - o0; dd1[tx + MW * e] = n1 - o1;
        }
        __syncthreads();
        if (j + 1 < nin) { request(j + 1, rawNew); request(j + 1 - taps, rawOld); }
        // exact prefix sum of the difference row: thread t owns elements 2t and 2t+1
        const double2 e0 = *reinterpret_cast<const double2*>(&dd0[2 * tx]);
        const double2 e1 = *reinterpret_cast<const double2*>(&dd1[2 * tx]);
        double s0 = e0.x + e0.y, s1 = e1.x + e1.y;
        s0 = wave_scan_f64(s0); s1 = wave_scan_f64(s1);
        if (lane == 63) { wtot[wave][0] = s0; wtot[wave][1] = s1; }
        __syncthreads();
        if (wave == 1) { s0 += wtot[0][0]; s1 += wtot[0][1]; }
        // s = inclusive sum through element 2t+1; pre[i] = sum of elements [0, i)
        pre0[2 * tx + 2] = s0; pre0[2 * tx + 1] = s0 - e0.y;
        pre1[2 * tx + 2] = s1; pre1[2 * tx + 1] = s1 - e1.y;
        if (tx == 0) { pre0[0] = 0.0; pre1[0] = 0.0; }
        __syncthreads();
        // the window sums change by (entering row's window sum) - (leaving row's window sum)
        c0 += pre0[tx + taps] - pre0[tx]; c1 += pre1[tx + taps] - pre1[tx];
        if (j >= taps - 1) {
            const int y = y0 + j - (taps - 1);
            if (x < W) st(x, y, k, c0, c1);
        }
    }
}

// Same filter for the common small odd windows (mode 0, TAPS = ksz in {3,5,7,9}): one wave per
// workgroup, two columns per lane, and the ring of horizontal sums lives in registers (the row loop
// is unrolled TAPS times so that ring slots are compile-time indices).  No LDS ring means occupancy
// is set by registers, and the only LDS traffic is the current input row.
#define MWF 128   // columns per wave
template <int TAPS, class Load, class Store>
__global__ __launch_bounds__(64) void dual_box_march_small_kernel(MarchGeom g, Load ld, Store st)
{
    constexpr int A = TAPS / 2, ROWW = MWF + TAPS - 1, ROWP = (ROWW + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float row0[ROWP];
    __shared__ __attribute__((aligned(16))) float row1[ROWP];
    const int H = g.H, W = g.W;
    const int x0 = blockIdx.x * MWF, y0 = g.yA + blockIdx.y * g.band, y1 = min(g.yB, y0 + g.band), k = blockIdx.z;
    const int lane = threadIdx.x, c0 = 2 * lane, x = x0 + c0;
    ld.begin(k);
    if (ld.skip(k)) return;      // (the same for the whole workgroup) nothing to do for this slice
    // source columns of the (at most three) row elements this lane loads, reflected once
    constexpr int NLD = (ROWW + 63) / 64;
    int xsrc[NLD];
#pragma unroll
    for (int e = 0; e < NLD; e++) xsrc[e] = r101(x0 - A + lane + 64 * e, W);
    double ring[TAPS][4];               // [slot][quantity*2 + column]
    double cs[4] = {0.0, 0.0, 0.0, 0.0};
    const int nin = (y1 - y0) + TAPS - 1;
    // input rows are requested two rows ahead of the one being summed
    typename Load::Raw preA[NLD], preB[NLD];
    auto request_row = [&](int j, typename Load::Raw (&pre)[NLD]) {
        const int ys = r101(y0 - A + j, H);
#pragma unroll
        for (int e = 0; e < NLD; e++)
            if (lane + 64 * e < ROWW) pre[e] = ld.fetch(xsrc[e], ys, k);
    };
    request_row(0, preA);
    if (nin > 1) request_row(1, preB);
    for (int j0 = 0; j0 < nin; j0 += TAPS) {
#pragma unroll
        for (int u = 0; u < TAPS; u++) {
            const int j = j0 + u;
            if (j < nin) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < NLD; e++) {
                    const int i = lane + 64 * e;
                    if (i < ROWW) {
                        float q0, q1;
                        ld.finish(preA[e], q0, q1);
                        row0[i] = q0; row1[i] = q1;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < NLD; e++) preA[e] = preB[e];
                if (j + 2 < nin) request_row(j + 2, preB);
                double h[4];
                {
                    double a0 = 0.0, a1 = 0.0;
#pragma unroll
                    for (int i = 0; i < TAPS; i++) { a0 += (double)row0[c0 + i]; a1 += (double)row1[c0 + i]; }
                    h[0] = a0; h[1] = a0 - (double)row0[c0] + (double)row0[c0 + TAPS];
                    h[2] = a1; h[3] = a1 - (double)row1[c0] + (double)row1[c0 + TAPS];
                }
#pragma unroll
                for (int q = 0; q < 4; q++) { ring[u][q] = h[q]; cs[q] += h[q]; }
                if (j >= TAPS - 1) {
                    const int y = y0 + j - (TAPS - 1);
                    if (x < W) st(x, y, k, cs[0], cs[2]);
                    if (x + 1 < W) st(x + 1, y, k, cs[1], cs[3]);
#pragma unroll
                    for (int q = 0; q < 4; q++) cs[q] -= ring[(u + 1) % TAPS][q];
                }
            }
        }
    }
}

static bool march_supported(int ksz) { return ksz >= 1 && ksz <= MW; }    // a row and its halo are two elements per thread

template <class Load, class Store>
static int dual_box_march(hipStream_t s, int n, int H, int W, int ksz, int mode, int yA, int yB, const Load& ld, const Store& st)
{
    const int taps = mode == 0 ? ksz : 2 * (ksz / 2);
    if (taps < 1) { mdisp_set_error("window %d too small", ksz); return MDISP_E_BADARG; }
    MarchGeom g;
    g.H = H; g.W = W; g.ksz = ksz; g.mode = mode;
    g.band = ksz <= 9 ? 64 : (ksz <= 21 ? 128 : 256);
    g.yA = yA; g.yB = yB;
    if (yB <= yA) return MDISP_OK;
    const int nbands = (yB - yA + g.band - 1) / g.band;
    if (mode == 0 && (ksz == 3 || ksz == 5 || ksz == 7 || ksz == 9)) {
        dim3 grid((W + MWF - 1) / MWF, nbands, n);
        switch (ksz) {
        case 3: dual_box_march_small_kernel<3, Load, Store><<<grid, 64, 0, s>>>(g, ld, st); break;
        case 5: dual_box_march_small_kernel<5, Load, Store><<<grid, 64, 0, s>>>(g, ld, st); break;
        case 7: dual_box_march_small_kernel<7, Load, Store><<<grid, 64, 0, s>>>(g, ld, st); break;
        default: dual_box_march_small_kernel<9, Load, Store><<<grid, 64, 0, s>>>(g, ld, st); break;
        }
        MD_HIP(hipGetLastError());
        return MDISP_OK;
    }
    dim3 grid((W + MW - 1) / MW, nbands, n);
    dual_box_march_kernel<Load, Store><<<grid, MW, 0, s>>>(g, ld, st);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

