// Fast global smoother = the post-filter of postProcess_ (disp_method/methods_disp.cpp:1299-1313:
// ximgproc::createDisparityWLSFilterGeneric(false), lambda 8000, sigmaColor 0.5, ROI = the whole frame; without
// confidence that filter is fastGlobalSmootherFilter(view, disparity, out, lambda, sigma_color), 3 iterations,
// lambda_attenuation 0.25).
//
// One iteration = a tridiagonal solve along every row, then along every column, in float, with the elimination order of
// the library's scalar code (see oracle/orc_fgs.c).  Each line is a serial recurrence, so a pass runs one lane per line:
// the vertical pass walks down the image with coalesced row accesses; the horizontal pass is the same kernel on
// transposed planes (weights transposed once, the signal before and after each pass through an LDS tile).
#include "common.h"
#include <math.h>
#include <vector>

namespace {

__global__ __launch_bounds__(256) void fgs_weights_kernel(const u8* guide, size_t gp, int H, int W, int cn, const float* lut,
                                                          float* ChT /* [W][H] */, float* Cv /* [H][W] */)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const u8* p = guide + (size_t)y * gp + (size_t)x * cn;
    int dh = 0, dv = 0;
    for (int c = 0; c < cn; c++) {
        if (x + 1 < W) { const int d = (int)p[c] - (int)p[cn + c]; dh += d * d; }
        if (y + 1 < H) { const int d = (int)p[c] - (int)p[gp + c]; dv += d * d; }
    }
    ChT[(size_t)x * H + y] = x + 1 < W ? lut[dh] : 0.0f;
    Cv[(size_t)y * W + x] = y + 1 < H ? lut[dv] : 0.0f;
}

// dst[x][y] = src[y][x]   (src is rows x cols)
__global__ __launch_bounds__(256) void fgs_transpose_kernel(const float* src, int rows, int cols, float* dst)
{
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int y = by + j, x = bx + tx;
        if (y < rows && x < cols) tile[j][tx] = src[(size_t)y * cols + x];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int x = bx + j, y = by + tx;
        if (y < rows && x < cols) dst[(size_t)x * rows + y] = tile[tx][j];
    }
}

// One lane per column of an n x m plane (n steps along the line, m lines): forward elimination, back substitution.
constexpr int FGS_PF = 8;
__global__ __launch_bounds__(64) void fgs_line_pass_kernel(float* u, const float* C, float* interD, int n, int m, float lambda)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= m) return;
    float coef_cur = lambda * C[j];
    float denom = 1.0f - coef_cur;
    float dprev = coef_cur / denom, uprev = u[j] / denom;
    interD[j] = dprev;
    u[j] = uprev;
    // Each line is a serial recurrence of a few operations per element, so a step must never wait for memory: the
    // operands of FGS_PF elements ahead are in flight, requested unconditionally (past the end the last element is asked
    // for again and dropped) so that the waits the compiler places cover only requests issued FGS_PF steps earlier.
    float cq[FGS_PF], uq[FGS_PF];
#pragma unroll
    for (int k = 0; k < FGS_PF; k++) {
        const size_t r = (size_t)min(1 + k, n - 1) * m + j;
        cq[k] = C[r]; uq[k] = u[r];
    }
    // whole groups of FGS_PF steps run without an exit in the middle (an early exit makes the compiler rotate the ring
    // through copies, which wait for the requests just issued); the last few steps load directly
    int i = 1;
    for (; i + FGS_PF <= n; i += FGS_PF) {
#pragma unroll
        for (int k = 0; k < FGS_PF; k++) {
            const float coef_prev = coef_cur;
            coef_cur = lambda * cq[k];
            const float ui = uq[k];
            const size_t r = (size_t)min(i + k + FGS_PF, n - 1) * m + j;
            cq[k] = C[r]; uq[k] = u[r];
            denom = (1.0f - coef_prev - coef_cur) - dprev * coef_prev;
            dprev = coef_cur / denom;
            uprev = (ui - uprev * coef_prev) / denom;
            interD[(size_t)(i + k) * m + j] = dprev;
            u[(size_t)(i + k) * m + j] = uprev;
        }
    }
    for (; i < n; i++) {
        const float coef_prev = coef_cur;
        coef_cur = lambda * C[(size_t)i * m + j];
        const float ui = u[(size_t)i * m + j];
        denom = (1.0f - coef_prev - coef_cur) - dprev * coef_prev;
        dprev = coef_cur / denom;
        uprev = (ui - uprev * coef_prev) / denom;
        interD[(size_t)i * m + j] = dprev;
        u[(size_t)i * m + j] = uprev;
    }
    float unext = uprev;
    float dq[FGS_PF];
#pragma unroll
    for (int k = 0; k < FGS_PF; k++) {
        const size_t r = (size_t)max(n - 2 - k, 0) * m + j;
        uq[k] = u[r]; dq[k] = interD[r];
    }
    i = n - 2;
    for (; i - (FGS_PF - 1) >= 0; i -= FGS_PF) {
#pragma unroll
        for (int k = 0; k < FGS_PF; k++) {
            unext = uq[k] - dq[k] * unext;
            const size_t r = (size_t)max(i - k - FGS_PF, 0) * m + j;
            uq[k] = u[r]; dq[k] = interD[r];
            u[(size_t)(i - k) * m + j] = unext;
        }
    }
    for (; i >= 0; i--) {
        unext = u[(size_t)i * m + j] - interD[(size_t)i * m + j] * unext;
        u[(size_t)i * m + j] = unext;
    }
}

__global__ __launch_bounds__(256) void fgs_s16_to_f32_kernel(const int16_t* s, float* d, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = (float)s[i];
}
__global__ __launch_bounds__(256) void fgs_f32_to_s16_kernel(const float* s, int16_t* d, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float r = rintf(s[i]);                                 // cv::saturate_cast<short>(cvRound(v))
        d[i] = (int16_t)(r < -32768.0f ? -32768 : r > 32767.0f ? 32767 : (int)r);
    }
}

}  // namespace

// lut + ChT + Cv + uT + interD
size_t md_fgs_bytes(int H, int W)
{
    const size_t plane = md_align((size_t)H * W * sizeof(float), 256);
    return md_align((3 * 255 * 255 + 1) * sizeof(float), 256) + 4 * plane;
}

// dGuide: 8-bit, cn = 1 or 3 interleaved; dU: H*W floats, filtered in place
int md_fgs_filter(hipStream_t s, const u8* dGuide, size_t gpitch, int H, int W, int cn, float* dU, double lambda_, double sigmaColor,
                  double attenuation, int iters, void* ws)
{
    MD_REQUIRE(lambda_ >= 0 && sigmaColor > 0 && iters >= 1, MDISP_E_BADARG, "lambda >= 0, sigma_color > 0, num_iter >= 1 are required");
    const size_t plane = md_align((size_t)H * W * sizeof(float), 256), nl = 3 * 255 * 255 + 1;
    u8* base = (u8*)ws;
    float* lut = (float*)base;
    float* ChT = (float*)(base + md_align(nl * sizeof(float), 256));
    float* Cv = (float*)((u8*)ChT + plane);
    float* uT = (float*)((u8*)Cv + plane);
    float* interD = (float*)((u8*)uT + plane);
    {
        // the table is evaluated on the host with the C library's expf / sqrtf, as the library does, and uploaded
        std::vector<float> h(nl);
        const float sc = (float)sigmaColor;
        const size_t used = (size_t)cn * 255 * 255 + 1;
        for (size_t i = 0; i < used; i++) h[i] = -expf(-sqrtf((float)i) / sc);
        MD_HIP(hipMemcpyAsync(lut, h.data(), used * sizeof(float), hipMemcpyHostToDevice, s));
        MD_HIP(hipStreamSynchronize(s));                             // h goes out of scope
    }
    fgs_weights_kernel<<<dim3((W + 255) / 256, H), 256, 0, s>>>(dGuide, gpitch, H, W, cn, lut, ChT, Cv);
    float lambda = (float)lambda_;
    const dim3 tg((W + 31) / 32, (H + 31) / 32), tgT((H + 31) / 32, (W + 31) / 32);
    for (int it = 0; it < iters; it++) {
        fgs_transpose_kernel<<<tg, 256, 0, s>>>(dU, H, W, uT);                                         // rows become columns
        fgs_line_pass_kernel<<<(H + 63) / 64, 64, 0, s>>>(uT, ChT, interD, W, H, lambda);             // horizontal pass
        fgs_transpose_kernel<<<tgT, 256, 0, s>>>(uT, W, H, dU);
        fgs_line_pass_kernel<<<(W + 63) / 64, 64, 0, s>>>(dU, Cv, interD, H, W, lambda);              // vertical pass
        lambda *= (float)attenuation;
    }
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

int md_fgs_convert(hipStream_t s, const int16_t* src, float* dst, size_t n)
{
    fgs_s16_to_f32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(src, dst, n);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}
int md_fgs_convert_back(hipStream_t s, const float* src, int16_t* dst, size_t n)
{
    fgs_f32_to_s16_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(src, dst, n);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}
