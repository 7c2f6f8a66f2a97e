// stereomatch_1::computeNCC (reference methods/method_StereoMatching.cpp:871-1106) for gfx950.
//
// The reference builds, for every pixel, a float support window (image padded by winSize/2 with BORDER_REFLECT, minus
// the BORDER_REFLECT_101 box mean at the pixel) and scores a candidate with
//     sum(l .* r) / (sum(l .* l) * sum(r .* r))        (float products, double sums, a PRODUCT of energies)
// Here the means and energies are computed once per view, and the cross sums per (pixel, offset) in the same order
// (row-major over the window, double accumulation of float products) as the CPU restatement this library is tested
// against.  Parity with the reference itself is unpinned: cv::boxFilter, cv::sum and cv::normalize may order their
// partial sums differently (or fuse a multiply-add), so a last-bit difference -- and with it a flipped strict '<' between
// two nearly equal scores -- is possible.
//   map overload:    offsets minD .. minD + numD - 2, strictly smallest value wins (left map); the right map is never
//                    assigned by the reference and comes out as zeros
//   volume overload: offsets minD .. minD + numD - 1, float slices normalised with cv::normalize(NORM_MINMAX)
#include "common.h"
#include <float.h>

namespace {

__device__ __forceinline__ int refl(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - p - 1;
    return p;
}
__device__ __forceinline__ int refl101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - p - 2;
    return p;
}

struct NccView { const u8* img; size_t pitch; int H, W; float* mean; double* energy; };

__device__ __forceinline__ float win_at(const NccView& v, int x, int y, int i, int j, int h, float mean)
{
    return __fsub_rn((float)v.img[(size_t)refl(y + j - h, v.H) * v.pitch + refl(x + i - h, v.W)], mean);
}

// one lane per pixel: box mean (integer sum, REFLECT_101) and the energy of the centred REFLECT window
__global__ __launch_bounds__(256) void ncc_prepare_kernel(NccView v, int win)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= v.W || y >= v.H) return;
    const int h = win / 2;
    long s = 0;
    for (int j = -h; j <= h; j++) {
        const u8* row = v.img + (size_t)refl101(y + j, v.H) * v.pitch;
        for (int i = -h; i <= h; i++) s += row[refl101(x + i, v.W)];
    }
    const float mean = (float)((double)s * (1.0 / ((double)win * win)));
    double e = 0.0;
    for (int j = 0; j < win; j++)
        for (int i = 0; i < win; i++) { const float a = win_at(v, x, y, i, j, h, mean); e += (double)__fmul_rn(a, a); }
    v.mean[(size_t)y * v.W + x] = mean;
    v.energy[(size_t)y * v.W + x] = e;
}

__device__ __forceinline__ double ncc_value(const NccView& L, int xl, const NccView& R, int xr, int y, int win)
{
    const int h = win / 2;
    const float ml = L.mean[(size_t)y * L.W + xl], mr = R.mean[(size_t)y * R.W + xr];
    double c = 0.0;
    // away from the image border (for every lane of the wave) no index is reflected: same taps in the same order, without
    // the index arithmetic per tap
    const bool inner = xl - h >= 0 && xl + h < L.W && xr - h >= 0 && xr + h < R.W && y - h >= 0 && y + h < L.H && y + h < R.H;
    if (__all(inner)) {
        const u8* rl = L.img + (size_t)(y - h) * L.pitch + (xl - h);
        const u8* rr = R.img + (size_t)(y - h) * R.pitch + (xr - h);
        for (int j = 0; j < win; j++, rl += L.pitch, rr += R.pitch) {
#pragma unroll 5
            for (int i = 0; i < win; i++) {
                const float a = __fsub_rn((float)rl[i], ml), b = __fsub_rn((float)rr[i], mr);
                c += (double)__fmul_rn(a, b);
            }
        }
    } else {
        for (int j = 0; j < win; j++) {
            const u8* rl = L.img + (size_t)refl(y + j - h, L.H) * L.pitch;
            const u8* rr = R.img + (size_t)refl(y + j - h, R.H) * R.pitch;
            for (int i = 0; i < win; i++) {
                const float a = __fsub_rn((float)rl[refl(xl + i - h, L.W)], ml), b = __fsub_rn((float)rr[refl(xr + i - h, R.W)], mr);
                c += (double)__fmul_rn(a, b);
            }
        }
    }
    return c / (L.energy[(size_t)y * L.W + xl] * R.energy[(size_t)y * R.W + xr]);
}

// left map: one lane per pixel, all offsets
__global__ __launch_bounds__(256) void ncc_map_kernel(NccView L, NccView R, int W, int win, int minD, int maxOff, float* __restrict__ out)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= L.H) return;
    double best = DBL_MAX;
    float d = 0.0f;
    for (int off = minD; off < maxOff; off++) {
        const double c = ncc_value(L, x, R, x + maxOff - off, y, win);
        if (c < best) { best = c; d = (float)off; }
    }
    out[(size_t)y * W + x] = d;
}

__device__ __forceinline__ u32 f2ord(float f) { const u32 u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(u32 o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

// one slice per blockIdx.z: raw values + the slice's minimum / maximum (ordered-integer atomics; NaN values do not
// take part, as in a comparison loop)
__global__ __launch_bounds__(256) void ncc_slice_kernel(NccView L, NccView R, int W, int win, int minD, int maxOff, int isLeft,
                                                        float* __restrict__ cost, u32* __restrict__ mm)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), k = blockIdx.z;
    const int H = isLeft ? L.H : R.H;
    float v = 0.0f;
    const bool in = x < W && y < H;
    if (in) {
        const int off = minD + k;
        v = (float)(isLeft ? ncc_value(L, x, R, x + maxOff - off, y, win) : ncc_value(L, x + off, R, x, y, win));
        cost[((size_t)k * H + y) * W + x] = v;
    }
    u32 lo = in && v == v ? f2ord(v) : 0xFFFFFFFFu, hi = in && v == v ? f2ord(v) : 0u;
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (u32)__shfl_xor((int)lo, o)); hi = max(hi, (u32)__shfl_xor((int)hi, o)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[2 * k], lo); atomicMax(&mm[2 * k + 1], hi); }
}

__global__ __launch_bounds__(256) void ncc_normalise_kernel(float* __restrict__ cost, size_t n, const u32* __restrict__ mm)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (i >= n) return;
    const double smin = (double)ord2f(mm[2 * k]), smax = (double)ord2f(mm[2 * k + 1]);
    const double scale = (smax - smin > DBL_EPSILON) ? 1.0 / (smax - smin) : 0.0;
    const float a = (float)scale, b = (float)(0.0 - smin * scale);
    float* p = cost + (size_t)k * n + i;
    *p = __fadd_rn(__fmul_rn(*p, a), b);
}

__global__ void ncc_mm_init_kernel(u32* mm, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { mm[2 * i] = 0xFFFFFFFFu; mm[2 * i + 1] = 0u; }
}

}  // namespace

size_t md_ncc_bytes(int H, int W, int minD, int numD)
{
    const size_t Wp = (size_t)W + minD + numD - 1, pp = md_align(Wp, 256);
    return pp * H + md_align((size_t)H * W * 4, 256) + md_align((size_t)H * W * 8, 256) + md_align(H * Wp * 4, 256) + md_align(H * Wp * 8, 256) +
           md_align((size_t)numD * 8, 256);
}

// dL / dR: gray views on the device; dOut: H*W floats (map) or null; dCost: numD*H*W floats (volume) or null
int md_ncc(hipStream_t s, const u8* dL, const u8* dR, size_t pitch, int H, int W, int isLeft, int win, int minD, int numD, void* ws,
           float* dOut, float* dCost)
{
    MD_REQUIRE(win >= 1 && win % 2 == 1, MDISP_E_BADARG, "winSize must be odd");
    MD_REQUIRE(numD >= 1 && minD >= 0, MDISP_E_BADARG, "numDisparity must be positive and minDisparity non-negative");
    const int maxOff = minD + numD - 1, Wp = W + maxOff;
    u8* base = (u8*)ws;
    auto take = [&](size_t bytes) { u8* r = base; base += md_align(bytes, 256); return r; };
    const size_t pp = md_align((size_t)Wp, 256);
    u8* pad = take(pp * H);
    NccView plain, padded;
    plain.H = padded.H = H;
    plain.W = W; plain.pitch = pitch; plain.mean = (float*)take((size_t)H * W * 4); plain.energy = (double*)take((size_t)H * W * 8);
    padded.W = Wp; padded.pitch = pp; padded.img = pad; padded.mean = (float*)take((size_t)H * Wp * 4); padded.energy = (double*)take((size_t)H * Wp * 8);
    u32* mm = (u32*)take((size_t)numD * 8);
    if (dOut && !isLeft) {        // the reference never assigns the right map
        MD_HIP(hipMemsetAsync(dOut, 0, (size_t)H * W * 4, s));
        if (!dCost) return MDISP_OK;
    }
    // left map: the right view padded on the left; right map: the left view padded on the right
    plain.img = isLeft ? dL : dR;
    MD_TRY(md_pad_reflect_lr(s, isLeft ? dR : dL, pitch, H, W, isLeft ? maxOff : 0, isLeft ? 0 : maxOff, pad, pp));
    ncc_prepare_kernel<<<dim3((W + 63) / 64, (H + 3) / 4), 256, 0, s>>>(plain, win);
    ncc_prepare_kernel<<<dim3((Wp + 63) / 64, (H + 3) / 4), 256, 0, s>>>(padded, win);
    const NccView& L = isLeft ? plain : padded;
    const NccView& R = isLeft ? padded : plain;
    if (dOut && isLeft) ncc_map_kernel<<<dim3((W + 63) / 64, (H + 3) / 4), 256, 0, s>>>(L, R, W, win, minD, maxOff, dOut);
    if (dCost) {
        MD_REQUIRE(numD <= 65535, MDISP_E_RANGE, "too many disparities for one launch");
        ncc_mm_init_kernel<<<(numD + 255) / 256, 256, 0, s>>>(mm, numD);
        ncc_slice_kernel<<<dim3((W + 63) / 64, (H + 3) / 4, numD), 256, 0, s>>>(L, R, W, win, minD, maxOff, isLeft, dCost, mm);
        const size_t n = (size_t)H * W;
        ncc_normalise_kernel<<<dim3((unsigned)((n + 255) / 256), numD), 256, 0, s>>>(dCost, n, mm);
    }
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}
