// Window-cost stereo kernels for gfx950: cost slice -> box aggregation -> winner-take-all fused in
// one LDS-tiled kernel, the integral-image SAD variant, the LR consistency check and the
// fixed-point bilinear remap.
//
// Replaces, in cameraCalibration/disp_method/methods_disp.cpp: the per-d absdiff / Hamming slices
// (:232-234, :336-338, :969-975), sqrBoxFilter aggregation (:343-344, :383-384, :979-980), the
// integral-image lookup (:238, :253-256), and the per-pixel minMaxLoc WTA loops (:246-262,
// :350-363, :984-1000); in methods/patchmatch.cpp:817-828 the LR check; in methods/methods.cpp:1738,1745
// cv::remap(INTER_LINEAR).
#include "common.h"
#include <algorithm>

#define BT_X 64
#define BT_Y 16

__device__ __forceinline__ int reflect101_idx(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - p - 2;
    return p;
}

struct BoxArgs {
    const u8* ref; size_t rp;     // reference-view image (raw intensities or census codes), H x W
    const u8* pad; size_t pp;     // other view padded by maxDisparity, H x Wp
    int H, W, Wp;
    int D, minD;                  // slice k <-> disparity minD + k
    int off0, offStep;            // slice k compares ref(x) with pad(x + off0 + k*offStep)
    int win, costKind, outKind;
    void* out; size_t op;         // output pitch in elements
    int RX, RY, SW, offmin;       // LDS tile geometry
};

__device__ __forceinline__ u32 pixel_cost(int kind, u32 r, u32 p)
{
    if (kind == MD_COST_HAMMING) return __popc(r ^ p);
    if (kind == MD_COST_HAMMING_SQR) { u32 c = __popc(r ^ p); return c * c; }
    u32 d = r > p ? r - p : p - r;
    return kind == MD_COST_ABSDIFF_SQR ? d * d : d;
}

// One workgroup owns a BT_X x BT_Y tile of output pixels and loops over all D slices:
//   A. per-pixel cost of the tile + halo from the LDS-resident input tiles  -> cost[] (u16)
//   B. sliding horizontal window sums                                      -> hs[]   (u32)
//   C. sliding vertical window sums, converted to float as the reference reads them
//      (methods_disp.cpp:357,399,993) and compared with the running best (strict '<' = first min).
// The window is centred with BORDER_REFLECT_101 applied to the cost slice, expressed through the
// xmap/ymap index tables so only in-image costs are ever computed.
__global__ __launch_bounds__(256) void cost_box_wta_kernel(BoxArgs a)
{
    extern __shared__ __align__(16) u8 smem[];
    const int h = a.win / 2, RX = a.RX, RY = a.RY, SW = a.SW;
    u8* refT = smem;                                             // [RY][RX]
    u8* padT = refT + md_align_dev(RY * RX, 16);                 // [RY][SW]
    u16* cost = reinterpret_cast<u16*>(padT + md_align_dev(RY * SW, 16));   // [RY][RX]
    u32* hs = reinterpret_cast<u32*>(reinterpret_cast<u8*>(cost) + md_align_dev(RY * RX * 2, 16));   // [RY][BT_X]
    short* xmap = reinterpret_cast<short*>(hs + RY * BT_X);      // [RX]
    short* ymap = xmap + md_align_dev(RX, 8);                    // [RY]

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * BT_X, y0 = blockIdx.y * BT_Y;
    const int gx0 = x0 - h, gy0 = y0 - h;
    // reflected source position in tile coordinates; the clamp only ever acts on window taps of
    // outputs that lie outside the image and are discarded
    for (int i = tid; i < RX; i += 256) xmap[i] = (short)min(max(reflect101_idx(gx0 + i, a.W) - gx0, 0), RX - 1);
    for (int i = tid; i < RY; i += 256) ymap[i] = (short)min(max(reflect101_idx(gy0 + i, a.H) - gy0, 0), RY - 1);
    for (int i = tid; i < RY * RX; i += 256) {
        int ry = i / RX, rx = i - ry * RX;
        int gy = gy0 + ry, gx = gx0 + rx;
        refT[i] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? a.ref[(size_t)gy * a.rp + gx] : 0;
    }
    for (int i = tid; i < RY * SW; i += 256) {
        int ry = i / SW, sx = i - ry * SW;
        int gy = gy0 + ry, px = gx0 + a.offmin + sx;
        padT[i] = (gy >= 0 && gy < a.H && px >= 0 && px < a.Wp) ? a.pad[(size_t)gy * a.pp + px] : 0;
    }
    __syncthreads();

    const int otx = tid & 63, oty = (tid >> 6) * 4;   // outputs (otx, oty .. oty+3)
    float best[4];
    int bestk[4] = {0, 0, 0, 0};
    const int crx = tid & 127, cry0 = tid >> 7;        // stage A mapping: column crx, rows cry0, cry0+2, ...
    const bool colIn = crx < RX && gx0 + crx >= 0 && gx0 + crx < a.W;

    for (int k = 0; k < a.D; k++) {
        const int rel = a.off0 + k * a.offStep - a.offmin;
        if (colIn) {
            for (int ry = cry0; ry < RY; ry += 2) {
                int gy = gy0 + ry;
                if (gy >= 0 && gy < a.H)
                    cost[ry * RX + crx] = (u16)pixel_cost(a.costKind, refT[ry * RX + crx], padT[ry * SW + crx + rel]);
            }
        }
        __syncthreads();
        for (int task = tid; task < RY * 4; task += 256) {
            const int ry = task >> 2, seg = task & 3;
            const int gy = gy0 + ry;
            if (gy >= 0 && gy < a.H) {
                const u16* crow = cost + ry * RX;
                int tx = seg * 16;
                u32 s = 0;
                for (int i = 0; i < a.win; i++) s += crow[xmap[tx + i]];
                hs[ry * BT_X + tx] = s;
                for (int j = 1; j < 16; j++) {
                    s += crow[xmap[tx + j - 1 + a.win]];
                    s -= crow[xmap[tx + j - 1]];
                    hs[ry * BT_X + tx + j] = s;
                }
            }
        }
        __syncthreads();
        {
            u32 s = 0;
            for (int i = 0; i < a.win; i++) s += hs[ymap[oty + i] * BT_X + otx];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j > 0) {
                    s += hs[ymap[oty + j - 1 + a.win] * BT_X + otx];
                    s -= hs[ymap[oty + j - 1] * BT_X + otx];
                }
                const float c = (float)s;
                if (k == 0 || c < best[j]) { best[j] = c; bestk[j] = k; }
            }
        }
        // the next iteration's stage-A writes touch cost[] only; hs[] is rewritten after its barrier
    }
    const int gx = x0 + otx;
    if (gx < a.W) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int gy = y0 + oty + j;
            if (gy >= a.H) break;
            const int d = bestk[j] + a.minD;
            if (a.outKind == MD_OUT_U8) reinterpret_cast<u8*>(a.out)[(size_t)gy * a.op + gx] = (u8)(d & 0xFF);
            else if (a.outKind == MD_OUT_F64) reinterpret_cast<double*>(a.out)[(size_t)gy * a.op + gx] = (double)d;
            else reinterpret_cast<float*>(a.out)[(size_t)gy * a.op + gx] = (float)d;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Marching variant of the same computation (census / absolute-difference cost -> w x w box with REFLECT_101 -> first
// minimum), without a workgroup barrier per disparity slice:
//   * a wave owns 64 consecutive columns (64 - 2h of them are outputs, h columns of halo on either side) and DCH
//     consecutive disparity slices, and marches down a band of rows;
//   * per slice a lane keeps the VERTICAL window sum of its column's costs in a register and updates it per row with the
//     cost of the row that enters the window minus the cost of the row that leaves it (both re-evaluated from the
//     census / intensity bytes: the cost is two or three instructions, cheaper than keeping w rows of it);
//   * the HORIZONTAL window sum is the difference of two entries of the wave-wide inclusive prefix sum of those
//     registers (DPP scan + two lane gathers), so any window up to the halo works with the same code;
//   * the winner over the wave's slices is kept as a key (sum << 9 | slice), the waves of a workgroup cover the
//     disparity range and meet once per row in LDS.
// Hamming costs run as packed u16 pairs (two slices per register); intensity costs as 32-bit values compared as floats,
// like the slices the reference reads (methods_disp.cpp:357,399).
__device__ __forceinline__ u32 wave_scan_incl(u32 v)
{
    // Hillis-Steele inside each row of 16 lanes, then the row totals across rows (lanes without a source add 0)
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);   // row_bcast:15 into rows 1 and 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);   // row_bcast:31 into rows 2 and 3
    return v;
}

struct MarchArgs {
    const u8* ref; size_t rp;
    const u8* pad; size_t pp;
    int H, W, D, minD, off0, offStep, win, band, outKind;
    void* out; size_t op;
};

template <int KIND, int DCH>
__global__ __launch_bounds__(512) void cost_box_march_kernel(MarchArgs a)
{
    constexpr bool PACKED = KIND == MD_COST_HAMMING || KIND == MD_COST_HAMMING_SQR;
    constexpr int NV = PACKED ? DCH / 2 : DCH, NQ = DCH / 4;
    __shared__ unsigned long long keys[2][8][64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    const int h = a.win / 2, ow = 64 - 2 * h;
    const int x0 = blockIdx.x * ow, xl = x0 - h + lane;
    const int xs = min(max(reflect101_idx(xl, a.W), 0), a.W - 1);          // the column whose costs this lane sums
    const int yb = blockIdx.y * a.band, ye = min(yb + a.band, a.H);
    const int k0 = min(wv * DCH, a.D - DCH);                                // the last chunk overlaps its predecessor
    // slice k compares ref(x) with pad(x + off0 + k*offStep): the chunk's DCH bytes are contiguous; byte j <-> slice kOf(j)
    const int base = a.offStep > 0 ? xs + a.off0 + k0 : xs + a.off0 - (k0 + DCH - 1);
    auto k_of = [&](int j) { return a.offStep > 0 ? k0 + j : k0 + DCH - 1 - j; };
    struct Row { u32 r; u32 p[NQ]; };
    auto fetch = [&](int y, Row& row) {                                     // y already reflected into the image
        row.r = a.ref[(size_t)y * a.rp + xs];
        __builtin_memcpy(row.p, a.pad + (size_t)y * a.pp + base, DCH);
    };
    u32 V[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) V[i] = 0;
    // V += sign * cost(row)
    auto accumulate = [&](const Row& row, bool add) {
        if (PACKED) {
            const u32 m = row.r * 0x01010101u;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const u32 w = row.p[q] ^ m;
                u32 c0 = (u32)__popc(w & 0x000000FFu) | ((u32)__popc(w & 0x0000FF00u) << 16);
                u32 c1 = (u32)__popc(w & 0x00FF0000u) | ((u32)__popc(w & 0xFF000000u) << 16);
                if (KIND == MD_COST_HAMMING_SQR) {
                    c0 = (c0 & 0xFFFFu) * (c0 & 0xFFFFu) | (((c0 >> 16) * (c0 >> 16)) << 16);
                    c1 = (c1 & 0xFFFFu) * (c1 & 0xFFFFu) | (((c1 >> 16) * (c1 >> 16)) << 16);
                }
                V[2 * q] = add ? V[2 * q] + c0 : V[2 * q] - c0;
                V[2 * q + 1] = add ? V[2 * q + 1] + c1 : V[2 * q + 1] - c1;
            }
        } else {
#pragma unroll
            for (int j = 0; j < DCH; j++) {
                const u32 pb = (row.p[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                u32 d = row.r > pb ? row.r - pb : pb - row.r;
                if (KIND == MD_COST_ABSDIFF_SQR) d *= d;
                V[j] = add ? V[j] + d : V[j] - d;
            }
        }
    };
    Row rin, rout;
    for (int j = -h; j <= h; j++) {                                         // the window of the band's first row
        fetch(reflect101_idx(yb + j, a.H), rin);
        accumulate(rin, true);
    }
    const int lo = lane - h - 1, hi = min(lane + h, 63);
    const bool outLane = lane >= h && lane < 64 - h && x0 + lane - h < a.W;
    for (int y = yb; y < ye; y++) {
        // requests for the next row's update go out before this row's sums are formed
        if (y + 1 < ye) {
            fetch(reflect101_idx(y + h + 1, a.H), rin);
            fetch(reflect101_idx(y - h, a.H), rout);
        }
        unsigned long long best = ~0ull;
        u32 best32 = 0xFFFFFFFFu;                                           // packed sums stay below 65536: 25-bit keys
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const u32 P = wave_scan_incl(V[i]);
            const u32 pa = (u32)__shfl((int)P, hi), pb = (u32)__shfl((int)P, max(lo, 0));
            const u32 hs = pa - (lo >= 0 ? pb : 0u);
            if (PACKED) {
                const u32 ka = ((hs & 0xFFFFu) << 9) | (u32)k_of(2 * i), kb = ((hs >> 16) << 9) | (u32)k_of(2 * i + 1);
                best32 = min(best32, min(ka, kb));
            } else {
                // the reference compares the sums as floats: order by the float's bit pattern (non-negative), then by slice
                const unsigned long long kk = ((unsigned long long)__float_as_uint((float)hs) << 9) | (u32)k_of(i);
                best = min(best, kk);
            }
        }
        if (PACKED) best = best32;
        const int par = (y - yb) & 1;
        keys[par][wv][lane] = best;
        __syncthreads();
        if (wv == (y - yb) % nw && outLane) {
            unsigned long long b = keys[par][0][lane];
            for (int w2 = 1; w2 < nw; w2++) b = min(b, keys[par][w2][lane]);
            const int d = (int)(b & 511u) + a.minD;
            const int gx = x0 + lane - h;
            if (a.outKind == MD_OUT_U8) reinterpret_cast<u8*>(a.out)[(size_t)y * a.op + gx] = (u8)(d & 0xFF);
            else if (a.outKind == MD_OUT_F64) reinterpret_cast<double*>(a.out)[(size_t)y * a.op + gx] = (double)d;
            else reinterpret_cast<float*>(a.out)[(size_t)y * a.op + gx] = (float)d;
        }
        if (y + 1 < ye) {
            accumulate(rin, true);
            accumulate(rout, false);
        }
    }
}

template <int KIND>
static int launch_march(hipStream_t s, const MarchArgs& a)
{
    const int h = a.win / 2, ow = 64 - 2 * h;
    const int dch = a.D <= 64 ? 16 : a.D <= 128 ? 32 : 64;
    const int nw = (a.D + dch - 1) / dch;
    dim3 grid((a.W + ow - 1) / ow, (a.H + a.band - 1) / a.band);
    if (dch == 16) cost_box_march_kernel<KIND, 16><<<grid, nw * 64, 0, s>>>(a);
    else if (dch == 32) cost_box_march_kernel<KIND, 32><<<grid, nw * 64, 0, s>>>(a);
    else cost_box_march_kernel<KIND, 64><<<grid, nw * 64, 0, s>>>(a);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// can the marching kernel take this problem?  (window within the halo, sums within the packed / keyed ranges, at least
// one full chunk of slices, at most eight waves)
static bool march_supported(int D, int win, int costKind)
{
    if (D < 16 || D > 512 || win > 33) return false;
    if (costKind == MD_COST_HAMMING_SQR && win > 15) return false;        // prefix sums of 64 columns stay below 65536
    return true;
}

int md_cost_box_wta(hipStream_t s, const u8* ref, size_t rp, const u8* pad, size_t pp, int H, int W,
                    int minD, int maxD, int off0, int offStep, int win, int costKind, int outKind, void* out, size_t op)
{
    BoxArgs a;
    a.ref = ref; a.rp = rp; a.pad = pad; a.pp = pp; a.H = H; a.W = W; a.Wp = W + maxD;
    a.D = maxD - minD + 1; a.minD = minD; a.off0 = off0; a.offStep = offStep;
    a.win = win; a.costKind = costKind; a.outKind = outKind; a.out = out; a.op = op;
    if (march_supported(a.D, win, costKind)) {
        MarchArgs m;
        m.ref = ref; m.rp = rp; m.pad = pad; m.pp = pp; m.H = H; m.W = W; m.D = a.D; m.minD = minD; m.off0 = off0; m.offStep = offStep;
        m.win = win; m.outKind = outKind; m.out = out; m.op = op;
        // a band re-creates its first window from w rows: bands of at least eight windows, and enough of them to fill the GPU
        m.band = std::max(64, 8 * win);
        if (costKind == MD_COST_HAMMING) return launch_march<MD_COST_HAMMING>(s, m);
        if (costKind == MD_COST_HAMMING_SQR) return launch_march<MD_COST_HAMMING_SQR>(s, m);
        if (costKind == MD_COST_ABSDIFF_SQR) return launch_march<MD_COST_ABSDIFF_SQR>(s, m);
        return launch_march<MD_COST_ABSDIFF>(s, m);
    }
    const int h = win / 2;
    a.RX = BT_X + 2 * h; a.RY = BT_Y + 2 * h;
    const int offLast = off0 + offStep * (a.D - 1);
    a.offmin = off0 < offLast ? off0 : offLast;
    const int offmax = off0 < offLast ? offLast : off0;
    a.SW = a.RX + (offmax - a.offmin);
    MD_REQUIRE(a.RX <= 128, MDISP_E_BADARG, "winSize %d too large for the box kernel (max 65)", win);
    size_t lds = md_align((size_t)a.RY * a.RX, 16) + md_align((size_t)a.RY * a.SW, 16) +
                 md_align((size_t)a.RY * a.RX * 2, 16) + (size_t)a.RY * BT_X * 4 +
                 md_align((size_t)a.RX, 8) * 2 + (size_t)a.RY * 2 + 64;
    MD_REQUIRE(lds <= 160 * 1024, MDISP_E_RANGE, "box kernel needs %zu bytes of LDS (max 163840)", lds);
    if (lds > 64 * 1024)
        MD_HIP(hipFuncSetAttribute((const void*)cost_box_wta_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((W + BT_X - 1) / BT_X, (H + BT_Y - 1) / BT_Y);
    cost_box_wta_kernel<<<grid, 256, lds, s>>>(a);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// ---------------------------------------------------------------------------------------------
// Integral-image SAD (computeSAD_inteOpti, methods_disp.cpp:207-309; stereomatch_1::computeSAD / STAD / SSD_inteOpti,
// methods/method_StereoMatching.cpp:155-221, 437-504, 722-789).  Per candidate: a difference image, its summed-area
// table integral(..., CV_32F), cost = I(x+h,y+h) - I(x+h,y-h) - I(x-h,y+h) + I(x-h,y-h) on the (H+1)x(W+1) table, first
// minimum.  The window is therefore 2h x 2h; in methods_disp.cpp any corner whose index falls outside the table reads 0
// (getMatVal, :159-171), so the last h columns / rows produce negative "garbage" costs, reproduced here exactly.
//
// Two table arithmetics.  FLOAT (the reference's): the library fills a CV_32F table as `s += src[x]` per row (float) and
// sum[y+1][x+1] = sum[y][x+1] + s, and the four corners combine left to right in float -- above 2^24 (a 255-valued
// difference image passes it after 65 793 pixels) the entries round, and the winner follows the rounding.  The row sum s
// stays below 255 * W < 2^24, so it is an exact integer and is formed as an integer prefix sum; the rounding happens in
// the serial walk down each column, which runs one lane per column (slices side by side).  INT: exact 32-bit tables
// with the same footprint and zero-outside rule; equal to FLOAT while every table entry stays below 2^24.

struct SatSlices {
    const u8* ref; size_t rp;      // the view the map is computed for
    const u8* oth; size_t op;      // kind 0: the other view padded by maxDisparity (REFLECT); kind 1: the other view itself
    int W, kind, off0, offStep;    // kind 0: slice k compares ref(x) with oth(off0 + k * offStep + x)
    int minD, square;              // kind 1: slice k compares ref(x) with oth(x - minD - k), 0 where x < minD + k; square: min(v^2, 255)
};

__device__ __forceinline__ int sat_diff(const SatSlices& g, const u8* r, const u8* o, int x, int d)
{
    if (g.kind == 0) { const int a = r[x], b = o[x]; return a > b ? a - b : b - a; }
    if (x < d) return 0;
    const int a = r[x], b = o[x - d];
    int v = a > b ? a - b : b - a;
    if (g.square) v = min(v * v, 255);
    return v;
}

// Slices are processed in batches of `nb` tables (grid y = slice) so that the serial column
// scan of one slice overlaps with those of the others.

// exclusive row prefix of the difference image: P[y][x] = sum_{x' < x} diff, x in [0, W]
__global__ __launch_bounds__(256) void sad_row_prefix_kernel(SatSlices g, int k0, int* __restrict__ Pbase, int PW, size_t tableStride)
{
    __shared__ int part[256];
    const int y = blockIdx.x, tid = threadIdx.x, k = blockIdx.y, W = g.W;
    const int chunk = (W + 255) / 256;
    const int xb = tid * chunk, xe = min(W, xb + chunk);
    const u8* r = g.ref + (size_t)y * g.rp;
    const u8* o = g.oth + (size_t)y * g.op + (g.kind == 0 ? g.off0 + (k0 + k) * g.offStep : 0);
    const int d = g.minD + k0 + k;
    int* P = Pbase + (size_t)k * tableStride;
    int s = 0;
    for (int x = xb; x < xe; x++) s += sat_diff(g, r, o, x, d);
    part[tid] = s;
    __syncthreads();
    for (int o2 = 1; o2 < 256; o2 <<= 1) {
        int v = tid >= o2 ? part[tid - o2] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;   // exclusive prefix of this thread's chunk
    int* row = P + (size_t)y * PW;
    for (int x = xb; x < xe; x++) {
        row[x] = run;
        run += sat_diff(g, r, o, x, d);
    }
    if (xe == W && xb < W) row[W] = run;
    if (W == 0 && tid == 0) row[0] = 0;
}

// in-place column prefix: I[y+1][x] = I[y][x] + P[y][x], in T's arithmetic (float: one rounding per row, the library's
// order); row 0 of I is zero.  I has H+1 rows; input row y lives at I[(y+1)*PW].  One lane per column; the loads of a
// group of rows are issued together, the sums and stores follow.
template <class T>
__global__ __launch_bounds__(64) void sad_col_prefix_kernel(int* __restrict__ Ibase, int H, int PW, size_t tableStride)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= PW) return;
    int* I = Ibase + (size_t)blockIdx.y * tableStride + x;
    T run = 0;
    auto put = [&](int y) {
        int bits;
        __builtin_memcpy(&bits, &run, 4);
        I[(size_t)y * PW] = bits;
    };
    int y = 1;
    for (; y + 7 <= H; y += 8) {
        int v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = I[(size_t)(y + q) * PW];
#pragma unroll
        for (int q = 0; q < 8; q++) { run = run + (T)v[q]; put(y + q); }
    }
    for (; y <= H; y++) { run = run + (T)I[(size_t)y * PW]; put(y); }
    I[0] = 0;
}

struct SatWta {
    int H, W, PW, h, first, d0;
    int legacy, clampT, T;       // legacy (stereomatch_1): interior pixels only, every corner inside the table, out = (char)(16 * index)
    int k0;                      // index of the batch's first slice
};

// four-corner lookups of the nb tables of a batch against the running best (strict '<': the lowest
// disparity wins ties, methods_disp.cpp:246-262)
template <class T>
__global__ __launch_bounds__(256) void sad_inte_wta_kernel(const int* __restrict__ Ibase, size_t tableStride, int nb, SatWta a,
                                                           int* __restrict__ best, int* __restrict__ bestIdx,
                                                           u8* __restrict__ out, size_t op, int last)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, h = a.h, H = a.H, W = a.W, PW = a.PW;
    if (x >= W) return;
    if (a.legacy && !(y >= h && y < H - h && x >= h && x < W - h)) { if (last) out[(size_t)y * op + x] = 0; return; }
    const bool xh = x + h <= W, xl = x - h >= 0, yh = y + h <= H, yl = y - h >= 0;   // a corner outside the table reads 0
    const size_t i = (size_t)y * W + x;
    T b = 0;
    int bk = 0;
    if (!a.first) { const int bits = best[i]; __builtin_memcpy(&b, &bits, 4); bk = bestIdx[i]; }
    bool have = !a.first;
    auto at = [&](const int* I, int yy, int xx) -> T { const int bits = I[(size_t)yy * PW + xx]; T v; __builtin_memcpy(&v, &bits, 4); return v; };
    for (int k = 0; k < nb; k++) {
        const int* I = Ibase + (size_t)k * tableStride;
        T c = xh && yh ? at(I, y + h, x + h) : (T)0;
        c = c - (xh && yl ? at(I, y - h, x + h) : (T)0);
        c = c - (xl && yh ? at(I, y + h, x - h) : (T)0);
        c = c + (xl && yl ? at(I, y - h, x - h) : (T)0);
        if (!have || c < b) { b = c; bk = a.k0 + k; have = true; }
    }
    if (last) {
        int r = a.legacy ? 16 * (a.clampT ? min(bk, a.T) : bk) : a.d0 + bk;
        out[(size_t)y * op + x] = (u8)(r & 0xFF);
    } else {
        int bits;
        __builtin_memcpy(&bits, &b, 4);
        best[i] = bits; bestIdx[i] = bk;
    }
}

// number of integral tables processed per batch: as many as fit half a gigabyte
int md_sad_inte_tables(int H, int W, int D)
{
    const size_t one = (size_t)(H + 1) * (W + 1) * sizeof(int);
    size_t n = ((size_t)512 << 20) / one;
    if (n < 1) n = 1;
    if (n > (size_t)D) n = D;
    return (int)n;
}

// I: md_sad_inte_tables() tables of (H+1)*(W+1) words, best: 2*H*W words of device scratch
static int sat_run(hipStream_t s, const SatSlices& g, int H, int D, SatWta w, bool floatTables, u8* out, size_t op, int* I, int* best)
{
    const int W = g.W;
    MD_REQUIRE((double)H * W * 255.0 < 2147483647.0, MDISP_E_RANGE, "integral table overflows 32 bits for %dx%d", W, H);
    // the float row sums are formed as integers: exact (and equal to the library's float `s`) below 2^24
    MD_REQUIRE(!floatTables || (double)W * 255.0 < 16777216.0, MDISP_E_RANGE, "image wider than 65793 columns");
    const int PW = W + 1;
    const int n = md_sad_inte_tables(H, W, D);
    const size_t stride = (size_t)(H + 1) * PW;
    w.H = H; w.W = W; w.PW = PW;
    for (int k0 = 0; k0 < D; k0 += n) {
        const int nb = D - k0 < n ? D - k0 : n;
        sad_row_prefix_kernel<<<dim3(H, nb), 256, 0, s>>>(g, k0, I + PW, PW, stride);
        const dim3 gc((PW + 63) / 64, nb), grid((W + 255) / 256, H);
        w.first = k0 == 0; w.k0 = k0;
        const int last = k0 + nb >= D;
        if (floatTables) {
            sad_col_prefix_kernel<float><<<gc, 64, 0, s>>>(I, H, PW, stride);
            sad_inte_wta_kernel<float><<<grid, 256, 0, s>>>(I, stride, nb, w, best, best + (size_t)H * W, out, op, last);
        } else {
            sad_col_prefix_kernel<int><<<gc, 64, 0, s>>>(I, H, PW, stride);
            sad_inte_wta_kernel<int><<<grid, 256, 0, s>>>(I, stride, nb, w, best, best + (size_t)H * W, out, op, last);
        }
    }
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

int md_sad_inte(hipStream_t s, const u8* ref, size_t rp, const u8* pad, size_t pp, int H, int W,
                int minD, int maxD, int off0, int offStep, int win, int floatTables, u8* out, size_t op, int* I, int* best)
{
    SatSlices g{};
    g.ref = ref; g.rp = rp; g.oth = pad; g.op = pp; g.W = W; g.kind = 0; g.off0 = off0; g.offStep = offStep; g.minD = minD;
    SatWta w{};
    w.h = win / 2; w.d0 = minD;
    return sat_run(s, g, H, maxD - minD + 1, w, floatTables != 0, out, op, I, best);
}

// the _inteOpti matchers of namespace stereomatch_1: difference 0 left of the candidate's offset, interior pixels only,
// out = (char)(16 * index) with the index clamped to T by the "T" variants
int md_legacy_inte(hipStream_t s, const u8* L, size_t lp, const u8* R, size_t rp, int H, int W, int minD, int numD, int win,
                   int square, int clampT, int T, int floatTables, u8* out, size_t op, int* I, int* best)
{
    SatSlices g{};
    g.ref = L; g.rp = lp; g.oth = R; g.op = rp; g.W = W; g.kind = 1; g.minD = minD; g.square = square;
    SatWta w{};
    w.h = win / 2; w.legacy = 1; w.clampT = clampT; w.T = T;
    return sat_run(s, g, H, numD, w, floatTables != 0, out, op, I, best);
}

// ---------------------------------------------------------------------------------------------
// PatchMatch::postProcess validity test, methods/patchmatch.cpp:817-828 (the right-view clamp uses
// `rows` as the reference does; an index of `cols` is clamped to the last column).
__global__ __launch_bounds__(256) void lr_check_kernel(const float* __restrict__ dL, const float* __restrict__ dR, int H, int W,
                                                       float thresh, u8* __restrict__ vL, u8* __restrict__ vR)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t row = (size_t)y * W;
    const float a = dL[row + x];
    int xr = (int)fmaxf(0.f, fminf((float)W, (float)x - a));
    xr = min(xr, W - 1);
    vL[row + x] = (u8)(fabsf(a - dR[row + xr]) <= thresh);
    const float b = dR[row + x];
    int xl = (int)fmaxf(0.f, fminf((float)H, (float)x + b));
    xl = min(xl, W - 1);
    vR[row + x] = (u8)(fabsf(b - dL[row + xl]) <= thresh);
}

int md_lr_check(hipStream_t s, const float* dL, const float* dR, int H, int W, float thresh, u8* vL, u8* vR)
{
    dim3 grid((W + 255) / 256, H);
    lr_check_kernel<<<grid, 256, 0, s>>>(dL, dR, H, W, thresh, vL, vR);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// ---------------------------------------------------------------------------------------------
// cv::remap(INTER_LINEAR) for 8-bit images with CV_16SC2 + CV_16UC1 fixed-point maps:
// 5 fractional bits, int16 weights summing to 32768, (sum + 2^14) >> 15, constant-0 border.
// One thread per destination pixel; map reads are coalesced, the four taps are gathered.
template <int CN>
__global__ __launch_bounds__(256) void remap_kernel(const u8* __restrict__ src, size_t sp, int sH, int sW,
                                                    const int16_t* __restrict__ m1, const u16* __restrict__ m2,
                                                    int dH, int dW, u8* __restrict__ dst, size_t dp)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dW) return;
    const size_t i = (size_t)y * dW + x;
    const u32 xy = reinterpret_cast<const u32*>(m1)[i];
    const int sx = (int16_t)(xy & 0xFFFF), sy = (int16_t)(xy >> 16);
    const int f = m2[i] & 1023;
    const int fa = f & 31, fb = f >> 5;
    int w0 = (32 - fa) * (32 - fb) * 32, w1 = fa * (32 - fb) * 32, w2 = (32 - fa) * fb * 32, w3 = fa * fb * 32;
    if (w0 > 32767) { w0 = 32767; w3 += 1; }   // int16 saturation of the (0,0) entry and its sum correction
    const bool x0ok = sx >= 0 && sx < sW, x1ok = sx + 1 >= 0 && sx + 1 < sW;
    const bool y0ok = sy >= 0 && sy < sH, y1ok = sy + 1 >= 0 && sy + 1 < sH;
#pragma unroll
    for (int c = 0; c < CN; c++) {
        int p00 = (x0ok && y0ok) ? src[(size_t)sy * sp + (size_t)sx * CN + c] : 0;
        int p01 = (x1ok && y0ok) ? src[(size_t)sy * sp + (size_t)(sx + 1) * CN + c] : 0;
        int p10 = (x0ok && y1ok) ? src[(size_t)(sy + 1) * sp + (size_t)sx * CN + c] : 0;
        int p11 = (x1ok && y1ok) ? src[(size_t)(sy + 1) * sp + (size_t)(sx + 1) * CN + c] : 0;
        int r = (p00 * w0 + p01 * w1 + p10 * w2 + p11 * w3 + (1 << 14)) >> 15;
        dst[(size_t)y * dp + (size_t)x * CN + c] = (u8)min(max(r, 0), 255);
    }
}

int md_remap(hipStream_t s, const u8* src, size_t sp, int sH, int sW, int cn, const int16_t* m1, const u16* m2,
             int dH, int dW, u8* dst, size_t dp)
{
    dim3 grid((dW + 255) / 256, dH);
    if (cn == 1) remap_kernel<1><<<grid, 256, 0, s>>>(src, sp, sH, sW, m1, m2, dH, dW, dst, dp);
    else if (cn == 3) remap_kernel<3><<<grid, 256, 0, s>>>(src, sp, sH, sW, m1, m2, dH, dW, dst, dp);
    else { mdisp_set_error("remap supports 1 or 3 channels, got %d", cn); return MDISP_E_BADARG; }
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// cv::initUndistortRectifyMap(..., CV_16SC2, map1, map2) on the device: one thread per destination row,
// walking the row with the library's incremental sums (_x += ir[0] ...), so that every map entry
// equals the host construction bit for bit (double arithmetic, no contraction).
struct MapArgs { double ir[9], k[8], fx, fy, u0, v0; int W, H; int16_t* m1; u16* m2; };
__global__ __launch_bounds__(64) void rectify_maps_kernel(MapArgs a)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.H) return;
    const double k1 = a.k[0], k2 = a.k[1], p1 = a.k[2], p2 = a.k[3], k3 = a.k[4], k4 = a.k[5], k5 = a.k[6], k6 = a.k[7];
    double _x = i * a.ir[1] + a.ir[2], _y = i * a.ir[4] + a.ir[5], _w = i * a.ir[7] + a.ir[8];
    for (int j = 0; j < a.W; j++, _x += a.ir[0], _y += a.ir[3], _w += a.ir[6]) {
        const double w = 1. / _w, x = _x * w, y = _y * w;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
        const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
        const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
        double u = (a.fx * xd + a.u0) * 32, v = (a.fy * yd + a.v0) * 32;
        u = fmin(fmax(u, -2147483648.0), 2147483647.0);
        v = fmin(fmax(v, -2147483648.0), 2147483647.0);
        const int iu = (int)rint(u), iv = (int)rint(v);        // round half to even, as cvRound
        const size_t o = (size_t)i * a.W + j;
        a.m1[o * 2] = (int16_t)(iu >> 5);
        a.m1[o * 2 + 1] = (int16_t)(iv >> 5);
        a.m2[o] = (u16)((iv & 31) * 32 + (iu & 31));
    }
}

int md_rectify_maps(hipStream_t s, const double ir[9], const double k[8], double fx, double fy, double u0, double v0, int W, int H,
                    int16_t* m1, u16* m2)
{
    MapArgs a;
    for (int i = 0; i < 9; i++) a.ir[i] = ir[i];
    for (int i = 0; i < 8; i++) a.k[i] = k[i];
    a.fx = fx; a.fy = fy; a.u0 = u0; a.v0 = v0; a.W = W; a.H = H; a.m1 = m1; a.m2 = m2;
    rectify_maps_kernel<<<(H + 63) / 64, 64, 0, s>>>(a);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// ---------------------------------------------------------------------------------------------
// Disparity -> 3-D points (cameraCalibration/methods/methods.cpp:2039-2076 and :2300-2352), double
// arithmetic per pixel, float output (x, y, z interleaved as in pcl::PointXYZ).
struct Mat44 { double m[16]; };
__global__ __launch_bounds__(256) void reproject_q_kernel(const float* __restrict__ disp, int H, int W, Mat44 Q, float* __restrict__ xyz)
{
    const int v = blockIdx.x * 256 + threadIdx.x, u = blockIdx.y;     // the reference multiplies Q by (row, col, d, 1)
    if (v >= W) return;
    const double d = (double)disp[(size_t)u * W + v];
    double X = 0, Y = 0, Z = 0;
    if (d != 0) {
        const double in[4] = {(double)u, (double)v, d, 1.0};
        double o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) s = __dadd_rn(s, __dmul_rn(Q.m[i * 4 + k], in[k]));
            o[i] = s;
        }
        X = __ddiv_rn(o[0], o[3]); Y = __ddiv_rn(o[1], o[3]); Z = __ddiv_rn(o[2], o[3]);
    }
    float* p = xyz + ((size_t)u * W + v) * 3;
    p[0] = (float)__ddiv_rn(X, 100.0); p[1] = (float)__ddiv_rn(Y, 100.0); p[2] = (float)__ddiv_rn(Z, 100.0);
}

__global__ __launch_bounds__(256) void reproject_tri_kernel(const float* __restrict__ disp, int H, int W, double f0, double f1, double u0,
                                                            double v0, double baseline, const float* __restrict__ bgr,
                                                            float* __restrict__ xyz)
{
    const int u = blockIdx.x * 256 + threadIdx.x, v = blockIdx.y;
    if (u >= W) return;
    const double dv = __dsub_rn((double)v, v0);
    const double temp1 = __dadd_rn(__dmul_rn(dv, dv), __dmul_rn(f1, f1));
    const double coeff = __ddiv_rn(__dmul_rn(f0, f1), sqrt(temp1));
    const double px0 = __ddiv_rn(__dsub_rn((double)u, u0), f0), px1 = __ddiv_rn(dv, f1);
    const double d = (double)disp[(size_t)v * W + u];
    double X = 0, Y = 0, Z = 0;
    if (d != 0) {
        Z = __ddiv_rn(__dmul_rn(baseline, coeff), fabs(d));
        Y = __dmul_rn(px1, Z);
        X = __dmul_rn(px0, sqrt(__dadd_rn(__dmul_rn(Z, Z), __dmul_rn(Y, Y))));
    }
    float* p = xyz + ((size_t)v * W + u) * 3;
    bool keep = !(Z > 20000);
    if (keep && bgr) {
        const float* c = bgr + ((size_t)v * W + u) * 3;
        keep = !(c[0] == 0 && c[1] == 0 && c[2] == 0);
    }
    p[0] = keep ? (float)X : 0.f; p[1] = keep ? (float)Y : 0.f; p[2] = keep ? (float)Z : 0.f;
}

int md_reproject_q(hipStream_t s, const float* disp, int H, int W, const double Q[16], float* xyz)
{
    Mat44 q;
    memcpy(q.m, Q, sizeof q.m);
    reproject_q_kernel<<<dim3((W + 255) / 256, H), 256, 0, s>>>(disp, H, W, q, xyz);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}
int md_reproject_tri(hipStream_t s, const float* disp, int H, int W, const double K[9], double baseline, const float* bgr, float* xyz)
{
    reproject_tri_kernel<<<dim3((W + 255) / 256, H), 256, 0, s>>>(disp, H, W, K[0], K[4], K[2], K[5], baseline, bgr, xyz);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}
