// Semi-global matching on gfx950: census -> Hamming cost -> 4/5/8 path scans -> WTA with
// uniqueness test, x16 parabola refinement and the disp2 left-right check.
//
// Replaces the cv::StereoSGBM::compute call sites of cameraCalibration/methods/methods.cpp:1883-1921,
// :1216-1219/:1291 and :1565-1568/:1650 (recursion, uniqueness, sub-pixel and disp12MaxDiff rules of
// OpenCV 4.1.0's StereoSGBM), with the matching cost the reference owns: the Hamming distance of
// census codes of the left image and of the reflect-padded right image, built exactly as
// censusStereo's left-disparity branch does (disp_method/methods_disp.cpp:859-982).
//
// Path costs are kept in the biased form L' = L + P2 >= 0 (so they fit 16 bits):
//   L'(p,d) = C(p,d) + min(L'(p-r,d), L'(p-r,d-1)+P1, L'(p-r,d+1)+P1, min_k L'(p-r,k)+P2) - min_k L'(p-r,k)
// and S' = sum of L' = S + npaths*P2; the bias is removed where S enters the uniqueness ratio.
#include "common.h"
#include <stdlib.h>

#define SGM_BIG 0x3FFF0000   // "infinite" path cost for disparities outside [0, D)

__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ u32 wave_min_u32(u32 v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, (u32)__shfl_xor((int)v, m, 64));
    return v;
}

struct SgmArgs {
    const u8* cL; size_t pL;      // census of left image,          H x W
    const u8* cR; size_t pR;      // census of padded right image,  H x (W + maxD)
    u16* S;                       // [H][W][D]
    int H, W, D, minD, maxD, P1, P2;
};

template <int ND> struct SVecT;
template <> struct SVecT<1> { typedef u16 type; };
template <> struct SVecT<2> { typedef u32 type; };
template <> struct SVecT<4> { typedef uint2 type; };
template <> struct SVecT<8> { typedef uint4 type; };

// per-step operands fetched ahead of the dependent recursion
template <int ND> struct StepIn {
    unsigned long long rbits;   // ND census bytes of the right image, byte j at bits 8*(ND-1-j)
    u32 cl;                     // census byte of the left pixel
    u16 sold[ND];               // S before this direction
};

template <int ND, bool FIRST>
__device__ __forceinline__ void sgm_fetch(const SgmArgs& a, int x, int y, int k0, bool full, StepIn<ND>& in)
{
    const int D = a.D;
    in.cl = a.cL[(size_t)y * a.pL + x];
    const u8* rb = a.cR + (size_t)y * a.pR + x + a.maxD - a.minD;   // disparity index k -> rb[-k]
    in.rbits = 0;
    if (full) {
        __builtin_memcpy(&in.rbits, rb - k0 - (ND - 1), ND);
    } else {
#pragma unroll
        for (int j = 0; j < ND; j++)
            if (k0 + j < D) in.rbits |= (unsigned long long)rb[-(k0 + j)] << (8 * (ND - 1 - j));
    }
    if (!FIRST) {
        const u16* sp = a.S + ((size_t)y * a.W + x) * D + k0;
        if (full && (D % ND == 0)) {
            typename SVecT<ND>::type v = *reinterpret_cast<const typename SVecT<ND>::type*>(sp);
            __builtin_memcpy(in.sold, &v, sizeof(v));
        } else {
#pragma unroll
            for (int j = 0; j < ND; j++) in.sold[j] = (k0 + j < D) ? sp[j] : 0;
        }
    }
}

// One wave per scan line, lane l owns disparities l*ND .. l*ND+ND-1.
// FIRST: this direction initialises S (store only) instead of accumulating into it.
// Operands of step t+4 are requested right after step t retires, so four S rows (512 B each at
// D = 256) stay in flight per wave while the serial recursion runs.
template <int ND, bool FIRST>
__global__ __launch_bounds__(256) void sgm_path_kernel(SgmArgs a, int dx, int dy, int nlines)
{
    const int lane = threadIdx.x & 63;
    const int line = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= nlines) return;
    const int H = a.H, W = a.W, D = a.D;
    int x0, y0;
    if (dy == 0) { y0 = line; x0 = dx > 0 ? 0 : W - 1; }
    else {
        const int ys = dy > 0 ? 0 : H - 1;
        if (line < W) { x0 = line; y0 = ys; }
        else { const int i = line - W; y0 = dy > 0 ? 1 + i : H - 2 - i; x0 = dx > 0 ? 0 : W - 1; }
    }
    int n;   // pixels on this line
    {
        int nx = dx > 0 ? W - x0 : (dx < 0 ? x0 + 1 : 0x7FFFFFFF);
        int ny = dy > 0 ? H - y0 : (dy < 0 ? y0 + 1 : 0x7FFFFFFF);
        n = min(nx, ny);
    }
    const int k0 = lane * ND;
    const bool full = (k0 + ND <= D);
    int Lp[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) Lp[j] = (k0 + j < D) ? 0 : SGM_BIG;   // predecessor outside the image: L' = 0
    int minPrev = 0;
    const int P1 = a.P1, P2 = a.P2;
    constexpr int PF = 4;
    StepIn<ND> ring[PF];
#pragma unroll
    for (int u = 0; u < PF; u++)
        if (u < n) sgm_fetch<ND, FIRST>(a, x0 + u * dx, y0 + u * dy, k0, full, ring[u]);

    for (int t0 = 0; t0 < n; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int t = t0 + u;
            if (t >= n) break;
            const int x = x0 + t * dx, y = y0 + t * dy;
            const StepIn<ND> in = ring[u];
            int left = __shfl_up(Lp[ND - 1], 1, 64);
            int right = __shfl_down(Lp[0], 1, 64);
            if (lane == 0) left = SGM_BIG;
            if (lane == 63) right = SGM_BIG;
            int Ln[ND];
            int mn = SGM_BIG;
            const int e = minPrev + P2;
#pragma unroll
            for (int j = 0; j < ND; j++) {
                const int k = k0 + j;
                const u32 rbyte = (u32)(in.rbits >> (8 * (ND - 1 - j))) & 0xFFu;
                const int c = __popc(in.cl ^ rbyte);
                const int lo = (j > 0) ? Lp[j - 1] : left;
                const int hi = (j < ND - 1) ? Lp[j + 1] : right;
                const int m = min(min(Lp[j], lo + P1), min(hi + P1, e));
                int L = c + m - minPrev;
                if (k >= D) L = SGM_BIG;
                Ln[j] = L;
                mn = min(mn, L);
            }
            mn = wave_min_i32(mn);
            u16 so[ND];
#pragma unroll
            for (int j = 0; j < ND; j++) {
                so[j] = (u16)((FIRST ? 0 : (int)in.sold[j]) + Ln[j]);
                Lp[j] = Ln[j];
            }
            u16* sp = a.S + ((size_t)y * W + x) * D + k0;
            if (full && (D % ND == 0)) {
                typename SVecT<ND>::type v;
                __builtin_memcpy(&v, so, sizeof(v));
                *reinterpret_cast<typename SVecT<ND>::type*>(sp) = v;
            } else {
#pragma unroll
                for (int j = 0; j < ND; j++)
                    if (k0 + j < D) sp[j] = so[j];
            }
            minPrev = mn;
            if (t + PF < n) sgm_fetch<ND, FIRST>(a, x0 + (t + PF) * dx, y0 + (t + PF) * dy, k0, full, ring[u]);
        }
    }
}

struct WtaArgs {
    const u16* S; int sstride;      // u16 elements between consecutive pixels of S
    int H, W, D, minD;
    int bias, uniq, maxDiff, subpixel;
    int16_t* disp; size_t dpitch;   // in elements
    size_t strideS, strideDisp;     // per-pair strides in elements (blockIdx.y = pair)
};

// One workgroup per image row.  Phase 1 (one wave per pixel, pixels strided over the 4 waves):
// first-minimum WTA over d, uniqueness test, sub-pixel parabola, and the disp2 slot election
// (minimum S, ties to the larger x -- StereoSGBM visits x right to left with a strict '>' test)
// by a packed atomicMin in LDS.  Phase 2: the disp12MaxDiff check against disp2.
template <int ND>
__global__ __launch_bounds__(256) void sgm_wta_kernel(WtaArgs a)
{
    extern __shared__ u32 lds[];
    u32* key = lds;                                   // [W] (S'min << 16) | (0xFFFF - x)
    int16_t* d1 = reinterpret_cast<int16_t*>(lds + a.W);   // [W]
    const int y = blockIdx.x;
    const int W = a.W, D = a.D;
    const int INVALID = (a.minD - 1) * 16;
    const u16* Sbase = a.S + (size_t)blockIdx.y * a.strideS;
    int16_t* dispBase = a.disp + (size_t)blockIdx.y * a.strideDisp;
    for (int i = threadIdx.x; i < W; i += 256) { key[i] = 0xFFFFFFFFu; d1[i] = (int16_t)INVALID; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k0 = lane * ND;
    for (int x = wv; x < W; x += 4) {
        const u16* sp = Sbase + ((size_t)y * W + x) * a.sstride;
        u32 s[ND];
        u32 best = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < ND; j++) {
            s[j] = (k0 + j < D) ? sp[k0 + j] : 0xFFFFu;
            u32 kk = (k0 + j < D) ? ((s[j] << 16) | (u32)(k0 + j)) : 0xFFFFFFFFu;
            best = min(best, kk);
        }
        best = wave_min_u32(best);
        const int bk = (int)(best & 0xFFFF);
        const int minS = (int)(best >> 16) - a.bias;
        bool bad = false;
#pragma unroll
        for (int j = 0; j < ND; j++) {
            const int k = k0 + j;
            if (k < D) bad |= (((int)s[j] - a.bias) * (100 - a.uniq) < minS * 100) && (abs(bk - k) > 1);
        }
        if (__any(bad)) continue;
        if (lane == 0) {
            int d = bk;
            const int x2 = x - d - a.minD;
            if (x2 >= 0 && x2 < W) atomicMin(&key[x2], ((best >> 16) << 16) | (u32)(0xFFFF - x));
            if (a.subpixel && d > 0 && d < D - 1) {
                const int sm = sp[d - 1], s0 = sp[d], spl = sp[d + 1];
                int den = sm + spl - 2 * s0;
                if (den < 1) den = 1;
                d = d * 16 + ((sm - spl) * 16 + den) / (den * 2);
            } else d *= 16;
            d1[x] = (int16_t)(d + a.minD * 16);
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        int v = d1[x];
        if (v != INVALID) {
            // disp2[xx]: disparity elected for column xx of the right view, INVALID when unset
            auto inconsistent = [&](int xx, int dd) -> bool {
                if (xx < 0 || xx >= W) return false;
                const u32 kk = key[xx];
                const int d2 = (kk == 0xFFFFFFFFu) ? INVALID : (0xFFFF - (int)(kk & 0xFFFF)) - xx;
                return d2 >= a.minD && abs(d2 - dd) > a.maxDiff;
            };
            const int _d = v >> 4, d_ = (v + 15) >> 4;
            if (inconsistent(x - _d, _d) && inconsistent(x - d_, d_)) v = INVALID;
        }
        dispBase[(size_t)y * a.dpitch + x] = (int16_t)v;
    }
}

// ---------------------------------------------------------------------------------------------

static const int DIRS[8][2] = {{+1, 0}, {-1, 0}, {0, +1}, {0, -1}, {+1, +1}, {-1, +1}, {+1, -1}, {-1, -1}};

int md_sgm_plan_make(int H, int W, const mdisp_sgbm_params* p, int batch, md_sgm_plan* plan, bool exportVolume, bool rowBands)
{
    plan->rowBands = false;
    plan->exportVolume = exportVolume;
    MD_REQUIRE(batch >= 1 && batch <= 4096, MDISP_E_BADARG, "batch must be in 1..4096");
    plan->batch = batch;
    MD_REQUIRE(p && plan, MDISP_E_BADARG, "null SGM parameters");
    MD_REQUIRE(H >= 3 && W >= 3 && W <= 65535, MDISP_E_BADARG, "image size %dx%d unsupported", W, H);
    MD_REQUIRE(p->numDisparities > 0 && p->numDisparities <= 512 && p->minDisparity >= 0, MDISP_E_BADARG,
               "numDisparities must be in 1..512 and minDisparity >= 0");
    MD_REQUIRE(p->paths == 8 || p->paths == 5 || p->paths == 4, MDISP_E_BADARG, "paths must be 4, 5 or 8");
    int cw = p->costWin > 0 ? p->costWin : 1;
    MD_REQUIRE(cw == 1, MDISP_E_UNSUPPORTED, "box-summed SGM cost (costWin > 1) is not built yet");
    MD_REQUIRE(p->censusMethod == MDISP_BASIC_CENSUS || p->censusMethod == MDISP_ROTATION_INVARIANT_CENSUS ||
               p->censusMethod == MDISP_UNIFORM_CENSUS, MDISP_E_UNSUPPORTED, "census method %d unsupported in SGM", p->censusMethod);
    plan->H = H; plan->W = W; plan->D = p->numDisparities; plan->minD = p->minDisparity;
    plan->maxD = p->minDisparity + p->numDisparities - 1;
    plan->Wp = W + plan->maxD;
    plan->P1 = p->P1 > 0 ? p->P1 : 2;
    plan->P2 = p->P2 > 0 ? p->P2 : 5;
    if (plan->P2 < plan->P1 + 1) plan->P2 = plan->P1 + 1;
    plan->uniq = p->uniquenessRatio >= 0 ? p->uniquenessRatio : 10;
    plan->maxDiff = p->disp12MaxDiff > 0 ? p->disp12MaxDiff : 1;
    plan->npaths = p->paths; plan->subpixel = p->subpixel; plan->costWin = cw; plan->census = p->censusMethod;
    MD_REQUIRE((long)plan->npaths * (8L * cw * cw + plan->P2) <= 65535, MDISP_E_RANGE,
               "npaths*(Cmax+P2) exceeds the 16-bit path-cost range");
    plan->pitchL = md_align((size_t)W, 256);
    plan->pitchR = md_align((size_t)plan->Wp, 256);
    // per-pair arrays are replicated `batch` times, back to back
    size_t off = 0;
    plan->off_cL = off; off += plan->pitchL * H * batch;
    plan->off_cR = off; off += plan->pitchR * H * batch;
    plan->off_pad = off; off += plan->pitchR * H;
    plan->off_scratch = off; off += plan->pitchR * H;
    off = md_align(off, 256);
    plan->use_v2 = md_sgm2_supported(*plan) && md_tune(MDISP_TUNE_SGM_PER_DIRECTION) <= 0;
    if (rowBands) {
        MD_REQUIRE(md_sgm2_supported(*plan) && plan->npaths == 8 && md_sgm2_u8(*plan) && batch == 1, MDISP_E_UNSUPPORTED,
                   "the row-band schedule needs eight paths, path states that fit a byte (8 + P2 <= 255) and one pair");
        plan->use_v2 = true; plan->rowBands = true;
    }
    plan->sstride = plan->use_v2 ? md_sgm2_slots(*plan) : plan->D;
    // the volume between the two sweeps: one byte per slot where the schedule's four-path sums fit a byte (and the
    // caller does not want the full 16-bit sums exported), else 16 bits
    const size_t slotBytes = plan->use_v2 && !exportVolume ? md_sgm2_partial_sum_bytes(*plan) : sizeof(u16);
    plan->strideS = md_align((size_t)H * W * plan->sstride * slotBytes, 256) / sizeof(u16);
    plan->off_S = off; off += plan->strideS * sizeof(u16) * batch;
    off = md_align(off, 256);
    plan->off_aux = off; off += 4096;
    plan->off_bnd = off;
    if (plan->use_v2) off += md_sgm2_extra_bytes(*plan, batch);
    plan->total = off;
    return MDISP_OK;
}

template <int ND>
static int launch_paths(hipStream_t s, const md_sgm_plan& pl, const SgmArgs& a)
{
    static const int L8[8] = {0, 1, 2, 3, 4, 5, 6, 7}, L5[5] = {0, 1, 2, 4, 5}, L4[4] = {0, 1, 2, 3};
    const int* list = pl.npaths == 8 ? L8 : (pl.npaths == 5 ? L5 : L4);
    for (int i = 0; i < pl.npaths; i++) {
        int dx = DIRS[list[i]][0], dy = DIRS[list[i]][1];
        int nlines = dy == 0 ? pl.H : (dx == 0 ? pl.W : pl.W + pl.H - 1);
        dim3 grid((nlines + 3) / 4);
        if (i == 0) sgm_path_kernel<ND, true><<<grid, 256, 0, s>>>(a, dx, dy, nlines);
        else sgm_path_kernel<ND, false><<<grid, 256, 0, s>>>(a, dx, dy, nlines);
        MD_HIP(hipGetLastError());
    }
    return MDISP_OK;
}

// census planes of every pair of the batch: the left view as it is, the right view padded by maxD reflected columns
static int sgm_census_planes(mdisp_ctx* ctx, const md_sgm_plan& pl, const u8* dL, const u8* dR, size_t pitch, size_t pairStride, void* ws)
{
    u8* base = (u8*)ws;
    // one launch for both views of every pair; the right view's reflected columns are read from the unpadded image
    return md_census_pairs(ctx->stream, dL, dR, pitch, pl.batch > 1 ? pairStride : 0, pl.batch, pl.H, pl.W, pl.maxD, pl.census,
                           base + pl.off_cL, pl.pitchL, pl.pitchL * pl.H, base + pl.off_cR, pl.pitchR, pl.pitchR * pl.H);
}

// dL/dR: first pair; pair i lives at +i*pairStride bytes
// dDisp == nullptr: leave S8 in the S volume (natural order); otherwise the strip-fused path also
// runs its fused winner-take-all and writes the final disparity maps.
static int sgm_aggregate(mdisp_ctx* ctx, const md_sgm_plan& pl, const u8* dL, const u8* dR, size_t pitch, size_t pairStride, void* ws,
                         int16_t* dDisp)
{
    hipStream_t s = ctx->stream;
    u8* base = (u8*)ws;
    u16* S = (u16*)(base + pl.off_S);
    const size_t strideL = pl.pitchL * pl.H, strideR = pl.pitchR * pl.H;
    int rc;
    if ((rc = sgm_census_planes(ctx, pl, dL, dR, pitch, pairStride, ws))) return rc;
    if (ctx->timing) MD_HIP(hipEventRecord(ctx->ev[2], s));
    if (pl.use_v2) {
        md_aux_streams aux;
        const bool haveAux = md_ctx_aux(ctx, &aux) == MDISP_OK;
        rc = md_sgm2_aggregate(s, pl, base + pl.off_cL, strideL, base + pl.off_cR, strideR, S, pl.strideS, base + pl.off_bnd, pl.batch,
                               dDisp ? 0 : 1, dDisp, ctx->sgm_err, &ctx->sgm_pending, haveAux ? &aux : nullptr);
        if (rc) return rc;
    } else {
        for (int i = 0; i < pl.batch; i++) {
            SgmArgs a;
            a.cL = base + pl.off_cL + (size_t)i * strideL; a.pL = pl.pitchL;
            a.cR = base + pl.off_cR + (size_t)i * strideR; a.pR = pl.pitchR; a.S = S + (size_t)i * pl.strideS;
            a.H = pl.H; a.W = pl.W; a.D = pl.D; a.minD = pl.minD; a.maxD = pl.maxD; a.P1 = pl.P1; a.P2 = pl.P2;
            if (pl.D <= 64) rc = launch_paths<1>(s, pl, a);
            else if (pl.D <= 128) rc = launch_paths<2>(s, pl, a);
            else if (pl.D <= 256) rc = launch_paths<4>(s, pl, a);
            else rc = launch_paths<8>(s, pl, a);
            if (rc) return rc;
        }
    }
    if (ctx->timing) { MD_HIP(hipEventRecord(ctx->ev[3], s)); ctx->ev_valid[1] = true; }
    return MDISP_OK;
}

int md_sgm_aggregate_only(mdisp_ctx* ctx, const md_sgm_plan& pl, const u8* dL, const u8* dR, size_t pitch, void* ws, const u16** Sout)
{
    *Sout = (const u16*)((u8*)ws + pl.off_S);
    return sgm_aggregate(ctx, pl, dL, dR, pitch, 0, ws, nullptr);
}

int md_sgm_run(mdisp_ctx* ctx, const md_sgm_plan& pl, const u8* dL, const u8* dR, size_t pitch, size_t pairStride,
               int16_t* dDisp, void* ws)
{
    hipStream_t s = ctx->stream;
    u16* S = (u16*)((u8*)ws + pl.off_S);
    int rc;
    if ((rc = sgm_aggregate(ctx, pl, dL, dR, pitch, pairStride, ws, pl.use_v2 ? dDisp : nullptr))) return rc;
    if (pl.use_v2) return MDISP_OK;   // winner-take-all, sub-pixel and LR check ran inside the fused path
    WtaArgs w;
    w.S = S; w.sstride = pl.sstride; w.H = pl.H; w.W = pl.W; w.D = pl.D; w.minD = pl.minD;
    w.bias = pl.npaths * pl.P2; w.uniq = pl.uniq; w.maxDiff = pl.maxDiff; w.subpixel = pl.subpixel;
    w.disp = dDisp; w.dpitch = pl.W;
    w.strideS = pl.strideS; w.strideDisp = (size_t)pl.H * pl.W;
    size_t lds = (size_t)pl.W * 4 + (size_t)pl.W * 2 + 16;
    if (ctx->timing) MD_HIP(hipEventRecord(ctx->ev[4], s));
    dim3 gw(pl.H, pl.batch);
    if (pl.D <= 64) sgm_wta_kernel<1><<<gw, 256, lds, s>>>(w);
    else if (pl.D <= 128) sgm_wta_kernel<2><<<gw, 256, lds, s>>>(w);
    else if (pl.D <= 256) sgm_wta_kernel<4><<<gw, 256, lds, s>>>(w);
    else sgm_wta_kernel<8><<<gw, 256, lds, s>>>(w);
    MD_HIP(hipGetLastError());
    if (ctx->timing) { MD_HIP(hipEventRecord(ctx->ev[5], s)); ctx->ev_valid[2] = true; }
    return MDISP_OK;
}

// One sweep over the rows [r0, r1) of one pair (see md_sgm2_rows_sweep); the first sweep also prepares the census planes
// (of the whole views: a few microseconds, and every row range needs its neighbours' rows anyway).
int md_sgm_rows_sweep(mdisp_ctx* ctx, const md_sgm_plan& pl, const u8* dL, const u8* dR, size_t pitch, void* ws, int r0, int r1, bool up, bool second,
                      const u8* bndIn, u8* bndOut, int16_t* dDisp)
{
    u8* base = (u8*)ws;
    int rc;
    if (!second && (rc = sgm_census_planes(ctx, pl, dL, dR, pitch, 0, ws))) return rc;
    return md_sgm2_rows_sweep(ctx->stream, pl, base + pl.off_cL, base + pl.off_cR, (u16*)(base + pl.off_S), base + pl.off_bnd, r0, r1, up, second,
                              bndIn, bndOut, dDisp, ctx->sgm_err, &ctx->sgm_pending);
}
