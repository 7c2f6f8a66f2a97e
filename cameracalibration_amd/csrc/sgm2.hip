// Strip-fused semi-global matching for gfx950 (same arithmetic as sgm.hip, far less HBM traffic).
//
// The eight path directions are folded into two sweeps over the image:
//   down sweep:  ->  (1,0),  v (0,1),  \ (1,1),  / (-1,1)     writes the 4-path partial sum S4 (u16)
//   up sweep:    <- (-1,0),  ^ (0,-1), (1,-1), (-1,-1)        reads S4, finishes S8 in registers
// so the aggregated volume crosses HBM once in each direction (2 B written + 2 B read per
// (x,y,d)) instead of being read-modify-written by every direction.
//
// A workgroup owns a strip of 16 image columns for the whole image height.  Vertical and diagonal
// path states live in registers / LDS and advance row by row; the horizontal path of a block of 16
// rows is scanned across the strip first.  What a strip cannot know locally -- the state of the
// horizontal and diagonal paths where they ENTER the strip -- is produced by a pre-pass that runs
// those six directions along their whole lines and keeps only the states at strip borders.
//
// Lane mapping: a "lane-row" of 16 lanes (one DPP row) owns one pixel at a time; each lane holds 16
// consecutive disparities as 8 registers of packed u16 pairs, interleaved as
//   A_g = [d(4g) | d(4g+2)],  B_g = [d(4g+1) | d(4g+3)]          (g = 0..3)
// so the d-1 / d+1 neighbours of a register are the sibling register or one v_alignbit away.  Path
// states are kept normalised (N = L' - min L'), which turns the recursion into
//   L'(d) = C(d) + min(N(d), P2, P1 + min(N(d-1), N(d+1)))
// in packed 16-bit arithmetic; the minimum over d is a register tree plus four DPP steps.
#include "common.h"

namespace {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 asv(u32 v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ u32 asu(us2 v) { return __builtin_bit_cast(u32, v); }
__device__ __forceinline__ u32 pkadd(u32 a, u32 b) { return asu(asv(a) + asv(b)); }
__device__ __forceinline__ u32 pksub(u32 a, u32 b) { return asu(asv(a) - asv(b)); }
__device__ __forceinline__ u32 pkmin(u32 a, u32 b) { return asu(__builtin_elementwise_min(asv(a), asv(b))); }

constexpr u32 INF2 = 0x7F007F00u;   // "infinite" packed pair; INF + P1 cannot wrap (P1 < 8192)

#define DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
#define DPP_ROW_SHL1 0x101     // lane i <- lane i+1 (within a row of 16)
#define DPP_ROW_SHR1 0x111     // lane i <- lane i-1

template <int CTRL> __device__ __forceinline__ u32 dpp_all(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ u32 dpp_keep(u32 old, u32 v)
{ return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, 0xF, false); }

// minimum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ u32 row_min16(u32 v)
{
    v = min(v, dpp_all<DPP_XOR1>(v));
    v = min(v, dpp_all<DPP_XOR2>(v));
    v = min(v, dpp_all<DPP_HALF_MIRROR>(v));
    v = min(v, dpp_all<DPP_MIRROR>(v));
    return v;
}

constexpr int NR = 8;          // packed registers per lane (16 disparities)
constexpr int DPL = 16;        // disparities per lane
constexpr int LPC = 16;        // lanes per pixel  -> up to 256 disparities
constexpr int CW = 16;         // strip width in columns
constexpr int RB = 16;         // rows per block (horizontal scan batch)

struct Sgm2Args {
    const u8* cL; size_t pL, strideL;     // census of the left image      [B][H][pL]
    const u8* cR; size_t pR, strideR;     // census of the padded right    [B][H][pR]
    u16* S; size_t strideS;               // partial sums                  [B][H][W][256] (register order)
    u16* bnd[6]; size_t strideB;          // strip-border path states      [B][H][nstrips][256]
    int H, W, D, minD, maxD, nstrips;
    u32 P1P1, P2P2;
    int tail;                             // 1 when D < 256: disparities >= D are masked off
    int naturalOut;                       // up sweep: store S8 in natural d order (for the separate WTA kernel)
};
// order of bnd[]: 0 (1,0)  1 (-1,0)  2 (1,1)  3 (-1,1)  4 (1,-1)  5 (-1,-1)

// masks that pin disparities >= D of this lane to "infinite"
__device__ __forceinline__ void make_tail(int lic, int D, u32 (&tail)[NR])
{
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int g = r >> 1, odd = r & 1;
        const int dlo = lic * DPL + 4 * g + odd, dhi = dlo + 2;
        tail[r] = (dlo >= D ? 0x00007F00u : 0u) | (dhi >= D ? 0x7F000000u : 0u);
    }
}

// matching cost of one pixel for this lane's 16 disparities:
// popcount(cL(x,y) ^ cR(x + maxD - d, y)), d = minD + lic*16 + e, in register order
__device__ __forceinline__ void load_cost_operands(const Sgm2Args& a, const u8* cLrow, const u8* cRrow, int x, int lic,
                                                   u32& cl, uint4& rb)
{
    cl = cLrow[x];
    // bytes at ascending addresses are descending disparities: byte j <-> e = 15 - j
    __builtin_memcpy(&rb, cRrow + x + a.maxD - a.minD - lic * DPL - 15, 16);
}
__device__ __forceinline__ void cost_from_operands(u32 cl, const uint4& rb, u32 (&C)[NR])
{
    const u32 m = cl * 0x01010101u;
    const u32 w[4] = {rb.x ^ m, rb.y ^ m, rb.z ^ m, rb.w ^ m};
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const u32 x = w[3 - g];   // dword holding e = 4g .. 4g+3 as bytes 3,2,1,0
        C[2 * g] = (u32)__popc(x & 0xFF000000u) | ((u32)__popc(x & 0x0000FF00u) << 16);       // [e=4g   | e=4g+2]
        C[2 * g + 1] = (u32)__popc(x & 0x00FF0000u) | ((u32)__popc(x & 0x000000FFu) << 16);   // [e=4g+1 | e=4g+3]
    }
}

// One step of one path for one pixel: consumes the predecessor's normalised state N, produces
// L' (returned in L) and leaves the new normalised state in N.
template <bool TAIL>
__device__ __forceinline__ void sgm_step(u32 (&N)[NR], const u32 (&C)[NR], u32 P1P1, u32 P2P2, const u32 (&tail)[NR], u32 (&L)[NR])
{
    // d-1 of this lane's first disparity lives in the previous lane's last B register (high half),
    // d+1 of its last disparity in the next lane's first A register (low half)
    const u32 prevB = dpp_keep<DPP_ROW_SHR1>(INF2, N[NR - 1]);
    const u32 nextA = dpp_keep<DPP_ROW_SHL1>(INF2, N[0]);
    u32 mn = 0xFFFFFFFFu;
    u32 O[NR];   // predecessor state (N is overwritten group by group below)
#pragma unroll
    for (int r = 0; r < NR; r++) O[r] = N[r];
#pragma unroll
    for (int g = 0; g < NR / 2; g++) {
        const u32 A = O[2 * g], B = O[2 * g + 1];
        const u32 Bm = g == 0 ? prevB : O[2 * g - 1];
        const u32 An = g == NR / 2 - 1 ? nextA : O[2 * g + 2];
        const u32 tA = pkmin(__builtin_amdgcn_alignbit(B, Bm, 16), B);       // min(N(d-1), N(d+1)) for A's pair
        const u32 tB = pkmin(A, __builtin_amdgcn_alignbit(An, A, 16));       // ... for B's pair
        u32 LA = pkadd(C[2 * g], pkmin(pkmin(A, P2P2), pkadd(tA, P1P1)));
        u32 LB = pkadd(C[2 * g + 1], pkmin(pkmin(B, P2P2), pkadd(tB, P1P1)));
        L[2 * g] = LA; L[2 * g + 1] = LB;
        if (TAIL) { LA |= tail[2 * g]; LB |= tail[2 * g + 1]; }
        N[2 * g] = LA; N[2 * g + 1] = LB;
        mn = g == 0 ? pkmin(LA, LB) : pkmin(mn, pkmin(LA, LB));
    }
    u32 m1 = min(mn & 0xFFFFu, mn >> 16);
    m1 = row_min16(m1);
    const u32 mm = m1 | (m1 << 16);
#pragma unroll
    for (int r = 0; r < NR; r++) N[r] = pksub(N[r], mm);
}

__device__ __forceinline__ void load8(const u16* p, u32 (&v)[NR])
{
    const uint4 a = reinterpret_cast<const uint4*>(p)[0], b = reinterpret_cast<const uint4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(u16* p, const u32 (&v)[NR])
{
    reinterpret_cast<uint4*>(p)[0] = make_uint4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<uint4*>(p)[1] = make_uint4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void zero8(u32 (&v)[NR])
{
#pragma unroll
    for (int r = 0; r < NR; r++) v[r] = 0;
}

// ---------------------------------------------------------------------------------------------
// Pre-pass: run a path along its whole line and keep the state wherever it leaves a strip.
// One lane-row per line, 16 lines per workgroup.  Line ids: [0,H) (1,0); [H,2H) (-1,0); then the four
// diagonal directions with W+H-1 lines each in the order (1,1) (-1,1) (1,-1) (-1,-1).
template <bool TAIL>
__global__ __launch_bounds__(256) void sgm2_prepass_kernel(Sgm2Args a)
{
    const int lic = threadIdx.x & 15, lr = threadIdx.x >> 4;
    const int pair = blockIdx.y;
    const int H = a.H, W = a.W;
    const int nd = W + H - 1;
    int line = blockIdx.x * 16 + lr;
    const int total = 2 * H + 4 * nd;
    if (line >= total) return;
    int dir, dx, dy, x, y;
    if (line < 2 * H) {
        dir = line / H; y = line - dir * H; dy = 0; dx = dir == 0 ? 1 : -1; x = dx > 0 ? 0 : W - 1;
    } else {
        line -= 2 * H;
        const int q = line / nd, i = line - q * nd;
        dir = 2 + q;
        dx = (q & 1) ? -1 : 1; dy = (q & 2) ? -1 : 1;
        const int ys = dy > 0 ? 0 : H - 1;
        if (i < W) { x = i; y = ys; }
        else { const int j = i - W; y = dy > 0 ? 1 + j : H - 2 - j; x = dx > 0 ? 0 : W - 1; }
    }
    int n;
    {
        const int nx = dx > 0 ? W - x : x + 1;
        const int ny = dy > 0 ? H - y : (dy < 0 ? y + 1 : 0x7FFFFFFF);
        n = min(nx, ny);
    }
    const u8* cLp = a.cL + (size_t)pair * a.strideL;
    const u8* cRp = a.cR + (size_t)pair * a.strideR;
    u16* bnd = a.bnd[dir] + (size_t)pair * a.strideB;
    u32 tail[NR];
    make_tail(lic, a.D, tail);
    u32 N[NR];
    zero8(N);                        // predecessor outside the image: all path costs equal
    if (TAIL) {
#pragma unroll
        for (int r = 0; r < NR; r++) N[r] |= tail[r];
    }
    u32 cl; uint4 rb;
    load_cost_operands(a, cLp + (size_t)y * a.pL, cRp + (size_t)y * a.pR, x, lic, cl, rb);
    for (int t = 0; t < n; t++) {
        u32 C[NR], L[NR];
        cost_from_operands(cl, rb, C);
        const int xn = x + dx, yn = y + dy;
        if (t + 1 < n) load_cost_operands(a, cLp + (size_t)yn * a.pL, cRp + (size_t)yn * a.pR, xn, lic, cl, rb);
        sgm_step<TAIL>(N, C, a.P1P1, a.P2P2, tail, L);
        // does the successor pixel open a new strip?  Then its workgroup will ask for this state,
        // indexed by the successor's row and strip.
        if (t + 1 < n) {
            bool edge; int strip;
            if (dx > 0) { edge = (xn % CW) == 0; strip = xn / CW; }
            else { edge = (x % CW) == 0; strip = x / CW - 1; }
            if (edge) store8(bnd + (((size_t)yn * a.nstrips + strip) * 256 + lic * DPL), N);
        }
        x = xn; y = yn;
    }
}

// ---------------------------------------------------------------------------------------------
// Main sweep.  UP = false: paths (1,0) (0,1) (1,1) (-1,1), top to bottom, S <- sum of the four.
//              UP = true : paths (-1,0) (0,-1) (1,-1) (-1,-1), bottom to top, S <- S + sum.
template <bool UP, bool TAIL>
__global__ __launch_bounds__(256) void sgm2_sweep_kernel(Sgm2Args a)
{
    __shared__ u32 diag[2][2][CW][128];   // [row parity][x-1 / x+1 family][column][packed pairs]
    const int lic = threadIdx.x & 15, lr = threadIdx.x >> 4;
    const int strip = blockIdx.x, pair = blockIdx.y;
    const int H = a.H, W = a.W;
    const int x0 = strip * CW;
    const int cw = min(CW, W - x0);
    const u8* cLp = a.cL + (size_t)pair * a.strideL;
    const u8* cRp = a.cR + (size_t)pair * a.strideR;
    u16* Sp = a.S + (size_t)pair * a.strideS;
    const u16* bH = a.bnd[UP ? 1 : 0] + (size_t)pair * a.strideB;   // horizontal path entering the strip
    const u16* bA = a.bnd[UP ? 4 : 2] + (size_t)pair * a.strideB;   // diagonal with dx = +1 (predecessor at x-1)
    const u16* bB = a.bnd[UP ? 5 : 3] + (size_t)pair * a.strideB;   // diagonal with dx = -1 (predecessor at x+1)
    u32 tail[NR];
    make_tail(lic, a.D, tail);
    const u32 P1P1 = a.P1P1, P2P2 = a.P2P2;

    u32 Nv[NR];   // vertical path state of column x0 + lr
    zero8(Nv);
    if (TAIL) {
#pragma unroll
        for (int r = 0; r < NR; r++) Nv[r] |= tail[r];
    }
    const int nblocks = (H + RB - 1) / RB;
    for (int jb = 0; jb < nblocks; jb++) {
        // rows of this block in processing order: ys, ys + sy, ...
        const int ys = UP ? H - 1 - jb * RB : jb * RB;
        const int sy = UP ? -1 : 1;
        const int nrows = min(RB, H - jb * RB);

        // ---- phase A: horizontal path, lane-row <-> image row, scanned across the strip ----
        if (lr < nrows) {
            const int y = ys + sy * lr;
            const u8* cLrow = cLp + (size_t)y * a.pL;
            const u8* cRrow = cRp + (size_t)y * a.pR;
            u32 N[NR];
            const bool fromImageBorder = UP ? (x0 + cw >= W) : (x0 == 0);
            if (fromImageBorder) {
                zero8(N);
                if (TAIL) {
#pragma unroll
                    for (int r = 0; r < NR; r++) N[r] |= tail[r];
                }
            } else load8(bH + (((size_t)y * a.nstrips + strip) * 256 + lic * DPL), N);
            int x = UP ? x0 + cw - 1 : x0;
            const int dx = UP ? -1 : 1;
            u32 cl; uint4 rb;
            load_cost_operands(a, cLrow, cRrow, x, lic, cl, rb);
            for (int t = 0; t < cw; t++) {
                u32 C[NR], L[NR];
                cost_from_operands(cl, rb, C);
                u16* sp = Sp + (((size_t)y * W + x) * 256 + lic * DPL);
                u32 Sold[NR];
                if (UP) load8(sp, Sold);
                if (t + 1 < cw) load_cost_operands(a, cLrow, cRrow, x + dx, lic, cl, rb);
                sgm_step<TAIL>(N, C, P1P1, P2P2, tail, L);
                if (UP) {
#pragma unroll
                    for (int r = 0; r < NR; r++) L[r] = pkadd(L[r], Sold[r]);
                }
                store8(sp, L);
                x += dx;
            }
        }
        __syncthreads();   // the horizontal sums of the block are read back by other waves below

        // ---- phase B: vertical and diagonal paths, lane-row <-> column ----
        const int x = x0 + lr;
        const bool colActive = lr < cw;
        for (int r = 0; r < nrows; r++) {
            const int y = ys + sy * r;
            const int yp = y - sy;                           // predecessor row
            const bool hasPrevRow = yp >= 0 && yp < H;
            const int par = r & 1;
            if (colActive) {
                u32 cl; uint4 rb;
                load_cost_operands(a, cLp + (size_t)y * a.pL, cRp + (size_t)y * a.pR, x, lic, cl, rb);
                u16* sp = Sp + (((size_t)y * W + x) * 256 + lic * DPL);
                u32 Sacc[NR];
                load8(sp, Sacc);
                u32 C[NR], L[NR];
                cost_from_operands(cl, rb, C);
                // vertical
                sgm_step<TAIL>(Nv, C, P1P1, P2P2, tail, L);
#pragma unroll
                for (int q = 0; q < NR; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
                // diagonal arriving from (x-1, yp)
                u32 Nd[NR];
                if (!hasPrevRow || x == 0) {
                    zero8(Nd);
                    if (TAIL) {
#pragma unroll
                        for (int q = 0; q < NR; q++) Nd[q] |= tail[q];
                    }
                } else if (lr == 0) load8(bA + (((size_t)y * a.nstrips + strip) * 256 + lic * DPL), Nd);
                else {
                    const uint4* src = reinterpret_cast<const uint4*>(&diag[par ^ 1][0][lr - 1][lic * 8]);
                    const uint4 v0 = src[0], v1 = src[1];
                    Nd[0] = v0.x; Nd[1] = v0.y; Nd[2] = v0.z; Nd[3] = v0.w; Nd[4] = v1.x; Nd[5] = v1.y; Nd[6] = v1.z; Nd[7] = v1.w;
                }
                sgm_step<TAIL>(Nd, C, P1P1, P2P2, tail, L);
#pragma unroll
                for (int q = 0; q < NR; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
                {
                    uint4* dst = reinterpret_cast<uint4*>(&diag[par][0][lr][lic * 8]);
                    dst[0] = make_uint4(Nd[0], Nd[1], Nd[2], Nd[3]);
                    dst[1] = make_uint4(Nd[4], Nd[5], Nd[6], Nd[7]);
                }
                // diagonal arriving from (x+1, yp)
                if (!hasPrevRow || x == W - 1) {
                    zero8(Nd);
                    if (TAIL) {
#pragma unroll
                        for (int q = 0; q < NR; q++) Nd[q] |= tail[q];
                    }
                } else if (lr == cw - 1) load8(bB + (((size_t)y * a.nstrips + strip) * 256 + lic * DPL), Nd);
                else {
                    const uint4* src = reinterpret_cast<const uint4*>(&diag[par ^ 1][1][lr + 1][lic * 8]);
                    const uint4 v0 = src[0], v1 = src[1];
                    Nd[0] = v0.x; Nd[1] = v0.y; Nd[2] = v0.z; Nd[3] = v0.w; Nd[4] = v1.x; Nd[5] = v1.y; Nd[6] = v1.z; Nd[7] = v1.w;
                }
                sgm_step<TAIL>(Nd, C, P1P1, P2P2, tail, L);
#pragma unroll
                for (int q = 0; q < NR; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
                {
                    uint4* dst = reinterpret_cast<uint4*>(&diag[par][1][lr][lic * 8]);
                    dst[0] = make_uint4(Nd[0], Nd[1], Nd[2], Nd[3]);
                    dst[1] = make_uint4(Nd[4], Nd[5], Nd[6], Nd[7]);
                }
                if (UP && a.naturalOut) {
                    // [A_g.lo|B_g.lo], [A_g.hi|B_g.hi]: pairs of consecutive disparities
                    u32 nat[NR];
#pragma unroll
                    for (int g = 0; g < NR / 2; g++) {
                        nat[2 * g] = __builtin_amdgcn_perm(Sacc[2 * g + 1], Sacc[2 * g], 0x05040100u);
                        nat[2 * g + 1] = __builtin_amdgcn_perm(Sacc[2 * g + 1], Sacc[2 * g], 0x07060302u);
                    }
                    store8(sp, nat);
                } else store8(sp, Sacc);
            }
            __syncthreads();   // diag[par] complete before the next row reads it
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------

size_t md_sgm2_extra_bytes(int H, int W, int batch)
{
    const int nstrips = (W + CW - 1) / CW;
    return md_align((size_t)H * nstrips * 256 * sizeof(u16), 256) * 6 * batch;
}

bool md_sgm2_supported(const md_sgm_plan& pl)
{
    return pl.npaths == 8 && pl.D <= 256 && pl.costWin == 1 && (long)pl.npaths * (8 + pl.P2) < 4096 && pl.P1 < 8192 &&
           pl.maxD - pl.minD + 1 == pl.D && pl.uniq <= 100;
}

// S must hold H*W*256 u16 per pair; bndBase md_sgm2_extra_bytes() bytes
int md_sgm2_aggregate(hipStream_t s, const md_sgm_plan& pl, const u8* cL, size_t strideL, const u8* cR, size_t strideR,
                      u16* S, size_t strideS, u8* bndBase, int batch, int naturalOut)
{
    Sgm2Args a;
    a.cL = cL; a.pL = pl.pitchL; a.strideL = strideL;
    a.cR = cR; a.pR = pl.pitchR; a.strideR = strideR;
    a.S = S; a.strideS = strideS;
    a.H = pl.H; a.W = pl.W; a.D = pl.D; a.minD = pl.minD; a.maxD = pl.maxD;
    a.nstrips = (pl.W + CW - 1) / CW;
    const size_t oneB = md_align((size_t)pl.H * a.nstrips * 256 * sizeof(u16), 256);
    for (int i = 0; i < 6; i++) a.bnd[i] = (u16*)(bndBase + (size_t)i * oneB * batch);
    a.strideB = oneB / sizeof(u16);
    a.P1P1 = (u32)pl.P1 | ((u32)pl.P1 << 16);
    a.P2P2 = (u32)pl.P2 | ((u32)pl.P2 << 16);
    a.tail = pl.D < 256;
    a.naturalOut = naturalOut;
    const int totalLines = 2 * pl.H + 4 * (pl.W + pl.H - 1);
    dim3 gp((totalLines + 15) / 16, batch), gs(a.nstrips, batch);
    if (a.tail) {
        sgm2_prepass_kernel<true><<<gp, 256, 0, s>>>(a);
        sgm2_sweep_kernel<false, true><<<gs, 256, 0, s>>>(a);
        sgm2_sweep_kernel<true, true><<<gs, 256, 0, s>>>(a);
    } else {
        sgm2_prepass_kernel<false><<<gp, 256, 0, s>>>(a);
        sgm2_sweep_kernel<false, false><<<gs, 256, 0, s>>>(a);
        sgm2_sweep_kernel<true, false><<<gs, 256, 0, s>>>(a);
    }
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}
