// Strip-fused semi-global matching for gfx950 (same arithmetic as sgm.hip, far less HBM traffic).
//
// The eight path directions are folded into two sweeps over the image:
//   down sweep:  v (0,1),  \ (1,1),  / (-1,1)   + the stored (1,0) path   -> 4-path partial sum S4 (u16)
//   up sweep:    ^ (0,-1), (1,-1), (-1,-1)       + the stored (-1,0) path  -> S4 + 4 = S8, in registers
// so the aggregated volume crosses HBM once in each direction (2 B written + 2 B read per
// (x,y,d)) instead of being read-modify-written by every direction.
//
// Kernels:
//   hscan    one lane-row per (image row, direction): the two horizontal paths over whole rows,
//            stored as a byte volume (their values never exceed Cmax + P2) or, for large P2, u16.
//   prepass  the four diagonal directions along their whole lines, keeping only the normalised
//            path state where a line crosses from one 16-column strip into the next.
//   sweep    a workgroup owns a strip of 16 columns for the whole image height; lane-row <-> column.
//            The vertical path state lives in registers, the two diagonal states travel through
//            LDS from the neighbouring column of the previous row (strip edges read the prepass
//            states), the horizontal path is read back from hscan's volume.
//
// Lane mapping: a "lane-row" of 16 lanes (one DPP row) owns one pixel at a time; each lane holds 16
// consecutive disparities as 8 registers of packed u16 pairs,
//   R_r = [e = r | e = r + 8]          (r = 0..7, e = disparity index inside the lane)
// so the d-1 / d+1 neighbours of a whole register are the registers R_(r-1) and R_(r+1); only R_0 and R_7
// need one v_alignbit each to splice in the neighbouring lane.  Path states are kept normalised
// (N = L' - min L'), which turns the recursion into
//   L'(d) = C(d) + min(N(d), P2, P1 + min(N(d-1), N(d+1)))
// in packed 16-bit arithmetic; the minimum over d is a register tree plus four DPP steps.
// Three-input minima use v_pk_minimum3_f16: bit patterns below 0x7C00 are non-negative finite halfs (subnormals
// included, which the default mode keeps) and order exactly like the integers they are, so every packed
// value here stays below 0x7C00 ("infinite" is 0x6000, P1 < 2048, sums < 4096).
#include "sgm_lanes.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Horizontal paths: LPC lanes per (row, direction), 256/LPC lines per workgroup; line id < H: (1,0), else (-1,0).
// A pixel occupies LPC*16 disparity slots in every volume (64, 128 or 256).
template <bool TAIL, bool U8, int LPC>
__global__ __launch_bounds__(256) void sgm2_hscan_kernel(Sgm2Args a)
{
    constexpr int DP = LPC * DPL, LINES = 256 / LPC;
    const int lic = threadIdx.x % LPC, lr = threadIdx.x / LPC;
    const int pair = blockIdx.y;
    const int H = a.H, W = a.W;
    const int line = blockIdx.x * LINES + lr;
    if (line >= 2 * H) return;
    const int dir = line >= H, y = line - dir * H;
    const int dx = dir ? -1 : 1;
    int x = dir ? W - 1 : 0;
    const u8* cLrow = a.cL + (size_t)pair * a.strideL + (size_t)y * a.pL;
    const u8* cRrow = a.cR + (size_t)pair * a.strideR + (size_t)y * a.pR;
    u8* hrow = a.Hvol[dir] + (size_t)pair * a.strideH + (size_t)y * W * (U8 ? DP : 2 * DP) + lic * (U8 ? 16 : 32);
    u32 tail[NR];
    make_tail(lic, a.D, tail);
    u32 N[NR];
    zero8(N);                        // predecessor outside the image: all path costs equal
    if (TAIL) {
#pragma unroll
        for (int r = 0; r < NR; r++) N[r] |= tail[r];
    }
    // One step = wait for this step's operands, store the previous step's values, request the next step's operands
    // (unconditionally: past the end the last pixel is asked for again), compute.  The compiler drains all outstanding
    // memory operations where the operands are first used; in this order the store it then waits for was issued a
    // whole step earlier instead of a few instructions earlier.
    u32 clw; uint4 rb;
    load_cost_operands_w(a, cLrow, cRrow, x, lic, clw, rb);
    uint4 pend = make_uint4(0, 0, 0, 0);
    u32 C[NR], L[NR];
    auto compute = [&](int t) {
        const int xn = t + 1 < W ? x + dx : x;
        load_cost_operands_w(a, cLrow, cRrow, xn, lic, clw, rb);
        sgm_step<TAIL, LPC>(N, C, a.P1P1, a.P2P2, tail, L, lic);
        if (U8) pend = pack_bytes(L);
        else store8(reinterpret_cast<u16*>(hrow + (size_t)x * 2 * DP), L);
        x += dx;
    };
    cost_from_operands(census_byte(clw, x), rb, C);
    compute(0);
    for (int t = 1; t < W; t++) {
        cost_from_operands(census_byte(clw, x), rb, C);
        if (U8) store16_stream(hrow + (size_t)(x - dx) * DP, pend);
        compute(t);
    }
    if (U8) store16_stream(hrow + (size_t)(x - dx) * DP, pend);
}

// ---------------------------------------------------------------------------------------------
// Diagonal pre-pass: run a path along its whole line and keep the state wherever it leaves a strip.
// One lane-row per slot, 16 slots per workgroup, directions in the order (1,1) (-1,1) (1,-1) (-1,-1).
// The W+H-1 lines of a direction have lengths 1, 2, .., m, m+1 (nd - 2m times), m, .., 1 with m = min(W,H) - 1;
// a slot runs either one full-length line or a line of the rising ramp followed by the line of the falling ramp
// that completes it to m + 1 steps, so every slot (and every workgroup) carries the same work.
__device__ __forceinline__ int wave_min_i32(int v)
{
    for (int o = 32; o; o >>= 1) v = min(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max_i32(int v)
{
    for (int o = 32; o; o >>= 1) v = max(v, __shfl_xor(v, o));
    return __builtin_amdgcn_readfirstlane(v);
}
__host__ __device__ inline int prepass_slots(int H, int W) { return W + H - 1 - (min(W, H) - 1); }

constexpr int PREPASS_PF = 4;
template <bool TAIL, int CW = 16>
__global__ __launch_bounds__(256) void sgm2_prepass_kernel(Sgm2Args a)
{
    const int lic = threadIdx.x & 15, lr = threadIdx.x >> 4;
    const int pair = blockIdx.y;
    const int H = a.H, W = a.W;
    const int nd = W + H - 1, m = min(W, H) - 1, ns = nd - m;
    const int slot = blockIdx.x * 16 + lr;
    if (slot >= 4 * ns) return;
    const int q = slot / ns, sl = slot - q * ns;
    const int dx = (q & 1) ? -1 : 1, dy = (q & 2) ? -1 : 1;
    const u8* cLp = a.cL + (size_t)pair * a.strideL;
    const u8* cRp = a.cR + (size_t)pair * a.strideR;
    u16* bnd = a.bnd[q] + (size_t)pair * a.strideB;
    u32 tail[NR];
    make_tail(lic, a.D, tail);
    const int full = nd - 2 * m;                          // lines of full length
    // line c of the direction (1,1) is x - y = c - (H - 1); the other directions mirror x and / or y
    auto line_start = [&](int c, int& x, int& y) {
        const int e = c - (H - 1);
        x = e >= 0 ? e : 0; y = e >= 0 ? 0 : -e;
        if (dx < 0) x = W - 1 - x;
        if (dy < 0) y = H - 1 - y;
    };
    const int T = m + 1;                                  // steps of every slot
    const bool paired = sl >= full;
    const int c1 = paired ? sl - full : m + sl;
    const int n1 = paired ? c1 + 1 : T;                   // steps of the first line; the second one takes the rest
    int x1, y1, x2, y2;
    line_start(c1, x1, y1);
    line_start(paired ? nd - m + c1 : c1, x2, y2);
    // The loop is uniform (every lane runs T steps) and its loads are unconditional: a load or a loop exit under a
    // per-lane condition makes the compiler wait for outstanding requests on the spot.  Positions and request
    // addresses advance incrementally; the jump to the second line is a per-lane select that only the steps inside the
    // wave's window of line ends [nlo, nhi] execute.
    const int nlo = wave_min_i32(n1), nhi = wave_max_i32(n1);
    const ptrdiff_t dL = (ptrdiff_t)dy * (ptrdiff_t)a.pL + dx, dR = (ptrdiff_t)dy * (ptrdiff_t)a.pR + dx;
    const int rofs = a.maxD - a.minD - lic * DPL - 15;
    const u8* fL = cLp + (size_t)y1 * a.pL + x1;
    const u8* fR = cRp + (size_t)y1 * a.pR + x1 + rofs;
    const u8* const fL2 = cLp + (size_t)y2 * a.pL + x2;
    const u8* const fR2 = cRp + (size_t)y2 * a.pR + x2 + rofs;
    auto issue = [&](u32& clw, uint4& rbv) {
        clw = *reinterpret_cast<const u32*>(fL - ((u32)reinterpret_cast<uintptr_t>(fL) & 3u));   // rows start 256-byte aligned
        __builtin_memcpy(&rbv, fR, 16);
    };
    // move the request position to step tf (uniform); past the end it stays on the last pixel
    auto advance_fetch = [&](int tf) {
        if (tf < T) {
            fL += dL; fR += dR;
            if (tf >= nlo && tf <= nhi) {
                const bool second = tf == n1;
                fL = second ? fL2 : fL; fR = second ? fR2 : fR;
            }
        }
    };
    // every step of a diagonal touches another image row: operands are requested PREPASS_PF steps ahead
    u32 cl[PREPASS_PF]; uint4 rb[PREPASS_PF];
#pragma unroll
    for (int u = 0; u < PREPASS_PF; u++) {
        if (u > 0) advance_fetch(u);
        issue(cl[u], rb[u]);
    }
    u32 N[NR];
    zero8(N);
    if (TAIL) {
#pragma unroll
        for (int r = 0; r < NR; r++) N[r] |= tail[r];
    }
    int x = x1, y = y1;
    // whole groups of PREPASS_PF steps run without an exit in the middle; the last steps use what the ring holds
    auto step = [&](int t, u32& clw, uint4& rbv, bool more) {
            if (t > 0) {
                x += dx; y += dy;
                if (t >= nlo && t <= nhi) {
                    const bool second = t == n1;          // the second line starts from the image border
                    x = second ? x2 : x; y = second ? y2 : y;
#pragma unroll
                    for (int r = 0; r < NR; r++) N[r] = second ? (TAIL ? tail[r] : 0u) : N[r];
                }
            }
            u32 C[NR], L[NR];
            cost_from_operands(census_byte(clw, x), rbv, C);
            if (more) {
                advance_fetch(t + PREPASS_PF);
                issue(clw, rbv);
            }
            sgm_step<TAIL>(N, C, a.P1P1, a.P2P2, tail, L);
            // does the successor pixel open a new strip?  Then its workgroup will ask for this state,
            // indexed by the successor's row and strip.
            const int xn = x + dx, yn = y + dy;
            bool edge; int strip;
            if (dx > 0) { edge = (xn % CW) == 0; strip = xn / CW; }
            else { edge = (x % CW) == 0; strip = x / CW - 1; }
            if (edge && t + 1 != n1 && t + 1 < T) store8(bnd + (((size_t)yn * a.nstrips + strip) * 256 + lic * DPL), N);
    };
    int t = 0;
    for (; t + PREPASS_PF <= T; t += PREPASS_PF) {
#pragma unroll
        for (int u = 0; u < PREPASS_PF; u++) step(t + u, cl[u], rb[u], true);
    }
#pragma unroll
    for (int u = 0; u < PREPASS_PF; u++)
        if (t + u < T) step(t + u, cl[u], rb[u], false);
}

// ---------------------------------------------------------------------------------------------
// Main sweep.  UP = false: (0,1) (1,1) (-1,1) top to bottom, S <- H(1,0) + the three.
//              UP = true : (0,-1) (1,-1) (-1,-1) bottom to top, S8 = S + H(-1,0) + the three.
// PFD rows of operands are in flight per lane-row: 1 when many waves share a SIMD, more when a small batch
// leaves one wave per SIMD and a row is shorter than a memory round trip.
template <bool UP, bool TAIL, bool U8, bool FUSE, int CW = 16, int PFD = 1>
__global__ __launch_bounds__(256) void sgm2_sweep_kernel(Sgm2Args a)
{
    // diagonal path states of the previous / current row: [row parity][x-1 / x+1 family][column][...],
    // as packed u16 pairs (128 dwords per pixel) or, when every value fits a byte, as 64 dwords
    constexpr int DW = U8 ? 64 : 128;
    __shared__ u32 diag[2][2][CW][DW];
    __shared__ u32 srow[FUSE ? CW : 1][FUSE ? 128 : 1];      // S8 of the current row, register order
    __shared__ u32 exclMask[FUSE ? 19 : 1][NR];              // 0xFFFF over {best-1, best, best+1} by (best - lane base + 1)
    if (FUSE) {
        init_excl_mask(exclMask);
        __syncthreads();
    }
    const int lic = threadIdx.x & 15, lr = threadIdx.x >> 4;
    const int strip = blockIdx.x, pair = blockIdx.y;
    const int H = a.H, W = a.W;
    const int x0 = strip * CW;
    const int cw = min(CW, W - x0);
    const u8* cLp = a.cL + (size_t)pair * a.strideL;
    const u8* cRp = a.cR + (size_t)pair * a.strideR;
    const bool readS = UP && !a.splitUp;
    u16* Sp = (UP && a.splitUp ? a.S2 : a.S) + (size_t)pair * a.strideS;
    const u8* Hp = a.Hvol[UP ? 1 : 0] + (size_t)pair * a.strideH;
    const u16* bA = a.bnd[UP ? 2 : 0] + (size_t)pair * a.strideB;   // diagonal with dx = +1 (predecessor at x-1)
    const u16* bB = a.bnd[UP ? 3 : 1] + (size_t)pair * a.strideB;   // diagonal with dx = -1 (predecessor at x+1)
    u32 tail[NR];
    make_tail(lic, a.D, tail);
    const u32 P1P1 = a.P1P1, P2P2 = a.P2P2;

    u32 Nv[NR];   // vertical path state of column x0 + lr
    zero8(Nv);
    if (TAIL) {
#pragma unroll
        for (int r = 0; r < NR; r++) Nv[r] |= tail[r];
    }
    const int x = x0 + lr;
    const bool colActive = lr < cw;
    const int sy = UP ? -1 : 1, ys = UP ? H - 1 : 0;
    // which strip-border state this lane-row consumes (none for interior columns)
    const bool edgeA = lr == 0 && x != 0;              // predecessor x-1 belongs to the strip on the left
    const bool edgeB = lr == cw - 1 && x != W - 1;     // predecessor x+1 belongs to the strip on the right
    // operands of later rows are requested before the current row is computed
    struct RowIn { u32 cl; uint4 rb, hb; u32 Snext[NR], Hnext[NR], BnA[NR], BnB[NR]; };
    RowIn ring[PFD];
    auto fetch_row = [&](int r, RowIn& in) {
        const int y = ys + sy * r;
        load_cost_operands(a, cLp + (size_t)y * a.pL, cRp + (size_t)y * a.pR, x, lic, in.cl, in.rb);
        const size_t pix = (size_t)y * W + x;
        if (U8) in.hb = *reinterpret_cast<const uint4*>(Hp + pix * 256 + lic * 16);
        else load8(reinterpret_cast<const u16*>(Hp + pix * 512) + lic * DPL, in.Hnext);
        if (readS) load8(Sp + (pix * 256 + lic * DPL), in.Snext);
        if (r > 0) {
            if (edgeA) load8(bA + (((size_t)y * a.nstrips + strip) * 256 + lic * DPL), in.BnA);
            if (edgeB) load8(bB + (((size_t)y * a.nstrips + strip) * 256 + lic * DPL), in.BnB);
        }
    };
#pragma unroll
    for (int u = 0; u < PFD; u++)
        if (colActive && u < H) fetch_row(u, ring[u]);
    auto process_row = [&](int r, RowIn& in) {
        const int y = ys + sy * r;
        const bool hasPrevRow = r > 0;
        const int par = r & 1;
        if (colActive) {
            u16* sp = Sp + (((size_t)y * W + x) * 256 + lic * DPL);
            u32 C[NR], L[NR], Sacc[NR], NA[NR], NB[NR];
            cost_from_operands(in.cl, in.rb, C);
            if (U8) unpack_bytes(in.hb, Sacc);
            else {
#pragma unroll
                for (int q = 0; q < NR; q++) Sacc[q] = in.Hnext[q];
            }
#pragma unroll
            for (int q = 0; q < NR; q++) {
                if (readS) Sacc[q] = pkadd(Sacc[q], in.Snext[q]);
                NA[q] = in.BnA[q]; NB[q] = in.BnB[q];
            }
            if (r + PFD < H) fetch_row(r + PFD, in);
            // vertical
            sgm_step<TAIL>(Nv, C, P1P1, P2P2, tail, L);
#pragma unroll
            for (int q = 0; q < NR; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
            // diagonal arriving from (x-1, previous row)
            if (!hasPrevRow || x == 0) {
                zero8(NA);
                if (TAIL) {
#pragma unroll
                    for (int q = 0; q < NR; q++) NA[q] |= tail[q];
                }
            } else if (lr != 0) {
                if (U8) {
                    unpack_bytes(*reinterpret_cast<const uint4*>(&diag[par ^ 1][0][lr - 1][lic * 4]), NA);
                    if (TAIL) {   // the byte form drops the "infinite" marker of disparities >= D
#pragma unroll
                        for (int q = 0; q < NR; q++) NA[q] |= tail[q];
                    }
                } else {
                    const uint4* src = reinterpret_cast<const uint4*>(&diag[par ^ 1][0][lr - 1][lic * 8]);
                    const uint4 v0 = src[0], v1 = src[1];
                    NA[0] = v0.x; NA[1] = v0.y; NA[2] = v0.z; NA[3] = v0.w; NA[4] = v1.x; NA[5] = v1.y; NA[6] = v1.z; NA[7] = v1.w;
                }
            }
            sgm_step<TAIL>(NA, C, P1P1, P2P2, tail, L);
#pragma unroll
            for (int q = 0; q < NR; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
            if (U8) *reinterpret_cast<uint4*>(&diag[par][0][lr][lic * 4]) = pack_bytes(NA);
            else {
                uint4* dst = reinterpret_cast<uint4*>(&diag[par][0][lr][lic * 8]);
                dst[0] = make_uint4(NA[0], NA[1], NA[2], NA[3]);
                dst[1] = make_uint4(NA[4], NA[5], NA[6], NA[7]);
            }
            // diagonal arriving from (x+1, previous row)
            if (!hasPrevRow || x == W - 1) {
                zero8(NB);
                if (TAIL) {
#pragma unroll
                    for (int q = 0; q < NR; q++) NB[q] |= tail[q];
                }
            } else if (lr != cw - 1) {
                if (U8) {
                    unpack_bytes(*reinterpret_cast<const uint4*>(&diag[par ^ 1][1][lr + 1][lic * 4]), NB);
                    if (TAIL) {
#pragma unroll
                        for (int q = 0; q < NR; q++) NB[q] |= tail[q];
                    }
                } else {
                    const uint4* src = reinterpret_cast<const uint4*>(&diag[par ^ 1][1][lr + 1][lic * 8]);
                    const uint4 v0 = src[0], v1 = src[1];
                    NB[0] = v0.x; NB[1] = v0.y; NB[2] = v0.z; NB[3] = v0.w; NB[4] = v1.x; NB[5] = v1.y; NB[6] = v1.z; NB[7] = v1.w;
                }
            }
            sgm_step<TAIL>(NB, C, P1P1, P2P2, tail, L);
#pragma unroll
            for (int q = 0; q < NR; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
            if (U8) *reinterpret_cast<uint4*>(&diag[par][1][lr][lic * 4]) = pack_bytes(NB);
            else {
                uint4* dst = reinterpret_cast<uint4*>(&diag[par][1][lr][lic * 8]);
                dst[0] = make_uint4(NB[0], NB[1], NB[2], NB[3]);
                dst[1] = make_uint4(NB[4], NB[5], NB[6], NB[7]);
            }
            if (FUSE) {
                fused_wta<TAIL>(a, Sacc, tail, lic, &srow[lr][0], exclMask, (size_t)pair * a.strideRec + (size_t)y * W + x);
            } else if (UP && a.naturalOut) {
                // [R_2g.lo|R_2g+1.lo] = e 2g, 2g+1;  [R_2g.hi|R_2g+1.hi] = e 2g+8, 2g+9
                u32 nat[NR];
#pragma unroll
                for (int g = 0; g < NR / 2; g++) {
                    nat[g] = __builtin_amdgcn_perm(Sacc[2 * g + 1], Sacc[2 * g], 0x05040100u);
                    nat[g + 4] = __builtin_amdgcn_perm(Sacc[2 * g + 1], Sacc[2 * g], 0x07060302u);
                }
                store8(sp, nat);
            } else store8(sp, Sacc);
        }
        __syncthreads();   // diag[par] complete before the next row reads it
    };
    for (int r0 = 0; r0 < H; r0 += PFD) {
#pragma unroll
        for (int u = 0; u < PFD; u++) {
            const int r = r0 + u;
            if (r >= H) break;
            process_row(r, ring[u]);
        }
    }
}

// Latency variant of sgm2_sweep_kernel for the smallest batches (byte path states, four-path sums that fit a byte):
// the three paths of a sweep run on three groups of four waves, so the serial chain of a row is one path step
// instead of three.  Every group evaluates the matching cost itself.  Group 0 runs the vertical path and keeps
// the row's partial sum (horizontal volume + vertical path, as bytes); groups 1 and 2 run the two diagonals,
// exchange their states through LDS as the one-group kernel does, and leave their path values in LDS, where
// group 0 picks them up one row later, adds them and stores the four-path sum as 16 bytes per lane.
inline __device__ uint4 add4(const uint4& p, const uint4& q) { return make_uint4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w); }

template <bool UP, bool TAIL, int PFD = 4>
__global__ __launch_bounds__(768) void sgm2_sweep3_kernel(Sgm2Args a)
{
    __shared__ u32 diag[2][2][CW][64];    // [row parity][diagonal][column]: normalised states, bytes
    __shared__ u32 part[2][2][CW][64];    // [row parity][diagonal][column]: path values, bytes
    // the group is the same for every lane of a wave: kept in a scalar register so that branches on it are jumps
    const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)), t = threadIdx.x & 255;
    const int lic = t & 15, lr = t >> 4;
    const int strip = blockIdx.x, pair = blockIdx.y;
    const int H = a.H, W = a.W;
    const int x0 = strip * CW;
    const int cw = min(CW, W - x0);
    const int x = x0 + lr;
    const bool colActive = lr < cw;
    const int xl = min(x, W - 1);             // the idle columns of the last strip request and compute a valid pixel
    const u8* cLp = a.cL + (size_t)pair * a.strideL;
    const u8* cRp = a.cR + (size_t)pair * a.strideR;
    u8* Sp = reinterpret_cast<u8*>((UP ? a.S2 : a.S) + (size_t)pair * a.strideS);
    const u8* Hp = a.Hvol[UP ? 1 : 0] + (size_t)pair * a.strideH;
    // group 1: diagonal with dx = +1 (predecessor at x-1), group 2: dx = -1 (predecessor at x+1)
    const u16* bnd = a.bnd[(UP ? 2 : 0) + (grp == 2 ? 1 : 0)] + (size_t)pair * a.strideB;
    const int fam = grp == 2 ? 1 : 0;
    const int nbCol = min(max(grp == 2 ? lr + 1 : lr - 1, 0), CW - 1);
    const bool atBorder = grp == 2 ? x == W - 1 : x == 0;                                  // the path starts at this pixel
    const bool atEdge = !atBorder && (grp == 2 ? lr == cw - 1 : lr == 0);                   // predecessor in the next strip
    const bool waveEdge = grp != 0 && __any(atEdge && colActive);                           // does this wave consume a border state?
    u32 tail[NR];
    make_tail(lic, a.D, tail);
    const u32 P1P1 = a.P1P1, P2P2 = a.P2P2;
    const int sy = UP ? -1 : 1, ys = UP ? H - 1 : 0;

    u32 N[NR];
    zero8(N);
    if (TAIL) {
#pragma unroll
        for (int r = 0; r < NR; r++) N[r] |= tail[r];
    }
    // A row is a short chain, so operands are requested PFD rows ahead.  The vertical group and the diagonal groups
    // run separate copies of the row loop (same number of barriers in each): with the two kinds of request in one
    // loop body the compiler waits for the cost operands before it issues the other group's load.  Every request sits
    // under wave-uniform conditions only (a load under a per-lane condition is waited for right behind it), and the
    // left census byte travels as its aligned dword (a byte carried around the loop is re-extended, i.e. waited for,
    // behind its load).
    if (grp == 0) {
        struct RowIn { u32 clw; uint4 rb, hb; };
        RowIn ring[PFD];
        auto fetch_row = [&](int r, RowIn& in) {
            const int y = ys + sy * r;
            load_cost_operands_w(a, cLp + (size_t)y * a.pL, cRp + (size_t)y * a.pR, xl, lic, in.clw, in.rb);
            in.hb = *reinterpret_cast<const uint4*>(Hp + ((size_t)y * W + xl) * 256 + lic * 16);
        };
#pragma unroll
        for (int u = 0; u < PFD; u++) fetch_row(min(u, H - 1), ring[u]);
        uint4 pend = make_uint4(0, 0, 0, 0);      // horizontal + vertical bytes of the previous row
        // whole groups of PFD rows run without an exit in the middle (an early exit makes the compiler rotate the ring
        // through copies, which wait for the requests just issued); the last rows use what the ring already holds
        auto row = [&](int r, RowIn& in, bool more) {
                const int y = ys + sy * r, par = r & 1;
                u32 C[NR], L[NR];
                cost_from_operands(census_byte(in.clw, xl), in.rb, C);
                const uint4 hcur = in.hb;
                if (more) fetch_row(min(r + PFD, H - 1), in);
                if (r > 0 && colActive) {
                    const uint4 pa = *reinterpret_cast<const uint4*>(&part[par ^ 1][0][lr][lic * 4]);
                    const uint4 pb = *reinterpret_cast<const uint4*>(&part[par ^ 1][1][lr][lic * 4]);
                    store16_stream(Sp + ((size_t)(y - sy) * W + x) * 256 + lic * 16, add4(pend, add4(pa, pb)));
                }
                sgm_step<TAIL>(N, C, P1P1, P2P2, tail, L);
                pend = add4(hcur, pack_bytes(L));
                __syncthreads();   // part[par] complete before the next row reads it
        };
        int r = 0;
        for (; r + PFD <= H; r += PFD) {
#pragma unroll
            for (int u = 0; u < PFD; u++) row(r + u, ring[u], true);
        }
#pragma unroll
        for (int u = 0; u < PFD; u++)
            if (r + u < H) row(r + u, ring[u], false);
        if (colActive) {
            const int par = (H - 1) & 1, y = ys + sy * (H - 1);
            const uint4 pa = *reinterpret_cast<const uint4*>(&part[par][0][lr][lic * 4]);
            const uint4 pb = *reinterpret_cast<const uint4*>(&part[par][1][lr][lic * 4]);
            store16_stream(Sp + ((size_t)y * W + x) * 256 + lic * 16, add4(pend, add4(pa, pb)));
        }
    } else {
        struct RowIn { u32 clw; uint4 rb; u32 Bn[NR]; };
        RowIn ring[PFD];
        auto fetch_row = [&](int r, RowIn& in) {
            const int y = ys + sy * r;
            load_cost_operands_w(a, cLp + (size_t)y * a.pL, cRp + (size_t)y * a.pR, xl, lic, in.clw, in.rb);
            if (waveEdge) load8(bnd + (((size_t)y * a.nstrips + strip) * 256 + lic * DPL), in.Bn);   // row 0's slot is never written nor used
        };
#pragma unroll
        for (int u = 0; u < PFD; u++) {
            zero8(ring[u].Bn);
            fetch_row(min(u, H - 1), ring[u]);
        }
        auto row = [&](int r, RowIn& in, bool more) {
                const int par = r & 1;
                u32 C[NR], L[NR], Bc[NR], Nd[NR];
                cost_from_operands(census_byte(in.clw, xl), in.rb, C);
#pragma unroll
                for (int q = 0; q < NR; q++) Bc[q] = in.Bn[q];
                if (more) fetch_row(min(r + PFD, H - 1), in);
                unpack_bytes(*reinterpret_cast<const uint4*>(&diag[par ^ 1][fam][nbCol][lic * 4]), Nd);
                const bool fresh = r == 0 || atBorder;
#pragma unroll
                for (int q = 0; q < NR; q++) {
                    // the byte form drops the "infinite" marker of disparities >= D
                    const u32 fromLds = TAIL ? Nd[q] | tail[q] : Nd[q];
                    N[q] = fresh ? (TAIL ? tail[q] : 0u) : atEdge ? Bc[q] : fromLds;
                }
                sgm_step<TAIL>(N, C, P1P1, P2P2, tail, L);
                if (colActive) {
                    *reinterpret_cast<uint4*>(&diag[par][fam][lr][lic * 4]) = pack_bytes(N);
                    *reinterpret_cast<uint4*>(&part[par][fam][lr][lic * 4]) = pack_bytes(L);
                }
                __syncthreads();   // diag[par] and part[par] complete before the next row reads them
        };
        int r = 0;
        for (; r + PFD <= H; r += PFD) {
#pragma unroll
            for (int u = 0; u < PFD; u++) row(r + u, ring[u], true);
        }
#pragma unroll
        for (int u = 0; u < PFD; u++)
            if (r + u < H) row(r + u, ring[u], false);
    }
}

// ---------------------------------------------------------------------------------------------
// Sheared sweep (byte-state mode only): the same three paths per sweep as sgm2_sweep_kernel, but a
// workgroup owns 256/LPC columns of the SHEARED frame c = x - y.  In that frame
//   down sweep:  (1,1)  stays in its column (registers),  (0,1) arrives from c+1,  (-1,1) from c+2
//   up sweep:    (-1,-1) stays in its column,             (0,-1) arrives from c-1, (1,-1) from c-2
// so every cross-strip dependency points the same way and neighbouring strips form a pipeline:
// a strip hands the states of its two outermost columns to the next strip through global memory,
// one row at a time, and no separate pass over the diagonals is needed.
//
// Protocol (one direction only, so it cannot deadlock): workgroups take a ticket when they start;
// ticket order is strip order along the dependency direction (all pairs of strip k before any
// pair of the strip that depends on it), so a workgroup only ever waits for one that started
// earlier.  A fifth "communication" wave does the exchange: it writes the edge states with
// agent-scope (write-through) stores, drains them (vmcnt(0)) and then publishes "rows done" in the
// strip's progress word; on the consuming side it reads the neighbour's progress word and states
// with agent-scope loads and parks them in LDS for the compute waves.  Every wait is bounded: on
// timeout the error word is set, all waits stop and the host reports the failure.
struct ShearArgs {
    Sgm2Args a;
    u8* hand;               // [B][H][nsh][3][LPC*16]: per row and strip, V of m=0, B of m=0, B of m=1
    size_t strideHand;      // bytes per pair
    int* sync;              // [0] ticket (down) [1] ticket (up), [16 + (sweep*B + pair)*nsh + k] progress
    int* err;               // sticky error word (owned by the context)
    int nsh, cmin, batch;
};
// compute lanes of a sheared-sweep workgroup (+ one communication wave), i.e. strips of NT / LPC columns: 256 lanes for
// large batches (a row barrier over five waves), 512 for small batches and for the five-path mode (half as many strips:
// half the hand-off traffic and half the pipeline ramp, at the price of a row barrier over nine waves)

template <bool UP, bool TAIL, bool FUSE, bool S4B, int LPC, bool FIVE = false, int SHEAR_NT = 256>
__global__ __launch_bounds__(SHEAR_NT + 64) void sgm2_shear_kernel(ShearArgs sa)
{
    // LPC lanes share a pixel (16 disparities each); a strip is CW = SHEAR_NT/LPC sheared columns wide
    constexpr int CW = SHEAR_NT / LPC, DP = LPC * DPL, SW = LPC * 4;
    const Sgm2Args& a = sa.a;
    __shared__ u32 vs[2][CW][SW];      // vertical-path states of the previous / current row, as bytes
    __shared__ u32 bs[2][CW][SW];      // states of the diagonal that crosses two sheared columns per row
    __shared__ u32 hin[2][3][SW];      // neighbour strip's states for the current / next row: V->m=CW-1, B->m=CW-2, B->m=CW-1
    __shared__ u32 srow[FUSE ? CW : 1][FUSE ? LPC * 8 : 1];
    __shared__ u32 exclMask[FUSE ? 19 : 1][NR];
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&sa.sync[UP ? 1 : 0], 1);
    if (FUSE) init_excl_mask(exclMask);
    __syncthreads();
    const int ticket = s_ticket;
    const int ord = ticket / sa.batch, pair = ticket - ord * sa.batch;
    const int k = UP ? ord : sa.nsh - 1 - ord;              // strips in dependency order
    const int H = a.H, W = a.W;
    const int c0 = sa.cmin + CW * k;
    const int ylo = max(0, -(c0 + CW - 1)), yhi = min(H, W - c0);   // rows in which the strip touches the image
    const int nrows = yhi - ylo;
    if (nrows <= 0) return;
    const int sy = UP ? -1 : 1, ys = UP ? yhi - 1 : ylo;
    u8* hand = sa.hand + (size_t)pair * sa.strideHand;
    const int nbk = UP ? k - 1 : k + 1;
    const bool hasNb = nbk >= 0 && nbk < sa.nsh;
    auto has_prev = [&](int y) { return UP ? y < H - 1 : y > 0; };
    auto hand_slot = [&](int y, int strip, int which, int lic) -> u8* {
        return hand + ((((size_t)y * sa.nsh + strip) * 3 + which) * DP + lic * 16);
    };

    if (threadIdx.x >= SHEAR_NT) {
        // ---- communication wave: publishes this strip's edge states one row behind the compute
        // waves and fetches the neighbour's one row ahead of them, so that neither the write-through
        // drain nor a wait for a slow neighbour sits in the compute waves' instruction stream
        const int lane = threadIdx.x - SHEAR_NT, which = lane / LPC, lic = lane % LPC;
        int* progMe = sa.sync + 16 + ((size_t)((UP ? sa.batch : 0) + pair) * sa.nsh + k);
        const int* progNb = progMe + (UP ? -1 : 1);
        int* errWord = sa.err;
        int seen = 0;
        // m = CW-1 sits in sheared column c0 + CW-1 (down) / c0 (up); m = CW-2 next to it
        const int c15 = UP ? c0 : c0 + CW - 1, c14 = UP ? c0 + 1 : c0 + CW - 2;
        // does this lane fetch a neighbour state for row i?
        auto wants = [&](int i) -> bool {
            const int y = ys + sy * i;
            if (!hasNb || !has_prev(y) || which > 2) return false;
            const int x = (which == 1 ? c14 : c15) + y;
            if (x < 0 || x >= W) return false;
            return which == 0 || (UP ? x != 0 : x != W - 1);
        };
        // request the neighbour states row i consumes; waits (bounded) until the neighbour has published them
        auto request = [&](int i, uint4& w) {
            const bool mine = wants(i);
            if (!__any(mine)) return;
            const int y = ys + sy * i;
            const int need = UP ? H - 1 - y : y;         // rows the neighbour must have published
            if (seen < need) {
                unsigned long long t0 = 0;
                for (int it = 0;; it++) {
                    seen = __hip_atomic_load(progNb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (seen >= need) break;
                    bool late = false;
                    if ((it & 63) == 63) {
                        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                        if (t0 == 0) t0 = now;
                        late = now - t0 > HANDOFF_TIMEOUT_TICKS;
                    }
                    if (late || __hip_atomic_load(errWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(errWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        seen = 0x7FFFFFFF;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
            }
            if (mine) w = hand_load(hand_slot(y - sy, nbk, which, lic));
        };
        auto park = [&](int i, const uint4& w) {          // hand the fetched states of row i to the compute waves
            if (wants(i)) *reinterpret_cast<uint4*>(&hin[i & 1][which][lic * 4]) = w;
        };
        auto send = [&](int i) {                           // row i has been computed: its edge states are in vs/bs[i & 1]
            if (which < 3) {
                const int par = i & 1;
                const uint4 w = which == 0 ? *reinterpret_cast<const uint4*>(&vs[par][0][lic * 4])
                                           : *reinterpret_cast<const uint4*>(&bs[par][which - 1][lic * 4]);
                hand_store(hand_slot(ys + sy * i, k, which, lic), w);
            }
        };
        auto announce = [&](int i) {                       // rows up to i are in memory
            const int y = ys + sy * i;
            if (lane == 0) __hip_atomic_store(progMe, (UP ? H - 1 - y : y) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // Software pipeline, one step per row, that never waits for an operation issued in the same
        // step: step i first drains what step i-1 issued (stores of row i-2, loads for row i+1, a
        // sample of the neighbour's progress), then announces row i-2, sends row i-1, parks row i+1's
        // states in LDS and requests row i+2's.  The progress sample makes the request's own poll
        // (which has to wait for its answer) the exception.
        uint4 w = make_uint4(0, 0, 0, 0);
        request(0, w);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        park(0, w);
        if (nrows > 1) request(1, w);
        int sample = 0;
        __syncthreads();
        for (int i = 0; i < nrows; i++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            seen = max(seen, sample);
            if (i >= 2) announce(i - 2);
            if (i >= 1) send(i - 1);
            if (i + 1 < nrows) park(i + 1, w);
            if (i + 2 < nrows) request(i + 2, w);
            if (hasNb) sample = __hip_atomic_load(progNb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        }
        send(nrows - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        announce(nrows - 1);
        return;
    }

    // ---- compute waves ----
    const int lic = threadIdx.x % LPC, lr = threadIdx.x / LPC;
    const int m = UP ? CW - 1 - lr : lr;                          // column index counted against the dependency direction
    const int c = c0 + lr;
    const u8* cLp = a.cL + (size_t)pair * a.strideL;
    const u8* cRp = a.cR + (size_t)pair * a.strideR;
    u16* Sp = a.S + (size_t)pair * a.strideS;
    const u8* Hp = a.Hvol[UP ? 1 : 0] + (size_t)pair * a.strideH;
    u32 tail[NR];
    make_tail(lic, a.D, tail);
    const u32 P1P1 = a.P1P1, P2P2 = a.P2P2;
    auto fresh = [&](u32 (&N)[NR]) {
#pragma unroll
        for (int q = 0; q < NR; q++) N[q] = TAIL ? tail[q] : 0u;
    };
    auto from_bytes = [&](const uint4& w, u32 (&N)[NR]) {
        unpack_bytes(w, N);
        if (TAIL) {
#pragma unroll
            for (int q = 0; q < NR; q++) N[q] |= tail[q];
        }
    };
    u32 Nr[NR];                        // the diagonal that stays in this sheared column
    fresh(Nr);
    u32 cl = 0; uint4 rb = make_uint4(0, 0, 0, 0), hb = make_uint4(0, 0, 0, 0), hb2 = make_uint4(0, 0, 0, 0);
    u32 Snext[NR];
    constexpr bool five = !UP && FIVE;   // this down sweep also adds the (-1,0) volume and finishes the sum
    const u8* Hp2 = a.Hvol[1] + (size_t)pair * a.strideH;
    uint4 sb = make_uint4(0, 0, 0, 0);
    // S4B: the four-path partial sum fits a byte per disparity and travels as one byte per slot
    u8* Sb = reinterpret_cast<u8*>(Sp);
    auto fetch_row = [&](int i) {
        const int y = ys + sy * i, x = c + y;
        if (x >= 0 && x < W) {
            load_cost_operands(a, cLp + (size_t)y * a.pL, cRp + (size_t)y * a.pR, x, lic, cl, rb);
            const size_t pix = (size_t)y * W + x;
            hb = load16_stream(Hp + pix * DP + lic * 16);
            if (five) hb2 = load16_stream(Hp2 + pix * DP + lic * 16);
            if (UP) { if (S4B) sb = load16_stream(Sb + pix * DP + lic * 16); else load8_stream(Sp + (pix * DP + lic * DPL), Snext); }
        }
    };
    fetch_row(0);
    __syncthreads();                   // the communication wave has fetched the first row's neighbour states
    for (int i = 0; i < nrows; i++) {
        const int y = ys + sy * i, x = c + y;
        const bool active = x >= 0 && x < W;
        const bool hasPrev = has_prev(y);
        const int par = i & 1;
        u32 C[NR], L[NR], Sacc[NR];
        if (active) {
            cost_from_operands(cl, rb, C);
            // byte volumes whose sum still fits a byte are added as bytes (four per plain 32-bit add) and unpacked once
            if (five && a.byteSum) unpack_bytes(make_uint4(hb.x + hb2.x, hb.y + hb2.y, hb.z + hb2.z, hb.w + hb2.w), Sacc);
            else if (UP && S4B && a.byteSum) unpack_bytes(make_uint4(hb.x + sb.x, hb.y + sb.y, hb.z + sb.z, hb.w + sb.w), Sacc);
            else {
                unpack_bytes(hb, Sacc);
                if (five) {
                    unpack_bytes(hb2, Snext);
#pragma unroll
                    for (int q = 0; q < NR; q++) Sacc[q] = pkadd(Sacc[q], Snext[q]);
                }
                if (UP) {
                    if (S4B) unpack_bytes(sb, Snext);
#pragma unroll
                    for (int q = 0; q < NR; q++) Sacc[q] = pkadd(Sacc[q], Snext[q]);
                }
            }
        }
        if (i + 1 < nrows) fetch_row(i + 1);
        if (active) {
            // diagonal resident in this column: predecessor (x-1, y-1) going down, (x+1, y+1) going up
            if (!hasPrev || (UP ? x == W - 1 : x == 0)) fresh(Nr);
            sgm_step<TAIL, LPC>(Nr, C, P1P1, P2P2, tail, L, lic);
#pragma unroll
            for (int q = 0; q < NR; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
            // vertical path: predecessor (x, y -+ 1) lives one sheared column further
            u32 N[NR];
            if (!hasPrev) fresh(N);
            else if (m < CW - 1) from_bytes(*reinterpret_cast<const uint4*>(&vs[par ^ 1][m + 1][lic * 4]), N);
            else from_bytes(*reinterpret_cast<const uint4*>(&hin[par][0][lic * 4]), N);
            sgm_step<TAIL, LPC>(N, C, P1P1, P2P2, tail, L, lic);
#pragma unroll
            for (int q = 0; q < NR; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
            *reinterpret_cast<uint4*>(&vs[par][m][lic * 4]) = pack_bytes(N);
            // the other diagonal: predecessor (x+1, y-1) going down, (x-1, y+1) going up, two columns further
            if (!hasPrev || (UP ? x == 0 : x == W - 1)) fresh(N);
            else if (m < CW - 2) from_bytes(*reinterpret_cast<const uint4*>(&bs[par ^ 1][m + 2][lic * 4]), N);
            else from_bytes(*reinterpret_cast<const uint4*>(&hin[par][m - (CW - 3)][lic * 4]), N);
            sgm_step<TAIL, LPC>(N, C, P1P1, P2P2, tail, L, lic);
#pragma unroll
            for (int q = 0; q < NR; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
            *reinterpret_cast<uint4*>(&bs[par][m][lic * 4]) = pack_bytes(N);
            const size_t pix = (size_t)y * W + x;
            if (FUSE) fused_wta<TAIL, LPC>(a, Sacc, tail, lic, &srow[lr][0], exclMask, (size_t)pair * a.strideRec + pix);
            else if ((UP || five) && a.naturalOut) {
                u32 nat[NR];
#pragma unroll
                for (int g = 0; g < NR / 2; g++) {
                    nat[g] = __builtin_amdgcn_perm(Sacc[2 * g + 1], Sacc[2 * g], 0x05040100u);
                    nat[g + 4] = __builtin_amdgcn_perm(Sacc[2 * g + 1], Sacc[2 * g], 0x07060302u);
                }
                store8(Sp + (pix * DP + lic * DPL), nat);
            } else if (S4B) store16_stream(Sb + pix * DP + lic * 16, pack_bytes(Sacc));
            else store8_stream(Sp + (pix * DP + lic * DPL), Sacc);
        }
        __syncthreads();   // vs/bs[par] complete before the next row reads them
    }
}

// Split mode (small batches): the down and the up sweep run side by side on two streams and each
// stores its four-path sum (register order); this kernel adds the two and runs the winner-take-all.
// One workgroup per image row.
template <bool TAIL, bool BYTES = false>
__global__ __launch_bounds__(256) void sgm2_wta_sum_kernel(Sgm2Args a)
{
    __shared__ u32 srow[CW][128];
    __shared__ u32 exclMask[19][NR];
    init_excl_mask(exclMask);
    __syncthreads();
    const int lic = threadIdx.x & 15, lr = threadIdx.x >> 4;
    const int y = blockIdx.x, pair = blockIdx.y, W = a.W;
    const u16* Sa = a.S + (size_t)pair * a.strideS;
    const u16* Sb = a.S2 + (size_t)pair * a.strideS;
    u32 tail[NR];
    make_tail(lic, a.D, tail);
    for (int x0 = 0; x0 < W; x0 += CW) {
        const int x = x0 + lr;
        if (x < W) {
            const size_t pix = (size_t)y * W + x;
            u32 A[NR], B[NR];
            if (BYTES) {      // four-path sums stored as bytes by sgm2_sweep3_kernel
                const uint4 wa = load16_stream(reinterpret_cast<const u8*>(Sa) + pix * 256 + lic * 16);
                const uint4 wb = load16_stream(reinterpret_cast<const u8*>(Sb) + pix * 256 + lic * 16);
                unpack_bytes(wa, A);
                unpack_bytes(wb, B);
            } else {
                load8_stream(Sa + (pix * 256 + lic * DPL), A);
                load8_stream(Sb + (pix * 256 + lic * DPL), B);
            }
#pragma unroll
            for (int q = 0; q < NR; q++) A[q] = pkadd(A[q], B[q]);
            fused_wta<TAIL>(a, A, tail, lic, &srow[lr][0], exclMask, (size_t)pair * a.strideRec + pix);
        }
    }
}

struct LrArgs {
    const uint2* rec; size_t strideRec;
    int H, W, D, minD, maxDiff, subpixel;
    int16_t* disp; size_t strideDisp;
    int y0;         // first image row of the launch
};

// One workgroup per image row: sub-pixel parabola, disp2 slot election (minimum S, ties to the
// larger x: StereoSGBM walks x right to left with a strict '>' test) by a packed atomicMin in
// LDS, then the disp12MaxDiff check.
__global__ __launch_bounds__(256) void sgm2_lr_kernel(LrArgs a)
{
    extern __shared__ u32 lds[];
    u32* key = lds;                                          // [W] (S'min << 16) | (0xFFFF - x)
    int16_t* d1 = reinterpret_cast<int16_t*>(lds + a.W);     // [W]
    const int y = a.y0 + blockIdx.x, W = a.W, D = a.D;
    const int INVALID = (a.minD - 1) * 16;
    const uint2* rrow = a.rec + (size_t)blockIdx.y * a.strideRec + (size_t)y * W;
    int16_t* drow = a.disp + (size_t)blockIdx.y * a.strideDisp + (size_t)y * W;
    for (int i = threadIdx.x; i < W; i += 256) key[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        const uint2 r = rrow[x];
        int v = INVALID;
        if (!(r.x & 0x80000000u)) {
            const int minS = (int)(r.x & 0xFFFFu);
            int d = (int)((r.x >> 16) & 0x7FFFu);
            const int x2 = x - d - a.minD;
            if (x2 >= 0 && x2 < W) atomicMin(&key[x2], ((u32)minS << 16) | (u32)(0xFFFF - x));
            if (a.subpixel && d > 0 && d < D - 1) {
                const int sm = (int)(r.y & 0xFFFFu), spl = (int)(r.y >> 16);
                int den = sm + spl - 2 * minS;
                if (den < 1) den = 1;
                d = d * 16 + ((sm - spl) * 16 + den) / (den * 2);
            } else d *= 16;
            v = d + a.minD * 16;
        }
        d1[x] = (int16_t)v;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        int v = d1[x];
        if (v != INVALID) {
            auto inconsistent = [&](int xx, int dd) -> bool {
                if (xx < 0 || xx >= W) return false;
                const u32 kk = key[xx];
                const int d2 = (kk == 0xFFFFFFFFu) ? INVALID : (0xFFFF - (int)(kk & 0xFFFF)) - xx;
                return d2 >= a.minD && abs(d2 - dd) > a.maxDiff;
            };
            const int _d = v >> 4, d_ = (v + 15) >> 4;
            if (inconsistent(x - _d, _d) && inconsistent(x - d_, d_)) v = INVALID;
        }
        drow[x] = (int16_t)v;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------

static bool fits_u8(const md_sgm_plan& pl) { return 8 * pl.costWin * pl.costWin + pl.P2 <= 255; }

// How the eight (five) paths are scheduled:
//   SHEAR    pipelined sheared sweeps (byte path states): the throughput mode for batches
//   PREPASS  upright strips + diagonal pre-pass, one sweep after the other (16-bit path states, or on request)
//   SPLIT    upright strips, the two sweeps side by side on two streams and a summing winner-take-all: a batch of
//            a few pairs cannot fill the strip pipeline, so the latency of one pair counts
//   ROWS     row bands with four paths per sweep (sgm3.hip): no horizontal volumes, one cost evaluation per sweep;
//            the throughput mode for eight paths (SHEAR remains for five paths and on request)
enum { SGM2_SHEAR = 0, SGM2_PREPASS = 1, SGM2_SPLIT = 2, SGM2_ROWS = 3 };
static int split_max_batch()
{
    const int e = md_tune(MDISP_TUNE_SGM_SPLIT_MAX);
    return e >= 0 ? e : 4;      // from 5 pairs on the strip pipeline (512-lane strips) is faster
}
static int path_waves_max_batch()
{
    const int e = md_tune(MDISP_TUNE_SGM_PATH_WAVES_MAX);
    return e >= 0 ? e : 4;      // largest batch whose sweeps run one path per wave group (every batch of the latency schedule)
}
static int sgm2_mode(const md_sgm_plan& pl)
{
    if (pl.npaths == 4) return SGM2_ROWS;             // the two horizontal and the two vertical paths: row bands without the diagonal steps
    if (pl.npaths == 5 && !pl.exportVolume && (!fits_u8(pl) || (md_tune(MDISP_TUNE_SGM_PIPELINE) != 1 && pl.batch > split_max_batch())))
        return SGM2_ROWS;                             // four paths in one sweep + the opposite horizontal path with the winner-take-all
    const bool rowsWanted = pl.npaths == 8 && md_tune(MDISP_TUNE_SGM_PIPELINE) != 1;
    if (!fits_u8(pl))      // 16-bit path states: row bands for batches, upright strips + pre-pass for a few pairs
        return rowsWanted && pl.batch > split_max_batch() && md_tune(MDISP_TUNE_SGM_FORCE_PREPASS) <= 0 ? SGM2_ROWS : SGM2_PREPASS;
    if (md_tune(MDISP_TUNE_SGM_FORCE_PREPASS) > 0) return SGM2_PREPASS;
    if (pl.D > 256 || pl.rowBands) return SGM2_ROWS;  // 512 slots per pixel exist only as row bands
    if (pl.npaths == 8 && pl.batch <= split_max_batch()) return SGM2_SPLIT;
    if (rowsWanted) return SGM2_ROWS;
    return SGM2_SHEAR;
}

// lanes per pixel and disparities per lane: the sheared kernels run 64 / 128 / 256 disparity slots per pixel at 16
// disparities per lane, the row-band kernels also 32 per lane (256 slots on 8 lanes, 512 on 16), the others always 16 x 16
static bool rows_wide_lanes(const md_sgm_plan& pl)
{
    // measured at 256 disparities: 32 per lane on eight lanes saves a tenth of the instructions but needs 140 to 220
    // registers per lane (two to three waves per SIMD) and runs 3 % slower than 16 per lane on sixteen lanes
    return sgm2_mode(pl) == SGM2_ROWS && pl.D > 256;
}
static int lanes_per_pixel(const md_sgm_plan& pl)
{
    const int mode = sgm2_mode(pl);
    if (rows_wide_lanes(pl)) return pl.D > 256 ? 16 : 8;
    if (pl.rowBands || (mode == SGM2_ROWS && (!fits_u8(pl) || pl.npaths == 4))) return 16;      // row-sharded calls, 16-bit path states: one geometry
    if ((mode != SGM2_SHEAR && mode != SGM2_ROWS) || md_tune(MDISP_TUNE_SGM_LANES16) > 0) return 16;
    return pl.D <= 64 ? 4 : pl.D <= 128 ? 8 : 16;
}
static int disp_per_lane(const md_sgm_plan& pl) { return rows_wide_lanes(pl) ? 32 : DPL; }
int md_sgm2_slots(const md_sgm_plan& pl) { return lanes_per_pixel(pl) * disp_per_lane(pl); }

static int shear_strips(int H, int W, int cw) { return (W + H - 1 + cw - 1) / cw; }
static int shear_nt(const md_sgm_plan& pl, int batch)
{
    const int e = md_tune(MDISP_TUNE_SGM_STRIP_LANES);
    if (e >= 0) return e == 512 ? 512 : 256;
    return (pl.npaths == 5 || batch <= 32) ? 512 : 256;
}
// sgm3.hip
int md_sgm3_lanes(int lpc, int dpl);
int md_sgm3_bands(int H, int lpc, int nt);
size_t md_sgm3_hand_bytes(int H, int W, int lpc, int dpl, int nt, bool s16);
size_t md_sgm3_sync_bytes(int H, int batch, int lpc, int nt);
int md_sgm3_sweeps(hipStream_t s, const Sgm2Args& a, u8* hand, size_t strideHand, int* sync, int* err, int batch, int lpc, int dpl, int nt,
                   int naturalOut, bool s4b, bool s16);
bool md_sgm3_meet_supported(const Sgm2Args& a, int lpc, int dpl, bool s4b, bool s16);
int md_sgm3_meet(hipStream_t s, const Sgm2Args& a, u8* handDown, u8* handUp, size_t strideHand, int* sync, int* err, int batch, int lpc, int dpl, int nt,
                 int firstDown, int firstUp);
int md_sgm3_sweep(hipStream_t s, const Sgm2Args& a, u8* hand, size_t strideHand, int* sync, int* err, int batch, int lpc, int dpl, int nt,
                  bool s4b, bool up, int role, int band0, int nbands, bool s16, int ticket);
int md_sgm3_ring();
size_t md_sgm3_boundary_bytes(int W, int lpc, int dpl, bool s16);
void md_sgm3_band_range(int H, int lpc, int nt, bool up, int r0, int r1, int* kb, int* ke);

static size_t shear_hand_bytes(int H, int W, int lpc, int nt) { return md_align((size_t)H * shear_strips(H, W, nt / lpc) * 3 * lpc * DPL, 256); }
static size_t shear_sync_bytes(int H, int W, int batch, int lpc, int nt)
{ return md_align((16 + 2 * (size_t)batch * shear_strips(H, W, nt / lpc)) * sizeof(int), 256); }

struct Sgm2Layout { size_t oneB, bndBytes, oneH, oneRec, oneS2, syncBytes; int nstrips, lpc, dpl, slots, mode, cw; bool u8m; };
static Sgm2Layout sgm2_layout(const md_sgm_plan& pl, int batch)
{
    Sgm2Layout l;
    l.u8m = fits_u8(pl);
    l.mode = sgm2_mode(pl);
    l.lpc = lanes_per_pixel(pl);
    l.dpl = disp_per_lane(pl);
    l.slots = l.lpc * l.dpl;
    l.cw = CW;
    l.nstrips = (pl.W + l.cw - 1) / l.cw;
    if (l.mode == SGM2_SHEAR) { l.oneB = 0; l.bndBytes = shear_hand_bytes(pl.H, pl.W, l.lpc, 256); }   // hand-off rows (sized for the narrower strips)
    else if (l.mode == SGM2_ROWS) { l.oneB = 0; l.bndBytes = 2 * md_sgm3_hand_bytes(pl.H, pl.W, l.lpc, l.dpl, md_sgm3_lanes(l.lpc, l.dpl), !l.u8m); }   // hand-off rings of the two sweep directions
    else { l.oneB = md_align((size_t)pl.H * l.nstrips * 256 * sizeof(u16), 256); l.bndBytes = 4 * l.oneB; }   // four border-state planes
    l.oneH = l.mode == SGM2_ROWS ? 0 : md_align((size_t)pl.H * pl.W * l.slots * (l.u8m ? 1 : 2), 256);
    l.oneRec = md_align((size_t)pl.H * pl.W * sizeof(uint2), 256);
    l.oneS2 = l.mode == SGM2_SPLIT ? md_align((size_t)pl.H * pl.W * 256 * sizeof(u16), 256) : 0;
    l.syncBytes = l.mode == SGM2_ROWS ? md_sgm3_sync_bytes(pl.H, batch, l.lpc, md_sgm3_lanes(l.lpc, l.dpl)) : shear_sync_bytes(pl.H, pl.W, batch, l.lpc, 256);
    return l;
}

size_t md_sgm2_extra_bytes(const md_sgm_plan& pl, int batch)
{
    const Sgm2Layout l = sgm2_layout(pl, batch);
    return (l.bndBytes + 2 * l.oneH + l.oneRec + l.oneS2) * batch + l.syncBytes;
}

bool md_sgm2_supported(const md_sgm_plan& pl)
{
    return (pl.npaths == 8 || pl.npaths == 4 || (pl.npaths == 5 && (fits_u8(pl) || !pl.exportVolume))) && (pl.D <= 256 || (pl.npaths == 8 && fits_u8(pl))) && pl.costWin == 1 && (long)pl.npaths * (8 + pl.P2) < 4096 && pl.P1 < 2048 &&
           pl.maxD - pl.minD + 1 == pl.D && pl.uniq <= 100;
}

bool md_sgm2_u8(const md_sgm_plan& pl) { return fits_u8(pl); }
size_t md_sgm2_partial_sum_bytes(const md_sgm_plan& pl)
{
    const int mode = sgm2_mode(pl);
    const bool byteSums = 4 * (8 * pl.costWin * pl.costWin + pl.P2) <= 255;
    return (mode == SGM2_ROWS || (mode == SGM2_SHEAR && pl.npaths == 8)) && byteSums ? 1 : 2;
}

namespace {
template <int LPC, int NT>
int launch_shear(hipStream_t s, const md_sgm_plan& pl, const ShearArgs& sa, int batch, int naturalOut)
{
    const Sgm2Args& a = sa.a;
    constexpr int LINES = 256 / LPC;
    dim3 gh((2 * pl.H + LINES - 1) / LINES, batch), gs((unsigned)sa.nsh * batch);
#define SGM2_SHEAR(TAILV, S4BV)                                                                           \
    do {                                                                                                  \
        sgm2_hscan_kernel<TAILV, true, LPC><<<gh, 256, 0, s>>>(a);                                        \
        sgm2_shear_kernel<false, TAILV, false, S4BV, LPC, false, NT><<<gs, NT + 64, 0, s>>>(sa);             \
        if (naturalOut) sgm2_shear_kernel<true, TAILV, false, S4BV, LPC, false, NT><<<gs, NT + 64, 0, s>>>(sa); \
        else sgm2_shear_kernel<true, TAILV, true, S4BV, LPC, false, NT><<<gs, NT + 64, 0, s>>>(sa);           \
    } while (0)
    // the down sweep's four-path sum stays below 256 for small P2: half the volume traffic
    const bool s4b = !naturalOut && 4 * (8 * pl.costWin * pl.costWin + pl.P2) <= 255;
    if (a.five) {
        // -> and <- from the horizontal scan, the three downward paths and the winner-take-all in ONE sweep
        sgm2_hscan_kernel<true, true, LPC><<<gh, 256, 0, s>>>(a);
        if (naturalOut) sgm2_shear_kernel<false, true, false, false, LPC, true, NT><<<gs, NT + 64, 0, s>>>(sa);
        else sgm2_shear_kernel<false, true, true, false, LPC, true, NT><<<gs, NT + 64, 0, s>>>(sa);
    } else if (a.tail) { if (s4b) SGM2_SHEAR(true, true); else SGM2_SHEAR(true, false); }
    else { if (s4b) SGM2_SHEAR(false, true); else SGM2_SHEAR(false, false); }
#undef SGM2_SHEAR
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}
}  // namespace

// S must hold H*W*md_sgm2_slots() u16 per pair; extra md_sgm2_extra_bytes() bytes
// naturalOut = 1: S8 is written to S in natural disparity order (volume export / separate WTA).
// naturalOut = 0: winner-take-all is fused into the up sweep and dDisp receives the final maps.
static void sgm2_fill_args(const md_sgm_plan& pl, const Sgm2Layout& l, const u8* cL, size_t strideL, const u8* cR, size_t strideR,
                           u16* S, size_t strideS, u8* extra, int batch, int naturalOut, Sgm2Args& a)
{
    a.cL = cL; a.pL = pl.pitchL; a.strideL = strideL;
    a.cR = cR; a.pR = pl.pitchR; a.strideR = strideR;
    a.S = S; a.strideS = strideS;
    a.H = pl.H; a.W = pl.W; a.D = pl.D; a.minD = pl.minD; a.maxD = pl.maxD;
    a.nstrips = l.nstrips;
    for (int i = 0; i < 4; i++) a.bnd[i] = (u16*)(extra + (size_t)i * l.oneB * batch);
    a.strideB = l.oneB / sizeof(u16);
    u8* hbase = extra + l.bndBytes * batch;
    a.Hvol[0] = hbase; a.Hvol[1] = hbase + l.oneH * batch;
    a.strideH = l.oneH;
    a.P1P1 = (u32)pl.P1 | ((u32)pl.P1 << 16);
    a.P2P2 = (u32)pl.P2 | ((u32)pl.P2 << 16);
    a.tail = pl.D < l.slots;
    a.naturalOut = naturalOut;
    a.five = pl.npaths == 5;
    a.four = pl.npaths == 4;
    a.byteSum = (pl.npaths == 5 ? 2 : 5) * (8 * pl.costWin * pl.costWin + pl.P2) <= 255;
    a.rec = (uint2*)(hbase + 2 * l.oneH * batch);
    a.strideRec = l.oneRec / sizeof(uint2);
    a.bias = pl.npaths * pl.P2; a.uniq = pl.uniq;
    a.smallS = ((long)pl.npaths * (8 * pl.costWin * pl.costWin + pl.P2) + 1) * l.dpl < 0x7C00;
    a.S2 = nullptr; a.splitUp = 0;
}

int md_sgm2_band_rows(const md_sgm_plan& pl)
{
    const Sgm2Layout l = sgm2_layout(pl, 1);
    return md_sgm3_lanes(l.lpc, l.dpl) / l.lpc;
}
size_t md_sgm2_boundary_bytes(const md_sgm_plan& pl)
{
    const Sgm2Layout l = sgm2_layout(pl, 1);
    return md_sgm3_boundary_bytes(pl.W, l.lpc, l.dpl, false);
}

// One sweep of the row-band schedule over the image rows [r0, r1) of one pair: the building block of a pair whose rows are
// spread over several devices.  `second` = false: the first sweep over these rows (stores the four-path sums in S);
// true: adds the other four paths, runs the winner-take-all and the left-right check, and writes rows [r0, r1) of dDisp.
// Either direction may come first.  bndIn: the states of the last row before the range in sweep order (the row r0 - 1 for a
// down sweep, r1 for an up sweep), as another call's bndOut delivered them; null when the range starts at the image
// edge.  bndOut (optional): receives the states of the range's last row in sweep order.
int md_sgm2_rows_sweep(hipStream_t s, const md_sgm_plan& pl, const u8* cL, const u8* cR, u16* S, u8* extra, int r0, int r1, bool up, bool second,
                       const u8* bndIn, u8* bndOut, int16_t* dDisp, int* errWord, bool* usedErrWord)
{
    const Sgm2Layout l = sgm2_layout(pl, 1);
    MD_REQUIRE(l.mode == SGM2_ROWS && pl.batch == 1, MDISP_E_UNSUPPORTED, "row-sharded SGM needs the row-band schedule (eight paths, byte path states) and one pair per call");
    const int nt = md_sgm3_lanes(l.lpc, l.dpl), rows = nt / l.lpc, nb = md_sgm3_bands(pl.H, l.lpc, nt);
    MD_REQUIRE(r0 >= 0 && r0 < r1 && r1 <= pl.H && r0 % rows == 0 && (r1 % rows == 0 || r1 == pl.H), MDISP_E_BADARG,
               "row range [%d, %d) must start at a multiple of %d rows and end at one or at the last row", r0, r1, rows);
    MD_REQUIRE(!second || dDisp, MDISP_E_BADARG, "the second sweep writes the disparity rows");
    Sgm2Args a;
    sgm2_fill_args(pl, l, cL, pl.pitchL * pl.H, cR, pl.pitchR * pl.H, S, pl.strideS, extra, 1, 0, a);
    int kb, ke;
    md_sgm3_band_range(pl.H, l.lpc, nt, up, r0, r1, &kb, &ke);
    MD_REQUIRE(kb == 0 || bndIn, MDISP_E_BADARG, "rows [%d, %d) do not start at the image edge in this direction: boundary states needed", r0, r1);
    const size_t bnd = md_sgm3_boundary_bytes(pl.W, l.lpc, l.dpl, false);
    const int ring = md_sgm3_ring();
    u8* hand = extra + (up ? l.bndBytes / 2 : 0);
    int* sync = (int*)(extra + l.bndBytes + 2 * l.oneH + l.oneRec);
    int* prog = sync + 16 + (up ? nb : 0);           // this direction's progress words (one pair)
    MD_HIP(hipMemsetAsync(sync + (up ? 1 : 0), 0, sizeof(int), s));
    MD_HIP(hipMemsetAsync(prog, 0, (size_t)nb * sizeof(int), s));
    if (kb > 0) {
        MD_HIP(hipMemcpyAsync(hand + (size_t)((kb - 1) & (ring - 1)) * bnd, bndIn, bnd, hipMemcpyDeviceToDevice, s));
        MD_HIP(hipMemsetD32Async((hipDeviceptr_t)(prog + kb - 1), pl.W, 1, s));     // "the row above is complete"
    }
    if (usedErrWord) *usedErrWord = true;
    const bool s4b = 4 * (8 * pl.costWin * pl.costWin + pl.P2) <= 255;
    MD_TRY(md_sgm3_sweep(s, a, hand, l.bndBytes, sync, errWord, 1, l.lpc, l.dpl, nt, s4b, up, second ? 1 : 0, kb, ke - kb, false, -1));
    if (bndOut && ke < nb) MD_HIP(hipMemcpyAsync(bndOut, hand + (size_t)((ke - 1) & (ring - 1)) * bnd, bnd, hipMemcpyDeviceToDevice, s));
    if (second) {
        LrArgs l2;
        l2.rec = a.rec; l2.strideRec = a.strideRec; l2.H = pl.H; l2.W = pl.W; l2.D = pl.D; l2.minD = pl.minD;
        l2.maxDiff = pl.maxDiff; l2.subpixel = pl.subpixel; l2.disp = dDisp; l2.strideDisp = (size_t)pl.H * pl.W; l2.y0 = r0;
        sgm2_lr_kernel<<<dim3(r1 - r0, 1), 256, (size_t)pl.W * 6 + 16, s>>>(l2);
        MD_HIP(hipGetLastError());
    }
    return MDISP_OK;
}

int md_sgm2_aggregate(hipStream_t s, const md_sgm_plan& pl, const u8* cL, size_t strideL, const u8* cR, size_t strideR,
                      u16* S, size_t strideS, u8* extra, int batch, int naturalOut, int16_t* dDisp, int* errWord, bool* usedErrWord,
                      const md_aux_streams* aux)
{
    Sgm2Args a;
    const Sgm2Layout l = sgm2_layout(pl, batch);
    sgm2_fill_args(pl, l, cL, strideL, cR, strideR, S, strideS, extra, batch, naturalOut, a);
    const bool u8m = l.u8m;
    u8* hbase = extra + l.bndBytes * batch;
    u8* afterRec = hbase + 2 * l.oneH * batch + l.oneRec * batch;
    if (l.mode == SGM2_SPLIT && !naturalOut && strideS * sizeof(u16) == l.oneS2) {
        // hscan || prepass, then down sweep || up sweep, then the summing winner-take-all
        a.S2 = (u16*)afterRec; a.splitUp = 1;
        hipStream_t s2 = aux ? aux->aux : s;
        dim3 gh((2 * pl.H + 15) / 16, batch), gp((4 * prepass_slots(pl.H, pl.W) + 15) / 16, batch), gs(a.nstrips, batch);
        if (aux) { MD_HIP(hipEventRecord(aux->ev[0], s)); MD_HIP(hipStreamWaitEvent(s2, aux->ev[0], 0)); }
#define SGM2_SPLIT_LAUNCH(TAILV)                                                                 \
        do {                                                                                     \
            sgm2_hscan_kernel<TAILV, true, 16><<<gh, 256, 0, s>>>(a);                            \
            sgm2_prepass_kernel<TAILV><<<gp, 256, 0, s2>>>(a);                                   \
            if (aux) {                                                                           \
                MD_HIP(hipEventRecord(aux->ev[1], s2)); MD_HIP(hipStreamWaitEvent(s, aux->ev[1], 0));   \
                MD_HIP(hipEventRecord(aux->ev[2], s)); MD_HIP(hipStreamWaitEvent(s2, aux->ev[2], 0));   \
            }                                                                                    \
            sgm2_sweep_kernel<false, TAILV, true, false><<<gs, 256, 0, s>>>(a);                  \
            sgm2_sweep_kernel<true, TAILV, true, false><<<gs, 256, 0, s2>>>(a);                  \
            if (aux) { MD_HIP(hipEventRecord(aux->ev[0], s2)); MD_HIP(hipStreamWaitEvent(s, aux->ev[0], 0)); }  \
            sgm2_wta_sum_kernel<TAILV><<<dim3(pl.H, batch), 256, 0, s>>>(a);                     \
        } while (0)
        // a few pairs leave most SIMDs with a single wave: spread the three paths of a sweep over three wave groups
        const bool pathWaves = batch <= path_waves_max_batch() && 4 * (8 * pl.costWin * pl.costWin + pl.P2) <= 255;
#define SGM2_SPLIT3_LAUNCH(TAILV)                                                                \
        do {                                                                                     \
            sgm2_hscan_kernel<TAILV, true, 16><<<gh, 256, 0, s>>>(a);                            \
            sgm2_prepass_kernel<TAILV><<<gp, 256, 0, s2>>>(a);                                   \
            if (aux) {                                                                           \
                MD_HIP(hipEventRecord(aux->ev[1], s2)); MD_HIP(hipStreamWaitEvent(s, aux->ev[1], 0));   \
                MD_HIP(hipEventRecord(aux->ev[2], s)); MD_HIP(hipStreamWaitEvent(s2, aux->ev[2], 0));   \
            }                                                                                    \
            sgm2_sweep3_kernel<false, TAILV><<<gs, 768, 0, s>>>(a);                              \
            sgm2_sweep3_kernel<true, TAILV><<<gs, 768, 0, s2>>>(a);                              \
            if (aux) { MD_HIP(hipEventRecord(aux->ev[0], s2)); MD_HIP(hipStreamWaitEvent(s, aux->ev[0], 0)); }  \
            sgm2_wta_sum_kernel<TAILV, true><<<dim3(pl.H, batch), 256, 0, s>>>(a);               \
        } while (0)
        if (pathWaves) { if (a.tail) SGM2_SPLIT3_LAUNCH(true); else SGM2_SPLIT3_LAUNCH(false); }
        else if (a.tail) SGM2_SPLIT_LAUNCH(true); else SGM2_SPLIT_LAUNCH(false);
#undef SGM2_SPLIT3_LAUNCH
#undef SGM2_SPLIT_LAUNCH
        MD_HIP(hipGetLastError());
    } else if (l.mode == SGM2_ROWS) {
        int* sync = (int*)afterRec;
        MD_HIP(hipMemsetAsync(sync, 0, l.syncBytes, s));
        if (usedErrWord) *usedErrWord = true;
        if (md_tune_take(MDISP_TUNE_SGM_PLANT_ERROR) > 0) {
            MD_HIP(hipMemsetAsync(errWord, 1, sizeof(int), s));
        }
        const bool s4b = !naturalOut && 4 * (8 * pl.costWin * pl.costWin + pl.P2) <= 255;
        const int nt = md_sgm3_lanes(l.lpc, l.dpl), rows = nt / l.lpc, nb = md_sgm3_bands(pl.H, l.lpc, nt);
        const int meetMax = md_tune(MDISP_TUNE_SGM_MEET_MAX) >= 0 ? md_tune(MDISP_TUNE_SGM_MEET_MAX) : 1 << 30;
        if (a.five) {
            // the four paths that run with the rows in one sweep, then the opposite horizontal path (no dependency between
            // rows) added to the stored sum, with the winner-take-all
            MD_REQUIRE(!naturalOut, MDISP_E_UNSUPPORTED, "five paths on the row bands: no volume export");
            MD_TRY(md_sgm3_sweep(s, a, extra, l.bndBytes, sync, errWord, batch, l.lpc, l.dpl, nt, s4b, false, 0, 0, -1, !l.u8m, 0));
            MD_TRY(md_sgm3_sweep(s, a, extra, l.bndBytes, sync, errWord, batch, l.lpc, l.dpl, nt, s4b, true, 3, 0, -1, !l.u8m, 1));
        } else if (batch <= meetMax && aux && !naturalOut && nb >= 4) {
            // The bands of a pair follow each other 2 * rows + 3 steps apart, so a sweep needs (bands of a pair in flight) x
            // that many steps to fill the GPU and as many to drain it: start from both image edges at once, which halves
            // the chain (64 pairs: 0.894 -> 0.877 ms per pair, 16 pairs: 1.51 -> 1.40, 8 pairs: 2.38 -> 1.73).  The down sweep runs first over the upper half while the up sweep runs first
            // over the lower half (each stores its four-path sum); then each continues into the other half as the second
            // sweep, starting from the states the other left in its hand-off ring -- the critical path of a pair is one
            // sweep over the image instead of two.
            const int hb = nb / 2, r0 = hb * rows;        // image rows [0, r0) and [r0, H)
            int kbd, ked, kbu, keu;
            md_sgm3_band_range(pl.H, l.lpc, nt, false, 0, r0, &kbd, &ked);        // down over the upper half: bands [0, ked)
            md_sgm3_band_range(pl.H, l.lpc, nt, true, r0, pl.H, &kbu, &keu);      // up over the lower half:   bands [0, keu)
            u8* handDown = extra; u8* handUp = extra + l.bndBytes / 2;
            // both directions and both roles in one launch: the GPU fills and drains once per call, not once per phase
            // (64 pairs: 0.797 -> 0.767 ms per pair, 16 pairs: 1.30 -> 1.15).  Very large batches amortise the phase change
            // anyway and run 1-3 % faster as two launches per phase (128 pairs: 0.720 against 0.730 ms per pair).
            const int oneLaunch = md_tune(MDISP_TUNE_SGM_ONE_LAUNCH);
            if ((oneLaunch > 0 || (oneLaunch < 0 && batch <= 96)) && md_sgm3_meet_supported(a, l.lpc, l.dpl, s4b, !l.u8m)) {
                MD_TRY(md_sgm3_meet(s, a, handDown, handUp, l.bndBytes, sync, errWord, batch, l.lpc, l.dpl, nt, ked, keu));
            } else {
            hipStream_t s2 = aux->aux;
            MD_HIP(hipEventRecord(aux->ev[0], s)); MD_HIP(hipStreamWaitEvent(s2, aux->ev[0], 0));
            MD_TRY(md_sgm3_sweep(s, a, handDown, l.bndBytes, sync, errWord, batch, l.lpc, l.dpl, nt, s4b, false, 0, 0, ked, !l.u8m, 0));
            MD_TRY(md_sgm3_sweep(s2, a, handUp, l.bndBytes, sync, errWord, batch, l.lpc, l.dpl, nt, s4b, true, 0, 0, keu, !l.u8m, 1));
            MD_HIP(hipEventRecord(aux->ev[1], s2)); MD_HIP(hipStreamWaitEvent(s, aux->ev[1], 0));
            MD_HIP(hipEventRecord(aux->ev[2], s)); MD_HIP(hipStreamWaitEvent(s2, aux->ev[2], 0));
            MD_TRY(md_sgm3_sweep(s, a, handUp, l.bndBytes, sync, errWord, batch, l.lpc, l.dpl, nt, s4b, true, 1, keu, nb - keu, !l.u8m, 2));
            MD_TRY(md_sgm3_sweep(s2, a, handDown, l.bndBytes, sync, errWord, batch, l.lpc, l.dpl, nt, s4b, false, 1, ked, nb - ked, !l.u8m, 3));
            MD_HIP(hipEventRecord(aux->ev[0], s2)); MD_HIP(hipStreamWaitEvent(s, aux->ev[0], 0));
            }
        } else {
            MD_TRY(md_sgm3_sweeps(s, a, extra, l.bndBytes, sync, errWord, batch, l.lpc, l.dpl, nt, naturalOut, s4b, !l.u8m));
        }
    } else if (l.mode == SGM2_SHEAR) {
        ShearArgs sa;
        sa.a = a;
        sa.hand = extra; sa.strideHand = l.bndBytes;
        sa.sync = (int*)afterRec;
        const int nt = shear_nt(pl, batch);
        sa.nsh = shear_strips(pl.H, pl.W, nt / l.lpc); sa.cmin = -(pl.H - 1); sa.batch = batch;
        MD_HIP(hipMemsetAsync(sa.sync, 0, l.syncBytes, s));
        // the error word belongs to the caller and is never cleared here: a timeout of an earlier call on the same
        // stream stays visible until the host reads it
        sa.err = errWord;
        if (usedErrWord) *usedErrWord = true;
        if (md_tune_take(MDISP_TUNE_SGM_PLANT_ERROR) > 0) {
            MD_HIP(hipMemsetAsync(errWord, 1, sizeof(int), s));
        }
        if (nt == 512) {
            if (l.lpc == 4) MD_TRY((launch_shear<4, 512>(s, pl, sa, batch, naturalOut)));
            else if (l.lpc == 8) MD_TRY((launch_shear<8, 512>(s, pl, sa, batch, naturalOut)));
            else MD_TRY((launch_shear<16, 512>(s, pl, sa, batch, naturalOut)));
        } else {
            if (l.lpc == 4) MD_TRY((launch_shear<4, 256>(s, pl, sa, batch, naturalOut)));
            else if (l.lpc == 8) MD_TRY((launch_shear<8, 256>(s, pl, sa, batch, naturalOut)));
            else MD_TRY((launch_shear<16, 256>(s, pl, sa, batch, naturalOut)));
        }
    } else {
    dim3 gh((2 * pl.H + 15) / 16, batch), gp((4 * prepass_slots(pl.H, pl.W) + 15) / 16, batch), gs(a.nstrips, batch);
#define SGM2_LAUNCH(TAILV, U8V)                                          \
    do {                                                                 \
        sgm2_hscan_kernel<TAILV, U8V, 16><<<gh, 256, 0, s>>>(a);         \
        sgm2_prepass_kernel<TAILV><<<gp, 256, 0, s>>>(a);                \
        sgm2_sweep_kernel<false, TAILV, U8V, false><<<gs, 256, 0, s>>>(a);                           \
        if (naturalOut) sgm2_sweep_kernel<true, TAILV, U8V, false><<<gs, 256, 0, s>>>(a);            \
        else sgm2_sweep_kernel<true, TAILV, U8V, true><<<gs, 256, 0, s>>>(a);                        \
    } while (0)
    if (a.tail) { if (u8m) SGM2_LAUNCH(true, true); else SGM2_LAUNCH(true, false); }
    else { if (u8m) SGM2_LAUNCH(false, true); else SGM2_LAUNCH(false, false); }
#undef SGM2_LAUNCH
    MD_HIP(hipGetLastError());
    }
    if (!naturalOut) {
        LrArgs l2;
        l2.rec = a.rec; l2.strideRec = a.strideRec; l2.H = pl.H; l2.W = pl.W; l2.D = pl.D; l2.minD = pl.minD;
        l2.maxDiff = pl.maxDiff; l2.subpixel = pl.subpixel; l2.disp = dDisp; l2.strideDisp = (size_t)pl.H * pl.W; l2.y0 = 0;
        sgm2_lr_kernel<<<dim3(pl.H, batch), 256, (size_t)pl.W * 6 + 16, s>>>(l2);
        MD_HIP(hipGetLastError());
    }
    return MDISP_OK;
}
