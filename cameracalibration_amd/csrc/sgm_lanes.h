// Device helpers shared by the SGM kernels (sgm2.hip, sgm3.hip): packed 16-bit path arithmetic on a "lane-row"
// (the LPC lanes that share one pixel, 16 disparities each), the census-Hamming cost, byte packing of path states
// and the fused winner-take-all.
//
// Lane mapping: each lane holds 16 consecutive disparities as 8 registers of packed u16 pairs,
//   R_r = [e = r | e = r + 8]          (r = 0..7, e = disparity index inside the lane)
// so the d-1 / d+1 neighbours of a whole register are the registers R_(r-1) and R_(r+1); only R_0 and R_7
// need one v_alignbit each to splice in the neighbouring lane.  Path states are kept normalised
// (N = L' - min L'), which turns the recursion into
//   L'(d) = C(d) + min(N(d), P2, P1 + min(N(d-1), N(d+1)))
// in packed 16-bit arithmetic; the minimum over d is a register tree plus DPP steps.
// Three-input minima use v_pk_minimum3_f16: bit patterns below 0x7C00 are non-negative finite halfs (subnormals
// included, which the default mode keeps) and order exactly like the integers they are, so every packed
// value here stays below 0x7C00 ("infinite" is 0x6000, P1 < 2048, sums < 4096).
#pragma once
#include "common.h"

struct Sgm2Args {
    const u8* cL; size_t pL, strideL;     // census of the left image      [B][H][pL]
    const u8* cR; size_t pR, strideR;     // census of the padded right    [B][H][pR]
    u16* S; size_t strideS;               // partial sums                  [B][H][W][256] (register order)
    u16* bnd[4]; size_t strideB;          // strip-border diagonal states  [B][H][nstrips][256]
    u8* Hvol[2]; size_t strideH;          // horizontal path volumes (1,0) and (-1,0): [B][H][W][256] u8 or u16
    int H, W, D, minD, maxD, nstrips;
    u32 P1P1, P2P2;
    int tail;                             // 1 when D < 256: disparities >= D are masked off
    int naturalOut;                       // last sweep: store the full sum in natural d order (volume export)
    int byteSum;                          // the two byte volumes a sweep adds first (8 paths: horizontal + four-path sum; 5 paths: both horizontals) sum to a byte
    int five;                             // 5-path mode (both horizontals + the three downward paths): the down sweep is the last one
    u16* S2; int splitUp;                 // split mode: the up sweep does not read S and stores its own four-path sum here
    uint2* rec; size_t strideRec;         // fused WTA output, one record per pixel [B][H][W]
    int bias, uniq;                       // npaths*P2 and the uniqueness ratio (percent)
    int four;                             // 4-path mode (the two horizontal and the two vertical paths): the row-band sweeps skip their diagonal paths
    int smallS;                           // every aggregated cost, times the disparities per lane, stays below 0x7C00 (row-band winner-take-all)
};
// WTA record: x = minS' | (best d << 16) | (not-unique << 31),  y = S'(best-1) | (S'(best+1) << 16)
// order of bnd[]: 0 (1,1)  1 (-1,1)  2 (1,-1)  3 (-1,-1)

namespace {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 asv(u32 v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ u32 asu(us2 v) { return __builtin_bit_cast(u32, v); }
// Sums and differences of packed pairs whose low halves neither carry nor borrow (every sum here stays below
// 0x7C00, every difference is value - minimum): plain 32-bit v_add_u32 / v_sub_u32, which issue in about 2.5
// cycles per wave64 where the packed 16-bit forms take about 4.4 (scripts/valu_ubench2.hip).
__device__ __forceinline__ u32 pkadd(u32 a, u32 b) { return a + b; }
__device__ __forceinline__ u32 pksub(u32 a, u32 b) { return a - b; }
__device__ __forceinline__ u32 pkmin(u32 a, u32 b) { return asu(__builtin_elementwise_min(asv(a), asv(b))); }

constexpr u32 INF1 = 0x6000u;       // "infinite": INF + 4095 + P1 stays below 0x7C00 (P1 < 2048)
constexpr u32 INF2 = INF1 | (INF1 << 16);
__device__ __forceinline__ u32 pkmin3(u32 a, u32 b, u32 c)
{
    u32 r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

#define DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
#define DPP_ROW_SHL1 0x101     // lane i <- lane i+1 (within a row of 16)
#define DPP_ROW_SHR1 0x111     // lane i <- lane i-1

template <int CTRL> __device__ __forceinline__ u32 dpp_all(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ u32 dpp_keep(u32 old, u32 v)
{ return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, 0xF, false); }

// minimum over the LPC (4, 8 or 16) aligned lanes that share a pixel, result in every one of them
template <int LPC = 16>
__device__ __forceinline__ u32 row_min16(u32 v)
{
    v = min(v, dpp_all<DPP_XOR1>(v));
    v = min(v, dpp_all<DPP_XOR2>(v));
    if (LPC >= 8) v = min(v, dpp_all<DPP_HALF_MIRROR>(v));
    if (LPC >= 16) v = min(v, dpp_all<DPP_MIRROR>(v));
    return v;
}

constexpr int NR = 8;          // packed registers per lane (16 disparities)
constexpr int DPL = 16;        // disparities per lane
constexpr int CW = 16;         // strip width in columns


// masks that pin disparities >= D of this lane to "infinite"
__device__ __forceinline__ void make_tail(int lic, int D, u32 (&tail)[NR])
{
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int dlo = lic * DPL + r, dhi = dlo + 8;
        tail[r] = (dlo >= D ? INF1 : 0u) | (dhi >= D ? INF1 << 16 : 0u);
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
// The same with the left census byte fetched as its aligned dword (rows start 256-byte aligned): a byte that is
// carried around a loop is kept as a byte by the compiler and re-extended right behind its load, which turns a request
// issued several steps ahead into a wait on the spot.  The byte is extracted with census_byte() where it is consumed.
__device__ __forceinline__ void load_cost_operands_w(const Sgm2Args& a, const u8* cLrow, const u8* cRrow, int x, int lic,
                                                     u32& clw, uint4& rb)
{
    clw = *reinterpret_cast<const u32*>(cLrow + (x & ~3));
    __builtin_memcpy(&rb, cRrow + x + a.maxD - a.minD - lic * DPL - 15, 16);
}
__device__ __forceinline__ u32 census_byte(u32 clw, int x) { return __builtin_amdgcn_ubfe(clw, 8 * (x & 3), 8); }

__device__ __forceinline__ void cost_from_operands(u32 cl, const uint4& rb, u32 (&C)[NR])
{
    const u32 m = cl * 0x01010101u;
    // dword 3 - k holds e = 4k .. 4k+3 as bytes 3,2,1,0
    const u32 w[4] = {rb.w ^ m, rb.z ^ m, rb.y ^ m, rb.x ^ m};
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const u32 lo = w[r >> 2], hi = w[(r >> 2) + 2];        // e = r and e = r + 8 sit at the same byte of their dwords
        const u32 mask = 0xFFu << (8 * (3 - (r & 3)));
        C[r] = (u32)__popc(lo & mask) | ((u32)__popc(hi & mask) << 16);
    }
}

// One step of one path for one pixel: consumes the predecessor's normalised state N, produces
// L' (returned in L) and leaves the new normalised state in N.
template <bool TAIL, int LPC = 16>
__device__ __forceinline__ void sgm_step(u32 (&N)[NR], const u32 (&C)[NR], u32 P1P1, u32 P2P2, const u32 (&tail)[NR], u32 (&L)[NR], int lic = 0)
{
    // d-1 of this lane's first disparity is the previous lane's e = 15 (high half of its R_7),
    // d+1 of its last disparity the next lane's e = 0 (low half of its R_0)
    u32 prev7 = dpp_keep<DPP_ROW_SHR1>(INF2, N[NR - 1]);
    u32 next0 = dpp_keep<DPP_ROW_SHL1>(INF2, N[0]);
    if (LPC < 16) {   // several pixels share a DPP row: cut the shifts at the pixel's first / last lane
        prev7 = lic == 0 ? INF2 : prev7;
        next0 = lic == LPC - 1 ? INF2 : next0;
    }
    u32 O[NR + 2];   // predecessor state framed by its d-1 / d+1 continuations: O[r] = N(d-1) of R_r, O[r+2] = N(d+1)
    O[0] = __builtin_amdgcn_alignbit(N[NR - 1], prev7, 16);     // [e = -1 | e = 7]
    O[NR + 1] = __builtin_amdgcn_alignbit(next0, N[0], 16);     // [e = 8  | e = 16]
#pragma unroll
    for (int r = 0; r < NR; r++) O[r + 1] = N[r];
    u32 mn = 0;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const u32 t = pkadd(pkmin(O[r], O[r + 2]), P1P1);
        u32 Lr = pkadd(C[r], pkmin3(O[r + 1], P2P2, t));
        L[r] = Lr;
        if (TAIL) Lr |= tail[r];
        N[r] = Lr;
        if (r & 1) mn = r == 1 ? pkmin(N[0], Lr) : pkmin3(mn, N[r - 1], Lr);
    }
    u32 m1 = min(mn & 0xFFFFu, mn >> 16);
    m1 = row_min16<LPC>(m1);
    u32 mm = m1 | (m1 << 16);
    // opaque to the compiler: it would otherwise fold the broadcast into the subtraction as v_mad_i32_i24(m1, -0x10001, N),
    // which issues at half the rate of a plain v_sub_u32
    asm("" : "+v"(mm));
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
// streaming variants: the volumes are written once and read once, far apart in time
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store8_stream(u16* p, const u32 (&v)[NR])
{
    u32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    __builtin_nontemporal_store(a, reinterpret_cast<u32x4*>(p));
    __builtin_nontemporal_store(b, reinterpret_cast<u32x4*>(p) + 1);
}
__device__ __forceinline__ void load8_stream(const u16* p, u32 (&v)[NR])
{
    const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + 1);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store16_stream(u8* p, const uint4& w)
{
    u32x4 a = {w.x, w.y, w.z, w.w};
    __builtin_nontemporal_store(a, reinterpret_cast<u32x4*>(p));
}
__device__ __forceinline__ uint4 load16_stream(const u8* p)
{
    const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(a.x, a.y, a.z, a.w);
}
__device__ __forceinline__ void zero8(u32 (&v)[NR])
{
#pragma unroll
    for (int r = 0; r < NR; r++) v[r] = 0;
}

// path values L' (<= Cmax + P2 <= 255) of one lane as 16 bytes (dword k: e = 2k, 2k+1, 2k+8, 2k+9), and back
__device__ __forceinline__ uint4 pack_bytes(const u32 (&L)[NR])
{
    uint4 w;
    w.x = __builtin_amdgcn_perm(L[1], L[0], 0x06020400u);   // [R_0.lo, R_1.lo, R_0.hi, R_1.hi] = e 0,1,8,9
    w.y = __builtin_amdgcn_perm(L[3], L[2], 0x06020400u);
    w.z = __builtin_amdgcn_perm(L[5], L[4], 0x06020400u);
    w.w = __builtin_amdgcn_perm(L[7], L[6], 0x06020400u);
    return w;
}
__device__ __forceinline__ void unpack_bytes(const uint4& w, u32 (&L)[NR])
{
    const u32 v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int g = 0; g < 4; g++) {
        L[2 * g] = v[g] & 0x00FF00FFu;                                   // bytes 0, 2 -> R_2g
        L[2 * g + 1] = __builtin_amdgcn_perm(v[g], v[g], 0x0C030C01u);   // bytes 1, 3 -> R_2g+1
    }
}

// Winner-take-all on the 16 aggregated costs each lane of a lane-row holds (register order): first
// minimum over d, the smallest cost outside {best-1, best, best+1} for the uniqueness test, and the
// two neighbours of the minimum for the parabola fit (through `srow`, this pixel's LPC*8 dwords
// of LDS).  The pixel's first lane writes the record.
template <bool TAIL, int LPC = 16>
__device__ __forceinline__ void fused_wta(const Sgm2Args& a, const u32 (&Sacc)[NR], const u32 (&tail)[NR], int lic, u32* srow,
                                          const u32 (*exclMask)[NR], size_t recIndex)
{
    // packed keys (S << 4 | local index), then (S, d) across the 16 lanes
    u32 kmin = 0xFFFFFFFFu;
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const u32 idx = (u32)q | ((u32)(q + 8) << 16);
        u32 k = asu(asv(Sacc[q]) * (unsigned short)16 + asv(idx));
        if (TAIL) k |= (tail[q] & 0x0000FFFFu ? 0x0000FFFFu : 0u) | (tail[q] & 0xFFFF0000u ? 0xFFFF0000u : 0u);
        kmin = q == 0 ? k : pkmin(kmin, k);
    }
    const u32 kl = min(kmin & 0xFFFFu, kmin >> 16);
    u32 key = ((kl >> 4) << 16) | (u32)(lic * DPL) | (kl & 15u);
    key = row_min16<LPC>(key);
    const int best = (int)(key & 0xFFFFu);
    const int minS = (int)(key >> 16);
    // smallest S outside {best-1, best, best+1}
    const int rel = best - lic * DPL;
    const int tsel = (unsigned)(rel + 1) <= 17u ? rel + 1 : 18;
    const uint4 m0 = *reinterpret_cast<const uint4*>(&exclMask[tsel][0]);
    const uint4 m1 = *reinterpret_cast<const uint4*>(&exclMask[tsel][4]);
    const u32 em[NR] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
    u32 smin = 0xFFFFFFFFu;
#pragma unroll
    for (int q = 0; q < NR; q++) {
        u32 v = Sacc[q] | em[q];
        if (TAIL) v |= (tail[q] & 0x0000FFFFu ? 0x0000FFFFu : 0u) | (tail[q] & 0xFFFF0000u ? 0xFFFF0000u : 0u);
        smin = q == 0 ? v : pkmin(smin, v);
    }
    const int mex = (int)row_min16<LPC>(min(smin & 0xFFFFu, smin >> 16));
    // uniqueness: some S(d) with |d - best| > 1 satisfies (S - bias)*(100 - u) < (minS - bias)*100
    const bool notUnique = mex != 0xFFFF && (mex - a.bias) * (100 - a.uniq) < (minS - a.bias) * 100;
    {
        uint4* dst = reinterpret_cast<uint4*>(&srow[lic * 8]);
        dst[0] = make_uint4(Sacc[0], Sacc[1], Sacc[2], Sacc[3]);
        dst[1] = make_uint4(Sacc[4], Sacc[5], Sacc[6], Sacc[7]);
    }
    auto s_at = [&](int d) -> u32 {
        d = min(max(d, 0), LPC * DPL - 1);
        const int e = d & 15;
        const int reg = e & 7, half = e >> 3;
        return (srow[(d >> 4) * 8 + reg] >> (16 * half)) & 0xFFFFu;
    };
    const u32 sm = s_at(best - 1), spl = s_at(best + 1);
    if (lic == 0) {
        uint2 rec;
        rec.x = (u32)minS | ((u32)best << 16) | (notUnique ? 0x80000000u : 0u);
        rec.y = sm | (spl << 16);
        a.rec[recIndex] = rec;
    }
}

// the 19 x NR table of masks fused_wta() uses: row t covers best - (lane base) = t - 1
__device__ __forceinline__ void init_excl_mask(u32 (*exclMask)[NR])
{
    for (int i = threadIdx.x; i < 19 * NR; i += blockDim.x) {
        const int t = i / NR, r = i % NR, rel = t - 1;
        const int elo = r, ehi = r + 8;
        u32 m = 0;
        if (t < 18) {
            if (elo - rel >= -1 && elo - rel <= 1) m |= 0x0000FFFFu;
            if (ehi - rel >= -1 && ehi - rel <= 1) m |= 0xFFFF0000u;
        }
        exclMask[t][r] = m;
    }
}

// ---- hand-off of path states between workgroups of one launch ----
// Every handed-off byte is written and read with agent-scope 8-byte atomics (write-through stores, cache-bypassing
// loads); the producer drains its stores (s_waitcnt vmcnt(0)) before it raises its progress word, the consumer polls
// that word and only then loads.  The compiler cannot move the accesses across those points: the drain is an asm
// statement with a memory clobber, the poll loop ends in a signal fence.
// a hand-off wait gives up after this many ticks of the 100 MHz real-time counter (4 s); it is read every 64 polls
constexpr unsigned long long HANDOFF_TIMEOUT_TICKS = 400000000ull;
__device__ __forceinline__ void hand_store(u8* p, const uint4& v)
{
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    __hip_atomic_store(q, (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, (unsigned long long)v.z | ((unsigned long long)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint4 hand_load(const u8* p)
{
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
}

// The same hand-off with ONE 16-byte write-through store / cache-bypassing load per lane (buffer instructions with the
// sc1 bit; `rsrc` describes the pair's hand-off rows, `off` is the byte offset inside them): half the memory
// instructions and fabric writes of the two 8-byte atomics above.
typedef u32 md_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t hand_rsrc(u8* base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void hand_store16(__amdgpu_buffer_rsrc_t rsrc, u32 off, const uint4& v)
{
    md_v4u d = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)off, 0, 16);
}
__device__ __forceinline__ uint4 hand_load16(__amdgpu_buffer_rsrc_t rsrc, u32 off)
{
    const md_v4u d = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 16);
    return make_uint4(d.x, d.y, d.z, d.w);
}

}  // namespace
