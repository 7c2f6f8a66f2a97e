// The lane helpers of sgm_lanes.h generalised to NRT packed registers per lane (DPL = 2 * NRT disparities per lane:
// 16 with NRT = 8, 32 with NRT = 16), for the row-band kernels of sgm3.hip.
//
// Register r of a lane holds the pair [e = r | e = r + NRT] of the lane's DPL disparities, so the d-1 / d+1 neighbours
// of a whole register are still the registers r-1 and r+1 and one step still needs only two DPP moves and two
// v_alignbit to splice in the neighbouring lanes -- per 32 disparities instead of per 16 when NRT = 16, and likewise
// for the minimum over d (one register tree, one DPP ladder per lane-row).  A lane's state travels as 2 * NRT bytes:
// dword k = e 2k, 2k+1, 2k+NRT, 2k+1+NRT.
#pragma once
#include "sgm_lanes.h"

namespace {

template <int NRT>
__device__ __forceinline__ void make_tail_n(int lic, int D, u32 (&tail)[NRT])
{
#pragma unroll
    for (int r = 0; r < NRT; r++) {
        const int dlo = lic * 2 * NRT + r, dhi = dlo + NRT;
        tail[r] = (dlo >= D ? INF1 : 0u) | (dhi >= D ? INF1 << 16 : 0u);
    }
}

// matching cost of one pixel for this lane's 2*NRT disparities, in register order.  rbw: the 2*NRT bytes of the padded
// right census that end at the lane's first disparity, as dwords at ascending addresses (= descending disparities:
// byte j <-> e = 2*NRT - 1 - j); cl: the left census byte.
template <int NRT>
__device__ __forceinline__ void cost_n(u32 cl, const u32 (&rbw)[NRT / 2], u32 (&C)[NRT])
{
    const u32 m = __builtin_amdgcn_perm(cl, cl, 0u);           // the byte in all four positions
    u32 w[NRT / 2];                                            // w[k] holds e = 4k .. 4k+3 as bytes 3,2,1,0
#pragma unroll
    for (int k = 0; k < NRT / 2; k++) w[k] = rbw[NRT / 2 - 1 - k] ^ m;
#pragma unroll
    for (int r = 0; r < NRT; r++) {
        const u32 lo = w[r >> 2], hi = w[(r >> 2) + NRT / 4];  // e = r and e = r + NRT sit at the same byte of their dwords
        const u32 mask = 0xFFu << (8 * (3 - (r & 3)));
        C[r] = (u32)__popc(lo & mask) | ((u32)__popc(hi & mask) << 16);
    }
}

template <int NRT>
__device__ __forceinline__ void pack_n(const u32 (&L)[NRT], u32 (&w)[NRT / 2])
{
#pragma unroll
    for (int k = 0; k < NRT / 2; k++) w[k] = __builtin_amdgcn_perm(L[2 * k + 1], L[2 * k], 0x06020400u);
}
template <int NRT>
__device__ __forceinline__ void unpack_n(const u32 (&w)[NRT / 2], u32 (&L)[NRT])
{
#pragma unroll
    for (int k = 0; k < NRT / 2; k++) {
        L[2 * k] = w[k] & 0x00FF00FFu;
        L[2 * k + 1] = __builtin_amdgcn_perm(w[k], w[k], 0x0C030C01u);
    }
}

// L'(d) = C(d) + min(N(d), P2, P1 + min(N(d-1), N(d+1))) for the lane's registers, and the minimum of L' over the
// pixel's disparities (disparities masked by `tail` excluded) broadcast to both halves of the returned word
// `edge` holds two registers that the row shifts write into.  A shift leaves the lanes without a source lane (the first /
// last lane of a DPP row) untouched, so once those lanes hold "infinite" they keep it from call to call and the
// registers never have to be re-initialised: the caller sets both to INF2 once and passes them to every step.
struct EdgeRegs { u32 prev, next; };
template <bool TAIL, int LPC, int NRT>
__device__ __forceinline__ u32 path_values_n(const u32 (&N)[NRT], const u32 (&C)[NRT], u32 P1P1, u32 P2P2, const u32 (&tail)[NRT],
                                             u32 (&L)[NRT], int lic, EdgeRegs& edge)
{
    edge.prev = dpp_keep<DPP_ROW_SHR1>(edge.prev, N[NRT - 1]); // the previous lane's last disparity (high half of its last register)
    edge.next = dpp_keep<DPP_ROW_SHL1>(edge.next, N[0]);       // the next lane's first disparity (low half of its first register)
    u32 prevL = edge.prev, nextF = edge.next;
    if (LPC < 16) {   // several pixels share a DPP row: cut the shifts at the pixel's first / last lane
        prevL = lic == 0 ? INF2 : prevL;
        nextF = lic == LPC - 1 ? INF2 : nextF;
    }
    u32 O[NRT + 2];
    O[0] = __builtin_amdgcn_alignbit(N[NRT - 1], prevL, 16);   // [e = -1 | e = NRT - 1]
    O[NRT + 1] = __builtin_amdgcn_alignbit(nextF, N[0], 16);   // [e = NRT | e = 2 NRT]
#pragma unroll
    for (int r = 0; r < NRT; r++) O[r + 1] = N[r];
    u32 mn = 0, held = 0;
#pragma unroll
    for (int r = 0; r < NRT; r++) {
        const u32 t = pkadd(pkmin(O[r], O[r + 2]), P1P1);
        const u32 Lr = pkadd(C[r], pkmin3(O[r + 1], P2P2, t));
        L[r] = Lr;
        const u32 Lm = TAIL ? Lr | tail[r] : Lr;
        if (r & 1) mn = r == 1 ? pkmin(held, Lm) : pkmin3(mn, held, Lm);
        else held = Lm;
    }
    u32 m1 = min(mn & 0xFFFFu, mn >> 16);
    m1 = row_min16<LPC>(m1);
    u32 mm = m1 | (m1 << 16);
    asm("" : "+v"(mm));       // keeps the broadcast out of the subtractions that follow (a v_mad_i32_i24 issues at half the rate of v_sub_u32)
    return mm;
}

// One step of a path whose state stays in registers: N <- L' - min L' (masked disparities stay "infinite")
template <bool TAIL, int LPC, int NRT>
__device__ __forceinline__ void step_n(u32 (&N)[NRT], const u32 (&C)[NRT], u32 P1P1, u32 P2P2, const u32 (&tail)[NRT], u32 (&L)[NRT], int lic,
                                       EdgeRegs& edge)
{
    const u32 mm = path_values_n<TAIL, LPC, NRT>(N, C, P1P1, P2P2, tail, L, lic, edge);
#pragma unroll
    for (int r = 0; r < NRT; r++) N[r] = pksub(TAIL ? L[r] | tail[r] : L[r], mm);
}

// The same step for a path without masked disparities whose state is kept as L' itself, together with oo = min L'
// (broadcast to both halves): L'(d) = C(d) + min(L(d), oo + P2, P1 + min(L(d-1), L(d+1))) - oo.  Every operand of the
// minimum is at least oo in both halves, so |m - oo| + C is one v_sad_u32 and nothing is subtracted afterwards.
template <int LPC, int NRT>
__device__ __forceinline__ void step_lazy_n(u32 (&N)[NRT], u32& oo, const u32 (&C)[NRT], u32 P1P1, u32 P2P2, u32 (&L)[NRT], int lic,
                                            EdgeRegs& edge)
{
    edge.prev = dpp_keep<DPP_ROW_SHR1>(edge.prev, N[NRT - 1]);
    edge.next = dpp_keep<DPP_ROW_SHL1>(edge.next, N[0]);
    u32 prevL = edge.prev, nextF = edge.next;
    if (LPC < 16) {
        prevL = lic == 0 ? INF2 : prevL;
        nextF = lic == LPC - 1 ? INF2 : nextF;
    }
    u32 O[NRT + 2];
    O[0] = __builtin_amdgcn_alignbit(N[NRT - 1], prevL, 16);
    O[NRT + 1] = __builtin_amdgcn_alignbit(nextF, N[0], 16);
#pragma unroll
    for (int r = 0; r < NRT; r++) O[r + 1] = N[r];
    const u32 P2o = pkadd(P2P2, oo);
    u32 mn = 0, held = 0;
#pragma unroll
    for (int r = 0; r < NRT; r++) {
        const u32 t = pkadd(pkmin(O[r], O[r + 2]), P1P1);
        const u32 m = pkmin3(O[r + 1], P2o, t);
        u32 Lr;
        asm("v_sad_u32 %0, %1, %2, %3" : "=v"(Lr) : "v"(m), "v"(oo), "v"(C[r]));
        L[r] = Lr;
        N[r] = Lr;
        if (r & 1) mn = r == 1 ? pkmin(held, Lr) : pkmin3(mn, held, Lr);
        else held = Lr;
    }
    u32 m1 = min(mn & 0xFFFFu, mn >> 16);
    m1 = row_min16<LPC>(m1);
    oo = m1 | (m1 << 16);
}

// One step of a path whose state is handed on as bytes: consumes the unpacked predecessor state N, returns L' and the
// packed normalised state.  Without masked disparities the minimum is subtracted from the packed bytes (every byte is
// at least the minimum, so no borrow crosses a byte): half the subtractions.
template <bool TAIL, int LPC, int NRT>
__device__ __forceinline__ void step_pack_n(u32 (&N)[NRT], const u32 (&C)[NRT], u32 P1P1, u32 P2P2, const u32 (&tail)[NRT], u32 (&L)[NRT],
                                            u32 (&outw)[NRT / 2], int lic, EdgeRegs& edge)
{
    const u32 mm = path_values_n<TAIL, LPC, NRT>(N, C, P1P1, P2P2, tail, L, lic, edge);
    if (TAIL) {
#pragma unroll
        for (int r = 0; r < NRT; r++) N[r] = pksub(L[r] | tail[r], mm);
        pack_n<NRT>(N, outw);
    } else {
        pack_n<NRT>(L, outw);
        u32 mb = mm | (mm << 8);
        asm("" : "+v"(mb));
#pragma unroll
        for (int k = 0; k < NRT / 2; k++) outw[k] -= mb;
    }
}

// the (DPL + 3) x NRT table of masks wta_n() uses: row t marks the disparities best-1, best, best+1 of a lane whose
// first disparity is best - (t - 1); the last row is empty
template <int NRT>
__device__ __forceinline__ void init_excl_mask_n(u32 (*exclMask)[NRT])
{
    constexpr int DPLn = 2 * NRT;
    for (int i = threadIdx.x; i < (DPLn + 3) * NRT; i += blockDim.x) {
        const int t = i / NRT, r = i % NRT, rel = t - 1;
        const int elo = r, ehi = r + NRT;
        u32 m = 0;
        if (t < DPLn + 2) {
            if (elo - rel >= -1 && elo - rel <= 1) m |= 0x0000FFFFu;
            if (ehi - rel >= -1 && ehi - rel <= 1) m |= 0xFFFF0000u;
        }
        exclMask[t][r] = m;
    }
}

// Winner-take-all on the aggregated costs of one pixel (register order, NRT registers in each of its LPC lanes): first
// minimum over d, the smallest cost outside {best-1, best, best+1} for the uniqueness test, and the two neighbours of
// the minimum for the parabola fit (through `srow`, this pixel's LPC*NRT dwords of LDS).  The pixel's first lane writes
// the record.  Costs must stay below 65536 / (2 NRT).
template <bool TAIL, int LPC, int NRT>
__device__ __forceinline__ void wta_n(const Sgm2Args& a, const u32 (&Sacc)[NRT], const u32 (&tail)[NRT], int lic, u32* srow,
                                      const u32 (*exclMask)[NRT], uint2* recOut, bool store = true)
{
    constexpr int DPLn = 2 * NRT, SH = NRT == 8 ? 4 : 5;
    // keys (S * DPL + local index) as packed pairs; while every key stays below 0x7C00 (a.smallS) their minimum is a
    // tree of packed three-input minima, and "excluded" is the bit pattern 0x7000 instead of 0xFFFF
    const bool small = !TAIL && a.smallS != 0;       // (masked disparities are OR-ed out, which could leave the ordered range)
    const u32 OUT1 = small ? 0x7000u : 0xFFFFu;
    u32 k[NRT];
#pragma unroll
    for (int q = 0; q < NRT; q++) {
        const u32 idx = (u32)q | ((u32)(q + NRT) << 16);
        asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(k[q]) : "v"(Sacc[q]), "v"((u32)DPLn | ((u32)DPLn << 16)), "v"(idx));
        if (TAIL) k[q] |= (tail[q] & 0x0000FFFFu ? OUT1 : 0u) | (tail[q] & 0xFFFF0000u ? OUT1 << 16 : 0u);
    }
    u32 kmin;
    if (small) {
        kmin = k[0];
#pragma unroll
        for (int q = 1; q + 1 < NRT; q += 2) kmin = pkmin3(kmin, k[q], k[q + 1]);
        kmin = pkmin(kmin, k[NRT - 1]);
    } else {
        kmin = k[0];
#pragma unroll
        for (int q = 1; q < NRT; q++) kmin = pkmin(kmin, k[q]);
    }
    const u32 kl = min(kmin & 0xFFFFu, kmin >> 16);
    u32 key = ((kl >> SH) << 16) | (u32)(lic * DPLn) | (kl & (u32)(DPLn - 1));
    key = row_min16<LPC>(key);
    const int best = (int)(key & 0xFFFFu);
    const int minS = (int)(key >> 16);
    const int rel = best - lic * DPLn;
    const int tsel = (unsigned)(rel + 1) <= (unsigned)(DPLn + 1) ? rel + 1 : DPLn + 2;
    const u32 OUT2 = OUT1 | (OUT1 << 16);
    u32 v[NRT];
#pragma unroll
    for (int q = 0; q < NRT; q++) {
        v[q] = Sacc[q] | (exclMask[tsel][q] & OUT2);
        if (TAIL) v[q] |= (tail[q] & 0x0000FFFFu ? OUT1 : 0u) | (tail[q] & 0xFFFF0000u ? OUT1 << 16 : 0u);
    }
    u32 smin;
    if (small) {
        smin = v[0];
#pragma unroll
        for (int q = 1; q + 1 < NRT; q += 2) smin = pkmin3(smin, v[q], v[q + 1]);
        smin = pkmin(smin, v[NRT - 1]);
    } else {
        smin = v[0];
#pragma unroll
        for (int q = 1; q < NRT; q++) smin = pkmin(smin, v[q]);
    }
    const int mex = (int)row_min16<LPC>(min(smin & 0xFFFFu, smin >> 16));
    // uniqueness: some S(d) with |d - best| > 1 satisfies (S - bias)*(100 - u) < (minS - bias)*100
    const bool notUnique = mex < (int)OUT1 && __mul24(mex - a.bias, 100 - a.uniq) < __mul24(minS - a.bias, 100);
    // the neighbours of the minimum for the parabola: the pixel's sums go to its LDS row in natural disparity order (16-bit
    // element d of the row is S(d)), so that the two look-ups are plain 16-bit reads at 2 (best -+ 1), clamped to the range
    u32 nat[NRT];
#pragma unroll
    for (int g = 0; g < NRT / 2; g++) {
        nat[g] = __builtin_amdgcn_perm(Sacc[2 * g + 1], Sacc[2 * g], 0x05040100u);
        nat[g + NRT / 2] = __builtin_amdgcn_perm(Sacc[2 * g + 1], Sacc[2 * g], 0x07060302u);
    }
#pragma unroll
    for (int q = 0; q < NRT; q += 4)
        *reinterpret_cast<uint4*>(&srow[lic * NRT + q]) = make_uint4(nat[q], nat[q + 1], nat[q + 2], nat[q + 3]);
    const unsigned short* s16row = reinterpret_cast<const unsigned short*>(srow);
    const u32 sm = s16row[max(best - 1, 0)], spl = s16row[min(best + 1, LPC * DPLn - 1)];
    if (lic == 0 && store) {
        uint2 rec;
        rec.x = (u32)minS | ((u32)best << 16) | (notUnique ? 0x80000000u : 0u);
        rec.y = sm | (spl << 16);
        *recOut = rec;
    }
}

}  // namespace
