// Row-band semi-global matching for gfx950: FOUR paths per sweep, two sweeps, no horizontal volumes.
//
//   down sweep:  -> (1,0)   \ (1,1)    v (0,1)    / (-1,1)      -> four-path partial sum S4 (one byte or u16 per slot)
//   up sweep:    <- (-1,0)  (-1,-1)    ^ (0,-1)   (1,-1)        -> S4 + four more = S8 in registers -> winner-take-all
//
// Both sweeps are the same program in "logical" coordinates (u, v): the up sweep mirrors both axes
// (x = W-1-u, y = H-1-v), which turns its four directions into the down sweep's.  In logical coordinates a pixel
// (u, v) depends on (u-1, v), (u-1, v-1), (u, v-1) and (u+1, v-1).  With the time skew t = u + 2v all four
// predecessors are one to three steps old, so a workgroup owns a BAND of image rows (one row per "slot": the LPC lanes
// that share a pixel), every slot marches along its row, and slot j runs two pixels behind slot j-1:
//   ->   state stays in the slot's registers,
//   /    needs the state of (u+1, v-1): what slot j-1 produced in the previous step,
//   v    needs (u, v-1):   produced two steps ago,
//   \    needs (u-1, v-1): produced three steps ago.
// Each step a slot publishes the three states of its pixel in LDS (bytes, double-buffered by step parity) and the slot
// below reads all three one step later: it uses the / state at once and keeps the other two in registers for one and
// two more steps.  The matching cost is evaluated once per pixel and sweep and shared by the four paths; nothing but
// the four-path sum (and, per band, the last row's states) ever reaches memory.
//
// Bands form a pipeline along the sweep: the last slot of band b hands its three states per pixel to slot 0 of band
// b+1 through global memory, and band b+1 follows about 2*SLOTS steps behind.  The protocol is the one of the sheared
// strips in sgm2.hip: workgroups take a ticket at start, tickets enumerate bands in dependency order (all pairs of band
// b before band b+1), so a workgroup only waits for one that started earlier; an extra "communication" wave per
// workgroup does the exchange (agent-scope write-through stores, drain, then the progress word; agent-scope loads on
// the consumer side, parked in LDS one step ahead); every wait is bounded and reports through a sticky error word.
//
// Launches: one kernel per sweep direction and role over a range of bands (sgm3_rows_kernel: large batches, the row-sharded
// entry point, geometries other than 16 lanes x 16 disparities), or -- batches of up to 96 pairs of the headline
// geometry -- ONE kernel for both directions and both roles (sgm3_meet_kernel), in which the four-path sums pass from a
// first-sweep workgroup to a second-sweep workgroup of the same launch.  Both share the sweep body sgm3_rows_body.
#include "sgm_lanes_n.h"
#include <algorithm>
#include <type_traits>

namespace {

// Waves per SIMD the register allocation aims at.  A workgroup is five waves on four SIMDs, so 5 waves per SIMD are four
// workgroups per CU and 4 are three.  The kernels with 16 disparities per lane, byte sums and no masked disparities (the
// headline geometry among them) fit 96 registers without spilling; the others keep the larger budget.
#ifndef SGM3_WPS0
#define SGM3_WPS0 5
#endif
constexpr int sgm3_waves_per_simd(int lpc, int nrt, bool tail, bool s4b, bool s16, int role)
{ return (lpc == 16 && nrt == 16) || s16 ? 2 : (nrt == 8 && !tail && s4b ? (role == 0 ? SGM3_WPS0 : 5) : 4); }

// compute lanes of a workgroup at 16 lanes x 16 disparities per pixel
#ifndef SGM3_NT16
#define SGM3_NT16 256
#endif
struct RowArgs {
    Sgm2Args a;
    u8* hand;               // [B][ring][W][3][state bytes]: states (v, \, /) of the last row of a band, per pixel; band b uses entry b & ringMask
    size_t strideHand;      // bytes per pair
    int* sync;              // [0] ticket (down) [1] ticket (up), [16 + (sweep*B + pair)*nb + band] pixels published by the band's last row
    int* err;               // sticky error word (owned by the caller)
    int nb, batch;          // bands per image, pairs per launch
    int band0;              // first band this launch runs (its predecessor's states and progress word are already in place)
    int nbands;             // bands this launch runs
    int group;              // pairs dispatched together: tickets run band-major inside a group of pairs, group after group
    int ringMask;           // entries of the hand-off ring - 1 (a power of two: an entry is rewritten only by a band that started after its reader finished with it)
    int ticket;             // which of the ticket words in sync[0..15] this launch counts on
    int slack;              // pixels a band stays behind its predecessor beyond the two it needs
    int vofs;               // logical rows of band 0 that lie before the image (so that band edges fall on the same image rows in both directions)
    // the one-launch schedule (both directions, both roles; sgm3_meet_kernel): the mirrored sweep's ring and padding, the
    // bands each direction sweeps first, and one word per direction, pair and band that is set when the band's four-path
    // sums are in memory
    u8* handUp; int vofsUp; int firstDown, firstUp; int* done;
#ifdef SGM3_TRACE
    int traceIdx;
#endif
};
#ifdef SGM3_TRACE
#define SGM3_TRACE_MAX (1 << 16)
__device__ unsigned long long g_trace[SGM3_TRACE_MAX][6];      // per ticket: start, ready, end, ticks the communication wave waited (100 MHz clock); shader clock at start and end
#endif

// LPC lanes share a pixel, each with NRT packed registers (2 NRT disparities); NT compute lanes per workgroup.
// Byte images of a pixel (LDS states, hand-off rows, byte four-path sums) are stored as NRT/8 planes of LPC*16 bytes:
// plane p holds dwords 4p .. 4p+3 of every lane's packed state, so that one 16-byte access per lane is contiguous
// across the pixel's lanes.
// UP: the sweep's direction (false: top to bottom, left to right; true: mirrored).  ROLE: 0 = the first sweep over the
// rows (stores the four-path sum), 1 = the second sweep (adds the stored sum, runs the winner-take-all), 2 = the
// second sweep with the full sum exported in natural disparity order, 3 = the last sweep of the five-path mode: only
// the path along the row (no dependency between rows: no hand-off, no barriers), added to the stored four-path sum,
// then the winner-take-all.  Either direction can be the first sweep.
// S16: the three states that leave the registers travel as 16-bit values (path values that do not fit a byte: 8 + P2 >
// 255) -- twice the LDS and hand-off bytes, no pack / unpack.
// FOUR: the 4-path mode (horizontal + vertical paths only): the two diagonal steps are left out.
// COH: the four-path sums are written by one workgroup and read by another of the SAME launch (one-launch schedule), so
// they travel with write-through stores and cache-bypassing loads like the hand-off rows, and a first-sweep band marks
// itself done when all its stores have completed.
constexpr int sgm3_state_dwords(int lpc, int nrt, bool s16) { return (s16 ? nrt : nrt / 2) * lpc; }
// One wait state between a 16-byte buffer store whose scalar offset is a register and the next vector instruction that
// overwrites one of its data registers.  The compiler leaves none for this form of the store (it does for an immediate
// scalar offset), and without it the stored dword can be the NEW register contents.  Measured with a 16-byte store of
// packed matching costs followed directly by the computation of the next column index into its first data register:
// the first dword of lanes 12-15 of every sixteen arrived in memory as that column index, on a loaded GPU only (nine or
// more full-size pairs in flight, never with four).  The data registers are operands of the statement, so they stay
// allocated up to it.  (Also tried with that store: keeping the matching costs of the first sweep in memory for the
// second one, 35 vector instructions fewer per second-sweep step -- bit-exact and SLOWER, 0.83 against 0.75 ms per
// pair at 64 pairs: the extra byte per candidate written and read takes the kernel to the memory system's limit.)
#define STORE_DATA_GUARD(d) asm volatile("s_nop 0" ::"v"((d).x), "v"((d).y), "v"((d).z), "v"((d).w) : "memory")
template <bool UP, int ROLE, bool TAIL, bool S4B, int LPC, int NRT, int NT, bool S16, bool FOUR, bool COH>
__device__ __forceinline__ void sgm3_rows_body(const RowArgs& ra, const int band, const int pair,
                                               u32 (*st)[NT / LPC + 1][3][sgm3_state_dwords(LPC, NRT, S16)], u32 (*srow)[LPC * NRT],
                                               const u32 (*exclMask)[NRT])
{
    constexpr bool SECOND = ROLE != 0, FUSE = ROLE == 1 || ROLE == 3, HONLY = ROLE == 3;
    constexpr int SLOTS = NT / LPC, DPLn = 2 * NRT, DP = LPC * DPLn, PL = NRT / 8, PW = LPC * 4, NW = NRT / 2;
    constexpr int SWN = S16 ? NRT : NRT / 2;      // dwords of one path state per lane
    constexpr int PLS = SWN / 4;                  // ... as 16-byte planes
    constexpr int SW = PLS * PW;                  // dwords of one path state per pixel
    constexpr int DPB = SW * 4;                   // bytes of one path state per pixel
    static_assert(!(S16 && S4B), "16-bit path states never have byte sums");
    static_assert(SW == sgm3_state_dwords(LPC, NRT, S16), "state size");
    static_assert(!COH || (S4B && (ROLE == 0 || ROLE == 1)), "one-launch schedule: byte sums, first and second sweep only");
    const Sgm2Args& a = ra.a;
    // a first-sweep band of the one-launch schedule ends here: every store of the workgroup has completed, then its word is set
    auto finish = [&]() {
        if constexpr (COH && ROLE == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_store(ra.done + ((size_t)((UP ? ra.batch : 0) + pair) * ra.nb + band), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    const int H = a.H, W = a.W;
    const int T = W + 2 * (SLOTS - 1);                    // steps: slot j works on u = t - 2j
    u8* hand = ra.hand + (size_t)pair * ra.strideHand;
    const bool hasPrev = band > 0, hasNext = band + 1 < ra.nb;
    auto lds_get = [&](const u32* p, u32 (&w)[SWN]) {      // p: the lane's first dword of plane 0
#pragma unroll
        for (int q = 0; q < PLS; q++) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + q * PW);
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    };
    auto lds_put = [&](u32* p, const u32 (&w)[SWN]) {
#pragma unroll
        for (int q = 0; q < PLS; q++) *reinterpret_cast<uint4*>(p + q * PW) = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    };

    if (threadIdx.x >= NT) {
        if (HONLY) return;      // rows are independent: nothing to exchange
#ifdef SGM3_COMM_PRIO
        __builtin_amdgcn_s_setprio(SGM3_COMM_PRIO);
#endif
        // ---- communication wave: publishes the last slot's states one step behind the compute waves and fetches
        // the states slot 0 will consume one step ahead of them
        const int lane = threadIdx.x - NT, which = lane / LPC, lic = lane % LPC;
        int* progMe = ra.sync + 16 + ((size_t)((UP ? ra.batch : 0) + pair) * ra.nb + band);
        const int* progNb = progMe - 1;
        int* errWord = ra.err;
        int seen = 0;
        const __amdgpu_buffer_rsrc_t rsrc = hand_rsrc(hand, ra.strideHand);      // the pair's rows: below 4 GB
        auto hand_slot = [&](int bnd, int u, int wh) -> u32 { return (u32)((((size_t)(bnd & ra.ringMask) * W + u) * 3 + wh) * DPB + lic * 16); };
        // slot 0 consumes in step i (i >= -1) the states of pixel i + 1 of the row above
        auto wants = [&](int i) -> bool { return hasPrev && which < 3 && i + 1 < W; };
        auto request = [&](int i, u32 (&w)[SWN]) {
            const bool mine = wants(i);
            if (!__any(mine)) return;
            const int need = min(i + 2 + ra.slack, W);    // pixels the band above must have published (two, plus the slack that decouples the bands)
#ifdef SGM3_KNOCK_NO_BAND_WAIT      // timing experiment only (results wrong): bands never wait for the band above
            seen = W;
#endif
            if (seen < need) {
                unsigned long long t0 = 0;
#ifdef SGM3_TRACE
                const unsigned long long tw0 = __builtin_amdgcn_s_memrealtime();
#endif
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
                    // back off: a band that waits for its predecessor's head start waits for tens of steps, and every
                    // poll is a memory-side access next to the streams of the working bands
                    if (it < 4) __builtin_amdgcn_s_sleep(2);
                    else if (it < 16) __builtin_amdgcn_s_sleep(16);
                    else __builtin_amdgcn_s_sleep(64);
                }
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
#ifdef SGM3_TRACE
                if (COH && lane == 0) g_trace[ra.traceIdx][3] += __builtin_amdgcn_s_memrealtime() - tw0;
#endif
            }
            if (mine) {
                const u32 src = hand_slot(band - 1, i + 1, which);
#pragma unroll
                for (int q = 0; q < PLS; q++) {
                    const uint4 v = hand_load16(rsrc, src + q * LPC * 16);
                    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
                }
            }
        };
        auto park = [&](int i, const u32 (&w)[SWN]) {      // hand the fetched states of step i to slot 0
            if (wants(i)) lds_put(&st[(i + 1) & 1][0][which][lic * 4], w);
        };
        auto send = [&](int t) {                           // step t has been computed: the last slot's states are in st[t & 1]
            const int u = t - 2 * (SLOTS - 1);
            if (hasNext && which < 3 && u >= 0 && u < W) {
                const u32 dst = hand_slot(band, u, which);
#pragma unroll
                for (int q = 0; q < PLS; q++)
                    hand_store16(rsrc, dst + q * LPC * 16, *reinterpret_cast<const uint4*>(&st[t & 1][SLOTS][which][q * PW + lic * 4]));
            }
        };
        auto announce = [&](int t) {                       // the states of steps up to t are in memory
            const int u = t - 2 * (SLOTS - 1);
            if (hasNext && lane == 0 && u >= 0) __hip_atomic_store(progMe, min(u, W - 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // Software pipeline, one step per pixel, that never waits for an operation issued in the same step: step t first
        // drains what step t-1 issued (stores of step t-2's states, loads for step t+1, a sample of the neighbour's
        // progress), then announces step t-2, sends step t-1, parks step t+1's states in LDS and requests step t+2's.
        u32 w[SWN];
#pragma unroll
        for (int q = 0; q < SWN; q++) w[q] = 0;
        request(-1, w);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        park(-1, w);
        request(0, w);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        park(0, w);
        request(1, w);
        int sample = 0;
        __syncthreads();      // the states of pixels 0 and 1 of the row above are parked
        __syncthreads();      // slot 0 has taken pixel 0's
        for (int t = 0; t < T; t++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            seen = max(seen, sample);
            if (t >= 2) announce(t - 2);
            if (t >= 1) send(t - 1);
            if (t + 1 < T) park(t + 1, w);
            if (t + 2 < T) request(t + 2, w);
            if (hasPrev) sample = __hip_atomic_load(progNb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if defined(SGM3_KNOCK_NO_BARRIERS)
#elif defined(SGM3_KNOCK_HALF_BARRIERS)
            if ((t & 1) == 0) __syncthreads();
#else
            __syncthreads();
#endif
        }
        send(T - 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        announce(T - 1);
        finish();
        return;
    }

    // ---- compute waves ----
    const int lic = threadIdx.x % LPC, j = threadIdx.x / LPC;
    const int v = band * SLOTS + j - ra.vofs;             // logical row; the first vofs slots of band 0 lie before the image
    const bool rowOk = v >= 0 && v < H;
    const int vc = min(max(v, 0), H - 1);                 // idle slots of the first / last band request (and drop) a valid row
    const int y = UP ? H - 1 - vc : vc;
    const bool top = v == 0;
    // addresses as a wave-uniform 64-bit base (the band's lowest image row of this pair) plus 32-bit lane offsets: a band
    // spans SLOTS rows, so the offsets stay far below 4 GB whatever the image size, and no 64-bit arithmetic or pointer
    // pair per lane is needed in the loop
    const int vFirst = band * SLOTS - ra.vofs, vLastC = min(max(vFirst + SLOTS - 1, 0), H - 1), vFirstC = min(max(vFirst, 0), H - 1);
    const int yLo = UP ? H - 1 - vLastC : vFirstC;
    const u32 dy = (u32)(y - yLo);
    const u8* const cLbase = a.cL + (size_t)pair * a.strideL + (size_t)yLo * a.pL;
    const u8* const cRbase = a.cR + (size_t)pair * a.strideR + (size_t)yLo * a.pR - DP;    // (biased so that the lane offsets below are never negative)
    constexpr u32 SB = S4B ? DP : 2 * DP;                                   // bytes of a pixel in the S volume
    u8* const Sbase = reinterpret_cast<u8*>(a.S + (size_t)pair * a.strideS) + (size_t)yLo * W * SB;
    uint2* const recBase = a.rec + (size_t)pair * a.strideRec + (size_t)yLo * W;
    const u32 cLoff = dy * (u32)a.pL;
    const u32 cRoff = dy * (u32)a.pR + (u32)(DP + a.maxD - a.minD - lic * DPLn - (DPLn - 1));
    const u32 Soff = dy * (u32)W * SB + (u32)lic * 16;
    const u32 recOff = dy * (u32)W;
    // byte sums: streaming accesses; in the one-launch schedule (COH) write-through stores and loads that bypass the
    // non-coherent caches, through a descriptor of the pair's volume (the band's first row as the scalar offset)
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<u8*>(a.S + (size_t)pair * a.strideS), 0,
                                                                           (int)(a.strideS * sizeof(u16)), 0x00020000);
    const u32 sRowBase = (u32)yLo * (u32)W * SB;
    auto s4_load = [&](u32 off) -> uint4 {
        if constexpr (COH) {
            const md_v4u d = __builtin_amdgcn_raw_buffer_load_b128(srsrc, (int)off, (int)sRowBase, 18);
            return make_uint4(d.x, d.y, d.z, d.w);
        } else return load16_stream(Sbase + off);
    };
    auto s4_store = [&](u32 off, const uint4& v) {
        if constexpr (COH) {
            md_v4u d = {v.x, v.y, v.z, v.w};
            __builtin_amdgcn_raw_buffer_store_b128(d, srsrc, (int)off, (int)sRowBase, 18);
            STORE_DATA_GUARD(d);
        } else store16_stream(Sbase + off, v);
    };
    u32 tail[NRT];
    make_tail_n<NRT>(lic, a.D, tail);
    const u32 P1P1 = a.P1P1, P2P2 = a.P2P2;
    constexpr bool four = FOUR;
    auto fresh = [&](u32 (&N)[NRT]) {
#pragma unroll
        for (int q = 0; q < NRT; q++) N[q] = TAIL ? tail[q] : 0u;
    };
    auto from_bytes = [&](const u32 (&w)[SWN], u32 (&N)[NRT]) {    // all-zero words are the state of a path that starts here
        if constexpr (S16) {
#pragma unroll
            for (int q = 0; q < NRT; q++) N[q] = w[q];
        } else unpack_n<NRT>(w, N);
        if (TAIL) {
#pragma unroll
            for (int q = 0; q < NRT; q++) N[q] |= tail[q];
        }
    };
    // where this slot reads the row above's states: slot j-1's LDS entry, or (slot 0) what the communication wave parked;
    // both alternate with the parity of t + 1
    const u32* const aboveBase = &st[0][j][0][lic * 4];
    constexpr int PARSTRIDE = (SLOTS + 1) * 3 * SW;      // dwords between the two parity entries

    u32 Nh[NRT];                       // the path that runs along the row
    EdgeRegs edge = {INF2, INF2};      // targets of the cross-lane shifts: their row-edge lanes stay "infinite"
    u32 ooh = 0;                       // its minimum (only where the state is kept un-normalised: no masked disparities)
    fresh(Nh);
    u32 clw = 0, rbw[NW], sbw[NW], Sn[NRT];
#pragma unroll
    for (int q = 0; q < NW; q++) { rbw[q] = 0; sbw[q] = 0; }
#pragma unroll
    for (int q = 0; q < NRT; q++) Sn[q] = 0;
    int xf = 0;                        // the column the operands in flight belong to
    auto fetch = [&](int t) {          // unconditional: out-of-range steps ask for the nearest pixel of the row again
        const int u = min(max(t - 2 * j, 0), W - 1);
        xf = UP ? W - 1 - u : u;
        clw = *reinterpret_cast<const u32*>(cLbase + (cLoff + (u32)(xf & ~3)));
        __builtin_memcpy(rbw, cRbase + (cRoff + (u32)xf), 2 * NRT);
        if (SECOND) {
            if (S4B) {
#pragma unroll
                for (int q = 0; q < PL; q++) {
                    const uint4 v4 = s4_load(Soff + (u32)xf * DP + q * LPC * 16);
                    sbw[4 * q] = v4.x; sbw[4 * q + 1] = v4.y; sbw[4 * q + 2] = v4.z; sbw[4 * q + 3] = v4.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < NRT / 4; q++) {
                    const uint4 v4 = load16_stream(Sbase + (Soff + (u32)xf * 2 * DP + q * LPC * 16));
                    Sn[4 * q] = v4.x; Sn[4 * q + 1] = v4.y; Sn[4 * q + 2] = v4.z; Sn[4 * q + 3] = v4.w;
                }
            }
        }
    };
    fetch(0);
    // states of the row above kept in registers: the vertical one for one step, the diagonal one for two.  The loop is
    // unrolled by two so that "keep" is a matter of which register set a step names, not of copies.
    u32 V0[SWN], V1[SWN], DA[SWN], DB[SWN];
#pragma unroll
    for (int q = 0; q < SWN; q++) { V0[q] = 0; V1[q] = 0; DA[q] = 0; DB[q] = 0; }
    if (!HONLY) __syncthreads();       // the communication wave has parked pixel 0 and pixel 1 of the row above
    if (!HONLY && j == 0) {                      // slot 0 has no step "-1": it takes pixel 0's states here
        lds_get(&st[0][0][0][lic * 4], V0);
        lds_get(&st[0][0][1][lic * 4], DB);
    }
    if (!HONLY) __syncthreads();
    // one step: Vuse was read in the previous step, D two steps ago; Vfill / D receive what the row above published in
    // the previous step (D only after its old contents have been consumed)
    auto step = [&](int t, auto parc, u32 (&Vuse)[SWN], u32 (&Vfill)[SWN], u32 (&D)[SWN]) {
        constexpr int PAR = decltype(parc)::value;        // t % 2, known at compile time: LDS addresses are base + immediate
        const int u = t - 2 * j;
        const bool active = rowOk && u >= 0 && u < W;
        const int x = xf;
        u32 C[NRT], L[NRT], Sacc[NRT];
        cost_n<NRT>(census_byte(clw, x), rbw, C);
        if (SECOND) {
            if (S4B) unpack_n<NRT>(sbw, Sacc);
            else {
#pragma unroll
                for (int q = 0; q < NRT; q++) Sacc[q] = Sn[q];
            }
        }
        fetch(t + 1);
        const u32* above = aboveBase + (PAR ^ 1) * PARSTRIDE;
        u32* const own = const_cast<u32*>(aboveBase) + 3 * SW + PAR * PARSTRIDE;
        u32 curA[SWN];
        if constexpr (!HONLY) {
            lds_get(above, Vfill);
            lds_get(above + 2 * SW, curA);              // anti-diagonal: used at once
        }
        // ALWAYS: every slot computes every step, inside the image or not (a slot before its row's first pixel, past its last
        // one or in a padding row works on garbage that no active slot ever uses: paths that enter the image are started
        // afresh by the border cases below); only the stores are conditional.  Without the branch around the step the row
        // path's registers need no copies where the two branches would meet.
        constexpr bool ALWAYS = true;
        if (ALWAYS || active) {
            // paths that start at this pixel (image border): rare, so the per-lane selects sit behind a wave-level test
            const bool firstCol = u == 0, lastCol = u == W - 1;
            if (__any(top || firstCol || lastCol)) {
                if constexpr (!HONLY) {
#pragma unroll
                    for (int q = 0; q < SWN; q++) {
                        if (top || lastCol) curA[q] = 0;
                        if (top) Vuse[q] = 0;
                        if (top || firstCol) D[q] = 0;
                    }
                }
                if (firstCol) { fresh(Nh); ooh = 0; }
            }
            if (TAIL) step_n<TAIL, LPC, NRT>(Nh, C, P1P1, P2P2, tail, L, lic, edge);
            else step_lazy_n<LPC, NRT>(Nh, ooh, C, P1P1, P2P2, L, lic, edge);
#pragma unroll
            for (int q = 0; q < NRT; q++) Sacc[q] = SECOND ? pkadd(Sacc[q], L[q]) : L[q];
            u32 N[NRT], ow[SWN];
            auto path_step = [&](u32 (&Nq)[NRT]) {          // consumes the predecessor state, leaves L' in L and the new state in ow
                if constexpr (S16) {
                    step_n<TAIL, LPC, NRT>(Nq, C, P1P1, P2P2, tail, L, lic, edge);
#pragma unroll
                    for (int q = 0; q < NRT; q++) ow[q] = Nq[q];
                } else step_pack_n<TAIL, LPC, NRT>(Nq, C, P1P1, P2P2, tail, L, ow, lic, edge);
            };
            if constexpr (!four && !HONLY) {
                from_bytes(curA, N);
                path_step(N);
#pragma unroll
                for (int q = 0; q < NRT; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
                lds_put(own + 2 * SW, ow);
            }
            if constexpr (!HONLY) {
                from_bytes(Vuse, N);
                path_step(N);
#pragma unroll
                for (int q = 0; q < NRT; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
                lds_put(own, ow);
            }
            if constexpr (!four && !HONLY) {
                from_bytes(D, N);
                path_step(N);
#pragma unroll
                for (int q = 0; q < NRT; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
                lds_put(own + SW, ow);
            }
            if (!SECOND) {
                if constexpr (S4B) {
                    u32 sw4[NW];
                    pack_n<NRT>(Sacc, sw4);
                    if constexpr (COH) {
#pragma unroll
                        for (int q = 0; q < PL; q++)
                            s4_store(!active ? 0xFFFFFFF0u : Soff + (u32)x * DP + q * LPC * 16,      // (beyond the descriptor's range: dropped)
                                     make_uint4(sw4[4 * q], sw4[4 * q + 1], sw4[4 * q + 2], sw4[4 * q + 3]));
                    } else if (active) {
#pragma unroll
                        for (int q = 0; q < PL; q++)
                            s4_store(Soff + (u32)x * DP + q * LPC * 16, make_uint4(sw4[4 * q], sw4[4 * q + 1], sw4[4 * q + 2], sw4[4 * q + 3]));
                    }
                } else if (active) {
#pragma unroll
                    for (int q = 0; q < NRT / 4; q++)
                        store16_stream(Sbase + (Soff + (u32)x * 2 * DP + q * LPC * 16), make_uint4(Sacc[4 * q], Sacc[4 * q + 1], Sacc[4 * q + 2], Sacc[4 * q + 3]));
                }
            } else if (FUSE) {
                wta_n<TAIL, LPC, NRT>(a, Sacc, tail, lic, &srow[j][0], exclMask, recBase + (recOff + (u32)x), active);
            } else if (active) {
                // volume export: the full sum in natural disparity order, in place of the four-path sum
                u32 nat[NRT];
#pragma unroll
                for (int g = 0; g < NRT / 2; g++) {
                    nat[g] = __builtin_amdgcn_perm(Sacc[2 * g + 1], Sacc[2 * g], 0x05040100u);
                    nat[g + NRT / 2] = __builtin_amdgcn_perm(Sacc[2 * g + 1], Sacc[2 * g], 0x07060302u);
                }
                uint4* dst = reinterpret_cast<uint4*>(Sbase + (Soff - (u32)lic * 16 + (u32)x * 2 * DP + (u32)lic * DPLn * 2));
#pragma unroll
                for (int q = 0; q < NRT / 4; q++) dst[q] = make_uint4(nat[4 * q], nat[4 * q + 1], nat[4 * q + 2], nat[4 * q + 3]);
            }
        }
        if constexpr (!HONLY) {
            lds_get(above + SW, D);
#if defined(SGM3_KNOCK_NO_BARRIERS)    // timing experiments only (results wrong): no step barrier / one per two steps
#elif defined(SGM3_KNOCK_HALF_BARRIERS)
            if (PAR == 0) __syncthreads();
#else
            __syncthreads();   // st[t & 1] complete before the next step reads it
#endif
        }
    };
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 2 <= T; t += 2) {
        step(t, P0{}, V0, V1, DA);
        step(t + 1, P1{}, V1, V0, DB);
    }
    if (t < T) step(t, P0{}, V0, V1, DA);
    finish();
}

// ---------------------------------------------------------------------------------------------------------------------
// The same sweep without a workgroup barrier per step (headline geometry: 16 lanes x 16 disparities, 256 compute lanes,
// byte sums, no masked disparities, eight paths; first and second sweep).
//
// A wave holds four neighbouring rows.  Three of the four hand-offs between neighbouring rows therefore stay inside one
// wave: the anti-diagonal and the vertical state move from lanes 16k .. 16k+15 to the next sixteen lanes through the
// cross-lane network (ds_bpermute_b32: no LDS memory access, no synchronisation -- it is the wave's own instruction
// stream), right after they are produced; the diagonal state, which is consumed two steps later, waits one step in an
// LDS row that only its own wave reads (LDS operations of one wave execute in order).  Only the hand-off from a wave's
// last row to the next wave's first row crosses waves.  It goes through a ring of RING pixels in LDS per link with two
// counters (pixels published, pixels released): the producer may run up to RING - 3 steps ahead of the minimum distance,
// so the waves of a workgroup no longer wait for the slowest of them in every step -- a wave waits only when its ring is
// empty or full.  The communication wave is the producer of wave 0's ring and the consumer of wave 3's; it moves up to
// four pixels per memory round trip in each direction and never blocks on one side while the other can progress.
// Every wait is bounded (HANDOFF_TIMEOUT_TICKS) and ends in the sticky error word.
constexpr int SGM3_RING = 8;
struct RingLds {
    u32 ring[5][SGM3_RING][3][64];    // link k: [pixel % RING][anti-diagonal, vertical, diagonal][16 lanes x 4 dwords]
    u32 stD[2][16][64];               // [step parity][row][...]: the diagonal state on its way to the next row of the same wave
    int prod[5], cons[5];             // per link: pixels published by the producer / released by the consumer
};

// polls an LDS counter until it reaches `need`; wave-uniform.  Returns the value seen, or a huge one after a timeout
// (the error word is set and the caller's later waits fall through).
// (The counters are addressed through LDS-typed pointers: through a generic pointer the accesses become flat operations,
// whose waits also drain the global loads in flight.)
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ int lds_peek(const int* word) { return __builtin_amdgcn_readfirstlane(*(const volatile lds_int*)word); }
__device__ __forceinline__ void lds_poke(int* word, int value) { *(volatile lds_int*)word = value; }
__device__ __forceinline__ int ring_wait(const int* word, int need, int* err)
{
    unsigned long long t0 = 0;
    int v;
    for (int it = 0;; it++) {
        v = lds_peek(word);
        if (v >= need) break;
        if ((it & 255) == 255) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            if (now - t0 > HANDOFF_TIMEOUT_TICKS || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = 0x3FFFFFFF;
                break;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void ring_post(int* word, int value)      // after the data it covers (same wave: LDS operations stay in order)
{
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) lds_poke(word, value);
    asm volatile("" ::: "memory");
}

template <bool UP, int ROLE, bool COH>
__device__ __forceinline__ void sgm3_ring_body(const RowArgs& ra, const int band, const int pair, RingLds& sh, u32 (*srow)[16 * 8],
                                               const u32 (*exclMask)[8])
{
    constexpr int LPC = 16, NRT = 8, NT = 256, SLOTS = 16, DPLn = 16, DP = 256, NW = 4, SWN = 4, R = SGM3_RING;
    constexpr bool SECOND = ROLE != 0;
    static_assert(ROLE == 0 || ROLE == 1, "first or second sweep");
    const Sgm2Args& a = ra.a;
    auto finish = [&]() {
        if constexpr (COH && ROLE == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0)
                __hip_atomic_store(ra.done + ((size_t)((UP ? ra.batch : 0) + pair) * ra.nb + band), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    const int H = a.H, W = a.W;
    u8* hand = ra.hand + (size_t)pair * ra.strideHand;
    const bool hasPrev = band > 0, hasNext = band + 1 < ra.nb;
    auto lds_get = [&](const u32* p, u32 (&w)[SWN]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    };
    auto lds_put = [&](u32* p, const u32 (&w)[SWN]) { *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]); };

    if (threadIdx.x >= NT) {
        // ---- communication wave: lanes 0..47 = three states x sixteen lanes of one pixel per memory operation
        const int lane = threadIdx.x - NT, which = lane / LPC, lic = lane % LPC;
        const bool mine = which < 3;
        int* progMe = ra.sync + 16 + ((size_t)((UP ? ra.batch : 0) + pair) * ra.nb + band);
        const int* progNb = progMe - 1;
        const __amdgpu_buffer_rsrc_t rsrc = hand_rsrc(hand, ra.strideHand);
        constexpr u32 DPB = 256;
        auto hand_slot = [&](int bnd, int u) -> u32 { return (u32)((((size_t)(bnd & ra.ringMask) * W + u) * 3 + which) * DPB + lic * 16); };
        if (lane == 0) {
            if (!hasPrev) lds_poke(&sh.prod[0], 0x3FFFFFFF);       // nothing to wait for: the first row's paths start afresh
            if (!hasNext) lds_poke(&sh.cons[4], 0x3FFFFFFF);       // nobody reads the last row's states
        }
        constexpr int KB = 4;                   // pixels per memory round trip, each way
        u32 w[KB][SWN];
#pragma unroll
        for (int q = 0; q < KB; q++)
#pragma unroll
            for (int e = 0; e < SWN; e++) w[q][e] = 0;
        int parked = 0, inflight = 0, sent = 0, announced = 0, seen = hasPrev ? 0 : W;
#if defined(SGM3_RING_KNOCK) && (SGM3_RING_KNOCK & 1)
        // timing experiment (results wrong): no dependency between bands
        if (lane == 0) { lds_poke(&sh.prod[0], 0x3FFFFFFF); lds_poke(&sh.cons[4], 0x3FFFFFFF); }
        const int wantIn = 0, wantOut = 0;
#else
        const int wantIn = hasPrev ? W : 0, wantOut = hasNext ? W : 0;
#endif
        unsigned long long t0 = 0;
        int idle = 0;
        while (parked < wantIn || sent < wantOut) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the loads of the previous round have arrived, its stores have completed
            if (sent > announced) {
                if (lane == 0) __hip_atomic_store(progMe, sent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                announced = sent;
            }
            bool moved = false;
            if (inflight > 0) {
#pragma unroll
                for (int q = 0; q < KB; q++)
                    if (q < inflight && mine) lds_put(&sh.ring[0][(parked + q) % R][which][lic * 4], w[q]);
                parked += inflight;
                inflight = 0;
                ring_post(&sh.prod[0], parked);
                moved = true;
            }
            if (parked < wantIn) {
                if (seen < W) seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(progNb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                __atomic_signal_fence(__ATOMIC_SEQ_CST);
                const int released = lds_peek(&sh.cons[0]);
                // pixel p may be fetched once the band above has published it (plus the configured slack) and its ring entry is free
                const int avail = seen >= W ? W : seen - ra.slack;
                const int n = min(KB, min(min(avail, released + R), W) - parked);
                if (n > 0) {
#pragma unroll
                    for (int q = 0; q < KB; q++)
                        if (q < n && mine) {
                            const uint4 v = hand_load16(rsrc, hand_slot(band - 1, parked + q));
                            w[q][0] = v.x; w[q][1] = v.y; w[q][2] = v.z; w[q][3] = v.w;
                        }
                    inflight = n;
                    moved = true;
                }
            }
            if (sent < wantOut) {
                const int ready = lds_peek(&sh.prod[4]);
                asm volatile("" ::: "memory");
                const int n = min(KB, min(ready, W) - sent);
                if (n > 0) {
#pragma unroll
                    for (int q = 0; q < KB; q++)
                        if (q < n && mine)
                            hand_store16(rsrc, hand_slot(band, sent + q), *reinterpret_cast<const uint4*>(&sh.ring[4][(sent + q) % R][which][lic * 4]));
                    sent += n;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the ring entries have been read
                    ring_post(&sh.cons[4], sent);
                    moved = true;
                }
            }
            if (!moved) {
                __builtin_amdgcn_s_sleep(2);
                if ((++idle & 255) == 0) {
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (t0 == 0) t0 = now;
                    if (now - t0 > HANDOFF_TIMEOUT_TICKS || __hip_atomic_load(ra.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(ra.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (lane == 0) {       // let the compute waves run to their end
                            lds_poke(&sh.prod[0], 0x3FFFFFFF);
                            lds_poke(&sh.cons[4], 0x3FFFFFFF);
                        }
                        break;
                    }
                }
            } else { idle = 0; t0 = 0; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (hasNext && lane == 0) __hip_atomic_store(progMe, W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        finish();
        return;
    }

    // ---- compute waves ----
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave 0..3: consumes link wv, produces link wv + 1
    const int lic = threadIdx.x % LPC, j = threadIdx.x / LPC, l = j & 3;
    const int T = W + 6;                                   // wave-local steps: row l of the wave works on u = t - 2l
    const int v = band * SLOTS + j - ra.vofs;
    const bool rowOk = v >= 0 && v < H;
    const int vc = min(max(v, 0), H - 1);
    const int y = UP ? H - 1 - vc : vc;
    const bool top = v == 0;
    const int vFirst = band * SLOTS - ra.vofs, vLastC = min(max(vFirst + SLOTS - 1, 0), H - 1), vFirstC = min(max(vFirst, 0), H - 1);
    const int yLo = UP ? H - 1 - vLastC : vFirstC;
    const u32 dy = (u32)(y - yLo);
    const u8* const cLbase = a.cL + (size_t)pair * a.strideL + (size_t)yLo * a.pL;
    const u8* const cRbase = a.cR + (size_t)pair * a.strideR + (size_t)yLo * a.pR - DP;
    constexpr u32 SB = DP;
    u8* const Sbase = reinterpret_cast<u8*>(a.S + (size_t)pair * a.strideS) + (size_t)yLo * W * SB;
    uint2* const recBase = a.rec + (size_t)pair * a.strideRec + (size_t)yLo * W;
    const u32 cLoff = dy * (u32)a.pL;
    const u32 cRoff = dy * (u32)a.pR + (u32)(DP + a.maxD - a.minD - lic * DPLn - (DPLn - 1));
    const u32 Soff = dy * (u32)W * SB + (u32)lic * 16;
    const u32 recOff = dy * (u32)W;
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<u8*>(a.S + (size_t)pair * a.strideS), 0,
                                                                           (int)(a.strideS * sizeof(u16)), 0x00020000);
    const u32 sRowBase = (u32)yLo * (u32)W * SB;
    auto s4_load = [&](u32 off) -> uint4 {
        if constexpr (COH) {
            const md_v4u d = __builtin_amdgcn_raw_buffer_load_b128(srsrc, (int)off, (int)sRowBase, 18);
            return make_uint4(d.x, d.y, d.z, d.w);
        } else return load16_stream(Sbase + off);
    };
    auto s4_store = [&](u32 off, const uint4& val) {
        if constexpr (COH) {
            md_v4u d = {val.x, val.y, val.z, val.w};
            __builtin_amdgcn_raw_buffer_store_b128(d, srsrc, (int)off, (int)sRowBase, 18);
            STORE_DATA_GUARD(d);
        } else store16_stream(Sbase + off, val);
    };
    u32 tail[NRT];
#pragma unroll
    for (int q = 0; q < NRT; q++) tail[q] = 0;
    const u32 P1P1 = a.P1P1, P2P2 = a.P2P2;
    auto fresh = [&](u32 (&N)[NRT]) {
#pragma unroll
        for (int q = 0; q < NRT; q++) N[q] = 0u;
    };
    // cross-lane source of the row above inside the wave: lane - 16 (the first row's lanes read themselves and are overwritten)
    const int upLane = (int)(((threadIdx.x & 63) >= 16 ? (threadIdx.x & 63) - 16 : (threadIdx.x & 63)) * 4);
    auto from_above = [&](const u32 (&src)[SWN], u32 (&dst)[SWN]) {
#pragma unroll
        for (int q = 0; q < SWN; q++) dst[q] = (u32)__builtin_amdgcn_ds_bpermute(upLane, (int)src[q]);
    };
    const bool firstRow = l == 0, lastRow = l == 3;
    u32* const ringIn = &sh.ring[0][0][0][0] + (size_t)wv * (R * 3 * 64) + lic * 4;         // this wave's incoming link
    u32* const ringOut = &sh.ring[0][0][0][0] + (size_t)(wv + 1) * (R * 3 * 64) + lic * 4;  // its outgoing link
    u32* const stDown = &sh.stD[0][j][lic * 4];                                              // own row's diagonal state (parity 0)
    const u32* const stUp = &sh.stD[0][j > 0 ? j - 1 : 0][lic * 4];                          // the row above's
    constexpr int STD_PAR = 16 * 64;

    u32 Nh[NRT];
    EdgeRegs edge = {INF2, INF2};
    u32 ooh = 0;
    fresh(Nh);
    u32 clw = 0, rbw[NW], sbw[NW];
#pragma unroll
    for (int q = 0; q < NW; q++) { rbw[q] = 0; sbw[q] = 0; }
    int xf = 0;
    auto fetch = [&](int t) {
        const int u = min(max(t - 2 * l, 0), W - 1);
        xf = UP ? W - 1 - u : u;
        clw = *reinterpret_cast<const u32*>(cLbase + (cLoff + (u32)(xf & ~3)));
        __builtin_memcpy(rbw, cRbase + (cRoff + (u32)xf), 2 * NRT);
        if (SECOND) {
            const uint4 v4 = s4_load(Soff + (u32)xf * DP);
            sbw[0] = v4.x; sbw[1] = v4.y; sbw[2] = v4.z; sbw[3] = v4.w;
        }
    };
    fetch(0);
    u32 V0[SWN], V1[SWN], DA[SWN], DB[SWN], curA[SWN];
#pragma unroll
    for (int q = 0; q < SWN; q++) { V0[q] = 0; V1[q] = 0; DA[q] = 0; DB[q] = 0; curA[q] = 0; }
    int seenProd = 0, seenCons = 0;
#if defined(SGM3_RING_KNOCK) && (SGM3_RING_KNOCK & 2)
    seenProd = 0x3FFFFFFF; seenCons = 0x3FFFFFFF;       // timing experiment (results wrong): the waves never wait for each other
#endif
    // the first row of the wave takes what a step "-1" and "-2" would have fetched: the vertical and diagonal state of pixel 0,
    // the anti-diagonal and vertical state of pixel 1
    if (seenProd < min(2, W)) seenProd = ring_wait(&sh.prod[wv], min(2, W), ra.err);
    if (firstRow) {
        lds_get(ringIn + 1 * 64, V0);
        lds_get(ringIn + 2 * 64, DB);
        lds_get(ringIn + (1 % R) * 192, curA);
        lds_get(ringIn + (1 % R) * 192 + 64, V1);
    }
    auto step = [&](int t, auto parc, u32 (&Vuse)[SWN], u32 (&D)[SWN]) {
        constexpr int PAR = decltype(parc)::value;
        const int u = t - 2 * l;
        const bool active = rowOk && u >= 0 && u < W;
        const int x = xf;
        u32 C[NRT], L[NRT], Sacc[NRT];
        cost_n<NRT>(census_byte(clw, x), rbw, C);
        if (SECOND) unpack_n<NRT>(sbw, Sacc);
        fetch(t + 1);
        // the last row publishes pixel u3 in this step: its ring entry must have been released
        const int u3 = t - 6;
        const bool publish = u3 >= 0 && u3 < W;
        if (publish && u3 >= seenCons + R) seenCons = ring_wait(&sh.cons[wv + 1], u3 - R + 1, ra.err);
        u32* const outEntry = ringOut + (u3 & (R - 1)) * 192;
        const bool firstCol = u == 0, lastCol = u == W - 1;
        if (__any(top || firstCol || lastCol)) {
#pragma unroll
            for (int q = 0; q < SWN; q++) {
                if (top || lastCol) curA[q] = 0;
                if (top) Vuse[q] = 0;
                if (top || firstCol) D[q] = 0;
            }
            if (firstCol) { fresh(Nh); ooh = 0; }
        }
        step_lazy_n<LPC, NRT>(Nh, ooh, C, P1P1, P2P2, L, lic, edge);
#pragma unroll
        for (int q = 0; q < NRT; q++) Sacc[q] = SECOND ? pkadd(Sacc[q], L[q]) : L[q];
        u32 N[NRT], ow[SWN];
        // anti-diagonal: consumed at once by the row below
        unpack_n<NRT>(curA, N);
        step_pack_n<false, LPC, NRT>(N, C, P1P1, P2P2, tail, L, ow, lic, edge);
#pragma unroll
        for (int q = 0; q < NRT; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
        if (lastRow && publish) lds_put(outEntry, ow);
        from_above(ow, curA);
        // vertical: the row below uses it two steps from now, out of the register set this step has just consumed
        unpack_n<NRT>(Vuse, N);
        step_pack_n<false, LPC, NRT>(N, C, P1P1, P2P2, tail, L, ow, lic, edge);
#pragma unroll
        for (int q = 0; q < NRT; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
        if (lastRow && publish) lds_put(outEntry + 64, ow);
        from_above(ow, Vuse);
        // diagonal: three steps from now; it spends one of them in the wave's own LDS row
        unpack_n<NRT>(D, N);
        step_pack_n<false, LPC, NRT>(N, C, P1P1, P2P2, tail, L, ow, lic, edge);
#pragma unroll
        for (int q = 0; q < NRT; q++) Sacc[q] = pkadd(Sacc[q], L[q]);
        if (lastRow && publish) lds_put(outEntry + 128, ow);
        lds_put(stDown + PAR * STD_PAR, ow);
        if (publish) ring_post(&sh.prod[wv + 1], u3 + 1);
        if (!SECOND) {
            u32 sw4[NW];
            pack_n<NRT>(Sacc, sw4);
            if constexpr (COH) {
                s4_store(!active ? 0xFFFFFFF0u : Soff + (u32)x * DP, make_uint4(sw4[0], sw4[1], sw4[2], sw4[3]));
            } else if (active) {
                s4_store(Soff + (u32)x * DP, make_uint4(sw4[0], sw4[1], sw4[2], sw4[3]));
            }
        } else {
            wta_n<false, LPC, NRT>(a, Sacc, tail, lic, &srow[j][0], exclMask, recBase + (recOff + (u32)x), active);
        }
        // what the next steps consume from the row above
        lds_get(stUp + (PAR ^ 1) * STD_PAR, D);
        {
            const int need = min(t + 3, W);
            if (seenProd < need) seenProd = ring_wait(&sh.prod[wv], need, ra.err);
            if (firstRow) {
                const u32* const e2 = ringIn + ((t + 2) & (R - 1)) * 192;
                lds_get(e2, curA);
                lds_get(e2 + 64, Vuse);
                lds_get(ringIn + ((t + 1) & (R - 1)) * 192 + 128, D);
            }
            // (LDS operations of a wave are performed in order: the counter is written after the reads above have been done)
            ring_post(&sh.cons[wv], t + 2);
        }
    };
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 2 <= T; t += 2) {
        step(t, P0{}, V0, DA);
        step(t + 1, P1{}, V1, DB);
    }
    if (t < T) step(t, P0{}, V0, DA);
    finish();
}

// One sweep (one direction, one role) over a range of bands: the launch counts tickets on its own word.
template <bool UP, int ROLE, bool TAIL, bool S4B, int LPC, int NRT, int NT, bool S16 = false, bool FOUR = false>
__global__ __launch_bounds__(NT + 64, sgm3_waves_per_simd(LPC, NRT, TAIL, S4B, S16, ROLE)) void sgm3_rows_kernel(RowArgs ra)
{
    constexpr bool FUSE = ROLE == 1 || ROLE == 3;
    constexpr int SLOTS = NT / LPC, DPLn = 2 * NRT, SW = sgm3_state_dwords(LPC, NRT, S16);
    // [step parity][row 0: the band above's states for slot 0, parked by the communication wave; row j + 1: slot j][v, \, /][packed bytes of the pixel]
    // Slot j reads row j and writes row j + 1, so every slot addresses the row above with the same constant strides.
    __shared__ u32 st[2][SLOTS + 1][3][SW];
    __shared__ u32 srow[FUSE ? SLOTS : 1][FUSE ? LPC * NRT : 1];
    __shared__ u32 exclMask[FUSE ? DPLn + 3 : 1][NRT];
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ra.sync[ra.ticket], 1);
    if (FUSE) init_excl_mask_n<NRT>(exclMask);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);      // wave-uniform: pair, band and everything derived from them stay in scalar registers
    // tickets: group after group of pairs; inside a group band after band (dependency order), pair after pair
    const int perGroup = ra.group * ra.nbands, grp = ticket / perGroup, rem = ticket - grp * perGroup;
    const int gsz = min(ra.group, ra.batch - grp * ra.group);
    const int ord = rem / gsz, pair = grp * ra.group + (rem - ord * gsz);
    const int band = ra.band0 + ord;
    sgm3_rows_body<UP, ROLE, TAIL, S4B, LPC, NRT, NT, S16, FOUR, false>(ra, band, pair, st, reinterpret_cast<u32 (*)[LPC * NRT]>(&srow[0][0]), exclMask);
}

// The one-launch schedule: both directions and both roles of every pair in ONE launch.  Tickets enumerate, band step after
// band step, the down sweep's and the up sweep's band of every pair; a direction's first bands store their four-path sums,
// the others -- past the row where the two sweeps meet -- add the sums the other direction stored there and run the
// winner-take-all.  A second-sweep band waits for the first-sweep band of the other direction that covers the same image
// rows (an earlier ticket: it has started), so the GPU fills once and drains once per call instead of once per phase.
template <bool TAIL, int LPC, int NRT, int NT>
#ifndef SGM3_MEET_WPS
#define SGM3_MEET_WPS sgm3_waves_per_simd(LPC, NRT, TAIL, true, false, 1)
#endif
__global__ __launch_bounds__(NT + 64, SGM3_MEET_WPS) void sgm3_meet_kernel(RowArgs ra)
{
    constexpr int SLOTS = NT / LPC, DPLn = 2 * NRT, SW = sgm3_state_dwords(LPC, NRT, false);
    __shared__ u32 st[2][SLOTS + 1][3][SW];
    __shared__ u32 srow[SLOTS][LPC * NRT];
    __shared__ u32 exclMask[DPLn + 3][NRT];
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ra.sync[ra.ticket], 1);
    init_excl_mask_n<NRT>(exclMask);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);
#ifdef SGM3_TRACE
    if (threadIdx.x == 0 && ticket < SGM3_TRACE_MAX) { g_trace[ticket][0] = __builtin_amdgcn_s_memrealtime(); g_trace[ticket][3] = 0; g_trace[ticket][4] = __builtin_amdgcn_s_memtime(); }
#endif
    // inside a band step the direction that sweeps first over more bands comes first: with an odd number of bands the middle
    // band is that direction's last first-sweep band and the other direction's first second-sweep band
    const int perStep = 2 * ra.batch, band = ticket / perStep, r = ticket - band * perStep;
    const bool up = (r >= ra.batch) != (ra.firstUp > ra.firstDown);
    const int pair = r >= ra.batch ? r - ra.batch : r;
    // (Both directions must change roles at the same band step: a second-sweep band waits for the other direction's band
    // nb - 1 - band, which has an earlier ticket only while band > nb - 1 - band.  Where the sweeps meet, that band
    // started just one band step earlier, so the first second-sweep bands wait for most of a band's duration.)
    const bool second = band >= (up ? ra.firstUp : ra.firstDown);
    if (second) {
        // the other direction's band over the same image rows (band edges coincide in the two directions)
        if (threadIdx.x == 0) {
            const int* flag = ra.done + ((size_t)((up ? 0 : ra.batch) + pair) * ra.nb + (ra.nb - 1 - band));
            unsigned long long t0 = 0;
            for (int it = 0; __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; it++) {
                if ((it & 63) == 63) {
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (t0 == 0) t0 = now;
                    if (now - t0 > HANDOFF_TIMEOUT_TICKS) { __hip_atomic_store(ra.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                if (__hip_atomic_load(ra.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                __builtin_amdgcn_s_sleep(32);
            }
        }
        __syncthreads();
    }
    RowArgs rl = ra;
#ifdef SGM3_TRACE
    rl.traceIdx = ticket < SGM3_TRACE_MAX ? ticket : 0;
    if (threadIdx.x == 0 && ticket < SGM3_TRACE_MAX) g_trace[ticket][1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (up) { rl.hand = ra.handUp; rl.vofs = ra.vofsUp; }
    if (!up && !second) sgm3_rows_body<false, 0, TAIL, true, LPC, NRT, NT, false, false, true>(rl, band, pair, st, srow, exclMask);
    else if (up && !second) sgm3_rows_body<true, 0, TAIL, true, LPC, NRT, NT, false, false, true>(rl, band, pair, st, srow, exclMask);
    else if (up) sgm3_rows_body<true, 1, TAIL, true, LPC, NRT, NT, false, false, true>(rl, band, pair, st, srow, exclMask);
    else sgm3_rows_body<false, 1, TAIL, true, LPC, NRT, NT, false, false, true>(rl, band, pair, st, srow, exclMask);
#ifdef SGM3_TRACE
    if (threadIdx.x == 0 && ticket < SGM3_TRACE_MAX) { g_trace[ticket][2] = __builtin_amdgcn_s_memrealtime(); g_trace[ticket][5] = __builtin_amdgcn_s_memtime(); }
#endif
}

// The one-launch schedule on the barrier-free sweep body (sgm3_ring_body): same tickets, same waits between the two
// directions.  48 KB of LDS per workgroup: three workgroups per CU.
__global__ __launch_bounds__(256 + 64, 4) void sgm3_meet_ring_kernel(RowArgs ra)
{
    __shared__ RingLds sh;
    __shared__ u32 srow[16][16 * 8];
    __shared__ u32 exclMask[16 + 3][8];
    __shared__ int s_ticket;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ra.sync[ra.ticket], 1);
    if (threadIdx.x < 5) { sh.prod[threadIdx.x] = 0; sh.cons[threadIdx.x] = 0; }
    init_excl_mask_n<8>(exclMask);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(s_ticket);
    const int perStep = 2 * ra.batch, band = ticket / perStep, r = ticket - band * perStep;
    const bool up = (r >= ra.batch) != (ra.firstUp > ra.firstDown);
    const int pair = r >= ra.batch ? r - ra.batch : r;
    const bool second = band >= (up ? ra.firstUp : ra.firstDown);
    if (second) {
        if (threadIdx.x == 0) {
            const int* flag = ra.done + ((size_t)((up ? 0 : ra.batch) + pair) * ra.nb + (ra.nb - 1 - band));
            unsigned long long t0 = 0;
            for (int it = 0; __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0; it++) {
                if ((it & 63) == 63) {
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (t0 == 0) t0 = now;
                    if (now - t0 > HANDOFF_TIMEOUT_TICKS) { __hip_atomic_store(ra.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                if (__hip_atomic_load(ra.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                __builtin_amdgcn_s_sleep(32);
            }
        }
        __syncthreads();
    }
    RowArgs rl = ra;
    if (up) { rl.hand = ra.handUp; rl.vofs = ra.vofsUp; }
    if (!up && !second) sgm3_ring_body<false, 0, true>(rl, band, pair, sh, srow, exclMask);
    else if (up && !second) sgm3_ring_body<true, 0, true>(rl, band, pair, sh, srow, exclMask);
    else if (up) sgm3_ring_body<true, 1, true>(rl, band, pair, sh, srow, exclMask);
    else sgm3_ring_body<false, 1, true>(rl, band, pair, sh, srow, exclMask);
}

template <int LPC, int NRT, int NT, bool BOTH, bool FOUR = false>
int launch_rows(hipStream_t s, const RowArgs& ra, int nbands, bool up, int role, bool s4b, bool tail)
{
    const dim3 g((unsigned)nbands * ra.batch);
#define SGM3_ROLE(UPV, ROLEV)                                                                                 \
    do {                                                                                                      \
        if (tail) { if (s4b) sgm3_rows_kernel<UPV, ROLEV, true, true, LPC, NRT, NT, false, FOUR><<<g, NT + 64, 0, s>>>(ra);   \
                    else sgm3_rows_kernel<UPV, ROLEV, true, false, LPC, NRT, NT, false, FOUR><<<g, NT + 64, 0, s>>>(ra); }    \
        else { if (s4b) sgm3_rows_kernel<UPV, ROLEV, false, true, LPC, NRT, NT, false, FOUR><<<g, NT + 64, 0, s>>>(ra);       \
               else sgm3_rows_kernel<UPV, ROLEV, false, false, LPC, NRT, NT, false, FOUR><<<g, NT + 64, 0, s>>>(ra); }        \
    } while (0)
    if (!up && role == 0) SGM3_ROLE(false, 0);
    else if (up && role == 1) SGM3_ROLE(true, 1);
    else if (up && role == 2) SGM3_ROLE(true, 2);
    else if (up && role == 3) SGM3_ROLE(true, 3);
    else if (BOTH && up && role == 0) { if constexpr (BOTH) SGM3_ROLE(true, 0); }
    else if (BOTH && !up && role == 1) { if constexpr (BOTH) SGM3_ROLE(false, 1); }
    else MD_REQUIRE(false, MDISP_E_UNSUPPORTED, "row bands: this sweep order is not built for the geometry");
#undef SGM3_ROLE
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

// 16-bit path states: 16 lanes x 16 disparities only
template <bool FOUR>
int launch_rows16(hipStream_t s, const RowArgs& ra, int nbands, bool up, int role, bool tail)
{
    const dim3 g((unsigned)nbands * ra.batch);
#define SGM3_ROLE16(UPV, ROLEV)                                                                           \
    do {                                                                                                  \
        if (tail) sgm3_rows_kernel<UPV, ROLEV, true, false, 16, 8, SGM3_NT16, true, FOUR><<<g, SGM3_NT16 + 64, 0, s>>>(ra);   \
        else sgm3_rows_kernel<UPV, ROLEV, false, false, 16, 8, SGM3_NT16, true, FOUR><<<g, SGM3_NT16 + 64, 0, s>>>(ra);       \
    } while (0)
    if (!up && role == 0) SGM3_ROLE16(false, 0);
    else if (up && role == 1) SGM3_ROLE16(true, 1);
    else if (up && role == 2) SGM3_ROLE16(true, 2);
    else if (up && role == 3) SGM3_ROLE16(true, 3);
    else if (up && role == 0) SGM3_ROLE16(true, 0);
    else if (!up && role == 1) SGM3_ROLE16(false, 1);
    else MD_REQUIRE(false, MDISP_E_UNSUPPORTED, "row bands: this sweep order is not built for 16-bit path states");
#undef SGM3_ROLE16
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

}  // namespace

// geometry of the row-band kernels: a pixel has lpc lanes x dpl disparity slots; compute lanes per workgroup (512 slots
// per pixel: eight rows per band, so that the states and the winner-take-all rows of a band fit the 64 KB of static LDS)
int md_sgm3_lanes(int lpc, int dpl) { return lpc == 16 && dpl == 32 ? 128 : lpc == 16 ? SGM3_NT16 : 256; }
// bands of the image: one more than the rows need, because the mirrored sweep pads its first band so that band edges fall
// on the same image rows in both directions
int md_sgm3_bands(int H, int lpc, int nt) { const int rows = nt / lpc; return (H + rows - 1) / rows; }
// bytes of the path states that cross a band edge (three states per pixel; s16: 16-bit states)
size_t md_sgm3_boundary_bytes(int W, int lpc, int dpl, bool s16) { return (size_t)W * 3 * lpc * dpl * (s16 ? 2 : 1); }
// The hand-off rows of one pair and one sweep direction: a ring of entries indexed by band & (entries - 1).  Band b + 1
// reads entry b pixel by pixel before it publishes the same pixel of its own last row, and so on down the chain, so by
// the time band b + entries writes a pixel of the entry every reader of the old contents is past it: two entries would
// do; 32 keep neighbouring bands in different cache sets and the whole ring below the 4 GB a buffer descriptor spans.
int md_sgm3_ring()
{
    const int e = md_tune(MDISP_TUNE_SGM_HAND_RING);
    return e >= 2 ? (e >= 32 ? 32 : e >= 16 ? 16 : e >= 8 ? 8 : e >= 4 ? 4 : 2) : 32;
}
size_t md_sgm3_hand_bytes(int H, int W, int lpc, int dpl, int nt, bool s16)
{ (void)H; (void)nt; return md_align((size_t)md_sgm3_ring() * md_sgm3_boundary_bytes(W, lpc, dpl, s16), 256); }
// tickets (16 words), progress words of the two directions, "band done" words of the two directions (one-launch schedule)
size_t md_sgm3_sync_bytes(int H, int batch, int lpc, int nt) { return md_align((16 + 4 * (size_t)batch * (md_sgm3_bands(H, lpc, nt) + 1)) * sizeof(int), 256); }

// Logical bands of one sweep that cover the image rows [r0, r1) (r0 a multiple of the band height, r1 a multiple or H)
void md_sgm3_band_range(int H, int lpc, int nt, bool up, int r0, int r1, int* kb, int* ke)
{
    const int rows = nt / lpc, nb = md_sgm3_bands(H, lpc, nt), Hp = nb * rows;
    if (!up) { *kb = r0 / rows; *ke = (r1 + rows - 1) / rows; }
    else { *kb = (Hp - std::min(Hp, (r1 + rows - 1) / rows * rows)) / rows; *ke = (Hp - r0) / rows; }
}

// One sweep over the logical bands [band0, band0 + nbands) of `batch` pairs.  `a` carries the census planes, the S
// volume (four-path sums: bytes when s4b, else u16 in register order; with role 2 the sweep leaves the full sum there as
// u16 in natural order) and the winner-take-all records.  The sweep's ticket and progress words in `sync` must be zero,
// except the progress word of band0 - 1, which must hold W when that band's states were put into `hand` from outside.
int md_sgm3_sweep(hipStream_t s, const Sgm2Args& a, u8* hand, size_t strideHand, int* sync, int* err, int batch, int lpc, int dpl, int nt,
                  bool s4b, bool up, int role, int band0, int nbands, bool s16, int ticket)
{
    RowArgs ra;
    ra.a = a; ra.hand = hand; ra.strideHand = strideHand; ra.sync = sync; ra.err = err;
    const int rows = nt / lpc;
    ra.nb = md_sgm3_bands(a.H, lpc, nt); ra.batch = batch; ra.band0 = band0;
    ra.vofs = up ? ra.nb * rows - a.H : 0;
    if (nbands < 0) nbands = ra.nb - band0;
    ra.nbands = nbands;
    ra.ringMask = md_sgm3_ring() - 1;
    ra.ticket = ticket >= 0 ? ticket : (up ? 1 : 0);
    MD_REQUIRE(strideHand < ((size_t)1 << 32) && ra.ticket < 16, MDISP_E_RANGE, "row bands: hand-off ring of %zu bytes per pair exceeds 4 GB", strideHand);
    { const int g = md_tune(MDISP_TUNE_SGM_PAIR_GROUP); ra.group = g > 0 ? std::min(g, batch) : batch; }
    { const int sl = md_tune(MDISP_TUNE_SGM_BAND_SLACK); ra.slack = sl > 0 ? sl : 0; }
    const bool tail = a.tail != 0;
    MD_REQUIRE(nt == md_sgm3_lanes(lpc, dpl) && (dpl == 16 || dpl == 32), MDISP_E_BADARG, "row bands: unsupported geometry");
    MD_REQUIRE(band0 >= 0 && nbands >= 1 && band0 + nbands <= ra.nb, MDISP_E_BADARG, "row bands: band range outside the image");
    if (s16) {
        MD_REQUIRE(lpc == 16 && dpl == 16 && !s4b, MDISP_E_UNSUPPORTED, "row bands: 16-bit path states need 16 lanes x 16 disparities");
        return a.four ? launch_rows16<true>(s, ra, nbands, up, role, tail) : launch_rows16<false>(s, ra, nbands, up, role, tail);
    }
    if (dpl == 32) {
        if (lpc == 16) return launch_rows<16, 16, 128, true>(s, ra, nbands, up, role, s4b, tail);
        MD_REQUIRE(false, MDISP_E_BADARG, "row bands: unsupported geometry");
    }
    if (a.four) {
        MD_REQUIRE(lpc == 16 && dpl == 16, MDISP_E_UNSUPPORTED, "row bands: the 4-path mode is built for 16 lanes x 16 disparities");
        return launch_rows<16, 8, SGM3_NT16, true, true>(s, ra, nbands, up, role, s4b, tail);
    }
    if (lpc == 4) return launch_rows<4, 8, 256, true>(s, ra, nbands, up, role, s4b, tail);
    if (lpc == 8) return launch_rows<8, 8, 256, true>(s, ra, nbands, up, role, s4b, tail);
    return launch_rows<16, 8, SGM3_NT16, true>(s, ra, nbands, up, role, s4b, tail);
}

// Both sweeps over all bands of `batch` pairs (down first); `sync` must be zero.  The two sweeps may share one hand-off ring.
int md_sgm3_sweeps(hipStream_t s, const Sgm2Args& a, u8* hand, size_t strideHand, int* sync, int* err, int batch, int lpc, int dpl, int nt,
                   int naturalOut, bool s4b, bool s16)
{
    MD_TRY(md_sgm3_sweep(s, a, hand, strideHand, sync, err, batch, lpc, dpl, nt, s4b, false, 0, 0, -1, s16, -1));
    return md_sgm3_sweep(s, a, hand, strideHand, sync, err, batch, lpc, dpl, nt, s4b, true, naturalOut ? 2 : 1, 0, -1, s16, -1);
}

// The one-launch schedule (sgm3_meet_kernel): the down sweep starts at the top, the up sweep at the bottom, each stores its
// four-path sums over its first `firstDown` / `firstUp` bands and continues past the meeting row as the second sweep.
// 16 lanes x 16 disparities, byte sums, no masked disparities, eight paths; a pair's sum volume must lie below 4 GB
// (one buffer descriptor).  `sync` must be zero.
bool md_sgm3_meet_supported(const Sgm2Args& a, int lpc, int dpl, bool s4b, bool s16)
{
    return lpc == 16 && dpl == 16 && s4b && !s16 && !a.tail && !a.four && !a.five && a.strideS * sizeof(u16) < ((size_t)1 << 32);
}
int md_sgm3_meet(hipStream_t s, const Sgm2Args& a, u8* handDown, u8* handUp, size_t strideHand, int* sync, int* err, int batch, int lpc, int dpl, int nt,
                 int firstDown, int firstUp)
{
    MD_REQUIRE(md_sgm3_meet_supported(a, lpc, dpl, true, false) && nt == SGM3_NT16, MDISP_E_UNSUPPORTED, "row bands: the one-launch schedule is not built for this geometry");
    RowArgs ra;
    ra.a = a; ra.hand = handDown; ra.handUp = handUp; ra.strideHand = strideHand; ra.sync = sync; ra.err = err;
    const int rows = nt / lpc;
    ra.nb = md_sgm3_bands(a.H, lpc, nt); ra.batch = batch; ra.band0 = 0; ra.nbands = ra.nb; ra.group = batch;
    ra.vofs = 0; ra.vofsUp = ra.nb * rows - a.H;
    ra.ringMask = md_sgm3_ring() - 1;
    ra.ticket = 0;
    ra.firstDown = firstDown; ra.firstUp = firstUp;
    ra.done = sync + 16 + 2 * (size_t)batch * (ra.nb + 1);
    // a second-sweep band waits for the other direction's band nb - 1 - band: that is an earlier ticket only if both
    // directions sweep first over (nearly) half of the bands
    MD_REQUIRE(firstDown + firstUp == ra.nb && 2 * firstDown >= ra.nb - 1 && 2 * firstUp >= ra.nb - 1, MDISP_E_BADARG,
               "row bands: the two first sweeps must cover the image exactly once and meet in the middle");
    MD_REQUIRE(strideHand < ((size_t)1 << 32), MDISP_E_RANGE, "row bands: hand-off ring of %zu bytes per pair exceeds 4 GB", strideHand);
    { const int sl = md_tune(MDISP_TUNE_SGM_BAND_SLACK); ra.slack = sl > 0 ? sl : 0; }
    const dim3 g((unsigned)(2 * ra.nb) * batch);
    // the barrier-free variant is bit-exact but measured slower (0.92 against 0.75 ms per pair at 64 pairs): opt-in
    if (SGM3_NT16 == 256 && md_tune(MDISP_TUNE_SGM_RING_HANDOFF) > 0) sgm3_meet_ring_kernel<<<g, 256 + 64, 0, s>>>(ra);
    else sgm3_meet_kernel<false, 16, 8, SGM3_NT16><<<g, SGM3_NT16 + 64, 0, s>>>(ra);
    MD_HIP(hipGetLastError());
    return MDISP_OK;
}

#ifdef SGM3_TRACE
extern "C" int md_sgm3_trace_read(void* out, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), bytes < sizeof(g_trace) ? bytes : sizeof(g_trace));
}
#endif
