// Issue cost per opcode of every vector instruction in the main loops of the headline SGM kernels (the opcodes of
// scripts/isa_ledger.py's listing), in nanoseconds per wave-instruction per SIMD with every SIMD of the chip busy, at 4 and
// at 5 waves per SIMD (the row-band kernel keeps 5).  Eight chains per lane, pinned with inline asm, so the figure is the
// issue rate and not a dependency stall.  Prints one JSON object; scripts/issue_bound.py multiplies it with the ledger.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_ubench4.bin scripts/valu_ubench4.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;

#define LOOP(STMT)                                                                                  \
    for (int it = 0; it < iters; it++) {                                                            \
        _Pragma("unroll") for (int u = 0; u < 8; u++) {                                             \
            _Pragma("unroll") for (int i = 0; i < 8; i++) { STMT; }                                 \
        }                                                                                           \
    }
#define B (a[(i + 3) & 7])
#define C (a[(i + 5) & 7])
#define A2(INS) asm volatile(INS " %0, %1" : "=v"(a[i]) : "v"(B))
#define A3(INS) asm volatile(INS " %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(B))
#define A4(INS) asm volatile(INS " %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(B), "v"(C))
#define A3X(INS, SUF) asm volatile(INS " %0, %1, %2 " SUF : "=v"(a[i]) : "v"(a[i]), "v"(B))
#define A2P(INS, SUF) asm volatile(INS " %0, %1 " SUF : "+v"(a[i]) : "v"(B))

// eight instructions in the proportions of the first sweep's loop: 2 plain adds, 2 packed minima, 1 three-input minimum,
// 1 permute, 1 population count, 1 cross-lane minimum (the figure printed is per instruction of the mix)
#define MIXED                                                                                                              \
    do {                                                                                                                   \
        if (i == 0 || i == 4) A3("v_add_u32"); else if (i == 1 || i == 5) A3("v_pk_min_u16"); else if (i == 2) A4("v_pk_minimum3_f16"); \
        else if (i == 3) A4("v_perm_b32"); else if (i == 6) A3("v_bcnt_u32_b32");                                          \
        else A3X("v_min_u32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf");                                                \
    } while (0)
#define OPS(X)                                                                                                             \
    X(0, "v_add_u32", A3("v_add_u32"))                                                                                     \
    X(1, "v_pk_minimum3_f16", A4("v_pk_minimum3_f16"))                                                                     \
    X(2, "v_pk_min_u16", A3("v_pk_min_u16"))                                                                               \
    X(3, "v_perm_b32", A4("v_perm_b32"))                                                                                   \
    X(4, "v_bitop3_b32", asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x48" : "=v"(a[i]) : "v"(a[i]), "v"(B), "v"(C))) \
    X(5, "v_bcnt_u32_b32", A3("v_bcnt_u32_b32"))                                                                           \
    X(6, "v_min_u32_dpp", A3X("v_min_u32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf"))                                    \
    X(7, "v_and_b32", A3("v_and_b32"))                                                                                     \
    X(8, "v_cndmask_b32", asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(B), "s"(mask)))     \
    X(9, "v_sub_u32", A3("v_sub_u32"))                                                                                     \
    X(10, "v_lshl_or_b32", A4("v_lshl_or_b32"))                                                                            \
    X(11, "v_mov_b32_dpp", A2P("v_mov_b32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf"))                                   \
    X(12, "v_alignbit_b32", A4("v_alignbit_b32"))                                                                          \
    X(13, "v_sad_u32", A4("v_sad_u32"))                                                                                    \
    X(14, "v_add3_u32", A4("v_add3_u32"))                                                                                  \
    X(15, "v_mov_b32", A2("v_mov_b32"))                                                                                    \
    X(16, "v_min_u32_sdwa", A3X("v_min_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0"))  \
    X(17, "v_mul_u32_u24", A3("v_mul_u32_u24"))                                                                            \
    X(18, "v_and_or_b32", A4("v_and_or_b32"))                                                                              \
    X(19, "v_pk_mad_u16", A4("v_pk_mad_u16"))                                                                              \
    X(20, "v_lshlrev_b32", A3("v_lshlrev_b32"))                                                                            \
    X(21, "v_min_u32", A3("v_min_u32"))                                                                                    \
    X(22, "v_cmp", asm volatile("v_cmp_lt_i32 vcc, %0, %1" ::"v"(a[i]), "v"(B) : "vcc"))                                   \
    X(23, "v_bfe_u32", A4("v_bfe_u32"))                                                                                    \
    X(24, "v_lshl_add_u32", A4("v_lshl_add_u32"))                                                                          \
    X(25, "v_mad_u32_u24", A4("v_mad_u32_u24"))                                                                            \
    X(26, "v_max_i32", A3("v_max_i32"))                                                                                    \
    X(27, "v_or_b32", A3("v_or_b32"))                                                                                      \
    X(28, "s_nop", asm volatile("s_nop 0"))                                                                                \
    X(29, "v_mul_i32_i24", A3("v_mul_i32_i24"))                                                                            \
    X(30, "v_pk_add_u16", A3("v_pk_add_u16"))                                                                              \
    X(31, "v_exp_f32", A2("v_exp_f32"))                                                                                    \
    X(32, "v_fma_f32", A4("v_fma_f32"))                                                                                    \
    X(33, "v_pk_fma_f32", asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[i & 3]) : "v"(p[i & 3]), "v"(p[(i + 1) & 3]), "v"(p[(i + 2) & 3]))) \
    X(34, "v_pk_min_u16_same_bank", asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 4) & 7])))                 \
    X(35, "mixed_stream_of_the_sweep", MIXED)

template <int OP>
__global__ __launch_bounds__(256) void k(u32* out, unsigned long long* clk, int iters, u32 seed)
{
    u32 a[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4];
    for (int i = 0; i < 8; i++) a[i] = (seed * (threadIdx.x + 1) + i * 0x9E3779B9u) & 0x03FF03FFu;
    for (int i = 0; i < 4; i++) p[i] = f2{(float)a[i] * 1e-9f, (float)a[i + 4] * 1e-9f};
    const unsigned long long mask = __builtin_amdgcn_read_exec() ^ (0x5555555555555555ull * seed);      // a select mask held in scalar registers
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define X(ID, NAME, STMT) if (OP == ID) LOOP(STMT)
    OPS(X)
#undef X
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    for (int i = 0; i < 4; i++) s ^= __float_as_uint(p[i].x + p[i].y);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static bool g_first = true;
template <int OP> void run(const char* name, int waves_per_simd)
{
    u32* d; unsigned long long* c;
    if (hipMalloc(&d, 8192 * 256 * 4) != hipSuccess || hipMalloc(&c, 8192 * 16) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(1); }
    const int blocks = 256 * waves_per_simd, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, c, 10, 3);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, c, iters, 3);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[8192 * 2];
    (void)hipMemcpy(h, c, blocks * 16, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; b++) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
    const double ghz = cyc / real * 0.1;       // s_memrealtime ticks at 100 MHz
    const double per_simd = (double)blocks * 4 * iters * 64.0 / 1024.0;       // wave-instructions each SIMD issued
    printf("%s\n  \"%s\": {\"ns\": %.4f, \"cycles\": %.3f, \"ghz\": %.3f}", g_first ? "" : ",", name, ms * 1e6 / per_simd, ms * 1e-3 * ghz * 1e9 / per_simd, ghz);
    g_first = false;
    (void)hipFree(d); (void)hipFree(c);
}

int main()
{
    printf("{");
    bool firstW = true;
    for (int w : {5, 4}) {
        printf("%s\n \"waves_per_simd_%d\": {", firstW ? "" : ",", w);
        firstW = false; g_first = true;
#define X(ID, NAME, STMT) run<ID>(NAME, w);
        OPS(X)
#undef X
        printf("\n }");
    }
    printf("\n}\n");
    return 0;
}
