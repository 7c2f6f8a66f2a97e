// Issue cost of further single VALU instructions on gfx950 (DPP / SDWA forms, moves, selects, float minima, s_nop) and the
// shader clock the chip holds while every SIMD issues them.  Same method as valu_ubench2.hip: chains pinned with inline asm.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_ubench3.bin valu_ubench3.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;

#define LOOP(STMT)                                                                                  \
    for (int it = 0; it < iters; it++) {                                                            \
        _Pragma("unroll") for (int u = 0; u < 8; u++) {                                             \
            _Pragma("unroll") for (int i = 0; i < 8; i++) { STMT; }                                 \
        }                                                                                           \
    }
#define A3(INS) asm volatile(INS " %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 3) & 7]))
#define A2(INS) asm volatile(INS " %0, %1" : "=v"(a[i]) : "v"(a[(i + 3) & 7]))
#define A4(INS) asm volatile(INS " %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 3) & 7]), "v"(a[(i + 5) & 7]))

template <int OP>
__global__ __launch_bounds__(256) void k(u32* out, unsigned long long* clk, int iters, u32 seed)
{
    u32 a[8];
    for (int i = 0; i < 8; i++) a[i] = (seed * (threadIdx.x + 1) + i * 0x9E3779B9u) & 0x03FF03FFu;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (OP == 0) LOOP(A2("v_mov_b32"))
    if (OP == 1) LOOP(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 3) & 7])))
    if (OP == 2) LOOP(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 3) & 7])))
    if (OP == 3) LOOP(asm volatile("v_min_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 3) & 7])))
    if (OP == 4) LOOP(asm volatile("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 3) & 7])))
    if (OP == 5) LOOP(asm volatile("s_nop 1"))
    if (OP == 6) LOOP(A3("v_or_b32"))
    if (OP == 7) LOOP(A4("v_mad_i32_i24"))
    if (OP == 8) LOOP(asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x28" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 3) & 7]), "v"(a[(i + 5) & 7])))
    if (OP == 9) LOOP(asm volatile("v_pk_min_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 3) & 7])))
    if (OP == 10) LOOP(A4("v_lshl_add_u32"))
    if (OP == 11) LOOP(A3("v_min_f32"))
    if (OP == 12) LOOP(A3("v_pk_min_f16"))
    if (OP == 13) LOOP(A3("v_min_u16"))
    if (OP == 14) LOOP(A3("v_min_f16"))
    if (OP == 15) LOOP(A3("v_add_f32"))
    if (OP == 16) LOOP(A3("v_max_f32"))
    if (OP == 17) LOOP(A4("v_min3_f32"))
    if (OP == 18) LOOP(A3("v_sub_u16"))
    if (OP == 19) LOOP(A3("v_add_u16"))
    if (OP == 20) LOOP(A4("v_sad_u8"))
    if (OP == 21) LOOP(A3("v_and_b32"))      // reference: fast class
    if (OP == 22) LOOP(A3("v_pk_min_u16"))   // reference: slow class
    if (OP == 23) LOOP(A4("v_bfi_b32"))
    if (OP == 24) LOOP(A4("v_med3_u32"))
    if (OP == 25) LOOP(A3("v_subrev_u32"))
    if (OP == 26) LOOP(asm volatile("v_add_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 3) & 7])))
    if (OP == 27) LOOP(asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 3) & 7])))
    if (OP == 28) LOOP(A3("v_max_u32"))
    if (OP == 29) LOOP(A4("v_xad_u32"))
    if (OP == 30) LOOP(A3("v_lshrrev_b32"))
    if (OP == 31) LOOP(A3("v_mul_u32_u24"))
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u32 s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP> void run(const char* name, int waves_per_simd)
{
    u32* d; unsigned long long* c;
    (void)hipMalloc(&d, 8192 * 256 * 4); (void)hipMalloc(&c, 8192 * 16);
    const int blocks = 256 * waves_per_simd, iters = 4000;
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
    const double per_simd = (double)blocks * 4 * iters * 64.0 / 1024.0;
    printf("%-22s %d waves/SIMD %8.3f ms  %.2f ns per wave-instruction per SIMD = %.2f shader cycles at the measured %.2f GHz\n", name,
           waves_per_simd, ms, ms * 1e6 / per_simd, ms * 1e-3 * ghz * 1e9 / per_simd, ghz);
    (void)hipFree(d); (void)hipFree(c);
}

int main()
{
    for (int w : {8, 4}) {
        run<21>("v_and_b32", w); run<22>("v_pk_min_u16", w);
        run<0>("v_mov_b32", w); run<1>("v_cndmask_b32", w); run<2>("v_mov_b32_dpp", w); run<3>("v_min_u32_dpp", w);
        run<4>("v_min_u32_sdwa", w); run<5>("s_nop 1", w); run<6>("v_or_b32", w); run<7>("v_mad_i32_i24", w);
        run<8>("v_bitop3_b32", w); run<9>("v_pk_min_u16 op_sel", w); run<10>("v_lshl_add_u32", w); run<11>("v_min_f32", w);
        run<12>("v_pk_min_f16", w); run<13>("v_min_u16", w); run<14>("v_min_f16", w); run<15>("v_add_f32", w);
        run<16>("v_max_f32", w); run<17>("v_min3_f32", w); run<18>("v_sub_u16", w); run<19>("v_add_u16", w);
        run<20>("v_sad_u8", w); run<23>("v_bfi_b32", w); run<24>("v_med3_u32", w); run<25>("v_subrev_u32", w);
        run<26>("v_add_u32_dpp", w); run<27>("v_add_u32_sdwa", w); run<28>("v_max_u32", w); run<29>("v_xad_u32", w);
        run<30>("v_lshrrev_b32", w); run<31>("v_mul_u32_u24", w);
    }
    return 0;
}
