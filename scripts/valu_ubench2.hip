// Issue cost of single VALU instructions on gfx950, pinned with inline asm so the compiler cannot fold the chains.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_ubench2.bin valu_ubench2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;

#define BODY3(INS)                                                                                  \
    for (int it = 0; it < iters; it++) {                                                            \
        _Pragma("unroll") for (int u = 0; u < 8; u++) {                                             \
            _Pragma("unroll") for (int i = 0; i < 8; i++)                                           \
                asm volatile(INS " %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 3) & 7]));      \
        }                                                                                           \
    }
#define BODY2(INS)                                                                                  \
    for (int it = 0; it < iters; it++) {                                                            \
        _Pragma("unroll") for (int u = 0; u < 8; u++) {                                             \
            _Pragma("unroll") for (int i = 0; i < 8; i++)                                           \
                asm volatile(INS " %0, %1" : "=v"(a[i]) : "v"(a[(i + 3) & 7]));                     \
        }                                                                                           \
    }
#define BODY4P(INS)                                                                                 \
    for (int it = 0; it < iters; it++) {                                                            \
        _Pragma("unroll") for (int u = 0; u < 8; u++) {                                             \
            _Pragma("unroll") for (int i = 0; i < 4; i++)                                           \
                asm volatile(INS " %0, %1, %2, %3" : "=v"(b[i]) : "v"(b[i]), "v"(b[(i + 1) & 3]), "v"(b[(i + 2) & 3])); \
        }                                                                                           \
    }
#define BODY3P(INS)                                                                                 \
    for (int it = 0; it < iters; it++) {                                                            \
        _Pragma("unroll") for (int u = 0; u < 8; u++) {                                             \
            _Pragma("unroll") for (int i = 0; i < 4; i++)                                           \
                asm volatile(INS " %0, %1, %2" : "=v"(b[i]) : "v"(b[i]), "v"(b[(i + 1) & 3]));      \
        }                                                                                           \
    }
#define BODY4(INS)                                                                                  \
    for (int it = 0; it < iters; it++) {                                                            \
        _Pragma("unroll") for (int u = 0; u < 8; u++) {                                             \
            _Pragma("unroll") for (int i = 0; i < 8; i++)                                           \
                asm volatile(INS " %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 3) & 7]), "v"(a[(i + 5) & 7])); \
        }                                                                                           \
    }

template <int OP>
__global__ __launch_bounds__(256) void k(u32* out, int iters, u32 seed)
{
    u32 a[8];
    for (int i = 0; i < 8; i++) a[i] = (seed * (threadIdx.x + 1) + i * 0x9E3779B9u) & 0x03FF03FFu;
    if (OP == 0) BODY3("v_add_u32")
    if (OP == 1) BODY3("v_pk_add_u16")
    if (OP == 2) BODY3("v_sub_u32")
    if (OP == 3) BODY3("v_pk_sub_i16")
    if (OP == 4) BODY3("v_min_u32")
    if (OP == 5) BODY3("v_pk_min_u16")
    if (OP == 6) BODY4("v_pk_minimum3_f16")
    if (OP == 7) BODY4("v_add3_u32")
    if (OP == 8) BODY4("v_perm_b32")
    if (OP == 9) BODY3("v_bcnt_u32_b32")
    if (OP == 10) BODY3("v_and_b32")
    if (OP == 11) BODY4("v_lshl_or_b32")
    if (OP == 12) BODY4("v_alignbit_b32")
    if (OP == 13) BODY4("v_min3_u32")
    if (OP == 14) BODY4("v_pk_mad_u16")
    if (OP == 15) BODY3("v_xor_b32")
    if (OP == 16) BODY4("v_and_or_b32")
    if (OP == 17) BODY4("v_bfe_u32")
    if (OP == 18) BODY3("v_lshlrev_b32")
    if (OP == 19) BODY4("v_mad_u32_u24")
    unsigned long long b[4];
    for (int i = 0; i < 4; i++) b[i] = ((unsigned long long)a[i] << 32) | a[i + 4];
    if (OP == 20) BODY2("v_exp_f32")
    if (OP == 21) BODY2("v_rcp_f32")
    if (OP == 22) BODY4P("v_pk_fma_f32")
    if (OP == 23) BODY3P("v_pk_mul_f32")
    if (OP == 24) BODY3P("v_pk_add_f32")
    if (OP == 25) BODY4("v_fma_f32")
    for (int i = 0; i < 4; i++) a[i] ^= (u32)b[i] ^ (u32)(b[i] >> 32);
    u32 s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> void run(const char* name, int waves_per_simd)
{
    u32* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    const int blocks = 256 * waves_per_simd, iters = 2000;   // one workgroup = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10, 3);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 3);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 * iters * 64.0 / 1024.0;   // wave-instructions per SIMD
    printf("%-22s %d waves/SIMD  %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / per_simd, ms * 1e-3 * 2.4e9 / per_simd);
    (void)hipFree(d);
}

int main()
{
    for (int w : {1, 8}) {
        switch (0) { default:
        run<0>("v_add_u32", w); run<1>("v_pk_add_u16", w); run<2>("v_sub_u32", w); run<3>("v_pk_sub_i16", w);
        run<4>("v_min_u32", w); run<5>("v_pk_min_u16", w); run<6>("v_pk_minimum3_f16", w); run<7>("v_add3_u32", w);
        run<8>("v_perm_b32", w); run<9>("v_bcnt_u32_b32", w); run<10>("v_and_b32", w); run<11>("v_lshl_or_b32", w);
        run<12>("v_alignbit_b32", w); run<13>("v_min3_u32", w); run<14>("v_pk_mad_u16", w); run<15>("v_xor_b32", w);
        run<16>("v_and_or_b32", w); run<17>("v_bfe_u32", w); run<18>("v_lshlrev_b32", w); run<19>("v_mad_u32_u24", w);
        run<20>("v_exp_f32", w); run<21>("v_rcp_f32", w); run<25>("v_fma_f32", w);
        run<22>("v_pk_fma_f32 (x0.5)", w); run<23>("v_pk_mul_f32 (x0.5)", w); run<24>("v_pk_add_f32 (x0.5)", w);
        }
    }
    return 0;
}
