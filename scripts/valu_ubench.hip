// Throughput of the VALU instruction kinds the SGM inner loop is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_ubench valu_ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 ash(u32 v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ u32 hsu(h2 v) { return __builtin_bit_cast(u32, v); }
__device__ __forceinline__ us2 asv(u32 v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ u32 asu(us2 v) { return __builtin_bit_cast(u32, v); }

// three-input minimum of packed u16 pairs below 0x7C00 (bit patterns of non-negative finite halfs order like integers)
__device__ __forceinline__ u32 pkmin3(u32 a, u32 b, u32 c)
{
    u32 r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__global__ void check_min3(u32* bad, u32 seed)
{
    u32 s = seed + (blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B9u, n = 0;
    for (int it = 0; it < 4096; it++) {
        u32 v[3];
        for (int j = 0; j < 3; j++) {
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            u32 lo = s & 0x7FFF, hi = (s >> 16) & 0x7FFF;
            if (it & 1) { lo &= 0x3FF; hi &= 0xFF; }          // small integers: subnormal halfs
            if (lo > 0x7C00) lo = 0x7C00;
            if (hi > 0x7C00) hi = 0x7C00;
            v[j] = lo | (hi << 16);
        }
        const u32 r = pkmin3(v[0], v[1], v[2]);
        const u32 elo = min(min(v[0] & 0xFFFF, v[1] & 0xFFFF), v[2] & 0xFFFF), ehi = min(min(v[0] >> 16, v[1] >> 16), v[2] >> 16);
        if (r != (elo | (ehi << 16))) n++;
    }
    if (n) atomicAdd(bad, n);
}

template <int OP>
__global__ __launch_bounds__(256) void k(u32* out, int iters, u32 seed)
{
    u32 a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;
    const u32 c = seed | 0x00010001u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = a[i] + a[(i + 3) & 7];                                                     // v_add_u32
                if (OP == 1) a[i] = asu(asv(a[i]) + asv(a[(i + 3) & 7]));                                      // v_pk_add_u16
                if (OP == 2) a[i] = asu(__builtin_elementwise_min(asv(a[i]), asv(a[(i + 3) & 7]))) ^ c;        // v_pk_min_u16
                if (OP == 3) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i + 1) & 7], 16);           // v_alignbit_b32
                if (OP == 4) a[i] = __popc(a[i]) + c;                                             // v_bcnt_u32_b32
                if (OP == 5) a[i] = min(a[i], (u32)__builtin_amdgcn_mov_dpp((int)a[i], 0xB1, 0xF, 0xF, true)) + 1;  // v_min dpp (+add)
                if (OP == 6) a[i] = __builtin_amdgcn_perm(a[i], c, 0x05040100u) ^ (u32)i;          // v_perm_b32 (+xor)
                if (OP == 7) a[i] = (a[i] & a[(i + 3) & 7]) | a[(i + 5) & 7];                                         // v_and_or
                if (OP == 8) a[i] = min(a[i], a[(i + 3) & 7]) + 1;                                             // v_min_u32
                if (OP == 9) a[i] = hsu(ash(a[i]) + ash(a[(i + 3) & 7]));                                       // v_pk_add_f16
                if (OP == 10) a[i] = hsu(__builtin_elementwise_min(ash(a[i]), ash(a[(i + 3) & 7])));             // v_pk_min_f16
                if (OP == 11) a[i] = __float_as_uint(__uint_as_float(a[i]) + __uint_as_float(a[(i + 3) & 7]));   // v_add_f32
                if (OP == 12) a[i] = __float_as_uint(fminf(__uint_as_float(a[i]), __uint_as_float(a[(i + 3) & 7])));  // v_min_f32
                if (OP == 13) a[i] = __float_as_uint(fmaf(__uint_as_float(a[i]), __uint_as_float(c), __uint_as_float(a[(i + 3) & 7])));  // v_fma_f32
                if (OP == 14) a[i] = hsu(ash(a[i]) * ash(c) + ash(a[(i + 3) & 7]));                               // v_pk_fma_f16
                if (OP == 15) a[i] = min(min(a[i], a[(i + 3) & 7]), a[(i + 5) & 7]);                              // v_min3_u32
                if (OP == 16) a[i] = a[i] + a[(i + 3) & 7] + a[(i + 5) & 7];                                      // v_add3_u32
                if (OP == 17) a[i] = asu(asv(a[i]) * asv(c) + asv(a[(i + 3) & 7]));                               // v_pk_mad_u16
                if (OP == 18) a[i] = pkmin3(a[i], a[(i + 3) & 7], a[(i + 5) & 7]);                                // v_pk_minimum3_f16
            }
        }
    }
    u32 s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP> void run(const char* name, int instr_per_inner)
{
    u32* d; hipMalloc(&d, 4096 * 256 * 4);
    const int blocks = 256 * 8, iters = 2000;   // 8 workgroups (32 waves) per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * 64.0 * instr_per_inner;   // wave-instructions
    double per_simd = winstr / 1024.0;
    printf("%-28s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz (%d instr counted per op)\n", name, ms,
           ms * 1e-3 * 2.4e9 / per_simd, instr_per_inner);
    hipFree(d);
}

int main()
{
    run<0>("v_add_u32", 1);
    run<1>("v_pk_add_u16", 1);
    run<2>("v_pk_min_u16 + v_xor", 2);
    run<3>("v_alignbit_b32", 1);
    run<4>("v_bcnt_u32_b32 (+add fused)", 1);
    run<5>("v_min_u32_dpp + v_add", 2);
    run<6>("v_perm_b32 + v_xor", 2);
    run<7>("v_and_or_b32", 1);
    run<8>("v_min_u32 + v_add", 2);
    run<9>("v_pk_add_f16", 1);
    run<10>("v_pk_min_f16", 1);
    run<11>("v_add_f32", 1);
    run<12>("v_min_f32", 1);
    run<13>("v_fma_f32", 1);
    run<14>("v_pk_fma_f16", 1);
    run<15>("v_min3_u32", 1);
    run<16>("v_add3_u32", 1);
    run<17>("v_pk_mad_u16", 1);
    run<18>("v_pk_minimum3_f16", 1);
    {
        u32* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
        check_min3<<<256, 256>>>(bad, 12345u);
        u32 h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("v_pk_minimum3_f16 as a u16 min3 (values <= 0x7C00, subnormals included): %u mismatches of %u\n", h, 256u * 256u * 4096u);
        hipFree(bad);
    }
    return 0;
}
