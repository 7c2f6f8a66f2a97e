// Streaming-copy variants on gfx950: which shape reaches the HBM rate a float4 copy is known to reach (about 6.3 TB/s)?
// Build: hipcc --offload-arch=gfx950 -O3 -o copy_ubench.bin copy_ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n; i += stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const size_t j = i + (size_t)u * 256;
            if (j < n) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const size_t j = i + (size_t)u * 256;
            if (j < n) { if (NT) __builtin_nontemporal_store(v[u], dst + j); else dst[j] = v[u]; }
        }
    }
}

template <int UNROLL, bool NT> void run(const char* name, u32x4* s, u32x4* d, size_t bytes, int blocks)
{
    const size_t n = bytes / 16;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    copy_k<UNROLL, NT><<<blocks, 256>>>(s, d, n);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++) copy_k<UNROLL, NT><<<blocks, 256>>>(s, d, n);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks %6d  %7.1f GB/s (read + written)\n", name, blocks, 2.0 * bytes * 5 / (ms * 1e-3) / 1e9);
}

int main()
{
    const size_t bytes = (size_t)2 << 30;
    u32x4 *s, *d;
    (void)hipMalloc(&s, bytes); (void)hipMalloc(&d, bytes);
    (void)hipMemset(s, 1, bytes); (void)hipMemset(d, 2, bytes);
    for (int blocks : {2048, 4096, 8192, 16384, 65536}) {
        run<1, false>("plain, 1 per lane", s, d, bytes, blocks);
        run<1, true>("nontemporal, 1 per lane", s, d, bytes, blocks);
        run<4, false>("plain, 4 per lane", s, d, bytes, blocks);
        run<4, true>("nontemporal, 4 per lane", s, d, bytes, blocks);
        run<8, false>("plain, 8 per lane", s, d, bytes, blocks);
    }
    { // one element per thread, no loop
        const int blocks = (int)(bytes / 16 / 256);
        run<1, false>("plain, one shot", s, d, bytes, blocks);
        run<1, true>("nontemporal, one shot", s, d, bytes, blocks);
    }
    float ms; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++) (void)hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s               %7.1f GB/s\n", "hipMemcpyAsync D2D", 2.0 * bytes * 5 / (ms * 1e-3) / 1e9);
    return 0;
}
