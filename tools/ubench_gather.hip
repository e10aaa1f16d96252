// Micro-benchmark: random 4-byte table lookups per clock per CU from (a) LDS, conflict-free replicated layout,
// (b) a 1-KiB table in global memory (vector L1 hits), (c) both at once.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>   // 0 = LDS, 1 = L1 (global), 2 = both interleaved (4 LDS : 1 L1)
__global__ __launch_bounds__(1024) void k(const uint32_t *__restrict__ gtab, uint32_t *out, int iters)
{
    __shared__ uint32_t tab[8192];      // 256 entries x 32 bank copies
    for (int e = threadIdx.x; e < 8192; e += 1024) tab[e] = gtab[e >> 5] ;
    __syncthreads();
    const uint32_t lane4 = (threadIdx.x & 31u);
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t v;
            if (MODE == 0 || (MODE == 2 && (i & 3) != 3)) v = tab[((x[i] & 255u) << 5) | lane4];
            else v = gtab[x[i] & 255u];
            x[i] = (x[i] >> 8) ^ v;     // next index depends on the loaded value: 8 independent chains
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= x[i];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, const uint32_t *gtab, uint32_t *out)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, gtab, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, gtab, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double lookups = 256.0 * 1024 * 8 * iters;
    printf("%-12s %.3f ms  %.2f lookups/ns chip-wide = %.1f lookups/clk/CU @2.4GHz\n", name, ms, lookups / (ms * 1e6),
           lookups / (ms * 1e6) / 256 / 2.4);
}

int main()
{
    uint32_t h[256];
    for (int i = 0; i < 256; i++) h[i] = i * 2654435761u;
    uint32_t *gtab, *out;
    hipMalloc(&gtab, 1024); hipMalloc(&out, 256 * 1024 * 4);
    hipMemcpy(gtab, h, 1024, hipMemcpyHostToDevice);
    run<0>("LDS", gtab, out);
    run<1>("L1 global", gtab, out);
    run<2>("3 LDS : 1 L1", gtab, out);
    run<0>("LDS", gtab, out);
    return 0;
}
