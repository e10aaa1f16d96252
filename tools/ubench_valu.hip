// Micro-benchmark: VALU issue rate of v_bitop3_b32 / v_xor_b32 / v_perm_b32 in (a) a tight loop and
// (b) long straight-line code, at 1 and 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__device__ __forceinline__ uint32_t op(uint32_t a, uint32_t b, uint32_t c)
{
    if (OP == 0) return __builtin_amdgcn_bitop3_b32(a, b, c, 0x6a);
    if (OP == 1) return a ^ b;
    if (OP == 2) return __builtin_amdgcn_perm(a, b, 0x06020400u);
    if (OP == 3) return (a & b) | (c & ~b);   // v_bfi
    if (OP == 5) {   // SDWA byte insert: byte 3 of b -> byte 1 of a, other bytes of a preserved
        asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(a) : "v"(b));
        return a;
    }
    if (OP == 6) return (a & 0xff00u) | c;   // v_and_or_b32
    return a + b;
}

// 16 independent chains; UNROLL ops per chain per loop iteration
template <int OP, int UNROLL>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = seed * (i + 1) + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = op<OP>(v[i], v[(i + 5) & 15], v[(i + 11) & 15]);
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r ^= v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP, int UNROLL>
void run(const char *name, int waves_per_simd)
{
    uint32_t *d;
    hipMalloc(&d, 256 * 4 * 4096);
    const int blocks = 256 * waves_per_simd;          // 256-thread blocks: 4 waves = 1 per SIMD per block
    const long ops_per_thread = 16L * UNROLL;
    const int iters = (int)(4000000L / ops_per_thread);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, UNROLL>), dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, UNROLL>), dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)iters * ops_per_thread;           // per wave
    const double ns_per_instr_per_simd = ms * 1e6 / (wave_instr * waves_per_simd);
    printf("%-10s unroll %4d (%6ld instr/iter) waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n",
           name, UNROLL, ops_per_thread, waves_per_simd, ms, ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
    hipFree(d);
}

int main()
{
    for (int w = 1; w <= 4; w *= 2) {
        run<0, 4>("bitop3", w);
        run<1, 4>("xor", w);
        run<2, 4>("perm", w);
        run<3, 4>("bfi", w);
        run<5, 4>("sdwa_ins", w);
        run<6, 4>("and_or", w);
        run<0, 256>("bitop3", w);
        run<1, 256>("xor", w);
        run<0, 1024>("bitop3", w);
    }
    return 0;
}
