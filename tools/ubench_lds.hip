// Micro-benchmark: the AES kernel's inner pattern without the AES -- per block-round 16 data-dependent ds_read_b32 from the
// conflict-free 32-copy layout, 8 x v_bitop3 (column XORs) and 16 x v_perm_b32 (next addresses) -- for NB independent blocks
// per lane and 4 / 8 / 16 waves per CU: how many lookups per clock does a CU sustain (LDS peak: 32)?
// ADDR = 1 (round 4): the addresses without v_perm -- three shifts per state word (two-source ops) and one v_bitop3 (x & 0xff00 | lane
// register) per lookup, the byte that already sits at bits 8..15 needing no shift: 12 x 1.9 + 16 x 2.5 VALU cycles against 16 x 4.3.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint8_t lds_u8;

template <int NB, int THREADS, int ADDR>
__global__ __launch_bounds__(THREADS) void k(uint32_t *out, int iters, uint32_t seed)
{
    __shared__ uint32_t tab[32768];
    for (int e = threadIdx.x; e < 32768; e += THREADS) tab[e] = e * 2654435761u ^ seed;
    __syncthreads();
    const lds_u8 *base = (const lds_u8 *)(lds_u32 *)tab;
    const uint32_t la = (threadIdx.x & 31u) * 4u, lb = la | 0x10000u;
    uint32_t s[NB][4];
#pragma unroll
    for (int q = 0; q < NB; q++)
#pragma unroll
        for (int j = 0; j < 4; j++) s[q][j] = seed * (4 * q + j + 1) + threadIdx.x * 2654435761u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < NB; q++) {
            uint32_t v[16];
            if (ADDR == 1) {
                const uint32_t m = 0xff00u;
                uint32_t hi[4], mid[4], lo[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { hi[j] = s[q][j] >> 16; mid[j] = s[q][j] >> 8; lo[j] = s[q][j] << 8; }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    v[4 * j + 0] = *reinterpret_cast<const lds_u32 *>(base + __builtin_amdgcn_bitop3_b32(hi[j], m, la, 0xea));
                    v[4 * j + 1] = *reinterpret_cast<const lds_u32 *>(base + __builtin_amdgcn_bitop3_b32(mid[(j + 1) & 3], m, la, 0xea) + 128);
                    v[4 * j + 2] = *reinterpret_cast<const lds_u32 *>(base + __builtin_amdgcn_bitop3_b32(s[q][(j + 2) & 3], m, lb, 0xea));
                    v[4 * j + 3] = *reinterpret_cast<const lds_u32 *>(base + __builtin_amdgcn_bitop3_b32(lo[(j + 3) & 3], m, lb, 0xea) + 128);
                }
            } else
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v[4 * j + 0] = *reinterpret_cast<const lds_u32 *>(base + __builtin_amdgcn_perm(s[q][j], la, 0x0c020700u));
                v[4 * j + 1] = *reinterpret_cast<const lds_u32 *>(base + __builtin_amdgcn_perm(s[q][(j + 1) & 3], la, 0x0c020600u) + 128);
                v[4 * j + 2] = *reinterpret_cast<const lds_u32 *>(base + __builtin_amdgcn_perm(s[q][(j + 2) & 3], lb, 0x0c020500u));
                v[4 * j + 3] = *reinterpret_cast<const lds_u32 *>(base + __builtin_amdgcn_perm(s[q][(j + 3) & 3], lb, 0x0c020400u) + 128);
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
                s[q][j] = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(v[4 * j], v[4 * j + 1], v[4 * j + 2], 0x96), v[4 * j + 3], seed + j, 0x96);
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int q = 0; q < NB; q++) r ^= s[q][0] ^ s[q][1] ^ s[q][2] ^ s[q][3];
    out[blockIdx.x * THREADS + threadIdx.x] = r;
}

template <int NB, int THREADS, int ADDR = 0>
void run()
{
    uint32_t *d;
    (void)hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 20000 / NB;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NB, THREADS, ADDR>), dim3(256), dim3(THREADS), 0, 0, d, 10, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NB, THREADS, ADDR>), dim3(256), dim3(THREADS), 0, 0, d, iters, 1u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double lookups = 256.0 * THREADS * 16 * NB * iters;
    printf("%s blocks per lane %d, waves per CU %2d: %.3f ms  %.1f lookups/clk/CU @2.4GHz (LDS peak 32)\n", ADDR ? "shift+bitop3" : "v_perm      ", NB, THREADS / 64, ms,
           lookups / (ms * 1e6) / 256 / 2.4);
    (void)hipFree(d);
}

int main()
{
    run<1, 1024>(); run<2, 1024>(); run<3, 1024>(); run<4, 1024>();
    run<1, 1024, 1>(); run<2, 1024, 1>(); run<3, 1024, 1>(); run<4, 1024, 1>();
    run<2, 512, 1>();
    run<1, 512>(); run<2, 512>(); run<4, 512>();
    run<2, 256>(); run<4, 256>();
    return 0;
}
