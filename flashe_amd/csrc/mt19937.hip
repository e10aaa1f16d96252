// np.random.random(n) on the device, bit for bit (the stochastic-rounding draws of
// _static_quantize_padding_asymmetric, jzf_quantize.py:55-67, come from NumPy's global MT19937 generator).
//
// The reference's quantiser is bit-exact only with NumPy's own stream, so round 2 drew the uniforms on the host and shipped them:
// 8 bytes per element over PCIe -- more than the fp32 input -- after ~5 ns per draw on one host core (50 ms per 1e7 elements, 25x
// the whole cipher round).  MT19937 is a 19937-bit linear recurrence: word k of the next 624-word block depends on words k, k + 1
// and k + 397 of the current one, so a block is three dependent phases of 227 + 227 + 170 independent words, and blocks are strictly
// sequential (jumping ahead costs more than generating).  One workgroup therefore walks the stream: ONE wave runs the twist chain
// (three LDS phases per block, wave-synchronous, no workgroup barrier on the critical path) eight blocks ahead into a ring, the other
// fifteen waves temper the finished blocks and emit doubles exactly as NumPy's mt19937_next_double does (a = next >> 5, b = next >> 6,
// (a * 2^26 + b) / 2^53), pairs that straddle a block boundary carried over.  The state the caller passes in (np.random.get_state())
// is advanced exactly as NumPy would have advanced it, so host draws continue the stream.
#include "ctx.h"

#include <cstring>

using flashe_host::fail;

namespace {

constexpr int kMtN = 624, kMtM = 397, kMtThreads = 1024;

__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__device__ __forceinline__ uint32_t mt_next(uint32_t cur, uint32_t nxt, uint32_t far)
{
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ double mt_double(uint32_t a, uint32_t b)
{
    return (static_cast<double>(a >> 5) * 67108864.0 + static_cast<double>(b >> 6)) / 9007199254740992.0;
}

// One 624-word block from the previous one, by ONE wave (no workgroup barrier on the critical path): three dependent phases of at
// most four words per lane; LDS operations of one wave execute in order, the fences only stop the compiler from reordering them.
__device__ __forceinline__ void mt_twist_wave(const uint32_t *__restrict__ o, uint32_t *__restrict__ nw, int lane)
{
    constexpr int D = kMtN - kMtM;                   // 227
    // a lane owns FOUR CONSECUTIVE words of a phase, so that its reads and writes merge into 8- and 16-byte LDS accesses (a lone
    // wave issues LDS instructions slowly: the instruction count, not the bytes, is what a phase costs)
    {
        const int k = 4 * lane;                      // 0 .. 224 (lane 56 owns 224, 225, 226)
        if (k < D) {
            uint32_t c[5], f[4], r[4];
#pragma unroll
            for (int j = 0; j < 5; j++) c[j] = o[k + j];
#pragma unroll
            for (int j = 0; j < 4; j++) f[j] = o[k + kMtM + j < kMtN ? k + kMtM + j : kMtN - 1];
#pragma unroll
            for (int j = 0; j < 4; j++) r[j] = mt_next(c[j], c[j + 1], f[j]);
#pragma unroll
            for (int j = 0; j < 4; j++) if (k + j < D) nw[k + j] = r[j];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        const int k = D + 4 * lane;                  // 227 .. 451 (lane 56 owns 451, 452, 453)
        if (k < 2 * D) {
            uint32_t c[5], f[4], r[4];
#pragma unroll
            for (int j = 0; j < 5; j++) c[j] = o[k + j];
#pragma unroll
            for (int j = 0; j < 4; j++) f[j] = nw[k - D + j];
#pragma unroll
            for (int j = 0; j < 4; j++) r[j] = mt_next(c[j], c[j + 1], f[j]);
#pragma unroll
            for (int j = 0; j < 4; j++) if (k + j < 2 * D) nw[k + j] = r[j];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        const int k = 2 * D + 4 * lane;              // 454 .. 622 (lane 42 owns 622, 623)
        if (k < kMtN) {
            uint32_t c[5], f[4], r[4];
#pragma unroll
            for (int j = 0; j < 5; j++) c[j] = k + j < kMtN ? o[k + j] : nw[0];          // word 624 of the old block = word 0 of the NEW one
#pragma unroll
            for (int j = 0; j < 4; j++) f[j] = nw[k - D + j < kMtN ? k - D + j : kMtN - 1];
#pragma unroll
            for (int j = 0; j < 4; j++) r[j] = mt_next(c[j], c[j + 1], f[j]);
#pragma unroll
            for (int j = 0; j < 4; j++) if (k + j < kMtN) nw[k + j] = r[j];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// state: 624 key words + the position (0..624) of the next unused word, updated in place.
// Wave 0 produces batches of kMtBatch blocks into one half of a two-batch ring while waves 1..15 temper the other half and emit
// its doubles: the sequential twist chain (~0.2 us per block for one wave) is the critical path, everything else hides under it.
// Block numbering: block 0 is the state as passed in (words pos.. are unused), block b + 1 = twist(block b); word position
// P = 624 * block + offset; double i uses positions pos + 2 i and pos + 2 i + 1 and is emitted by the batch that holds its SECOND
// word (the first word of a pair that straddles two batches travels through `carry`).
constexpr int kMtBatch = 8;

__global__ __launch_bounds__(kMtThreads) void mt19937_random_kernel(uint32_t *__restrict__ state, uint64_t n, double *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint32_t ring[2][kMtBatch][kMtN];
    __shared__ uint32_t carry[2];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const uint64_t pos0 = state[kMtN];
    const uint64_t end = pos0 + 2 * n;                               // one past the last word position used
    const uint64_t n_blocks = (end + kMtN - 1) / kMtN;               // blocks 0 .. n_blocks - 1 are needed
    const uint64_t n_batches = (n_blocks + kMtBatch - 1) / kMtBatch;
    constexpr uint64_t kBW = static_cast<uint64_t>(kMtBatch) * kMtN; // words per batch

    auto produce = [&](uint64_t g) {                                 // wave 0: the blocks of batch g
        for (int j = 0; j < kMtBatch; j++) {
            const uint64_t bi = g * kMtBatch + j;
            if (bi >= n_blocks) break;
            uint32_t *dst = ring[g & 1][j];
            if (bi == 0) {
                for (int k = lane; k < kMtN; k += 64) dst[k] = state[k];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
                const uint32_t *src = j ? ring[g & 1][j - 1] : ring[(g - 1) & 1][kMtBatch - 1];
                mt_twist_wave(src, dst, lane);
            }
        }
    };
    auto consume = [&](uint64_t g) {                                 // waves 1..15: the doubles whose second word lies in batch g
        const uint64_t base = g * kBW;
        const uint32_t(*blk)[kMtN] = ring[g & 1];
        constexpr int kConsumers = kMtThreads - 64, kPairs = kMtN / 2;              // at most 312 second words per block
        for (int u = t - 64; u < kMtBatch * kPairs; u += kConsumers) {
            const int j = u / kPairs, q = u - j * kPairs;                           // block of the batch, pair slot of the block
            const uint64_t b0 = base + static_cast<uint64_t>(j) * kMtN;             // position of the block's word 0
            if (b0 >= end) break;
            // second words sit at the block offsets x with (b0 + x - pos0) odd
            const uint32_t x = 2u * q + (((b0 - pos0) & 1u) ? 0u : 1u);
            const uint64_t P2 = b0 + x;
            if (x >= kMtN || P2 <= pos0 || P2 >= end) continue;
            const uint32_t wb = mt_temper(blk[j][x]);
            const uint32_t wa = x ? mt_temper(blk[j][x - 1]) : j ? mt_temper(blk[j - 1][kMtN - 1]) : carry[g & 1];
            out[(P2 - pos0 - 1) >> 1] = mt_double(wa, wb);
        }
        // the batch ends with the FIRST word of a pair: hand it to the next batch
        if (t == 64 && base + kBW < end && ((base + kBW - 1 - pos0) & 1u) == 0) carry[(g + 1) & 1] = mt_temper(blk[kMtBatch - 1][kMtN - 1]);
    };

    if (wave == 0) produce(0);
    __syncthreads();
    for (uint64_t g = 0; g < n_batches; g++) {
        if (wave == 0) { if (g + 1 < n_batches) produce(g + 1); }
        else consume(g);
        __syncthreads();
    }
    // the generator state NumPy would be left with: the last block, positioned behind the last word used
    const uint64_t last = n_blocks - 1;
    const uint32_t *fin = ring[(last / kMtBatch) & 1][last % kMtBatch];
    for (int k = t; k < kMtN; k += kMtThreads) state[k] = fin[k];
    if (t == 0) state[kMtN] = static_cast<uint32_t>(end - last * kMtN);
}

}  // namespace

extern "C" int flashe_mt19937_random_dev(flashe_ctx *ctx, uint32_t key[624], uint32_t *pos, uint64_t n, double *u_dev)
{
    CHECK_CTX(ctx);
    if (!key || !pos || *pos > 624u || (n && !u_dev)) return fail(ctx, FLASHE_EINVAL, "flashe_mt19937_random_dev: bad arguments");
    if (n == 0) return FLASHE_OK;
    if (ctx->capturing) return fail(ctx, FLASHE_EINVAL, "flashe_mt19937_random_dev is synchronous (it returns the advanced state): not capturable");
    uint32_t host[625];
    memcpy(host, key, 624 * sizeof(uint32_t));
    host[624] = *pos;
    uint32_t *st = nullptr;
    HIP_TRY(ctx, hipMalloc(&st, sizeof host));
    hipError_t e = hipMemcpyAsync(st, host, sizeof host, hipMemcpyHostToDevice, ctx->env.stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mt19937_random_kernel, dim3(1), dim3(kMtThreads), 0, ctx->env.stream, st, n, u_dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, st, sizeof host, hipMemcpyDeviceToHost, ctx->env.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->env.stream);
    (void)hipFree(st);
    HIP_TRY(ctx, e);
    memcpy(key, host, 624 * sizeof(uint32_t));
    *pos = host[624];
    return FLASHE_OK;
}
