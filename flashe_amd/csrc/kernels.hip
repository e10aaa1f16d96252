// gfx950 (MI355X / CDNA4) kernels of the FLASHE cipher engine.
//
// Everything here is integer / byte work, bounded either by the AES-256 PRF rate (LDS
// T-table lookups + VALU) or by HBM streaming -- no MFMA.  Design notes:
//   * PRF = AES-256 over (iter | idx | counter) blocks.  The four T-tables live in LDS,
//     replicated 32x so that every lane of a 32-lane LDS service group owns a private bank
//     (ds_read_b32 banks = (addr/4) mod 32): entry stride 256 B, two tables per 64-KiB half.
//     A lookup address is built by ONE v_perm_b32 (state byte -> address byte 1, lane offset ->
//     byte 0, half select -> byte 2) and the table choice rides in the ds_read immediate offset.
//   * one lane = one element (b > 64) or one AES block = m elements (b <= 64); 16 B per lane
//     on every global access of the wide path, so loads and stores are full 1-KiB wave bursts.
//   * persistent launch: one 1024-thread workgroup per CU (128 KiB of LDS tables), tiles dealt
//     round-robin; there is no inter-workgroup reuse, so no XCD remap is needed.
#include "device_common.h"

namespace flashe {

struct IdxLists {
    uint32_t add[kMaxIdx];
    uint32_t minus[kMaxIdx];
};

struct PrfParams {
    const uint64_t *in;    // may be null
    uint64_t *out;
    const uint32_t *te0;
    uint64_t n;            // length of the WHOLE vector (defines the chunking)
    uint64_t first, count; // this launch covers global elements [first, first + count); in/out are
                           // indexed by (element - first)
    uint64_t blk_first, blk_count;   // b <= 64: AES blocks intersecting that range
    uint64_t mask_lo, mask_hi;
    uint32_t iter;
    int in_limbs;
    int n_add, n_minus;
    // b <= 64 only:
    uint32_t n_jobs;
    int b, m;
    Codec cq;              // optional fused quantise front end / unquantise back end
};

// ---- b > 64: one element per lane, counter = element index (m = 1 makes chunks irrelevant) ----
// Generic prefix lists (dropout decrypts, single-mask decrypts).  One add and at most one minus prefix -- every
// encrypt, the no-dropout decrypt, the mask streams -- go through prf_wide_batch_kernel below.
__global__ __launch_bounds__(kPrfThreads) void prf_wide_kernel(const RoundKeys rk, const PrfParams p, const IdxLists lists)
{
    const uint32_t iter = p.iter + p.te0[kIterShiftWord];
    __shared__ uint32_t tab[kTabWords];
    fill_tables(tab, p.te0);
    const LaneRegs lr = lane_regs(tab);
    const u128 mask = (static_cast<u128>(p.mask_hi) << 64) | p.mask_lo;

    for (uint64_t base = static_cast<uint64_t>(blockIdx.x) * kPrfThreads; base < p.count;
         base += static_cast<uint64_t>(gridDim.x) * kPrfThreads) {
        const uint64_t e = base + threadIdx.x;          // index into in / out
        if (e >= p.count) continue;
        const uint64_t j = p.first + e;                 // global element = PRF counter (m = 1)
        u128 acc = 0;
        if (p.cq.x) acc = codec_quantize(p.cq, e);
        else if (p.in) acc = p.in_limbs == 2 ? ld128(p.in + 2 * e) : static_cast<u128>(p.in[e]);
        {
            int k = 0;
            for (; k + 1 < p.n_add; k += 2) {
                uint32_t s[2][4];
                set_block(s[0], iter, lists.add[k], j);
                set_block(s[1], iter, lists.add[k + 1], j);
                aes256_encrypt<2>(rk, lr, s);
                acc += words_to_u128(s[0]);
                acc += words_to_u128(s[1]);
            }
            if (k < p.n_add) {
                uint32_t s[1][4];
                set_block(s[0], iter, lists.add[k], j);
                aes256_encrypt<1>(rk, lr, s);
                acc += words_to_u128(s[0]);
            }
            k = 0;
            for (; k + 1 < p.n_minus; k += 2) {
                uint32_t s[2][4];
                set_block(s[0], iter, lists.minus[k], j);
                set_block(s[1], iter, lists.minus[k + 1], j);
                aes256_encrypt<2>(rk, lr, s);
                acc -= words_to_u128(s[0]);
                acc -= words_to_u128(s[1]);
            }
            if (k < p.n_minus) {
                uint32_t s[1][4];
                set_block(s[0], iter, lists.minus[k], j);
                aes256_encrypt<1>(rk, lr, s);
                acc -= words_to_u128(s[0]);
            }
        }
        if (p.cq.fout) p.cq.fout[e] = codec_unquantize(p.cq, e, acc & mask);
        else st128(p.out + 2 * e, acc & mask);
    }
}

// ---- one add / at most one minus prefix: many independent jobs in one launch ----
// One launch = up to kMaxBatch jobs; job v covers global elements [first, first + count) of an n-element
// vector: out[k] = in[k] + term(iter, add, first + k) - [DBL] term(iter, minus, first + k), in = 0 when null.
// encrypt: (idx, idx + 1, pt); telescoped decrypt: (C, 0, aggregate); mask precompute: in = null.
// Tiling: job v starts with big[v] BIG tiles (THREADS * 4 elements; every wave walks 4 x 64 consecutive elements, so
// its 256 counters share bytes 1..3 and the wave-uniform part of rounds 1-2 is computed once per 256 elements) and
// finishes with SMALL tiles (THREADS elements).  Tile index space of the launch: all big tiles of all jobs, then all
// small tiles; workgroup i takes tiles i, i + grid, ...  The host makes the number of big tiles a multiple of the
// grid, so every workgroup gets the same number of them and the remainder is balanced in units of 1024 elements.
struct JobTable {
    uint32_t add[kMaxBatch], minus[kMaxBatch];
    uint64_t first[kMaxBatch], count[kMaxBatch];
    uint64_t big_end[kMaxBatch], small_end[kMaxBatch];      // running totals of big / small tiles
    const uint64_t *in[kMaxBatch];
    uint64_t *out[kMaxBatch];
    uint64_t in_stride[kMaxBatch];      // limbs between the n_in input vectors of a summed input
    uint64_t *sum_out[kMaxBatch];       // optional: where the summed input goes
    uint8_t in_limbs[kMaxBatch], n_in[kMaxBatch];
    __device__ uint32_t add_of(int v) const { return add[v]; }
    __device__ uint32_t minus_of(int v) const { return minus[v]; }
    __device__ uint64_t first_of(int v) const { return first[v]; }
    __device__ uint64_t count_of(int v) const { return count[v]; }
    __device__ uint64_t big_end_of(int v) const { return big_end[v]; }
    __device__ uint64_t small_end_of(int v) const { return small_end[v]; }
    __device__ const uint64_t *in_of(int v) const { return in[v]; }
    __device__ uint64_t *out_of(int v) const { return out[v]; }
    __device__ int in_limbs_of(int v) const { return in_limbs[v]; }
    __device__ int n_in_of(int v) const { return n_in[v]; }
    __device__ uint64_t in_stride_of(int v) const { return in_stride[v]; }
    __device__ uint64_t *sum_out_of(int v) const { return sum_out[v]; }
};

// Compact table for the common batch: up to kMaxUniform whole vectors of EQUAL length (the clients a simulation hosts, the
// layers of a model) -- four words per job instead of eleven, so 128 of them fit the kernel-argument block and a hundred
// LeNet-sized encrypts are one launch.
constexpr int kMaxUniform = kMaxUniformBatch;
struct UniformJobTable {
    uint32_t add[kMaxUniform], minus[kMaxUniform];
    const uint64_t *in[kMaxUniform];
    uint64_t *out[kMaxUniform];
    uint64_t count, big_per, small_per;   // elements, big tiles and small tiles of every job
    int in_limbs;
    __device__ uint32_t add_of(int v) const { return add[v]; }
    __device__ uint32_t minus_of(int v) const { return minus[v]; }
    __device__ uint64_t first_of(int) const { return 0; }
    __device__ uint64_t count_of(int) const { return count; }
    __device__ uint64_t big_end_of(int v) const { return static_cast<uint64_t>(v + 1) * big_per; }
    __device__ uint64_t small_end_of(int v) const { return static_cast<uint64_t>(v + 1) * small_per; }
    __device__ const uint64_t *in_of(int v) const { return in[v]; }
    __device__ uint64_t *out_of(int v) const { return out[v]; }
    __device__ int in_limbs_of(int) const { return in_limbs; }
    __device__ int n_in_of(int) const { return 1; }
    __device__ uint64_t in_stride_of(int) const { return 0; }
    __device__ uint64_t *sum_out_of(int) const { return nullptr; }
};
constexpr int kBigEpl = 4;

// KIND 0 / 1 / 3 only name the instantiation (one job / several jobs per launch / a lone latency-bound job in 256-thread
// workgroups), so that profiles list them as separate kernels; KIND 2 additionally compiles the summed
// input (reduce fused in): its operands are loaded before the AES rounds and added after them, so their HBM
// latency hides under the lookups of the same element.
constexpr int kSumRegs = 10;            // operands held in registers across the rounds (more are added up front)
template <bool DBL, int THREADS, int KIND, class Table>
__global__ __launch_bounds__(THREADS) void prf_wide_batch_kernel(const RoundKeys rk, const Table tb, int n_vec, uint64_t n,
                                                                     uint32_t iter0, uint64_t mask_lo, uint64_t mask_hi,
                                                                     const uint32_t *__restrict__ te0)
{
    const uint32_t iter = iter0 + te0[kIterShiftWord];
    __shared__ uint32_t tab[kTabWords];
    fill_tables(tab, te0);
    const LaneRegs lr = lane_regs(tab);
    const u128 mask = (static_cast<u128>(mask_hi) << 64) | mask_lo;
    const uint64_t n_big = tb.big_end_of(n_vec - 1), total_tiles = n_big + tb.small_end_of(n_vec - 1);
    const uint32_t wave64 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(threadIdx.x & ~63u));
    const uint32_t lane = threadIdx.x & 63u;
    int cur = -1, v = 0;
    bool in_small = false;
    uint64_t tile0 = 0;                                                  // first tile (of the current kind) of job v
    CtrPrefix pre_a{}, pre_b{};
    // workgroup g takes a CONTIGUOUS share of the big tiles and a contiguous share of the small ones (equal counts: the
    // host made the big tiles a multiple of the grid): consecutive tiles mostly belong to the same job, so the prefix
    // words are rebuilt at job boundaries only, and a workgroup streams through adjacent memory
    const uint64_t G = gridDim.x, g = blockIdx.x, n_small = total_tiles - n_big;
    const uint64_t big_lo = n_big / G * g + (g < n_big % G ? g : n_big % G), big_cnt = n_big / G + (g < n_big % G ? 1 : 0);
    const uint64_t small_lo = n_small / G * g + (g < n_small % G ? g : n_small % G), small_cnt = n_small / G + (g < n_small % G ? 1 : 0);
    for (uint64_t kk = 0; kk < big_cnt + small_cnt; kk++) {
        const uint64_t t = kk < big_cnt ? big_lo + kk : n_big + small_lo + (kk - big_cnt);
        // locate the tile: wave-uniform, and t only grows within a kind
        uint64_t kw;                                                     // first element of this wave's share
        int epl;
        if (t < n_big) {
            while (t >= tb.big_end_of(v)) tile0 = tb.big_end_of(v++);
            epl = kBigEpl;
            kw = (t - tile0) * (THREADS * kBigEpl) + wave64 * kBigEpl;
        } else {
            if (!in_small) { in_small = true; v = 0; tile0 = 0; }
            const uint64_t ts = t - n_big;
            while (ts >= tb.small_end_of(v)) tile0 = tb.small_end_of(v++);
            const uint64_t big_v = tb.big_end_of(v) - (v ? tb.big_end_of(v - 1) : 0);
            epl = 1;
            kw = big_v * (THREADS * kBigEpl) + (ts - tile0) * THREADS + wave64;
        }
        const uint32_t ia = tb.add_of(v), im = tb.minus_of(v);
        const uint64_t count = tb.count_of(v), first = tb.first_of(v);
        // the shortcuts need the job to stay inside one 2^32 counter window (only the low counter word varies)
        const bool ctr_fast = ((first + count - 1) >> 32) == (first >> 32);
        if (v != cur && ctr_fast) {
            pre_a = scalar_prefix(ctr_prefix(rk, lr, iter, ia, static_cast<uint32_t>(first >> 32)));
            if (DBL) pre_b = scalar_prefix(ctr_prefix(rk, lr, iter, im, static_cast<uint32_t>(first >> 32)));
            cur = v;
        }
        if (kw >= count) continue;
        const uint64_t *in = tb.in_of(v);
        uint64_t *out = tb.out_of(v);
        const int in_limbs = tb.in_limbs_of(v);
        const int n_in = tb.n_in_of(v);
        const uint64_t in_stride = tb.in_stride_of(v);
        uint64_t *sum_out = tb.sum_out_of(v);
        // wave-uniform part of rounds 1-2: valid while the wave's 256 counters share bytes 1..3
        const bool uni = FLASHE_CTR2 && epl == kBigEpl && ctr_fast && ((first + kw) & 255u) == 0;
        CtrUniform Ua{}, Ub{};
        if (uni) {
            const uint32_t x3 = static_cast<uint32_t>(first + kw) ^ rk.w[3];
            Ua = ctr_uniform(rk, te0, pre_a, x3);
            if (DBL) Ub = ctr_uniform(rk, te0, pre_b, x3);
        }
        for (int e = 0; e < epl; e++) {
            const uint64_t k = kw + static_cast<uint64_t>(e) * 64u + lane;
            if (k >= count) break;
            const uint64_t j = first + k;
            u128 acc = !in ? static_cast<u128>(0) : in_limbs == 2 ? ld128(in + 2 * k) : static_cast<u128>(in[k]);
            u64x2 held[KIND == 2 ? kSumRegs : 1];
            if (KIND == 2 && n_in > 1) {        // the reduce fused in: sum of n_in vectors (wave-uniform trip count)
#pragma unroll
                for (int c = 0; c < kSumRegs; c++) {
                    const int cc = c + 1 < n_in ? c + 1 : 0;        // surplus slots re-read operand 0 and are not added
                    held[c] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(in + static_cast<uint64_t>(cc) * in_stride + 2 * k));
                }
                for (int c = kSumRegs + 1; c < n_in; c++) acc += ld128_nt(in + static_cast<uint64_t>(c) * in_stride + 2 * k);
            }
            uint32_t s[DBL ? 2 : 1][4];
            if (uni) {
                const uint32_t v0 = T3(static_cast<uint32_t>(j) ^ rk.w[3], SEL_B0);
                ctr_round2(lr, pre_a.u[0], v0, Ua, s[0]);
                if (DBL) ctr_round2(lr, pre_b.u[0], v0, Ub, s[DBL ? 1 : 0]);
                aes256_rounds<DBL ? 2 : 1, 3>(rk, lr, s, true);
            } else if (ctr_fast) {
                const CtrVar x = ctr_var(rk, lr, static_cast<uint32_t>(j));
                ctr_round1(pre_a, x, s[0]);
                if (DBL) ctr_round1(pre_b, x, s[DBL ? 1 : 0]);
                aes256_rounds<DBL ? 2 : 1, 2>(rk, lr, s, true);
            } else {
                set_block(s[0], iter, ia, j);
                if (DBL) set_block(s[DBL ? 1 : 0], iter, im, j);
                aes256_encrypt<DBL ? 2 : 1>(rk, lr, s);
            }
            if (KIND == 2 && n_in > 1) {
#pragma unroll
                for (int c = 0; c < kSumRegs; c++)
                    if (c + 1 < n_in) acc += (static_cast<u128>(held[c][1]) << 64) | held[c][0];
                acc &= mask;
                if (sum_out) st128_nt(sum_out + 2 * k, acc);
            }
            acc += words_to_u128(s[0]);
            if (DBL) acc -= words_to_u128(s[DBL ? 1 : 0]);
            // plain store: measured, part of a ciphertext is still in the Infinity Cache for the reduce that follows; the result of the
            // fused reduce + decrypt is read by nobody on the device: non-temporal (0.379 -> 0.367 ms for ten 1e7-element operands)
            if (KIND == 2) st128_nt(out + 2 * k, acc & mask);
            else st128(out + 2 * k, acc & mask);
        }
    }
}

// ---- chained jobs (b > 64): C consecutive clients share their PRF streams -------------------------------------
// Client c of the double-mask scheme encrypts with term(idx_c) - term(idx_c + 1) (jzf_flashe.py:349-353,480-481):
// the minus stream of client c IS the add stream of client c + 1.  A CHAIN of `len` outputs over one element range
// has len + 1 streams s_0 .. s_len and out_c = in_c + S(s_c) - S(s_{c+1}); every stream is computed ONCE per
// element (11 AES blocks per element-position for ten clients instead of 20).  A plain (add, minus) job is a chain
// of length 1; a SINGLE chain has one stream per output and no subtraction (single-mask encrypts, mask precompute).
//
// Work unit = a wave tile: 256 consecutive counters aligned to 256, so bytes 1..3 of the low counter word are
// wave-uniform (the scalar-cache step of the CTR shortcut), walked as two PAIRS of elements per lane; a pair runs
// through the software-pipelined two-block rounds, both blocks on the SAME prefix.  The stream loop is the outer
// loop: per (tile, stream) one wave-uniform prefix fetch + 15 scalar-cache lookups, then 4 blocks per lane; the
// previous stream's blocks of the lane's 4 elements stay in registers (16 VGPRs) for the subtraction.
// Dealing: tiles are weighted by their stream count; workgroup g owns the tiles whose weight offset falls into
// [W g / G, W (g + 1) / G) -- contiguous memory per workgroup -- and its 16 waves take them round-robin; what is
// left of a workgroup's share after whole rounds is cut into HALF tiles (one pair per lane, one-step shortcut) so
// that no wave ends up with a whole 256 x (len + 1)-block tile more than its neighbours.  Short launches
// (all_half) run entirely in half tiles.
#ifndef FLASHE_HALF_U
#define FLASHE_HALF_U 1        // round 6: the half tiles of prf_chain_kernel take the second counter shortcut too (a half tile is 128 aligned counters: ctr_uniform per
                               // item and stream, 196 lookups per block instead of 208) -- launches of 1e6 .. 2e6 elements -3.5 ... -5 %, the headline unchanged (0: A/B builds)
#endif
constexpr int kMaxChains = 16;       // chains per launch
constexpr int kMaxLinks = 128;       // outputs per launch, all chains together
struct ChainTable {
    uint64_t first[kMaxChains], count[kMaxChains];     // element range (global indices = PRF counters)
    uint64_t wend[kMaxChains];                         // running total of tiles x streams
    uint16_t link0[kMaxChains], sbase[kMaxChains];     // first output / first stream of the chain in the flat arrays
    uint8_t len[kMaxChains];                           // outputs of the chain
    uint8_t flags[kMaxChains];                         // bit 0: SINGLE, bit 1: two-limb inputs
    uint32_t idx[kMaxLinks + kMaxChains];              // prefix index of every stream
    const uint64_t *in[kMaxLinks];                     // may be null (zeros); addresses element first[chain]
    uint64_t *out[kMaxLinks];
    uint64_t *sum_out[kMaxChains];                     // SUM kernels only: where sum_c out_c of the chain goes (null = nowhere)
};

__device__ __forceinline__ CtrPrefix load_prefix(const uint32_t *pre_lds, int s)
{
    const uint4 v = *reinterpret_cast<const uint4 *>(pre_lds + 4 * s);       // wave-uniform address: a broadcast read
    return CtrPrefix{{static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(v.x)), static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(v.y)),
                      static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(v.z)), static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(v.w))}};
}

// An unconditional use of two loaded values: the compiler places the wait for their loads HERE, on the straight-line path,
// instead of inside the exec-masked store branches (where a skipped branch leaves the loads "possibly in flight" in its
// path-insensitive bookkeeping and every later wait becomes vmcnt(0), which also waits for the stores just issued).
__device__ __forceinline__ void loads_landed(const u128 &a, const u128 &b)
{
    asm volatile("" ::"v"(static_cast<uint64_t>(a)), "v"(static_cast<uint64_t>(a >> 64)), "v"(static_cast<uint64_t>(b)),
                 "v"(static_cast<uint64_t>(b >> 64)));
}

template <class T> __device__ __forceinline__ void swap_regs(T &a, T &b) { const T t = a; a = b; b = t; }

// a wave-uniform 64-bit value read from LDS, moved to SGPRs so that what is derived from it stays scalar
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v))) |
           (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32)))) << 32);
}

// SUM: the launch also writes the chain's LOCAL PARTIAL AGGREGATE sum_c out_c mod 2^b ("each GPU encrypts and
// locally mod-adds its share"; the arbiter's reduce jzf_aggregator.py:424-430 applied to the ciphertexts this GPU has just produced):
// every out_c of an element passes through the lane's registers in turn, so the running sum costs 4 VGPRs per element and one
// non-temporal 16-byte store, and the C ciphertexts are never re-read for the reduce.
// CODEC: the launch carries a fused quantise front end / unquantise back end (a one-output job); the plain instantiations -- every
// encrypt of a round -- do not even see the descriptor.
template <int THREADS, bool SUM, bool CODEC>
__global__ __launch_bounds__(THREADS) void prf_chain_kernel(const RoundKeys rk, const ChainTable tb, int n_chains, int all_half_arg,
                                                              uint32_t iter0, uint64_t mask_lo, uint64_t mask_hi,
                                                              const uint32_t *__restrict__ te0, const Codec cq)
{
    const uint32_t iter = iter0 + te0[kIterShiftWord];
    constexpr uint32_t WAVES = THREADS / 64;
    __shared__ uint32_t tab[kTabWords];
    __shared__ __attribute__((aligned(16))) uint32_t pre_lds[(kMaxLinks + kMaxChains) * 4];
    __shared__ uint64_t d_tlo[kMaxChains], d_cend[kMaxChains];
    int all_half = all_half_arg;
    fill_tables(tab, te0);
#ifdef FLASHE_TUNING
    if (all_half & 0x100) return;                  // timing probes of the prologue (FLASHE_CHAIN_TUNE / FLASHE_CHAIN_PROBE only)
#endif
    const LaneRegs lr = lane_regs(tab);
    const u128 mask = (static_cast<u128>(mask_hi) << 64) | mask_lo;
    {
        // round-1 prefix words of every stream (chains never straddle a 2^32 counter window: host-checked)
        const int last = n_chains - 1;
        const int n_streams = tb.sbase[last] + tb.len[last] + ((tb.flags[last] & 1) ? 0 : 1);
        for (int s = threadIdx.x; s < n_streams; s += THREADS) {
            int i = 0;
            while (i < last && s >= tb.sbase[i + 1]) i++;
            const CtrPrefix c = ctr_prefix(rk, lr, iter, tb.idx[s], static_cast<uint32_t>(tb.first[i] >> 32));
            *reinterpret_cast<uint4 *>(pre_lds + 4 * s) = make_uint4(c.u[0], c.u[1], c.u[2], c.u[3]);
        }
        // this workgroup's tiles of every chain: lane i works out chain i (32-bit arithmetic whenever the launch's total
        // weight fits -- a 64-bit division is ~150 dependent instructions, and a short launch is all prologue)
        if (threadIdx.x < static_cast<unsigned>(n_chains)) {
            const int i = threadIdx.x;
            const uint64_t Wt = tb.wend[last], cw = i ? tb.wend[i - 1] : 0;
            const uint32_t w = tb.len[i] + ((tb.flags[i] & 1) ? 0u : 1u);
            uint64_t a, b, T;
            if (Wt <= 0xffffffffull && gridDim.x <= 0xffffu) {
                const uint32_t W32 = static_cast<uint32_t>(Wt), G = gridDim.x, g = blockIdx.x, c32 = static_cast<uint32_t>(cw);
                const uint32_t q = W32 / G, r = W32 % G;
                const uint32_t lo = q * g + r * g / G, hi = q * (g + 1) + r * (g + 1) / G;
                const uint32_t T32 = (static_cast<uint32_t>(tb.wend[i]) - c32) / w;
                const uint32_t a32 = lo > c32 ? (lo - c32 + w - 1) / w : 0, b32 = hi > c32 ? (hi - c32 + w - 1) / w : 0;
                T = T32; a = a32; b = b32;
            } else {
                const uint64_t G = gridDim.x, g = blockIdx.x;
                const uint64_t lo = Wt / G * g + (Wt % G) * g / G, hi = Wt / G * (g + 1) + (Wt % G) * (g + 1) / G;
                T = (tb.wend[i] - cw) / w;
                a = lo > cw ? (lo - cw + w - 1) / w : 0; b = hi > cw ? (hi - cw + w - 1) / w : 0;
            }
            if (a > T) a = T;
            if (b > T) b = T;
            d_tlo[i] = a; d_cend[i] = b - a;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t acc = 0;
            for (int i = 0; i < n_chains; i++) { acc += d_cend[i]; d_cend[i] = acc; }
        }
        __syncthreads();
    }
#ifdef FLASHE_TUNING
    if (all_half & 0x200) return;
#endif
    // (bit 2, round 6: the whole launch in QUARTER tiles -- 64 counters, one block per lane and stream, two STREAMS per step -- for
    // launches too short to give every wave a half tile: see the quarter branch below)
    const bool quarter = !CODEC && !SUM && (all_half & 4) != 0;            // (a summed chain is only launched with two whole tiles per wave: launch_prf_batch_sum)
    all_half &= 1;
    const uint32_t wave = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t Ng = uniform64(d_cend[n_chains - 1]);
    const uint64_t n_full = (all_half || quarter) ? 0 : Ng - Ng % WAVES;
    const uint64_t n_items = quarter ? 4 * Ng : n_full + 2 * (Ng - n_full);
    int cur = 0;
    uint64_t cbeg = 0;                                                     // local index of chain cur's first tile
    for (uint64_t q = wave; q < n_items; q += WAVES) {
        const bool whole = q < n_full;
        const uint64_t L = quarter ? (q >> 2) : whole ? q : n_full + ((q - n_full) >> 1);
        const uint32_t half = whole || quarter ? 0u : static_cast<uint32_t>((q - n_full) & 1u);
        while (L >= uniform64(d_cend[cur])) cbeg = uniform64(d_cend[cur++]);
        const uint64_t first = tb.first[cur], end = first + tb.count[cur];
        const uint64_t tj = (first & ~255ull) + 256u * (uniform64(d_tlo[cur]) + (L - cbeg)) + 128u * half +
                            (quarter ? 64u * static_cast<uint32_t>(q & 3u) : 0u);                             // first counter of the item
        const int link0 = tb.link0[cur], sbase = tb.sbase[cur];
        const bool single = tb.flags[cur] & 1, in2 = tb.flags[cur] & 2;
        const int n_streams = tb.len[cur] + (single ? 0 : 1);
        if (whole) {
            // ---- 256 elements: two pairs per lane, wave-uniform part of rounds 1-2 through the scalar cache ----
            const uint32_t x3 = static_cast<uint32_t>(tj) ^ rk.w[3];
            const uint32_t jl = static_cast<uint32_t>(tj) + lane;
            uint32_t vA0 = T3(jl ^ rk.w[3], SEL_B0), vA1 = T3((jl + 64u) ^ rk.w[3], SEL_B0);
            uint32_t vB0 = T3((jl + 128u) ^ rk.w[3], SEL_B0), vB1 = T3((jl + 192u) ^ rk.w[3], SEL_B0);
            u128 pA0 = 0, pA1 = 0, pB0 = 0, pB1 = 0;
            u128 qA0 = 0, qA1 = 0, qB0 = 0, qB1 = 0;                   // SUM: running sum of the outputs of the lane's four elements
            uint64_t *const sum_out = SUM ? tb.sum_out[cur] : nullptr;
            for (int c = 0; c < n_streams; c++) {
                const CtrPrefix pre = load_prefix(pre_lds, sbase + c);
                const CtrUniform U = ctr_uniform(rk, te0, pre, x3);
                const int link = single ? c : c - 1;                   // the output this stream completes
                const uint64_t *in = link >= 0 ? tb.in[link0 + link] : nullptr;
                uint64_t *out = link >= 0 ? tb.out[link0 + link] : nullptr;
                const bool last_stream = c == n_streams - 1;
#pragma unroll 1
                for (int p = 0; p < 2; p++) {
                    const uint64_t jb = tj + 128u * p;
                    if (jb < end && jb + 128u > first) {
                        const uint64_t j0 = jb + lane, j1 = j0 + 64u, k0 = j0 - first, k1 = j1 - first;
                        const bool a0 = j0 >= first && j0 < end, a1 = j1 >= first && j1 < end;
                        // every load is consumed on every path (the adds below are unconditional, only the stores are
                        // predicated): otherwise the compiler must assume a load may still be in flight at the loop's back
                        // edge and drains the memory queue -- stores included -- every iteration
                        u128 x0 = 0, x1 = 0;
                        if (CODEC && cq.x != nullptr && link >= 0) {
                            if (a0) x0 = codec_quantize(cq, k0);
                            if (a1) x1 = codec_quantize(cq, k1);
                        } else if (in != nullptr && in2) {
                            if (a0) x0 = ld128(in + 2 * k0);
                            if (a1) x1 = ld128(in + 2 * k1);
                        } else if (in != nullptr) {
                            if (a0) x0 = static_cast<u128>(in[k0]);
                            if (a1) x1 = static_cast<u128>(in[k1]);
                        }
                        uint32_t s[2][4];
                        ctr_round2(lr, pre.u[0], vA0, U, s[0]);
                        ctr_round2(lr, pre.u[0], vA1, U, s[1]);
                        aes256_rounds<2, 3>(rk, lr, s, true);
                        loads_landed(x0, x1);
                        const u128 c0 = words_to_u128(s[0]), c1 = words_to_u128(s[1]);
                        const u128 r0 = x0 + (single ? c0 : pA0 - c0), r1 = x1 + (single ? c1 : pA1 - c1);
                        if (CODEC && cq.fout != nullptr) {
                            if (a0 && link >= 0) cq.fout[k0] = codec_unquantize(cq, k0, r0 & mask);
                            if (a1 && link >= 0) cq.fout[k1] = codec_unquantize(cq, k1, r1 & mask);
                        } else {
                            if (a0 && out != nullptr) st128(out + 2 * k0, r0 & mask);
                            if (a1 && out != nullptr) st128(out + 2 * k1, r1 & mask);
                        }
                        if constexpr (SUM) {
                            // (the sums do not take part in the register rotation of the rolled pair loop: p is wave-uniform, a scalar
                            // branch around four adds is cheaper than eight more moves per pair)
                            if (link >= 0) {
                                if (p == 0) { qA0 += r0; qA1 += r1; } else { qB0 += r0; qB1 += r1; }
                            }
                            if (last_stream && sum_out != nullptr) {
                                if (p == 0) {
                                    if (a0) st128_nt(sum_out + 2 * k0, qA0 & mask);
                                    if (a1) st128_nt(sum_out + 2 * k1, qA1 & mask);
                                } else {
                                    if (a0) st128_nt(sum_out + 2 * k0, qB0 & mask);
                                    if (a1) st128_nt(sum_out + 2 * k1, qB1 & mask);
                                }
                            }
                        }
                        pA0 = c0; pA1 = c1;
                    }
                    swap_regs(pA0, pB0); swap_regs(pA1, pB1); swap_regs(vA0, vB0); swap_regs(vA1, vB1);
                }
            }
        } else if (quarter) {
            // ---- 64 elements (round 6): ONE block per lane and stream, the software-pipelined pair is two consecutive STREAMS of the
            // chain.  For launches that cannot give every wave of the chip a half tile (config 3: a hundred LeNet-sized vectors are
            // 242 tiles): four times the items, so a chain is cut into a third as many pieces (a cut costs a stream) and every wave
            // gets ONE item of the same length instead of one or two; and 64 aligned counters share bytes 1 .. 3, so both counter-mode
            // shortcuts apply (196 lookups per block; the half tiles take only the first: 208).
            if (!CODEC && !SUM && tj < end && tj + 64u > first) {
                const uint64_t j0 = tj + lane, k0 = j0 - first;
                const bool a0 = j0 >= first && j0 < end;
                const uint32_t x3 = static_cast<uint32_t>(tj) ^ rk.w[3];
                const uint32_t v0 = T3(static_cast<uint32_t>(j0) ^ rk.w[3], SEL_B0);
                u128 pv = 0;
                for (int c = 0; c < n_streams; c += 2) {
                    const bool has1 = c + 1 < n_streams;             // (an odd stream count computes its last stream twice)
                    const CtrPrefix pre0 = load_prefix(pre_lds, sbase + c), pre1 = load_prefix(pre_lds, sbase + (has1 ? c + 1 : c));
                    const CtrUniform U0 = ctr_uniform(rk, te0, pre0, x3), U1 = ctr_uniform(rk, te0, pre1, x3);
                    const int l0 = single ? c : c - 1;               // the output stream c completes; stream c + 1 completes l0 + 1
                    const uint64_t *in0 = l0 >= 0 ? tb.in[link0 + l0] : nullptr, *in1 = has1 ? tb.in[link0 + l0 + 1] : nullptr;
                    u128 x0 = 0, x1 = 0;
                    if (in0 != nullptr && a0) x0 = in2 ? ld128(in0 + 2 * k0) : static_cast<u128>(in0[k0]);
                    if (in1 != nullptr && a0) x1 = in2 ? ld128(in1 + 2 * k0) : static_cast<u128>(in1[k0]);
                    uint32_t s[2][4];
                    ctr_round2(lr, pre0.u[0], v0, U0, s[0]);
                    ctr_round2(lr, pre1.u[0], v0, U1, s[1]);
                    aes256_rounds<2, 3>(rk, lr, s, FLASHE_SWP_PRIO_HALF != 0);
                    loads_landed(x0, x1);
                    const u128 c0 = words_to_u128(s[0]), c1 = words_to_u128(s[1]);
                    if (l0 >= 0) {
                        const u128 r0 = (x0 + (single ? c0 : pv - c0)) & mask;
                        if (a0 && tb.out[link0 + l0] != nullptr) st128(tb.out[link0 + l0] + 2 * k0, r0);
                    }
                    if (has1) {
                        const u128 r1 = (x1 + (single ? c1 : c0 - c1)) & mask;
                        if (a0 && tb.out[link0 + l0 + 1] != nullptr) st128(tb.out[link0 + l0 + 1] + 2 * k0, r1);
                    }
                    pv = has1 ? c1 : c0;
                }
            }
        } else if (tj < end && tj + 128u > first) {
            // ---- 128 elements: one pair per lane; the counter-dependent lookup of round 1 is shared by all streams ----
            const uint64_t j0 = tj + lane, j1 = j0 + 64u, k0 = j0 - first, k1 = j1 - first;
            const bool a0 = j0 >= first && j0 < end, a1 = j1 >= first && j1 < end;
            const CtrVar xv0 = ctr_var(rk, lr, static_cast<uint32_t>(j0)), xv1 = ctr_var(rk, lr, static_cast<uint32_t>(j1));
#if FLASHE_HALF_U
            const uint32_t x3h = static_cast<uint32_t>(tj) ^ rk.w[3];        // (a half tile is 128 aligned counters: bytes 1 .. 3 are the wave's)
#endif
            u128 p0 = 0, p1 = 0, q0 = 0, q1 = 0;
            uint64_t *const sum_out = SUM ? tb.sum_out[cur] : nullptr;
            for (int c = 0; c < n_streams; c++) {
                const CtrPrefix pre = load_prefix(pre_lds, sbase + c);
                const int link = single ? c : c - 1;
                const uint64_t *in = link >= 0 ? tb.in[link0 + link] : nullptr;
                uint64_t *out = link >= 0 ? tb.out[link0 + link] : nullptr;
                u128 x0 = 0, x1 = 0;
                if (CODEC && cq.x != nullptr && link >= 0) {
                    if (a0) x0 = codec_quantize(cq, k0);
                    if (a1) x1 = codec_quantize(cq, k1);
                } else if (in != nullptr && in2) {
                    if (a0) x0 = ld128(in + 2 * k0);
                    if (a1) x1 = ld128(in + 2 * k1);
                } else if (in != nullptr) {
                    if (a0) x0 = static_cast<u128>(in[k0]);
                    if (a1) x1 = static_cast<u128>(in[k1]);
                }
                uint32_t s[2][4];
#if FLASHE_HALF_U
                {
                    const CtrUniform U = ctr_uniform(rk, te0, pre, x3h);
                    ctr_round2(lr, pre.u[0], xv0.v[0], U, s[0]);
                    ctr_round2(lr, pre.u[0], xv1.v[0], U, s[1]);
                }
                aes256_rounds<2, 3>(rk, lr, s, FLASHE_SWP_PRIO_HALF != 0);
#else
                ctr_round1(pre, xv0, s[0]);
                ctr_round1(pre, xv1, s[1]);
                aes256_rounds<2, 2>(rk, lr, s, FLASHE_SWP_PRIO_HALF != 0);
#endif
                loads_landed(x0, x1);
                const u128 c0 = words_to_u128(s[0]), c1 = words_to_u128(s[1]);
                const u128 r0 = x0 + (single ? c0 : p0 - c0), r1 = x1 + (single ? c1 : p1 - c1);
                if (CODEC && cq.fout != nullptr) {
                    if (a0 && link >= 0) cq.fout[k0] = codec_unquantize(cq, k0, r0 & mask);
                    if (a1 && link >= 0) cq.fout[k1] = codec_unquantize(cq, k1, r1 & mask);
                } else {
                    if (a0 && out != nullptr) st128(out + 2 * k0, r0 & mask);
                    if (a1 && out != nullptr) st128(out + 2 * k1, r1 & mask);
                }
                if constexpr (SUM) {
                    if (link >= 0) { q0 += r0; q1 += r1; }
                    if (c == n_streams - 1 && sum_out != nullptr) {
                        if (a0) st128_nt(sum_out + 2 * k0, q0 & mask);
                        if (a1) st128_nt(sum_out + 2 * k1, q1 & mask);
                    }
                }
                p0 = c0; p1 = c1;
            }
        }
    }
}

// ---- b <= 64: one AES block (m = 128 / b elements) per lane, chunk-dependent counters ----
// Bits [sh, sh + 64) of S (caller masks to b bits).
__device__ __forceinline__ uint64_t extract64(u128 S, int sh)
{
    return sh >= 128 ? 0ull : static_cast<uint64_t>(S >> sh);
}

constexpr int kSmallThreads = 1024;
constexpr int kTch = 8;    // elements accumulated in registers per pass

__global__ __launch_bounds__(kSmallThreads) void prf_small_kernel(const RoundKeys rk, const PrfParams p, const IdxLists lists)
{
    const uint32_t iter = p.iter + p.te0[kIterShiftWord];
    __shared__ uint32_t tab[kTabWords];
    fill_tables(tab, p.te0);
    const LaneRegs lr = lane_regs(tab);

    // chunks_idx(range(n), n_jobs) in closed form: the first r chunks have d + 1 elements.
    const uint64_t J = p.n_jobs, d = p.n / J, r = p.n % J;
    const uint64_t m = static_cast<uint64_t>(p.m);
    const uint64_t nb1 = (d + 1 + m - 1) / m;           // AES blocks in a (d+1)-element chunk
    const uint64_t nb0 = d ? (d + m - 1) / m : 0;       // ... in a d-element chunk
    const uint64_t range_end = p.first + p.count;

    for (uint64_t Bl = static_cast<uint64_t>(blockIdx.x) * kSmallThreads + threadIdx.x; Bl < p.blk_count;
         Bl += static_cast<uint64_t>(gridDim.x) * kSmallThreads) {
        const uint64_t B = p.blk_first + Bl;
        uint64_t begin, len, i;
        if (B < r * nb1) {
            const uint64_t c = B / nb1;
            i = B - c * nb1; begin = c * (d + 1); len = d + 1;
        } else {
            const uint64_t B2 = B - r * nb1, c = B2 / nb0;
            i = B2 - c * nb0; begin = r * (d + 1) + c * d; len = d;
        }
        const uint64_t j0 = begin + i * m;
        const uint64_t rem = len - i * m;
        const int cnt = rem < m ? static_cast<int>(rem) : static_cast<int>(m);
        const uint64_t ctr = begin + i;

        for (int t0 = 0; t0 < cnt; t0 += kTch) {
            uint64_t acc[kTch];
#pragma unroll
            for (int u = 0; u < kTch; u++) {
                const uint64_t j = j0 + t0 + u;
                const bool on = t0 + u < cnt && j >= p.first && j < range_end;
                acc[u] = !on ? 0ull : p.cq.x ? codec_quantize(p.cq, j - p.first) : p.in ? p.in[j - p.first] : 0ull;
            }
            int k = 0;
            for (; k + 1 < p.n_add; k += 2) {
                uint32_t s[2][4];
                set_block(s[0], iter, lists.add[k], ctr);
                set_block(s[1], iter, lists.add[k + 1], ctr);
                aes256_encrypt<2>(rk, lr, s);
                const u128 S0 = words_to_u128(s[0]), S1 = words_to_u128(s[1]);
#pragma unroll
                for (int u = 0; u < kTch; u++) acc[u] += extract64(S0, p.b * (t0 + u)) + extract64(S1, p.b * (t0 + u));
            }
            if (k < p.n_add) {
                uint32_t s[1][4];
                set_block(s[0], iter, lists.add[k], ctr);
                aes256_encrypt<1>(rk, lr, s);
                const u128 S0 = words_to_u128(s[0]);
#pragma unroll
                for (int u = 0; u < kTch; u++) acc[u] += extract64(S0, p.b * (t0 + u));
            }
            k = 0;
            for (; k + 1 < p.n_minus; k += 2) {
                uint32_t s[2][4];
                set_block(s[0], iter, lists.minus[k], ctr);
                set_block(s[1], iter, lists.minus[k + 1], ctr);
                aes256_encrypt<2>(rk, lr, s);
                const u128 S0 = words_to_u128(s[0]), S1 = words_to_u128(s[1]);
#pragma unroll
                for (int u = 0; u < kTch; u++) acc[u] -= extract64(S0, p.b * (t0 + u)) + extract64(S1, p.b * (t0 + u));
            }
            if (k < p.n_minus) {
                uint32_t s[1][4];
                set_block(s[0], iter, lists.minus[k], ctr);
                aes256_encrypt<1>(rk, lr, s);
                const u128 S0 = words_to_u128(s[0]);
#pragma unroll
                for (int u = 0; u < kTch; u++) acc[u] -= extract64(S0, p.b * (t0 + u));
            }
#pragma unroll
            for (int u = 0; u < kTch; u++) {
                const uint64_t j = j0 + t0 + u;
                if (t0 + u < cnt && j >= p.first && j < range_end) {
                    if (p.cq.fout) p.cq.fout[j - p.first] = codec_unquantize(p.cq, j - p.first, acc[u] & p.mask_lo);
                    else p.out[j - p.first] = acc[u] & p.mask_lo;
                }
            }
        }
    }
}

// The bit-sliced PRF backends are NOT part of the default library: they measured 2.5x the VALU work of the table kernel (DESIGN.md
// section 4.3) and nothing selects them by default.  `make bitslice` builds libflashe_hip_bitslice.so with them (-DFLASHE_WITH_BITSLICE,
// the generated header comes from tools/bitslice/gen_bitslice.py at build time); FLASHE_LIB_NAME selects that library.
#ifdef FLASHE_WITH_BITSLICE
// ------------------------------------------------------------------------------------------
// Bit-sliced AES-256 PRF (b > 64): pure VALU, no LDS.
//
// One lane holds 32 AES blocks as 128 bit planes (plane[8*B + k] = bit k of state byte B; bit p of
// the 32-bit word = block p); a wave therefore runs 2048 blocks per pass through straight-line
// v_bitop3_b32 code generated by tools/bitslice/gen_bitslice.py (S-box = Boyar-Peralta circuit
// mapped to 3-input LUTs, MixColumns / AddRoundKey merged into the same netlist).  With two streams
// (encrypt double: idx and idx + 1) blocks p < 16 carry the add stream and p >= 16 the minus
// stream of the SAME 16 elements, so the subtraction stays inside the lane: a wave-pass covers
// elements tile*1024 + q*64 + lane (q = 0..15) -- for fixed q the 64 lanes are contiguous, i.e.
// every global access is a full coalesced wave burst.  Planes <-> per-block words go through
// 32x32 bit transposes (v_perm_b32 for the byte-granular stages, v_bfi_b32 for the rest).
// ------------------------------------------------------------------------------------------
}  // namespace flashe
#include "aes_bitslice_gen.h"
#include "bitslice_core.h"
namespace flashe {

constexpr int kBsThreads = 256;

// NSTREAM = 2: one add + one minus prefix (encrypt double / no-dropout decrypt), 16 elements per lane
// per pass.  NSTREAM = 1: one add prefix only, 32 elements per lane per pass.
template <int NSTREAM>
__global__ __launch_bounds__(kBsThreads, 2) void prf_wide_bs_kernel(const uint32_t *__restrict__ rkw, const PrfParams p,
                                                                    const uint32_t idx_a, const uint32_t idx_b)
{
    const uint32_t iter = p.iter + p.te0[kIterShiftWord];
    constexpr int EPL = 32 / NSTREAM;                   // elements per lane per pass
    constexpr uint64_t TILE = 64ull * EPL;              // elements per wave-pass
    const int lane = threadIdx.x & 63;
    const uint64_t wave = static_cast<uint64_t>(blockIdx.x) * (kBsThreads / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = static_cast<uint64_t>(gridDim.x) * (kBsThreads / 64);
    const u128 mask = (static_cast<u128>(p.mask_hi) << 64) | p.mask_lo;
    const uint64_t n_tiles = (p.count + TILE - 1) / TILE;

    for (uint64_t tile = wave; tile < n_tiles; tile += n_waves) {
        uint32_t s[128];
        const uint64_t t_first = p.first + tile * TILE;
        bs::load_planes<NSTREAM>(s, iter, idx_a, idx_b, t_first + lane, t_first, t_first + TILE - 1);
        bs::encrypt_planes(s, rkw);
        u128 S[32];
        bs::planes_to_blocks(s, S);
        // ---- out = in + S_a - S_b ----
#pragma unroll
        for (int q = 0; q < EPL; q++) {
            const uint64_t e = tile * TILE + static_cast<uint64_t>(q) * 64 + lane;
            if (e < p.count) {
                u128 acc = 0;
                if (p.in) acc = p.in_limbs == 2 ? ld128(p.in + 2 * e) : static_cast<u128>(p.in[e]);
                acc += S[q];
                if (NSTREAM == 2) acc -= S[q + 16];
                st128(p.out + 2 * e, acc & mask);
            }
        }
    }
}

// Packed bit-sliced PRF: 16 blocks per lane in 64 plane registers (two state bytes per register), so the
// kernel needs ~half the VGPRs of prf_wide_bs_kernel and 3-4 waves fit per SIMD.  NSTREAM = 2: blocks
// 0..7 / 8..15 are the add / minus stream of the same 8 elements (wave-pass = 512 elements).
template <int NSTREAM, int WAVES>
__global__ __launch_bounds__(kBsThreads, WAVES) void prf_wide_bsp_kernel(const uint32_t *__restrict__ rkp, const PrfParams p,
                                                                         const uint32_t idx_a, const uint32_t idx_b)
{
    const uint32_t iter = p.iter + p.te0[kIterShiftWord];
    constexpr int EPL = 16 / NSTREAM;
    constexpr uint64_t TILE = 64ull * EPL;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = static_cast<uint64_t>(blockIdx.x) * (kBsThreads / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = static_cast<uint64_t>(gridDim.x) * (kBsThreads / 64);
    const u128 mask = (static_cast<u128>(p.mask_hi) << 64) | p.mask_lo;
    const uint64_t n_tiles = (p.count + TILE - 1) / TILE;

    for (uint64_t tile = wave; tile < n_tiles; tile += n_waves) {
        uint32_t s[64];
        bs::load_planes_p<NSTREAM>(s, iter, idx_a, idx_b, p.first + tile * TILE + lane);
        bs::encrypt_planes_p(s, rkp);
        u128 S[16];
        bs::planes_to_blocks_p(s, S);
#pragma unroll
        for (int q = 0; q < EPL; q++) {
            const uint64_t e = tile * TILE + static_cast<uint64_t>(q) * 64 + lane;
            if (e < p.count) {
                u128 acc = 0;
                if (p.in) acc = p.in_limbs == 2 ? ld128(p.in + 2 * e) : static_cast<u128>(p.in[e]);
                acc += S[q];
                if (NSTREAM == 2) acc -= S[q + 8];
                st128(p.out + 2 * e, acc & mask);
            }
        }
    }
}
#endif  // FLASHE_WITH_BITSLICE

// ---- b <= 64, one add / at most one minus prefix: many jobs per launch, coalesced element traffic ----
// A lane still encrypts one AES block (m = 128 / b elements, counter = chunk begin + block index), but the m
// elements are no longer loaded / stored by that lane (lane-strided 8-byte accesses, m * 8 bytes apart).  A wave
// owns 64 consecutive blocks = 64 * m consecutive elements: every lane packs its block's per-slot result
//     D = slot-wise (S_add - S_minus) mod 2^b        (one SWAR subtraction on the 128-bit word)
// into a 16-byte row of a per-wave LDS scratch, and the wave then walks its elements 64 at a time, lane-contiguous:
// element x of the tile reads the b-bit window of row x / m at bit b * (x % m), adds the plaintext and stores --
// full coalesced bursts.  Tiles that contain a partial block (chunk ends) or run past the job take the per-lane path.
struct SmallJobTable {
    uint32_t add[kMaxBatch], minus[kMaxBatch];
    uint64_t first[kMaxBatch], count[kMaxBatch];            // element range of the job (global indices)
    uint64_t blk_first[kMaxBatch], blk_count[kMaxBatch];    // AES blocks intersecting it (global block numbering)
    uint64_t tile_end[kMaxBatch];                           // running total of 64-block wave tiles
    const uint64_t *in[kMaxBatch];                          // may be null; indexed by (element - first)
    uint64_t *out[kMaxBatch];
};

struct SmallParams {
    uint64_t n;               // length of the whole vector (defines the chunking)
    uint32_t n_jobs, iter;
    int b, m;
    int no_direct;            // A/B knob FLASHE_SMALL_DIRECT: 0 = stage every output through the LDS rows, 2 = general walk instead of the b <= 32 fast walk
    int swp_prio;             // wave priority rising through the rounds of a block pair (device_common.h): set per launch from the measured table
    int no_fixed_width;       // A/B knob FLASHE_SMALL_FIXED=0: the compact layout's kernels with int_bits at run time even at the compiled-in widths
    uint32_t m_magic;         // ceil(2^32 / m): x / m == (x * m_magic) >> 32 for x < 2^13
    uint64_t mask_lo;
    uint64_t top_lo, top_hi;  // the top bit of every b-bit slot of the 128-bit word (SWAR subtraction)
    uint32_t nb1_magic, nb0_magic;   // floor(2^32 / nb1), floor(2^32 / nb0): 32-bit block -> chunk division without a divide
    const uint32_t *te0;
    Codec cq;                 // optional fused quantise front end / unquantise back end (single-job launches)
};

// x / dsr for 32-bit operands with magic = floor(2^32 / dsr) (dsr >= 2; dsr == 1 is handled by the caller): the estimate is at most
// two short, fixed up by comparisons -- ~8 instructions instead of the ~100 of a 64-bit division per AES block.
__device__ __forceinline__ uint32_t udiv_magic(uint32_t x, uint32_t dsr, uint32_t magic)
{
    uint32_t q = __umulhi(x, magic);
    uint32_t rem = x - q * dsr;
    if (rem >= dsr) { q++; rem -= dsr; }
    if (rem >= dsr) { q++; }
    return q;
}

template <bool DBL>
__global__ __launch_bounds__(kSmallThreads) void prf_small_jobs_kernel(const RoundKeys rk, const SmallJobTable tb, int n_vec, const SmallParams p)
{
    const uint32_t iter = p.iter + p.te0[kIterShiftWord];
    __shared__ uint32_t tab[kTabWords];
    __shared__ uint32_t scratch[(kSmallThreads / 64) * 256 + 8];
    fill_tables(tab, p.te0);
    const LaneRegs lr = lane_regs(tab);
    const uint64_t J = p.n_jobs, d = p.n / J, r = p.n % J;
    const uint64_t m = static_cast<uint64_t>(p.m);
    const uint64_t nb1 = (d + 1 + m - 1) / m;
    const uint64_t nb0 = d ? (d + m - 1) / m : 0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    uint32_t *row0 = scratch + wave * 256;
    const bool ctr_fast = ((p.n - 1) >> 32) == 0;           // every counter (chunk begin + block index) is below n
    const u128 top = (static_cast<u128>(p.top_hi) << 64) | p.top_lo;
    // the unit of distribution is a WAVE tile of 64 blocks (all jobs in one index space): short vectors, e.g. a hundred
    // LeNet-sized models, still spread evenly over the 16 x 256 waves of the chip
    const uint64_t total_tiles = tb.tile_end[n_vec - 1];
    const uint64_t n_waves = static_cast<uint64_t>(gridDim.x) * (kSmallThreads / 64);
    int cur = -1, v = 0;
    uint64_t tile0 = 0;
    CtrPrefix pre_a{}, pre_b{};
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * (kSmallThreads / 64) + wave; t < total_tiles; t += n_waves) {
        while (t >= tb.tile_end[v]) tile0 = tb.tile_end[v++];
        const uint32_t ia = tb.add[v], im = tb.minus[v];
        if (v != cur && ctr_fast) {
            pre_a = ctr_prefix(rk, lr, iter, ia, 0u);
            if (DBL) pre_b = ctr_prefix(rk, lr, iter, im, 0u);
            cur = v;
        }
        const uint64_t Bw = (t - tile0) * 64u;                              // this wave's first block (job-local)
        const uint64_t blk_count = tb.blk_count[v];
        const uint64_t first = tb.first[v], range_end = first + tb.count[v];
        const uint64_t *in = tb.in[v];
        uint64_t *out = tb.out[v];
        const uint64_t Bl = Bw + lane;
        const bool valid = Bl < blk_count;
        const uint64_t B = tb.blk_first[v] + (valid ? Bl : 0);
        uint64_t begin, len, i;
        if (ctr_fast) {
            // n < 2^32: block, chunk and element indices fit 32 bits
            const uint32_t B32 = static_cast<uint32_t>(B), nb1_32 = static_cast<uint32_t>(nb1), nb0_32 = static_cast<uint32_t>(nb0);
            const uint32_t d32 = static_cast<uint32_t>(d), r32 = static_cast<uint32_t>(r);
            if (B32 < r32 * nb1_32) {
                const uint32_t c = nb1_32 == 1 ? B32 : udiv_magic(B32, nb1_32, p.nb1_magic);
                i = B32 - c * nb1_32; begin = static_cast<uint64_t>(c) * (d32 + 1u); len = d + 1;
            } else {
                const uint32_t B2 = B32 - r32 * nb1_32, c = nb0_32 == 1 ? B2 : udiv_magic(B2, nb0_32, p.nb0_magic);
                i = B2 - c * nb0_32; begin = static_cast<uint64_t>(r32) * (d32 + 1u) + static_cast<uint64_t>(c) * d32; len = d;
            }
        } else if (B < r * nb1) {
            const uint64_t c = B / nb1;
            i = B - c * nb1; begin = c * (d + 1); len = d + 1;
        } else {
            const uint64_t B2 = B - r * nb1, c = B2 / nb0;
            i = B2 - c * nb0; begin = r * (d + 1) + c * d; len = d;
        }
        const uint64_t j0 = begin + i * m;
        const uint64_t rem = len - i * m;
        const int cnt = rem < m ? static_cast<int>(rem) : static_cast<int>(m);
        const uint64_t ctr = begin + i;
        uint32_t s[DBL ? 2 : 1][4];
        if (ctr_fast) {
            const CtrVar x = ctr_var(rk, lr, static_cast<uint32_t>(ctr));
            ctr_round1(pre_a, x, s[0]);
            if (DBL) ctr_round1(pre_b, x, s[DBL ? 1 : 0]);
            aes256_rounds<DBL ? 2 : 1, 2>(rk, lr, s, FLASHE_SMALL_NP_PRIO != 0);
        } else {
            set_block(s[0], iter, ia, ctr);
            if (DBL) set_block(s[DBL ? 1 : 0], iter, im, ctr);
            aes256_encrypt<DBL ? 2 : 1>(rk, lr, s);
        }
        const u128 S0 = words_to_u128(s[0]);
        u128 D = S0;
        if (DBL) {
            const u128 S1 = words_to_u128(s[DBL ? 1 : 0]);
            D = ((S0 | top) - (S1 & ~top)) ^ ((S0 ^ ~S1) & top);            // per slot: (a - b) mod 2^b
        }
        // Runs of consecutive elements inside the wave tile: blocks are full except the last one of a chunk, so with at most
        // one partial block (lane P) the tile is run A = lanes 0..P and run B = the lanes after it (next chunk).
        const uint64_t valid_mask = __ballot(valid), partial_mask = __ballot(valid && cnt < p.m);
        if (__popcll(partial_mask) <= 1) {
            *reinterpret_cast<uint4 *>(row0 + 4 * lane) = make_uint4(static_cast<uint32_t>(D), static_cast<uint32_t>(D >> 32),
                                                                     static_cast<uint32_t>(D >> 64), static_cast<uint32_t>(D >> 96));
            __builtin_amdgcn_wave_barrier();
            const int n_valid = __popcll(valid_mask);
            const int P = partial_mask ? static_cast<int>(__ffsll(static_cast<unsigned long long>(partial_mask))) - 1 : -1;
            const uint32_t j0_lo = static_cast<uint32_t>(j0), j0_hi = static_cast<uint32_t>(j0 >> 32);
#pragma unroll 1
            for (int run = 0; run < 2; run++) {
                const int lane_base = run == 0 ? 0 : P + 1;
                if (run == 1 && (P < 0 || lane_base >= n_valid)) break;
                const uint64_t e0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(j0_lo, lane_base)) |
                                    (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(j0_hi, lane_base))) << 32);
                const uint32_t n_elems = run == 0 ? (P >= 0 ? static_cast<uint32_t>(P) * p.m + static_cast<uint32_t>(__builtin_amdgcn_readlane(cnt, P))
                                                            : static_cast<uint32_t>(n_valid) * p.m)
                                                  : static_cast<uint32_t>(n_valid - lane_base) * p.m;
                for (uint32_t x = lane; x < n_elems; x += 64u) {
                    const uint32_t blk = static_cast<uint32_t>((static_cast<uint64_t>(x) * p.m_magic) >> 32);
                    const uint32_t o = static_cast<uint32_t>(p.b) * (x - blk * static_cast<uint32_t>(p.m));
                    const uint32_t *w = row0 + 4 * (lane_base + blk) + (o >> 5);
                    const uint32_t sh = o & 31u;
                    uint64_t val = ((static_cast<uint64_t>(w[1]) << 32) | w[0]) >> sh;
                    if (sh) val |= static_cast<uint64_t>(w[2]) << (64u - sh);
                    const uint64_t j = e0 + x;
                    if (j >= first && j < range_end) {
                        const uint64_t pt = p.cq.x ? codec_quantize(p.cq, j - first) : in ? __builtin_nontemporal_load(in + (j - first)) : 0ull;
                        if (p.cq.fout) p.cq.fout[j - first] = codec_unquantize(p.cq, j - first, (pt + val) & p.mask_lo);
                        else __builtin_nontemporal_store((pt + val) & p.mask_lo, out + (j - first));
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else if (valid) {
            for (int tt = 0; tt < cnt; tt++) {
                const uint64_t j = j0 + tt;
                if (j < first || j >= range_end) continue;
                const uint64_t val = extract64(D, p.b * tt);
                const uint64_t pt = p.cq.x ? codec_quantize(p.cq, j - first) : in ? in[j - first] : 0ull;
                if (p.cq.fout) p.cq.fout[j - first] = codec_unquantize(p.cq, j - first, (pt + val) & p.mask_lo);
                else out[j - first] = (pt + val) & p.mask_lo;
            }
        }
    }
}

// ---- b <= 64, chained: consecutive clients share their streams (see prf_chain_kernel) ----
// A lane owns AES block(s) of the vector for ALL streams of its chain: the chunk arithmetic is done once per tile, the four
// counter-dependent lookups of round 1 once per block (they do not depend on the prefix), and per stream the lane runs one
// block (PAIR = false) or two blocks software pipelined on the same prefix (PAIR = true: blocks L and L + 64 of a 128-block
// tile).  Per output the slot-wise difference of the previous and the current stream goes through the per-wave LDS rows and
// the wave walks its 64 * m consecutive elements lane-contiguously, exactly like prf_small_jobs_kernel.
struct SmallChainTable {
    uint64_t first[kMaxChains], count[kMaxChains];           // element range of the chain (global indices)
    uint64_t blk_first[kMaxChains], blk_count[kMaxChains];   // AES blocks intersecting it (global block numbering)
    uint64_t wend[kMaxChains];                               // running total of tiles x streams
    uint16_t link0[kMaxChains], sbase[kMaxChains];
    uint8_t len[kMaxChains], flags[kMaxChains];              // bit 0: SINGLE
    uint32_t idx[kMaxLinks + kMaxChains];
    const uint64_t *in[kMaxLinks];
    uint64_t *out[kMaxLinks];
    uint64_t *sum_out[kMaxChains];                           // optional (compact layout at a compile-time width): sum of the chain's outputs
};

// block B of the vector (global block numbering) -> first element j0, elements in the block cnt, PRF counter (n < 2^32)
__device__ __forceinline__ void small_block_params(uint32_t B32, uint32_t nb1_32, uint32_t nb0_32, uint32_t d32, uint32_t r32, uint32_t m,
                                                   const SmallParams &p, uint64_t *j0, int *cnt, uint32_t *ctr)
{
    uint32_t begin, len, i;
    if (B32 < r32 * nb1_32) {
        const uint32_t c = nb1_32 == 1 ? B32 : udiv_magic(B32, nb1_32, p.nb1_magic);
        i = B32 - c * nb1_32; begin = c * (d32 + 1u); len = d32 + 1u;
    } else {
        const uint32_t B2 = B32 - r32 * nb1_32, c = nb0_32 == 1 ? B2 : udiv_magic(B2, nb0_32, p.nb0_magic);
        i = B2 - c * nb0_32; begin = r32 * (d32 + 1u) + c * d32; len = d32;
    }
    const uint32_t rem = len - i * m;
    *j0 = static_cast<uint64_t>(begin) + static_cast<uint64_t>(i) * m;
    *cnt = static_cast<int>(rem < m ? rem : m);
    *ctr = begin + i;
}

// The wave's 64 blocks hold D (16 bytes per lane, b-bit slots): out[j] = (in[j] + slot) mod 2^b for every element of the tile
// that lies in [first, range_end), coalesced whenever the tile has at most one partial block (a chunk end).
// ET: the element type of the vectors in memory -- uint64_t (one limb per element, the ABI's layout) or, for int_bits <= 32,
// uint32_t (the compact layout of the *_u32_dev entry points: half the bytes of a kernel that is bound by them)
// sum / sum_first (summed chains, the tiles that do not take the direct path): sum[j] = out[j] for the chain's first output, += for the
// later ones -- the element -> lane mapping of a tile is the same for every output, so each word is read and written by one lane only
template <class ET>
__device__ __forceinline__ void small_walk(uint32_t *row0, uint32_t lane, bool valid, int cnt, uint64_t j0, u128 D, const ET *in,
                                           ET *out, uint64_t first, uint64_t range_end, const SmallParams &p, ET *sum = nullptr, bool sum_first = false)
{
    const uint64_t valid_mask = __ballot(valid), partial_mask = __ballot(valid && cnt < p.m);
    if (__popcll(partial_mask) <= 1) {
        *reinterpret_cast<uint4 *>(row0 + 4 * lane) = make_uint4(static_cast<uint32_t>(D), static_cast<uint32_t>(D >> 32),
                                                                 static_cast<uint32_t>(D >> 64), static_cast<uint32_t>(D >> 96));
        __builtin_amdgcn_wave_barrier();
        const int n_valid = __popcll(valid_mask);
        const int P = partial_mask ? static_cast<int>(__ffsll(static_cast<unsigned long long>(partial_mask))) - 1 : -1;
        const uint32_t j0_lo = static_cast<uint32_t>(j0), j0_hi = static_cast<uint32_t>(j0 >> 32);
#pragma unroll 1
        for (int run = 0; run < 2; run++) {
            const int lane_base = run == 0 ? 0 : P + 1;
            if (run == 1 && (P < 0 || lane_base >= n_valid)) break;
            const uint64_t e0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(j0_lo, lane_base)) |
                                (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(j0_hi, lane_base))) << 32);
            const uint32_t n_elems = run == 0 ? (P >= 0 ? static_cast<uint32_t>(P) * p.m + static_cast<uint32_t>(__builtin_amdgcn_readlane(cnt, P))
                                                        : static_cast<uint32_t>(n_valid) * p.m)
                                              : static_cast<uint32_t>(n_valid - lane_base) * p.m;
            for (uint32_t x = lane; x < n_elems; x += 64u) {
                const uint32_t blk = static_cast<uint32_t>((static_cast<uint64_t>(x) * p.m_magic) >> 32);
                const uint32_t o = static_cast<uint32_t>(p.b) * (x - blk * static_cast<uint32_t>(p.m));
                const uint32_t *w = row0 + 4 * (lane_base + blk) + (o >> 5);
                const uint32_t sh = o & 31u;
                uint64_t val = ((static_cast<uint64_t>(w[1]) << 32) | w[0]) >> sh;
                if (sh) val |= static_cast<uint64_t>(w[2]) << (64u - sh);
                const uint64_t j = e0 + x;
                if (j >= first && j < range_end) {
                    const uint64_t pt = in ? static_cast<uint64_t>(__builtin_nontemporal_load(in + (j - first))) : 0ull;
                    const uint64_t r = (pt + val) & p.mask_lo;
                    __builtin_nontemporal_store(static_cast<ET>(r), out + (j - first));
                    if (sum) sum[j - first] = static_cast<ET>((r + (sum_first ? 0ull : static_cast<uint64_t>(sum[j - first]))) & p.mask_lo);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    } else if (valid) {
        for (int tt = 0; tt < cnt; tt++) {
            const uint64_t j = j0 + tt;
            if (j < first || j >= range_end) continue;
            const uint64_t val = extract64(D, p.b * tt);
            const uint64_t r = ((in ? static_cast<uint64_t>(in[j - first]) : 0ull) + val) & p.mask_lo;
            out[j - first] = static_cast<ET>(r);
            if (sum) sum[j - first] = static_cast<ET>((r + (sum_first ? 0ull : static_cast<uint64_t>(sum[j - first]))) & p.mask_lo);
        }
    }
}

// b <= 32, the common tile (64 whole blocks inside the range): the walk of small_walk with everything per element that can be
// hoisted hoisted -- the (block, slot) of element lane + 64 i advances by additions (no multiply: v_mul_hi / v_mul_lo are
// quarter-rate), a value is one 32-bit funnel shift of two row words, the sum needs only its low word (b <= 32), no range
// checks.  The per-element VALU work of the general walk cost as much as the AES rounds it follows (b = 20: 0.53 ms against
// 0.29 ms without outputs).
constexpr uint32_t kWalkBatch = 8;                      // plaintext loads in flight per lane (a rolled loop would wait for each)
struct WalkPt { uint32_t v[kWalkBatch]; };

// the first kWalkBatch plaintext words of the lane's walk, requested BEFORE the AES rounds of the stream that completes this
// output: their latency hides under the rounds
template <class ET>
__device__ __forceinline__ WalkPt small_walk32_load(const ET *__restrict__ in, uint64_t e0, uint64_t first, uint32_t lane, uint32_t m)
{
    WalkPt r;
    const ET *pin = in ? in + (e0 - first) + lane : nullptr;
#pragma unroll
    for (uint32_t u = 0; u < kWalkBatch; u++) {
        r.v[u] = 0u;
        // b <= 32: only the low word of the 8-byte element takes part (little endian: the first four bytes); a 4-byte load holds one
        // VGPR while it is in flight instead of two -- up to sixteen fewer live registers across the AES rounds of a pair in a
        // kernel that sits at 128 VGPRs with spills.  Two builds alternated in one process (tests/perf/ab_two_libs.py), ten
        // 1e7-element vectors: b = 16 0.343 -> 0.310 ms, b = 20 0.351 -> 0.346, b = 25 and b = 8 unchanged
#ifdef FLASHE_WALK_LOAD64       // (A/B build: tests/perf/ab_two_libs.py)
        if (pin && u < m) r.v[u] = static_cast<uint32_t>(__builtin_nontemporal_load(pin + 64u * u));
#else
        if (pin && u < m) r.v[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(pin + 64u * u));
#endif
    }
    return r;
}

template <class ET>
__device__ __forceinline__ void small_walk32(uint32_t *row0, uint32_t lane, uint64_t e0, u128 D, const WalkPt &pt0, const ET *__restrict__ in,
                                             ET *__restrict__ out, uint64_t first, const SmallParams &p, uint32_t blk0, uint32_t o0)
{
    *reinterpret_cast<uint4 *>(row0 + 4 * lane) = make_uint4(static_cast<uint32_t>(D), static_cast<uint32_t>(D >> 32),
                                                             static_cast<uint32_t>(D >> 64), static_cast<uint32_t>(D >> 96));
    __builtin_amdgcn_wave_barrier();
    const uint32_t m = static_cast<uint32_t>(p.m), b = static_cast<uint32_t>(p.b);
    const uint32_t q64 = 64u / m, r64b = (64u % m) * b, mb = m * b, mask = static_cast<uint32_t>(p.mask_lo);
    const ET *pin = in ? in + (e0 - first) + lane : nullptr;
    ET *pout = out + (e0 - first) + lane;
    uint32_t blk = blk0, o = o0;
    for (uint32_t i0 = 0; i0 < m; i0 += kWalkBatch) {
        uint32_t pt[kWalkBatch], val[kWalkBatch];
#pragma unroll
        for (uint32_t u = 0; u < kWalkBatch; u++) {
            pt[u] = pt0.v[u];
            if (i0) {                                      // (m > kWalkBatch: b < 16) later batches are loaded here
                pt[u] = 0u;
                if (pin && i0 + u < m) pt[u] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(pin + 64u * (i0 + u)));
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < kWalkBatch; u++) {
            const uint32_t *w = row0 + 4 * blk + (o >> 5);
            // bits o .. o + 31 of the block: o + b <= 128, so past word 3 only bits that the mask removes are read
            val[u] = __builtin_amdgcn_alignbit(w[1], w[0], o & 31u);
            o += r64b; blk += q64;
            if (o >= mb) { o -= mb; blk++; }
            if (blk > 63u) blk = 63u;                     // (slots beyond the tile's last element: not stored)
        }
#pragma unroll
        for (uint32_t u = 0; u < kWalkBatch; u++)
            if (i0 + u < m) __builtin_nontemporal_store(static_cast<ET>((pt[u] + val[u]) & mask), pout + 64u * (i0 + u));
    }
    __builtin_amdgcn_wave_barrier();
}

// m <= 4 (26 <= b <= 64): the elements of a lane's block are adjacent 8-byte words in memory, so the lane adds and stores them
// itself in 16-byte accesses -- one (m = 2) or two (m = 3, 4) per lane for a whole block inside the range, no staging through
// LDS, no index arithmetic (-9.5 % at b = 64, -12 % at b = 40 and 32 on ten 1e7-element vectors; the same in 8-byte accesses
// was 60-130 % SLOWER, and 16-byte accesses at m >= 5 lose too); chunk ends and range ends take the per-element form.
__device__ __forceinline__ void small_direct(bool valid, int cnt, uint64_t j0, u128 D, const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                             uint64_t first, uint64_t range_end, const SmallParams &p)
{
    if (!valid) return;
    if (p.m == 2 && cnt == 2 && j0 >= first && j0 + 2 <= range_end) {
        const uint64_t k = j0 - first;
        u64x2 pt = {0ull, 0ull};
        if (in) pt = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(in + k));
        u64x2 r;
        if (p.b == 64) {            // the two slots are the two halves: no variable 128-bit shift, no mask
            r[0] = pt[0] + static_cast<uint64_t>(D);
            r[1] = pt[1] + static_cast<uint64_t>(D >> 64);
        } else {
            r[0] = (pt[0] + static_cast<uint64_t>(D)) & p.mask_lo;
            r[1] = (pt[1] + static_cast<uint64_t>(D >> p.b)) & p.mask_lo;
        }
        __builtin_nontemporal_store(r, reinterpret_cast<u64x2 *>(out + k));
        return;
    }
    if (cnt == p.m && p.m >= 3 && j0 >= first && j0 + p.m <= range_end) {      // whole block of 3 or 4 elements: two 16-byte accesses
        const uint64_t k = j0 - first;
        u64x2 a = {0ull, 0ull}, c = {0ull, 0ull};
        if (in) {
            a = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(in + k));
            if (p.m == 4) c = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(in + k + 2));
            else c[0] = __builtin_nontemporal_load(in + k + 2);
        }
        u64x2 r0, r1;
        r0[0] = (a[0] + static_cast<uint64_t>(D)) & p.mask_lo;
        r0[1] = (a[1] + static_cast<uint64_t>(D >> p.b)) & p.mask_lo;
        r1[0] = (c[0] + static_cast<uint64_t>(D >> (2 * p.b))) & p.mask_lo;
        r1[1] = (c[1] + static_cast<uint64_t>(D >> (3 * p.b))) & p.mask_lo;
        __builtin_nontemporal_store(r0, reinterpret_cast<u64x2 *>(out + k));
        if (p.m == 4) __builtin_nontemporal_store(r1, reinterpret_cast<u64x2 *>(out + k + 2));
        else __builtin_nontemporal_store(r1[0], out + k + 2);
        return;
    }
    uint64_t pt[4] = {0ull, 0ull, 0ull, 0ull};
    bool ok[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint64_t j = j0 + t;
        ok[t] = t < cnt && j >= first && j < range_end;
        if (ok[t] && in) pt[t] = __builtin_nontemporal_load(in + (j - first));
    }
#pragma unroll
    for (int t = 0; t < 4; t++)
        if (ok[t]) __builtin_nontemporal_store((pt[t] + static_cast<uint64_t>(D >> (p.b * t))) & p.mask_lo, out + (j0 + t - first));
}

// ---- the compact layout at a COMPILE-TIME width (round 5): int_bits in FLASHE_FIXED32_WIDTHS ----
// In the uint32 layout the m = 128 / B elements of a lane's AES block are m adjacent 4-byte words, and consecutive lanes of a whole tile
// hold consecutive blocks: the lane loads, adds and stores ITS OWN block's elements -- 16 + 8 bytes at m = 6, 16 + 4 at m = 5, 16 + 16 at
// m = 8, a wave covers one contiguous run of 64 m words -- with every slot position a compile-time constant: slot t of a stream is one
// v_alignbit / shift of two state words, the client's mask term is slot(add stream) - slot(minus stream) per element (no SWAR
// subtraction with borrow fix-ups over the 128-bit word, no staging of the difference through an LDS row, no funnel shifts at lane-
// dependent word offsets, no walk state), and the plaintext prefetch holds m registers per block across the AES rounds instead of
// eight.  Whole tiles inside the range only; chunk ends and range ends keep the general walk.
// the widths compiled in for the compact layout: what the reference's job configurations run (examples/configs/*_flashe_q16_*: int_bits 20;
// 100 clients: 23; 16) and the byte-sized ones a caller with uint32 arrays picks (8: sixteen elements per
// block in registers spill, measured slower than the run-time width)
#ifndef FLASHE_FIXED32_WIDTHS
#define FLASHE_FIXED32_WIDTHS(X) X(16) X(20) X(23) X(24) X(32)
#endif
static bool fixed32_width(int b)
{
    switch (b) {
#define FLASHE_FIXED32_CASE(B) case B:
        FLASHE_FIXED32_WIDTHS(FLASHE_FIXED32_CASE)
#undef FLASHE_FIXED32_CASE
        return true;
    default: return false;
    }
}
#ifndef FLASHE_SMALL_U2
#define FLASHE_SMALL_U2 1      // small_chain_fast32 takes the second counter shortcut where both of a lane's blocks lie in one 256-counter window each -- on chains of
                               // at most FLASHE_SMALL_U2_STREAMS streams (the decrypt of one vector: 0.0587 -> 0.0571 ms at int_bits 20, -3 ... -5 % at every width).  On the
                               // ten-client chain it LOSES 10-15 % (0.278 -> 0.320 ms: thirty dependent scalar loads and two unpipelined lookup steps at the head
                               // of every one of eleven steps, with one pair per step to spread them over; the wide kernel has two) -- ab_compact_libs.py, round 6
#ifndef FLASHE_SMALL_U2_STREAMS
#define FLASHE_SMALL_U2_STREAMS 2
#endif
#endif
#ifndef FLASHE_SMALL_FAST32
#define FLASHE_SMALL_FAST32 1   // whole tiles of the compile-time-width compact kernels in a loop of their own (small_chain_fast32; 0: the general loop, for A/B builds)
#endif
template <int M> struct DirectPt { uint32_t v[M]; };
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

template <int M>
__device__ __forceinline__ DirectPt<M> direct32_load(const uint32_t *__restrict__ in, uint64_t k)
{
    DirectPt<M> r;
#pragma unroll
    for (int t = 0; t < M; t++) r.v[t] = 0u;
    if (!in) return r;
    const uint32_t *q = in + k;
    int t = 0;
#pragma unroll
    for (; t + 4 <= M; t += 4) {
        const u32x4_a4 x = *reinterpret_cast<const u32x4_a4 *>(q + t);
        r.v[t] = x[0]; r.v[t + 1] = x[1]; r.v[t + 2] = x[2]; r.v[t + 3] = x[3];
    }
    if (M - t >= 2) {
        const u32x2_a4 x = *reinterpret_cast<const u32x2_a4 *>(q + t);
        r.v[t] = x[0]; r.v[t + 1] = x[1];
        t += 2;
    }
    if (t < M) r.v[t] = q[t];
    return r;
}

// out[k + t] = (pt[t] + slot_t(add) - slot_t(minus)) mod 2^B for the M = 128 / B elements of the lane's block; `single`: no minus stream
// (acc: the running sum of the chain's outputs for this block, kept in registers -- a summed chain)
template <int B>
__device__ __forceinline__ void direct32_store(uint32_t *__restrict__ out, uint64_t k, const DirectPt<128 / B> &pt, u128 add, u128 minus, bool single,
                                               uint32_t (&acc)[128 / B])
{
    constexpr int M = 128 / B;
    constexpr uint32_t mask = B >= 32 ? 0xffffffffu : ((1u << (B & 31)) - 1u);
    uint32_t r[M];
#pragma unroll
    for (int t = 0; t < M; t++) {
        const uint32_t a = static_cast<uint32_t>(add >> (B * t)), m = static_cast<uint32_t>(minus >> (B * t));      // (constant shifts: one v_alignbit each)
        r[t] = (pt.v[t] + a - (single ? 0u : m)) & mask;
        acc[t] += r[t];
    }
    uint32_t *q = out + k;
    int t = 0;
#pragma unroll
    for (; t + 4 <= M; t += 4) {
        u32x4_a4 x;
        x[0] = r[t]; x[1] = r[t + 1]; x[2] = r[t + 2]; x[3] = r[t + 3];
        *reinterpret_cast<u32x4_a4 *>(q + t) = x;
    }
    if (M - t >= 2) {
        u32x2_a4 x;
        x[0] = r[t]; x[1] = r[t + 1];
        *reinterpret_cast<u32x2_a4 *>(q + t) = x;
        t += 2;
    }
    if (t < M) q[t] = r[t];
}

// int_bits = 64 in the one-limb layout at compile time (round 5): the two elements of a lane's block are one 16-byte access, the slots are
// the two halves of the block -- the plaintext pair is requested BEFORE the AES rounds of the stream that completes the output (the
// run-time-width kernel loads it behind them), and the instantiation carries none of the walks of the narrower widths (its register
// budget goes to the rounds: the general kernel sits at 128 VGPRs with scratch).  Whole blocks inside the range; the rest: small_direct.
__device__ __forceinline__ u64x2 direct64_load(const uint64_t *__restrict__ in, uint64_t k)
{
    u64x2 r = {0ull, 0ull};
    if (in) r = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(in + k));
    return r;
}
__device__ __forceinline__ void direct64_store(uint64_t *__restrict__ out, uint64_t k, u64x2 pt, u128 add, u128 minus, bool single)
{
    u64x2 r;
    r[0] = pt[0] + static_cast<uint64_t>(add) - (single ? 0ull : static_cast<uint64_t>(minus));
    r[1] = pt[1] + static_cast<uint64_t>(add >> 64) - (single ? 0ull : static_cast<uint64_t>(minus >> 64));
    __builtin_nontemporal_store(r, reinterpret_cast<u64x2 *>(out + k));
}

template <int M>
__device__ __forceinline__ void direct32_put(uint32_t *__restrict__ q, const uint32_t (&r)[M], uint32_t mask)
{
    int t = 0;
#pragma unroll
    for (; t + 4 <= M; t += 4) {
        u32x4_a4 x;
        x[0] = r[t] & mask; x[1] = r[t + 1] & mask; x[2] = r[t + 2] & mask; x[3] = r[t + 3] & mask;
        *reinterpret_cast<u32x4_a4 *>(q + t) = x;
    }
    if (M - t >= 2) {
        u32x2_a4 x;
        x[0] = r[t] & mask; x[1] = r[t + 1] & mask;
        *reinterpret_cast<u32x2_a4 *>(q + t) = x;
        t += 2;
    }
    if (t < M) q[t] = r[t] & mask;
}

// Round 6: the chain of a tile whose TWO blocks per lane both take the direct path (whole blocks inside the range: all but the chunk
// ends) as a loop of its own.  What it drops from the general loop, per stream and block: the selects on `single` (a launch-wide
// flag: fourteen v_cndmask per block at m = 6 -- here a template parameter), the second slot extraction of every stream (a stream is
// client c's minus term AND client c + 1's add term: its slots stay in registers for the next step instead of being shifted out of
// the 128-bit word twice), the per-step tests of the two tile kinds; and the stream's prefix is requested one step ahead (its LDS
// round trip used to stand between a wave and the first lookups of every step).  The running sum of a block's outputs is kept
// UNMASKED (it is masked when it is stored: sums mod 2^32 agree with sums mod 2^B).  m = 6: 30 slot instructions per block and stream
// instead of 52.  In-process A/B (ab_compact_libs.py, AB_SUM=1, ten 1e7-element clients): int_bits 20 0.283-0.296 -> 0.274-0.280 ms,
// 16: 0.2234 -> 0.2156, 23 / 24 / 32: within 1 %.  Measured and dropped (tests/perf/experiments/r06_small_chain_fast32_pipelined.patch):
// the streams software-pipelined against each other (the next stream's first lookups issued before this stream's outputs): 0.2769
// against 0.2740 at int_bits 20, 127 VGPRs -- what is left between two steps is not what holds these kernels back.
template <int B, bool SINGLE>
__device__ __forceinline__ void small_chain_fast32(const RoundKeys &rk, const LaneRegs lr, const uint32_t *pre_lds, int sbase, int n_streams,
                                                   const uint64_t *const *in_tab, uint64_t *const *out_tab, const CtrVar &xA, const CtrVar &xB,
                                                   uint64_t kA, uint64_t kB, uint32_t *sum32, bool prio,
                                                   const uint32_t *__restrict__ te4, bool uni, uint32_t x3A, uint32_t x3B)
{
    constexpr int M = 128 / B;
    constexpr uint32_t mask = B >= 32 ? 0xffffffffu : ((1u << (B & 31)) - 1u);
    uint32_t psA[M], psB[M], accA[M], accB[M];
#pragma unroll
    for (int t = 0; t < M; t++) { psA[t] = 0u; psB[t] = 0u; accA[t] = 0u; accB[t] = 0u; }
    auto prefix_of = [&](const uint4 &v) {
        return CtrPrefix{{static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(v.x)), static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(v.y)),
                          static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(v.z)), static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(v.w))}};
    };
    uint4 pv = *reinterpret_cast<const uint4 *>(pre_lds + 4 * sbase);
    // the outputs of the stream whose blocks are SA / SB (it completes output `link`), and its slots kept for the next step
    auto emit = [&](int link, const u128 SA, const u128 SB, const DirectPt<M> &dA, const DirectPt<M> &dB) {
        uint32_t csA[M], csB[M];
#pragma unroll
        for (int t = 0; t < M; t++) {                            // (constant shifts: one v_alignbit / shift each)
            csA[t] = static_cast<uint32_t>(SA >> (B * t));
            csB[t] = static_cast<uint32_t>(SB >> (B * t));
        }
        if (link >= 0) {
            uint32_t rA[M], rB[M];
#pragma unroll
            for (int t = 0; t < M; t++) {
                const uint32_t ua = dA.v[t] + (SINGLE ? csA[t] : psA[t] - csA[t]), ub = dB.v[t] + (SINGLE ? csB[t] : psB[t] - csB[t]);
                accA[t] += ua; accB[t] += ub;
                rA[t] = ua; rB[t] = ub;
            }
            uint32_t *out = reinterpret_cast<uint32_t *>(out_tab[link]);
            direct32_put<M>(out + kA, rA, mask);
            direct32_put<M>(out + kB, rB, mask);
        }
#pragma unroll
        for (int t = 0; t < M; t++) { psA[t] = csA[t]; psB[t] = csB[t]; }
    };
    for (int c = 0; c < n_streams; c++) {
        const CtrPrefix pre = prefix_of(pv);
        // the next stream's prefix: on its way under this stream's rounds (the last step re-reads its own)
        pv = *reinterpret_cast<const uint4 *>(pre_lds + 4 * (sbase + (c + 1 < n_streams ? c + 1 : c)));
        const int link = SINGLE ? c : c - 1;
        DirectPt<M> dA{}, dB{};
        if (link >= 0) {
            const uint32_t *in = reinterpret_cast<const uint32_t *>(in_tab[link]);
            dA = direct32_load<M>(in, kA);
            dB = direct32_load<M>(in, kB);
        }
        uint32_t s[2][4];
        if (FLASHE_SMALL_U2 && uni) {
            // both blocks' sixty-four counters share bytes 1 .. 3 (x3A / x3B): the second counter shortcut of the wide kernel, 196 lookups
            // per block instead of 208 (the wave-uniform part of rounds 1-2 through the scalar cache, device_common.h)
            const CtrUniform UA = ctr_uniform(rk, te4, pre, x3A), UB = ctr_uniform(rk, te4, pre, x3B);
            ctr_round2(lr, pre.u[0], xA.v[0], UA, s[0]);
            ctr_round2(lr, pre.u[0], xB.v[0], UB, s[1]);
            aes256_rounds<2, 3>(rk, lr, s, prio);
        } else {
            ctr_round1(pre, xA, s[0]);
            ctr_round1(pre, xB, s[1]);
            aes256_rounds<2, 2>(rk, lr, s, prio);
        }
        emit(link, words_to_u128(s[0]), words_to_u128(s[1]), dA, dB);
    }
    if (sum32) {
        direct32_put<M>(sum32 + kA, accA, mask);
        direct32_put<M>(sum32 + kB, accB, mask);
    }
}

// per b-bit slot (prev - cur) mod 2^b of two 128-bit words (SWAR: borrows must not cross slots); b == 64: the slots are the two
// halves and two plain 64-bit subtractions do it (-2 % on ten 1e7-element vectors: the output arithmetic, not the lookup count,
// is what separates this kernel from the wide one -- tests/perf/experiments/r03_small_win_kernel.patch)
__device__ __forceinline__ u128 slot_diff(u128 prev, u128 cur, u128 top, int b)
{
    if (b == 64) {
        const uint64_t lo = static_cast<uint64_t>(prev) - static_cast<uint64_t>(cur), hi = static_cast<uint64_t>(prev >> 64) - static_cast<uint64_t>(cur >> 64);
        return (static_cast<u128>(hi) << 64) | lo;
    }
    return ((prev | top) - (cur & ~top)) ^ ((prev ^ ~cur) & top);
}

// B: 0 = int_bits at run time; FLASHE_FIXED32_WIDTHS (compact layout, PAIR only) = the width as a compile-time constant, whole tiles through
// direct32_load / direct32_store
template <bool PAIR, class ET = uint64_t, int B = 0>
__global__ __launch_bounds__(kSmallThreads) void prf_small_chain_kernel(const RoundKeys rk, const SmallChainTable tb, int n_chains, const SmallParams p)
{
    static_assert(B == 0 || (PAIR && ((sizeof(ET) == 4 && B <= 32) || (sizeof(ET) == 8 && B == 64))), "compile-time widths: the paired kernel, compact layout or int_bits 64");
    constexpr uint32_t WAVES = kSmallThreads / 64, TILE = PAIR ? 128u : 64u;
    constexpr int MB = B ? 128 / B : 1;
    __shared__ uint32_t tab[kTabWords];
    __shared__ uint32_t scratch[(kSmallThreads / 64) * 256 + 8];
    __shared__ __attribute__((aligned(16))) uint32_t pre_lds[(kMaxLinks + kMaxChains) * 4];
    __shared__ uint64_t d_tlo[kMaxChains], d_cend[kMaxChains];
    const uint32_t iter = p.iter + p.te0[kIterShiftWord];
    fill_tables(tab, p.te0);
    const LaneRegs lr = lane_regs(tab);
    {
        const int last = n_chains - 1;
        const int n_streams = tb.sbase[last] + tb.len[last] + ((tb.flags[last] & 1) ? 0 : 1);
        for (int s = threadIdx.x; s < n_streams; s += kSmallThreads) {
            const CtrPrefix c = ctr_prefix(rk, lr, iter, tb.idx[s], 0u);          // n < 2^32 (host-checked): the high counter word is 0
            *reinterpret_cast<uint4 *>(pre_lds + 4 * s) = make_uint4(c.u[0], c.u[1], c.u[2], c.u[3]);
        }
        if (threadIdx.x < static_cast<unsigned>(n_chains)) {                          // this workgroup's tiles of every chain (see prf_chain_kernel)
            const int i = threadIdx.x;
            const uint64_t Wt = tb.wend[last], cw = i ? tb.wend[i - 1] : 0;
            const uint32_t w = tb.len[i] + ((tb.flags[i] & 1) ? 0u : 1u);
            uint64_t a, b, T;
            if (Wt <= 0xffffffffull && gridDim.x <= 0xffffu) {
                const uint32_t W32 = static_cast<uint32_t>(Wt), G = gridDim.x, g = blockIdx.x, c32 = static_cast<uint32_t>(cw);
                const uint32_t q = W32 / G, r = W32 % G;
                const uint32_t lo = q * g + r * g / G, hi = q * (g + 1) + r * (g + 1) / G;
                T = (static_cast<uint32_t>(tb.wend[i]) - c32) / w;
                a = lo > c32 ? (lo - c32 + w - 1) / w : 0; b = hi > c32 ? (hi - c32 + w - 1) / w : 0;
            } else {
                const uint64_t G = gridDim.x, g = blockIdx.x;
                const uint64_t lo = Wt / G * g + (Wt % G) * g / G, hi = Wt / G * (g + 1) + (Wt % G) * (g + 1) / G;
                T = (tb.wend[i] - cw) / w;
                a = lo > cw ? (lo - cw + w - 1) / w : 0; b = hi > cw ? (hi - cw + w - 1) / w : 0;
            }
            if (a > T) a = T;
            if (b > T) b = T;
            d_tlo[i] = a; d_cend[i] = b - a;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t acc = 0;
            for (int i = 0; i < n_chains; i++) { acc += d_cend[i]; d_cend[i] = acc; }
        }
        __syncthreads();
    }
    const uint64_t J = p.n_jobs, d = p.n / J, r = p.n % J, m64 = static_cast<uint64_t>(p.m);
    const uint32_t nb1_32 = static_cast<uint32_t>((d + 1 + m64 - 1) / m64), nb0_32 = static_cast<uint32_t>(d ? (d + m64 - 1) / m64 : 0);
    const uint32_t d32 = static_cast<uint32_t>(d), r32 = static_cast<uint32_t>(r), m32 = static_cast<uint32_t>(p.m);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    uint32_t *row0 = scratch + wave * 256;
    const u128 top = (static_cast<u128>(p.top_hi) << 64) | p.top_lo;
    const bool direct = p.m <= 4 && !p.no_direct;
    const bool walk32 = p.b <= 32 && !direct && p.no_direct != 2;     // (FLASHE_SMALL_DIRECT=2: A/B knob, general walk everywhere)
    const uint32_t wblk0 = lane / m32, wo0 = (lane - wblk0 * m32) * static_cast<uint32_t>(p.b);       // m >= 5 measured 35-150 % slower than the staged walk (40 .. 64-byte lane stride)      // m = 3, 4 measured: 8-byte accesses at a 24 / 32-byte lane stride lose 60-130 % against the staged walk
    const uint64_t Ng = uniform64(d_cend[n_chains - 1]);
    int cur = 0;
    uint64_t cbeg = 0;
    for (uint64_t q = wave; q < Ng; q += WAVES) {
        while (q >= uniform64(d_cend[cur])) cbeg = uniform64(d_cend[cur++]);
        const uint64_t first = tb.first[cur], range_end = first + tb.count[cur], blk_count = tb.blk_count[cur];
        const uint64_t Bw = (uniform64(d_tlo[cur]) + (q - cbeg)) * TILE;             // the tile's first block (chain-local)
        const int link0 = tb.link0[cur], sbase = tb.sbase[cur];
        const bool single = tb.flags[cur] & 1;
        const int n_streams = tb.len[cur] + (single ? 0 : 1);
        // per-lane block(s): chunk arithmetic and the counter-dependent quarter of round 1, once for all streams
        const bool vA = Bw + lane < blk_count, vB = PAIR && Bw + 64u + lane < blk_count;
        uint64_t j0A = 0, j0B = 0;
        int cntA = 0, cntB = 0;
        uint32_t ctrA = 0, ctrB = 0;
        small_block_params(static_cast<uint32_t>(tb.blk_first[cur] + (vA ? Bw + lane : 0)), nb1_32, nb0_32, d32, r32, m32, p, &j0A, &cntA, &ctrA);
        if (PAIR) small_block_params(static_cast<uint32_t>(tb.blk_first[cur] + (vB ? Bw + 64u + lane : 0)), nb1_32, nb0_32, d32, r32, m32, p, &j0B, &cntB, &ctrB);
        const CtrVar xA = ctr_var(rk, lr, ctrA);
        CtrVar xB{};
        if (PAIR) xB = ctr_var(rk, lr, ctrB);
        // the common tile: 64 whole blocks, all inside the range -> their 64 m elements are one contiguous run
        uint64_t e0A = 0, e0B = 0;
        bool fastA = false, fastB = false;
        if (walk32) {
            e0A = uniform64(j0A);
            fastA = __ballot(vA && cntA == p.m) == ~0ull && e0A >= first && e0A + 64u * m64 <= range_end;
            if (PAIR) {
                e0B = uniform64(j0B);
                fastB = __ballot(vB && cntB == p.m) == ~0ull && e0B >= first && e0B + 64u * m64 <= range_end;
            }
        }
        // (int_bits 64 at compile time: the lane's block is whole and inside the range -> one 16-byte access)
        const bool wholeA = B == 64 && vA && cntA == 2 && j0A >= first && j0A + 2 <= range_end;
        const bool wholeB = B == 64 && vB && cntB == 2 && j0B >= first && j0B + 2 <= range_end;
        u128 prevA = 0, prevB = 0;
        // (a summed chain, compile-time width: the blocks' running sums; irregular tiles keep theirs in memory, see small_walk)
        uint32_t accA[MB], accB[MB];
#pragma unroll
        for (int t = 0; t < MB; t++) { accA[t] = 0u; accB[t] = 0u; }
        uint32_t *const sum32 = B != 0 && B != 64 ? reinterpret_cast<uint32_t *>(tb.sum_out[cur]) : nullptr;
#if FLASHE_SMALL_FAST32
        if constexpr (PAIR && B != 0 && B != 64) {
            if (fastA && fastB) {                                  // (wave-uniform) both blocks of every lane whole and inside the range
                // the lanes' counters are consecutive inside a chunk; where a set of sixty-four does not cross a multiple of 256 its
                // bytes 1 .. 3 are the wave's (three sets of four in the chunks whose first counter is not a multiple of 64)
                const uint32_t bA = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(ctrA)), bB = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(ctrB));
                const bool uni = FLASHE_SMALL_U2 && n_streams <= FLASHE_SMALL_U2_STREAMS && __ballot(ctrA - bA == lane && ctrB - bB == lane) == ~0ull &&
                                 (bA & 255u) <= 192u && (bB & 255u) <= 192u;
                if (single) small_chain_fast32<B, true>(rk, lr, pre_lds, sbase, n_streams, tb.in + link0, tb.out + link0, xA, xB, j0A - first, j0B - first, sum32, p.swp_prio != 0,
                                                        p.te0, uni, bA ^ rk.w[3], bB ^ rk.w[3]);
                else small_chain_fast32<B, false>(rk, lr, pre_lds, sbase, n_streams, tb.in + link0, tb.out + link0, xA, xB, j0A - first, j0B - first, sum32, p.swp_prio != 0,
                                                  p.te0, uni, bA ^ rk.w[3], bB ^ rk.w[3]);
                continue;
            }
        }
#endif
        if (PAIR) {
            // two blocks per lane on the same prefix, one stream per step
            for (int c = 0; c < n_streams; c++) {
                const CtrPrefix pre = load_prefix(pre_lds, sbase + c);
                const int link = single ? c : c - 1;
                WalkPt ptA{}, ptB{};
                DirectPt<MB> dA{}, dB{};
                u64x2 qA = {0ull, 0ull}, qB = {0ull, 0ull};
                if (B == 64) {
                    if (link >= 0 && wholeA) qA = direct64_load(tb.in[link0 + link], j0A - first);
                    if (link >= 0 && wholeB) qB = direct64_load(tb.in[link0 + link], j0B - first);
                } else if (B) {
                    if (link >= 0 && fastA) dA = direct32_load<MB>(reinterpret_cast<const uint32_t *>(tb.in[link0 + link]), j0A - first);
                    if (link >= 0 && fastB) dB = direct32_load<MB>(reinterpret_cast<const uint32_t *>(tb.in[link0 + link]), j0B - first);
                } else {
                    if (link >= 0 && fastA) ptA = small_walk32_load(reinterpret_cast<const ET *>(tb.in[link0 + link]), e0A, first, lane, m32);
                    if (link >= 0 && fastB) ptB = small_walk32_load(reinterpret_cast<const ET *>(tb.in[link0 + link]), e0B, first, lane, m32);
                }
                uint32_t s[2][4];
                ctr_round1(pre, xA, s[0]);
                ctr_round1(pre, xB, s[1]);
                aes256_rounds<2, 2>(rk, lr, s, p.swp_prio != 0);
                const u128 SA = words_to_u128(s[0]), SB = words_to_u128(s[1]);
                if (link >= 0) {
                    const uint64_t *in = tb.in[link0 + link];
                    uint64_t *out = tb.out[link0 + link];
                    const ET *ein = reinterpret_cast<const ET *>(in);
                    ET *eout = reinterpret_cast<ET *>(out);
                    if (B == 64) {
                        if constexpr (B == 64) {
                            if (wholeA) direct64_store(out, j0A - first, qA, single ? SA : prevA, SA, single);
                            else small_direct(vA, cntA, j0A, single ? SA : slot_diff(prevA, SA, top, 64), in, out, first, range_end, p);
                            if (wholeB) direct64_store(out, j0B - first, qB, single ? SB : prevB, SB, single);
                            else small_direct(vB, cntB, j0B, single ? SB : slot_diff(prevB, SB, top, 64), in, out, first, range_end, p);
                        }
                        prevA = SA; prevB = SB;
                        continue;
                    }
                    if (B) {
                        // compile-time width: whole tiles element by element from the two streams' slots, everything else the general walk
                        if constexpr (B != 0 && B != 64) {
                            uint32_t *o32 = reinterpret_cast<uint32_t *>(out);
                            ET *const sm = reinterpret_cast<ET *>(sum32);
                            if (fastA) direct32_store<B>(o32, j0A - first, dA, single ? SA : prevA, SA, single, accA);
                            else small_walk(row0, lane, vA, cntA, j0A, single ? SA : slot_diff(prevA, SA, top, p.b), ein, eout, first, range_end, p, sm, link == 0);
                            if (fastB) direct32_store<B>(o32, j0B - first, dB, single ? SB : prevB, SB, single, accB);
                            else small_walk(row0, lane, vB, cntB, j0B, single ? SB : slot_diff(prevB, SB, top, p.b), ein, eout, first, range_end, p, sm, link == 0);
                        }
                        prevA = SA; prevB = SB;
                        continue;
                    }
                    // per slot (previous - current) mod 2^b: the previous stream is this client's add stream, the current its minus stream
                    const u128 DA = single ? SA : slot_diff(prevA, SA, top, p.b);
                    const u128 DB = single ? SB : slot_diff(prevB, SB, top, p.b);
                    if (direct) {                                  // (never with the compact layout: the host turns `direct` off)
                        small_direct(vA, cntA, j0A, DA, in, out, first, range_end, p);
                        small_direct(vB, cntB, j0B, DB, in, out, first, range_end, p);
                    } else {
                        if (fastA) small_walk32(row0, lane, e0A, DA, ptA, ein, eout, first, p, wblk0, wo0);
                        else small_walk(row0, lane, vA, cntA, j0A, DA, ein, eout, first, range_end, p);
                        if (fastB) small_walk32(row0, lane, e0B, DB, ptB, ein, eout, first, p, wblk0, wo0);
                        else small_walk(row0, lane, vB, cntB, j0B, DB, ein, eout, first, range_end, p);
                    }
                }
                prevA = SA; prevB = SB;
            }
            if constexpr (B != 0 && B != 64) {
                if (sum32) {                                         // the sums of the blocks that took the direct path, one store each
                    constexpr uint32_t bmask = B >= 32 ? 0xffffffffu : ((1u << (B & 31)) - 1u);
                    if (fastA) direct32_put<MB>(sum32 + (j0A - first), accA, bmask);
                    if (fastB) direct32_put<MB>(sum32 + (j0B - first), accB, bmask);
                }
            }
        } else {
            // one block per lane, TWO STREAMS per step (short launches: half the dependent AES depth per wave; an odd stream count
            // computes its last stream twice)
            for (int c = 0; c < n_streams; c += 2) {
                const bool has1 = c + 1 < n_streams;
                const CtrPrefix pre0 = load_prefix(pre_lds, sbase + c), pre1 = load_prefix(pre_lds, sbase + (has1 ? c + 1 : c));
                const int l0 = single ? c : c - 1;
                WalkPt pt0{}, pt1{};
                if (fastA && l0 >= 0) pt0 = small_walk32_load(reinterpret_cast<const ET *>(tb.in[link0 + l0]), e0A, first, lane, m32);
                if (fastA && has1) pt1 = small_walk32_load(reinterpret_cast<const ET *>(tb.in[link0 + l0 + 1]), e0A, first, lane, m32);
                uint32_t s[2][4];
                ctr_round1(pre0, xA, s[0]);
                ctr_round1(pre1, xA, s[1]);
                aes256_rounds<2, 2>(rk, lr, s, FLASHE_SMALL_NP_PRIO != 0);
                const u128 S0 = words_to_u128(s[0]), S1 = words_to_u128(s[1]);
                if (l0 >= 0) {
                    const u128 D = single ? S0 : slot_diff(prevA, S0, top, p.b);
                    const ET *ein = reinterpret_cast<const ET *>(tb.in[link0 + l0]);
                    ET *eout = reinterpret_cast<ET *>(tb.out[link0 + l0]);
                    if (direct) small_direct(vA, cntA, j0A, D, tb.in[link0 + l0], tb.out[link0 + l0], first, range_end, p);
                    else if (fastA) small_walk32(row0, lane, e0A, D, pt0, ein, eout, first, p, wblk0, wo0);
                    else small_walk(row0, lane, vA, cntA, j0A, D, ein, eout, first, range_end, p);
                }
                if (has1) {
                    const u128 D = single ? S1 : slot_diff(S0, S1, top, p.b);
                    const ET *ein = reinterpret_cast<const ET *>(tb.in[link0 + l0 + 1]);
                    ET *eout = reinterpret_cast<ET *>(tb.out[link0 + l0 + 1]);
                    if (direct) small_direct(vA, cntA, j0A, D, tb.in[link0 + l0 + 1], tb.out[link0 + l0 + 1], first, range_end, p);
                    else if (fastA) small_walk32(row0, lane, e0A, D, pt1, ein, eout, first, p, wblk0, wo0);
                    else small_walk(row0, lane, vA, cntA, j0A, D, ein, eout, first, range_end, p);
                }
                prevA = has1 ? S1 : S0;
            }
        }
    }
}


// ---- b <= 64: the reduce fused with the decrypt of its result (one add, at most one minus prefix) ----
// out[j] = (sum_c ct_c[j] + term(add, j) - term(minus, j)) mod 2^b (jzf_aggregator.py:424-430 followed by jzf_flashe.py:570-571 with the
// telescoped prefixes of :356-367) in ONE pass: a wave owns 64 consecutive AES blocks = 64 m consecutive elements, runs the add and the
// minus block of its lane as one software-pipelined pair, puts the per-slot difference into its LDS row and then walks the 64 m
// elements lane-contiguously, adding the C operands as it goes.  The element-wise reduce streams the C ciphertexts either way
// (8 (C + 1) bytes per element); what this saves is the aggregate's round trip (write + read + write of 8 bytes per element) and a
// launch, and the AES of one wave hides under the operand stream of the others.
template <int CB, int EPL>
__global__ __launch_bounds__(kSmallThreads) void small_reduce_decrypt_kernel(const RoundKeys rk, const SmallParams p, uint32_t add_idx, uint32_t minus_idx,
                                                                               int has_minus, uint64_t first, uint64_t count, uint64_t blk_first,
                                                                               uint64_t blk_count, int C, const PtrTable ops, uint64_t *agg_out,
                                                                               uint64_t *out)
{
    constexpr uint32_t WAVES = kSmallThreads / 64;
    __shared__ uint32_t tab[kTabWords];
    __shared__ uint32_t scratch[WAVES * 256 + 8];
    const uint32_t iter = p.iter + p.te0[kIterShiftWord];
    fill_tables(tab, p.te0);
    const LaneRegs lr = lane_regs(tab);
    const CtrPrefix pre_a = scalar_prefix(ctr_prefix(rk, lr, iter, add_idx, 0u));            // n < 2^32 (host-checked)
    const CtrPrefix pre_m = scalar_prefix(ctr_prefix(rk, lr, iter, minus_idx, 0u));
    const uint64_t J = p.n_jobs, d = p.n / J, r = p.n % J, m64 = static_cast<uint64_t>(p.m);
    const uint32_t nb1_32 = static_cast<uint32_t>((d + 1 + m64 - 1) / m64), nb0_32 = static_cast<uint32_t>(d ? (d + m64 - 1) / m64 : 0);
    const uint32_t d32 = static_cast<uint32_t>(d), r32 = static_cast<uint32_t>(r), m32 = static_cast<uint32_t>(p.m);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    uint32_t *row0 = scratch + wave * 256;
    const u128 top = (static_cast<u128>(p.top_hi) << 64) | p.top_lo;
    const uint64_t range_end = first + count;
    const uint64_t n_tiles = (blk_count + 63u) / 64u;
    const uint64_t *const *tab_ops = ops.p;
    // a workgroup's sixteen waves take sixteen consecutive tiles at a time: 16 x 64 m contiguous elements of every operand
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * WAVES + wave; t < n_tiles; t += static_cast<uint64_t>(gridDim.x) * WAVES) {
        const bool valid = t * 64u + lane < blk_count;
        uint64_t j0 = 0;
        int cnt = 0;
        uint32_t ctr = 0;
        small_block_params(static_cast<uint32_t>(blk_first + (valid ? t * 64u + lane : 0)), nb1_32, nb0_32, d32, r32, m32, p, &j0, &cnt, &ctr);
        const CtrVar x = ctr_var(rk, lr, ctr);
        uint32_t s[2][4];
        ctr_round1(pre_a, x, s[0]);
        ctr_round1(pre_m, x, s[1]);
#ifdef FLASHE_TUNING
        if (!(has_minus & 0x100))                                          // (0x100: timing probe without the rounds, FLASHE_SMALL_REDUCE_PROBE)
#endif
        aes256_rounds<2, 2>(rk, lr, s, p.swp_prio != 0);
        const u128 Sa = words_to_u128(s[0]), Sm = words_to_u128(s[1]);
        const u128 D = (has_minus & 1) ? slot_diff(Sa, Sm, top, p.b) : Sa;
        *reinterpret_cast<uint4 *>(row0 + 4 * lane) = make_uint4(static_cast<uint32_t>(D), static_cast<uint32_t>(D >> 32),
                                                                 static_cast<uint32_t>(D >> 64), static_cast<uint32_t>(D >> 96));
        __builtin_amdgcn_wave_barrier();
        // the tile's elements: blocks are whole except the last one of a chunk, so the 64 blocks are runs of consecutive elements
        // broken only after a partial block; each run is walked lane-contiguously
        const uint64_t valid_mask = __ballot(valid);
        uint64_t partial_mask = __ballot(valid && cnt < p.m);
        const int n_valid = __popcll(valid_mask);
        const uint32_t j0_lo = static_cast<uint32_t>(j0), j0_hi = static_cast<uint32_t>(j0 >> 32);
        int lane_base = 0;
        while (lane_base < n_valid) {
            const int P = partial_mask ? static_cast<int>(__ffsll(static_cast<unsigned long long>(partial_mask))) - 1 : -1;
            const uint64_t e0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(j0_lo, lane_base)) |
                                (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(j0_hi, lane_base))) << 32);
            const uint32_t n_elems = P >= 0 ? static_cast<uint32_t>(P - lane_base) * m32 + static_cast<uint32_t>(__builtin_amdgcn_readlane(cnt, P))
                                            : static_cast<uint32_t>(n_valid - lane_base) * m32;
            for (uint32_t x0 = lane; x0 < n_elems; x0 += 64u * EPL) {
                // EPL elements per lane per step (x0, x0 + 64, ...), CB operands of each requested before the first is used
                uint64_t k[EPL], sum[EPL];
                bool ok[EPL];
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    const uint32_t x = x0 + 64u * e;
                    const uint64_t j = e0 + x;
                    ok[e] = x < n_elems && j >= first && j < range_end;
                    k[e] = ok[e] ? j - first : (e ? k[e - 1] : 0);
                    sum[e] = 0;
                }
                for (int c = 0; c < C; c += CB) {
                    uint64_t v[CB][EPL];
#pragma unroll
                    for (int u = 0; u < CB; u++) {
                        const uint64_t *src = tab_ops[c + u < C ? c + u : C - 1];       // surplus slots of the last step re-read an operand and are not added
#pragma unroll
                        for (int e = 0; e < EPL; e++) v[u][e] = __builtin_nontemporal_load(src + k[e]);
                    }
#pragma unroll
                    for (int u = 0; u < CB; u++) {
                        if (c + u < C) {
#pragma unroll
                            for (int e = 0; e < EPL; e++) sum[e] += v[u][e];
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < EPL; e++) {
                    if (!ok[e]) continue;
                    const uint32_t x = x0 + 64u * e;
                    const uint32_t blk = static_cast<uint32_t>((static_cast<uint64_t>(x) * p.m_magic) >> 32);
                    const uint32_t o = static_cast<uint32_t>(p.b) * (x - blk * m32);
                    const uint32_t *w = row0 + 4 * (lane_base + blk) + (o >> 5);
                    const uint32_t sh = o & 31u;
                    uint64_t val = ((static_cast<uint64_t>(w[1]) << 32) | w[0]) >> sh;
                    if (sh) val |= static_cast<uint64_t>(w[2]) << (64u - sh);
                    if (agg_out) __builtin_nontemporal_store(sum[e] & p.mask_lo, agg_out + k[e]);
                    __builtin_nontemporal_store((sum[e] + val) & p.mask_lo, out + k[e]);
                }
            }
            if (P < 0) break;
            lane_base = P + 1;
            partial_mask &= partial_mask - 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}


// The same for b <= 32 (m >= 4) in tiles of 32 blocks: lanes 0-31 run the add-stream block, lanes 32-63 the minus-stream block of the
// SAME 32 AES blocks (one block per lane), both go to the wave's LDS row, and the walk takes slot(add) - slot(minus) per element.
// A 64-block tile is 64 m elements -- 1,024 at b = 8 -- and a 1e7-element vector then has only 2.4 tiles per wave of the chip: the
// last, partly filled round of tiles cost up to 26 % (a wave streams no faster because its neighbours are idle).  Half the tile size
// halves that, and with b <= 32 the sums and the slots are 32-bit: operands are read as 4-byte low words.
// (Built and dropped: the sixteen waves of a workgroup meeting at a barrier and walking their 512 m elements TOGETHER, 1,024
// consecutive elements of every operand per step -- 0.19 ms without the AES rounds where the per-wave walk takes 0.18, and with them
// 0.23 against 0.187: one workgroup per CU in lockstep means nobody streams while everybody runs its rounds.)
// IT / OT: element type of the operands / of agg_out and out in memory (uint64_t, or uint32_t for the compact layout)
template <int CB, class IT = uint64_t, class OT = uint64_t>
__global__ __launch_bounds__(kSmallThreads) void small_reduce_decrypt_split_kernel(const RoundKeys rk, const SmallParams p, uint32_t add_idx,
                                                                                     uint32_t minus_idx, int has_minus, uint64_t first, uint64_t count,
                                                                                     uint64_t blk_first, uint64_t blk_count, int C, const PtrTable ops,
                                                                                     uint64_t *agg_out_, uint64_t *out_)
{
    OT *const agg_out = reinterpret_cast<OT *>(agg_out_);
    OT *const out = reinterpret_cast<OT *>(out_);
    constexpr uint32_t WAVES = kSmallThreads / 64, EPL = 2;
    __shared__ uint32_t tab[kTabWords];
    __shared__ uint32_t scratch[WAVES * 256 + 8];
    const uint32_t iter = p.iter + p.te0[kIterShiftWord];
    fill_tables(tab, p.te0);
    const LaneRegs lr = lane_regs(tab);
    const uint32_t lane = threadIdx.x & 63u, half = lane >> 5, l32 = lane & 31u;
    const CtrPrefix pre = ctr_prefix(rk, lr, iter, half ? minus_idx : add_idx, 0u);          // n < 2^32 (host-checked)
    const uint64_t J = p.n_jobs, d = p.n / J, r = p.n % J, m64 = static_cast<uint64_t>(p.m);
    const uint32_t nb1_32 = static_cast<uint32_t>((d + 1 + m64 - 1) / m64), nb0_32 = static_cast<uint32_t>(d ? (d + m64 - 1) / m64 : 0);
    const uint32_t d32 = static_cast<uint32_t>(d), r32 = static_cast<uint32_t>(r), m32 = static_cast<uint32_t>(p.m);
    const uint32_t wave = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    uint32_t *row0 = scratch + wave * 256;
    const uint32_t mask = static_cast<uint32_t>(p.mask_lo);
    const uint64_t range_end = first + count;
    const uint64_t n_tiles = (blk_count + 31u) / 32u;
    const uint64_t *const *tab_ops = ops.p;
    const bool drop = half && !(has_minus & 1);                   // no minus prefix: the upper half contributes zeros
    for (uint64_t t = static_cast<uint64_t>(blockIdx.x) * WAVES + wave; t < n_tiles; t += static_cast<uint64_t>(gridDim.x) * WAVES) {
        const bool valid = t * 32u + l32 < blk_count;
        uint64_t j0 = 0;
        int cnt = 0;
        uint32_t ctr = 0;
        small_block_params(static_cast<uint32_t>(blk_first + (valid ? t * 32u + l32 : 0)), nb1_32, nb0_32, d32, r32, m32, p, &j0, &cnt, &ctr);
        const CtrVar x = ctr_var(rk, lr, ctr);
        uint32_t s[1][4];
        ctr_round1(pre, x, s[0]);
#ifdef FLASHE_TUNING
        if (!(has_minus & 0x100))                                        // (0x100: timing probe without the rounds)
#endif
        aes256_rounds1_deep<2>(rk, lr, s[0], FLASHE_DEEP_PRIO != 0);     // (one block per lane: all sixteen lookups of a round in flight)
        // row word order = little-endian words of the 128-bit block value (word 0 = bits 0..31)
        *reinterpret_cast<uint4 *>(row0 + 4 * lane) = drop ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(s[0][3], s[0][2], s[0][1], s[0][0]);
        __builtin_amdgcn_wave_barrier();
        const uint32_t valid_mask = static_cast<uint32_t>(__ballot(valid));
        uint32_t partial_mask = static_cast<uint32_t>(__ballot(valid && cnt < p.m));
        const int n_valid = __popc(valid_mask);
        const uint32_t j0_lo = static_cast<uint32_t>(j0), j0_hi = static_cast<uint32_t>(j0 >> 32);
        int lane_base = 0;
        if constexpr (sizeof(IT) == 4 && sizeof(OT) == 4) {      // (measured without gain in the one-limb layout: 0.212 against 0.208 ms at b = 20 -- that walk streams 880 MB at 4.2 TB/s as it is)
            // The regular tile -- 32 whole blocks inside the range (0x200: the host checked that every pointer is 16-byte aligned): a lane
            // takes E consecutive elements of its operands in ONE access of 4 E bytes and makes one store, up to eight operands in flight
            // -- a fraction of the memory instructions of the element-per-lane walk and more bytes in flight per lane, which is what this
            // HBM-bound pass is short of (b <= 32: the sums and the slots are 32-bit).  E = 2, 3 or 4 (bits 10-11 of has_minus, chosen by the
            // host, round 5) so that the tile's 32 m elements FILL the wave's accesses: at m = 6 (int_bits 20) a tile is 192 elements -- 48
            // lanes of 16-byte accesses (round 4: 75 % of every access, 0.119 ms for ten 1e7-element operands), exactly 64 lanes of 12-byte
            // ones; m = 4: 128 elements = 64 lanes of 8 bytes.
            const uint64_t e0t = static_cast<uint32_t>(__builtin_amdgcn_readlane(j0_lo, 0)) |
                                 (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(j0_hi, 0))) << 32);
            const uint32_t epl = (static_cast<uint32_t>(has_minus) >> 10) & 3u ? (static_cast<uint32_t>(has_minus) >> 10) & 3u : 4u;      // 0 = 4
            if ((has_minus & 0x200) && valid_mask == 0xffffffffu && partial_mask == 0 && e0t >= first && e0t + 32u * m32 <= range_end &&
                (epl != 4u || ((e0t - first) & 3u) == 0)) {
                const uint64_t k0 = e0t - first;
                auto regular_tile = [&](auto e_tag) {
                    constexpr uint32_t E = decltype(e_tag)::value;
                    // (16-byte accesses are aligned: checked above; the narrower ones need their 4 bytes only)
                    typedef uint32_t vecE __attribute__((ext_vector_type(E), aligned(E == 4 ? 16 : 4)));
                    constexpr int QB = 8;                              // operands in flight per lane (C is wave-uniform: surplus slots issue nothing)
                    for (uint32_t qd = lane; E * qd < 32u * m32; qd += 64u) {
                        const uint64_t kq = k0 + E * qd;
                        uint32_t sum[E];
#pragma unroll
                        for (uint32_t tq = 0; tq < E; tq++) sum[tq] = 0u;
                        for (int c = 0; c < C; c += QB) {
                            vecE v[QB];
#pragma unroll
                            for (int u = 0; u < QB; u++) {
                                v[u] = vecE(0u);
                                if (c + u < C) v[u] = __builtin_nontemporal_load(reinterpret_cast<const vecE *>(reinterpret_cast<const IT *>(tab_ops[c + u]) + kq));
                            }
#pragma unroll
                            for (int u = 0; u < QB; u++) {
#pragma unroll
                                for (uint32_t tq = 0; tq < E; tq++) sum[tq] += v[u][tq];
                            }
                        }
                        vecE res = vecE(0u), ag = vecE(0u);
#pragma unroll
                        for (uint32_t tq = 0; tq < E; tq++) {
                            const uint32_t xx = E * qd + tq;
                            const uint32_t blk = static_cast<uint32_t>((static_cast<uint64_t>(xx) * p.m_magic) >> 32);
                            const uint32_t o = static_cast<uint32_t>(p.b) * (xx - blk * m32);
                            const uint32_t *wa = row0 + 4 * blk + (o >> 5), *wm = wa + 128;
                            res[tq] = (sum[tq] + __builtin_amdgcn_alignbit(wa[1], wa[0], o & 31u) - __builtin_amdgcn_alignbit(wm[1], wm[0], o & 31u)) & mask;
                            ag[tq] = sum[tq] & mask;
                        }
                        if (agg_out) __builtin_nontemporal_store(ag, reinterpret_cast<vecE *>(agg_out + kq));
                        __builtin_nontemporal_store(res, reinterpret_cast<vecE *>(out + kq));
                    }
                };
                if (epl == 3u) regular_tile(std::integral_constant<uint32_t, 3>{});
                else if (epl == 2u) regular_tile(std::integral_constant<uint32_t, 2>{});
                else regular_tile(std::integral_constant<uint32_t, 4>{});
                __builtin_amdgcn_wave_barrier();
                continue;
            }
        }
        while (lane_base < n_valid) {
            const int P = partial_mask ? static_cast<int>(__ffs(partial_mask)) - 1 : -1;
            const uint64_t e0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(j0_lo, lane_base)) |
                                (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(j0_hi, lane_base))) << 32);
            const uint32_t n_elems = P >= 0 ? static_cast<uint32_t>(P - lane_base) * m32 + static_cast<uint32_t>(__builtin_amdgcn_readlane(cnt, P))
                                            : static_cast<uint32_t>(n_valid - lane_base) * m32;
            for (uint32_t x0 = lane; x0 < n_elems; x0 += 64u * EPL) {
                uint64_t k[EPL];
                uint32_t sum[EPL];
                bool ok[EPL];
#pragma unroll
                for (uint32_t e = 0; e < EPL; e++) {
                    const uint32_t xx = x0 + 64u * e;
                    const uint64_t j = e0 + xx;
                    ok[e] = xx < n_elems && j >= first && j < range_end;
                    k[e] = ok[e] ? j - first : (e ? k[e - 1] : 0);
                    sum[e] = 0;
                }
                for (int c = 0; c < C; c += CB) {
                    uint32_t v[CB][EPL];
#pragma unroll
                    for (int u = 0; u < CB; u++) {
                        const uint64_t *src = tab_ops[c + u < C ? c + u : C - 1];       // surplus slots of the last step re-read an operand and are not added
#pragma unroll
                        for (uint32_t e = 0; e < EPL; e++) v[u][e] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(reinterpret_cast<const IT *>(src) + k[e]));
                    }
#pragma unroll
                    for (int u = 0; u < CB; u++) {
                        if (c + u < C) {
#pragma unroll
                            for (uint32_t e = 0; e < EPL; e++) sum[e] += v[u][e];
                        }
                    }
                }
#pragma unroll
                for (uint32_t e = 0; e < EPL; e++) {
                    if (!ok[e]) continue;
                    const uint32_t xx = x0 + 64u * e;
                    const uint32_t blk = static_cast<uint32_t>((static_cast<uint64_t>(xx) * p.m_magic) >> 32);
                    const uint32_t o = static_cast<uint32_t>(p.b) * (xx - blk * m32);
                    const uint32_t *wa = row0 + 4 * (lane_base + blk) + (o >> 5), *wm = wa + 128;
                    // bits o .. o + 31 of the block: o + b <= 128, so past word 3 only bits that the mask removes are read
                    const uint32_t val = __builtin_amdgcn_alignbit(wa[1], wa[0], o & 31u) - __builtin_amdgcn_alignbit(wm[1], wm[0], o & 31u);
                    if (agg_out) __builtin_nontemporal_store(static_cast<OT>(sum[e] & mask), agg_out + k[e]);
                    __builtin_nontemporal_store(static_cast<OT>((sum[e] + val) & mask), out + k[e]);
                }
            }
            if (P < 0) break;
            lane_base = P + 1;
            partial_mask &= partial_mask - 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Known-answer helper: raw AES of nblk blocks given as big-endian words.
__global__ __launch_bounds__(kPrfThreads) void aes_blocks_kernel(const RoundKeys rk, const uint32_t *te0, uint32_t nblk,
                                                                 const uint32_t *in, uint32_t *out)
{
    __shared__ uint32_t tab[kTabWords];
    fill_tables(tab, te0);
    const LaneRegs lr = lane_regs(tab);
    for (uint32_t i = blockIdx.x * kPrfThreads + threadIdx.x; i < nblk; i += gridDim.x * kPrfThreads) {
        uint32_t s[1][4] = {{in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]}};
        aes256_encrypt<1>(rk, lr, s);
        out[4 * i] = s[0][0]; out[4 * i + 1] = s[0][1]; out[4 * i + 2] = s[0][2]; out[4 * i + 3] = s[0][3];
    }
}

// AES-block index (in the kernel's global block numbering) of element j of an n-vector.
static uint64_t block_of(uint64_t j, uint64_t n, uint64_t J, uint64_t m)
{
    const uint64_t d = n / J, r = n % J;
    const uint64_t nb1 = (d + 1 + m - 1) / m, nb0 = d ? (d + m - 1) / m : 0;
    if (j < r * (d + 1)) { const uint64_t c = j / (d + 1); return c * nb1 + (j - c * (d + 1)) / m; }
    const uint64_t j2 = j - r * (d + 1), c = j2 / d;
    return r * nb1 + c * nb0 + (j2 - c * d) / m;
}

hipError_t launch_prf(const LaunchEnv &env, uint32_t iter, const uint32_t *add, int n_add,
                      const uint32_t *minus, int n_minus, uint64_t n, uint32_t n_jobs,
                      uint64_t first, uint64_t count,
                      const uint64_t *in_dev, int in_limbs, uint64_t *out_dev)
{
    if (count == 0) return hipSuccess;
    if (env.codec && env.prf_backend != PRF_AUTO && env.prf_backend != PRF_TABLE) {
        LaunchEnv e2 = env;                  // the fused codec lives in the table kernels
        e2.prf_backend = PRF_TABLE;
        return launch_prf(e2, iter, add, n_add, minus, n_minus, n, n_jobs, first, count, in_dev, in_limbs, out_dev);
    }
#ifdef FLASHE_WITH_BITSLICE
    if (env.prf_backend == PRF_HYBRID && env.b > 64 && n_add == 1 && n_minus <= 1 && env.stream2) {
        // split [first, first + count): tail -> bit-sliced kernel on stream2, head -> table kernel here.
        // The bit-sliced share is a whole number of passes of all its waves (1 wave per SIMD).
        const uint64_t tile = n_minus ? 1024 : 2048;
        const uint64_t wave_slots = static_cast<uint64_t>(env.num_cus) * 4;
        uint64_t passes = (count * env.hybrid_bs_permille / 1000 + tile * wave_slots / 2) / (tile * wave_slots);
        uint64_t bs_count = passes * tile * wave_slots;
        if (bs_count == 0 && count * env.hybrid_bs_permille / 1000 >= tile) bs_count = (count * env.hybrid_bs_permille / 1000) / tile * tile;
        if (bs_count > count) bs_count = count / tile * tile;
        const uint64_t tt_count = count - bs_count;
        LaunchEnv e_tt = env, e_bs = env;
        e_tt.prf_backend = PRF_TABLE;
        e_bs.prf_backend = PRF_BITSLICE;
        e_bs.stream = env.stream2;
        hipError_t err = hipSuccess;
        if (bs_count) {
            if ((err = hipEventRecord(env.ev_fork, env.stream)) != hipSuccess) return err;
            if ((err = hipStreamWaitEvent(env.stream2, env.ev_fork, 0)) != hipSuccess) return err;
            const int il = in_limbs ? in_limbs : 0;
            err = launch_prf(e_bs, iter, add, n_add, minus, n_minus, n, n_jobs, first + tt_count, bs_count,
                             in_dev ? in_dev + tt_count * il : nullptr, in_limbs, out_dev + tt_count * 2);
            if (err != hipSuccess) return err;
        }
        if (tt_count) {
            err = launch_prf(e_tt, iter, add, n_add, minus, n_minus, n, n_jobs, first, tt_count, in_dev, in_limbs, out_dev);
            if (err != hipSuccess) return err;
        }
        if (bs_count) {
            if ((err = hipEventRecord(env.ev_join, env.stream2)) != hipSuccess) return err;
            if ((err = hipStreamWaitEvent(env.stream, env.ev_join, 0)) != hipSuccess) return err;
        }
        return hipSuccess;
    }
#endif
    IdxLists lists;
    for (int k = 0; k < kMaxIdx; k++) { lists.add[k] = k < n_add ? add[k] : 0; lists.minus[k] = k < n_minus ? minus[k] : 0; }
    PrfParams p{};
    p.in = in_dev; p.out = out_dev; p.te0 = env.te0_dev; p.n = n; p.iter = iter;
    p.first = first; p.count = count;
    p.in_limbs = in_limbs; p.n_add = n_add; p.n_minus = n_minus; p.n_jobs = n_jobs;
    p.b = env.b; p.m = 128 / env.b;
    if (env.codec) p.cq = *env.codec;
    masks_of(env.b, &p.mask_lo, &p.mask_hi);
    const bool bs_shape = env.b > 64 && n_add == 1 && n_minus <= 1;
#ifdef FLASHE_WITH_BITSLICE
    if (bs_shape && env.prf_backend == PRF_BITSLICE16) {
        // waves per SIMD the kernel variant is compiled for (register budget 256 / 168 / 128 VGPRs)
        static const int kWaves = [] { const char *e = FLASHE_TUNE_ENV("FLASHE_BS16_WAVES"); int w = e ? atoi(e) : 3; return w < 2 || w > 4 ? 3 : w; }();
        const uint64_t tile = n_minus ? 512 : 1024;
        uint64_t waves = (count + tile - 1) / tile;
        uint64_t blocks = (waves + 3) / 4;
        const uint64_t cap = static_cast<uint64_t>(env.num_cus) * kWaves;
        if (blocks > cap) blocks = cap;
        const dim3 g(static_cast<unsigned>(blocks)), t(kBsThreads);
#define BSP_LAUNCH(NS, W) hipLaunchKernelGGL((prf_wide_bsp_kernel<NS, W>), g, t, 0, env.stream, env.rkp_dev, p, lists.add[0], lists.minus[0])
        if (n_minus) { if (kWaves == 2) BSP_LAUNCH(2, 2); else if (kWaves == 3) BSP_LAUNCH(2, 3); else BSP_LAUNCH(2, 4); }
        else { if (kWaves == 2) BSP_LAUNCH(1, 2); else if (kWaves == 3) BSP_LAUNCH(1, 3); else BSP_LAUNCH(1, 4); }
#undef BSP_LAUNCH
    } else if (bs_shape && env.prf_backend == PRF_BITSLICE) {
        // 8 waves per CU (2 per SIMD at <= 256 VGPRs): two 256-thread blocks per CU
        const uint64_t tile = n_minus ? 1024 : 2048;
        uint64_t waves = (count + tile - 1) / tile;
        uint64_t blocks = (waves + 3) / 4;
        const uint64_t cap = static_cast<uint64_t>(env.num_cus) * 2;
        if (blocks > cap) blocks = cap;
        if (n_minus)
            hipLaunchKernelGGL(prf_wide_bs_kernel<2>, dim3(static_cast<unsigned>(blocks)), dim3(kBsThreads), 0, env.stream,
                               env.rkw_dev, p, lists.add[0], lists.minus[0]);
        else
            hipLaunchKernelGGL(prf_wide_bs_kernel<1>, dim3(static_cast<unsigned>(blocks)), dim3(kBsThreads), 0, env.stream,
                               env.rkw_dev, p, lists.add[0], 0u);
    } else
#endif
    if (bs_shape && !(env.codec && (!env.use_chain || ((first + count - 1) >> 32) != (first >> 32)))) {
        const PrfJob job{lists.add[0], lists.minus[0], first, count, in_dev, in_limbs, out_dev};
        return launch_prf_jobs(env, iter, n_minus == 1, 1, &job, n, n_jobs);
    } else if (env.b > 64) {
        const int grid = grid_for(env, count, kPrfThreads);
        hipLaunchKernelGGL(prf_wide_kernel, dim3(grid), dim3(kPrfThreads), 0, env.stream, env.rk, p, lists);
    } else if (n_add == 1 && n_minus <= 1) {
        const PrfJob job{lists.add[0], lists.minus[0], first, count, in_dev, in_limbs, out_dev};
        return launch_prf_jobs(env, iter, n_minus == 1, 1, &job, n, n_jobs);
    } else {
        p.blk_first = block_of(first, n, n_jobs, p.m);
        p.blk_count = block_of(first + count - 1, n, n_jobs, p.m) - p.blk_first + 1;
        const int grid = grid_for(env, p.blk_count, kSmallThreads);
        hipLaunchKernelGGL(prf_small_kernel, dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, p, lists);
    }
    return hipGetLastError();
}

hipError_t launch_prf_batch(const LaunchEnv &env, uint32_t iter, bool dbl, int n_vec, const uint32_t *idx,
                            const uint64_t *const *in_dev, int in_limbs, uint64_t *const *out_dev, uint64_t n, uint32_t n_jobs)
{
    return launch_prf_batch_range(env, iter, dbl, n_vec, idx, in_dev, in_limbs, out_dev, n, n_jobs, 0, n);
}

hipError_t launch_prf_batch_range(const LaunchEnv &env, uint32_t iter, bool dbl, int n_vec, const uint32_t *idx, const uint64_t *const *in_dev,
                                  int in_limbs, uint64_t *const *out_dev, uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count)
{
    if (count == 0 || n_vec == 0) return hipSuccess;
    if (n_vec > ((env.b > 64 || env.use_chain) ? kMaxUniform : kMaxBatch)) return hipErrorInvalidValue;
    if (env.use_chain) {
        // runs of consecutive clients share their streams (double mask); single mask: one stream per vector
        std::vector<uint32_t> sidx;
        std::vector<PrfChain> chains;
        std::vector<int> starts;
        for (int v = 0; v < n_vec;) {
            int w = v + 1;
            while (w < n_vec && (!dbl || idx[w] == idx[w - 1] + 1u)) w++;
            starts.push_back(static_cast<int>(sidx.size()));
            for (int k = v; k < w; k++) sidx.push_back(idx[k]);
            if (dbl) sidx.push_back(idx[w - 1] + 1u);
            chains.push_back(PrfChain{nullptr, w - v, !dbl, first, count, in_dev + v, in_limbs, out_dev + v});
            v = w;
        }
        for (size_t c = 0; c < chains.size(); c++) chains[c].idx = sidx.data() + starts[c];
        const hipError_t e = launch_prf_chains(env, iter, static_cast<int>(chains.size()), chains.data(), n, n_jobs);
        if (e != hipErrorNotSupported) return e;
    }
    if (env.b > 64 && n_vec > kMaxBatch && first == 0 && count == n) {
        // many equal vectors: compact table, one launch
        if (n_vec > kMaxUniform) return hipErrorInvalidValue;
        UniformJobTable tb{};
        for (int v = 0; v < n_vec; v++) { tb.add[v] = idx[v]; tb.minus[v] = idx[v] + 1u; tb.in[v] = in_dev[v]; tb.out[v] = out_dev[v]; }
        tb.count = n; tb.in_limbs = in_limbs;
        constexpr uint64_t kBigTile = static_cast<uint64_t>(kPrfThreads) * kBigEpl;
        const uint64_t cus = static_cast<uint64_t>(env.num_cus);
        // big tiles only if they come in whole rounds of the grid (else everything in 1024-element tiles)
        uint64_t big_per = n / kBigTile;
        while (big_per && (big_per * n_vec) % cus) big_per--;
        tb.big_per = big_per;
        tb.small_per = (n - big_per * kBigTile + kPrfThreads - 1) / kPrfThreads;
        const uint64_t tiles = (tb.big_per + tb.small_per) * n_vec;
        uint64_t lo, hi;
        masks_of(env.b, &lo, &hi);
        const int grid = static_cast<int>(tiles < cus ? tiles : cus);
        if (dbl)
            hipLaunchKernelGGL((prf_wide_batch_kernel<true, kPrfThreads, 1, UniformJobTable>), dim3(grid), dim3(kPrfThreads), 0, env.stream,
                               env.rk, tb, n_vec, n, iter, lo, hi, env.te0_dev);
        else
            hipLaunchKernelGGL((prf_wide_batch_kernel<false, kPrfThreads, 1, UniformJobTable>), dim3(grid), dim3(kPrfThreads), 0, env.stream,
                               env.rk, tb, n_vec, n, iter, lo, hi, env.te0_dev);
        return hipGetLastError();
    }
    std::vector<PrfJob> jobs(n_vec);
    for (int v = 0; v < n_vec; v++)
        jobs[v] = PrfJob{idx[v], idx[v] + 1u, first, count, in_dev[v], in_limbs, out_dev[v]};
    LaunchEnv e2 = env;
    e2.use_chain = 0;                        // the chained form was tried above
    return launch_prf_jobs(e2, iter, dbl, n_vec, jobs.data(), n, n_jobs);
}

hipError_t launch_prf_batch_sum(const LaunchEnv &env, uint32_t iter, int n_vec, const uint32_t *idx, const uint64_t *const *in_dev,
                                int in_limbs, uint64_t *const *out_dev, uint64_t *sum_out_dev, uint64_t n, uint32_t n_jobs, uint64_t first,
                                uint64_t count)
{
    if (count == 0 || n_vec == 0) return hipSuccess;
    if (!env.use_chain || env.b <= 64 || env.codec || n_vec > kMaxLinks || !sum_out_dev) return hipErrorNotSupported;
    if (env.prf_backend != PRF_AUTO && env.prf_backend != PRF_TABLE) return hipErrorNotSupported;
    for (int v = 1; v < n_vec; v++) if (idx[v] != idx[v - 1] + 1u) return hipErrorNotSupported;     // one run of consecutive clients
    // an uncut chain of a short vector leaves most waves idle (launch_prf_chains cuts such chains for parallelism, a summed chain
    // cannot be cut): below two whole tiles per wave the separate reduce is the better plan
    const uint64_t waves = static_cast<uint64_t>(env.num_cus) * (kPrfThreads / 64);
    if ((count + 255) / 256 < 2 * waves) return hipErrorNotSupported;
    std::vector<uint32_t> sidx(idx, idx + n_vec);
    sidx.push_back(idx[n_vec - 1] + 1u);
    PrfChain ch{sidx.data(), n_vec, false, first, count, in_dev, in_limbs, out_dev};
    ch.sum_out_dev = sum_out_dev;
    return launch_prf_chains(env, iter, 1, &ch, n, n_jobs);
}

// b <= 64 form of launch_prf_jobs
static hipError_t launch_prf_jobs_small(const LaunchEnv &env, uint32_t iter, bool dbl, int n_entries, const PrfJob *jobs, uint64_t n,
                                        uint32_t n_jobs)
{
    SmallJobTable tb{};
    SmallParams p{};
    p.n = n; p.n_jobs = n_jobs; p.iter = iter; p.b = env.b; p.m = 128 / env.b; p.te0 = env.te0_dev;
    if (env.codec) {
        if (n_entries != 1) return hipErrorInvalidValue;
        p.cq = *env.codec;
    }
    p.m_magic = static_cast<uint32_t>(((1ull << 32) + p.m - 1) / p.m);
    uint64_t hi;
    masks_of(env.b, &p.mask_lo, &hi);
    unsigned __int128 top = 0;
    for (int t = 0; t < p.m; t++) top |= static_cast<unsigned __int128>(1) << (env.b * t + env.b - 1);
    p.top_lo = static_cast<uint64_t>(top); p.top_hi = static_cast<uint64_t>(top >> 64);
    {
        const uint64_t mm = p.m, d = n / n_jobs, nb1 = (d + 1 + mm - 1) / mm, nb0 = d ? (d + mm - 1) / mm : 0;
        p.nb1_magic = nb1 > 1 && nb1 < (1ull << 32) ? static_cast<uint32_t>((1ull << 32) / nb1) : 0;
        p.nb0_magic = nb0 > 1 && nb0 < (1ull << 32) ? static_cast<uint32_t>((1ull << 32) / nb0) : 0;
    }
    uint64_t tiles = 0;
    int nv = 0;
    for (int e = 0; e < n_entries; e++) {
        if (jobs[e].count == 0) continue;
        if (jobs[e].n_in > 1 || (jobs[e].in_dev && jobs[e].in_limbs != 1)) return hipErrorInvalidValue;
        tb.add[nv] = jobs[e].add_idx; tb.minus[nv] = jobs[e].minus_idx;
        tb.first[nv] = jobs[e].first; tb.count[nv] = jobs[e].count;
        tb.in[nv] = jobs[e].in_dev; tb.out[nv] = jobs[e].out_dev;
        tb.blk_first[nv] = block_of(jobs[e].first, n, n_jobs, p.m);
        tb.blk_count[nv] = block_of(jobs[e].first + jobs[e].count - 1, n, n_jobs, p.m) - tb.blk_first[nv] + 1;
        tiles += (tb.blk_count[nv] + 63) / 64;
        tb.tile_end[nv++] = tiles;
    }
    if (nv == 0) return hipSuccess;
    const uint64_t cus = static_cast<uint64_t>(env.num_cus), wg_tiles = (tiles + kSmallThreads / 64 - 1) / (kSmallThreads / 64);
    const int grid = static_cast<int>(wg_tiles < cus ? wg_tiles : cus);
    if (dbl) hipLaunchKernelGGL(prf_small_jobs_kernel<true>, dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, tb, nv, p);
    else hipLaunchKernelGGL(prf_small_jobs_kernel<false>, dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, tb, nv, p);
    return hipGetLastError();
}

// Jobs -> chains: neighbours over the same element range are linked when the minus prefix of one is the add prefix of the
// next (double mask), or simply collected (single mask).  hipErrorNotSupported = use the job-table kernel.
// A lone double-mask job that cannot fill the chip is latency bound: the chained kernel would run its two streams one after the
// other (two dependent 14-round passes per lane, ~3.2 us each on a lone wave), the job-table kernel runs an element's add and minus
// block as ONE software-pipelined pair.  Such jobs take the job-table kernel in 256-thread workgroups (one element per lane, a
// LeNet-sized vector spreads over 241 CUs instead of 61).
static bool lone_small_double_job(const LaunchEnv &env, bool dbl, int n_entries, const PrfJob *jobs)
{
    static const bool on = !(FLASHE_TUNE_ENV("FLASHE_SMALL_LATENCY") && atoi(FLASHE_TUNE_ENV("FLASHE_SMALL_LATENCY")) == 0);
    return on && env.b > 64 && dbl && n_entries == 1 && !env.codec && jobs[0].n_in <= 1 && jobs[0].count &&
           jobs[0].count <= 256ull * static_cast<uint64_t>(env.num_cus) && ((jobs[0].first + jobs[0].count - 1) >> 32) == (jobs[0].first >> 32);
}

static hipError_t launch_jobs_as_chains(const LaunchEnv &env, uint32_t iter, bool dbl, int n_entries, const PrfJob *jobs, uint64_t n, uint32_t n_jobs)
{
    if (!env.use_chain) return hipErrorNotSupported;
    if (lone_small_double_job(env, dbl, n_entries, jobs)) return hipErrorNotSupported;
    struct Build { std::vector<uint32_t> idx; std::vector<const uint64_t *> in; std::vector<uint64_t *> out; uint64_t first, count; int in_limbs; };
    std::vector<Build> bs;
    for (int e = 0; e < n_entries; e++) {
        const PrfJob &j = jobs[e];
        if (j.n_in > 1) return hipErrorNotSupported;
        if (j.count == 0) continue;
        const int il = j.in_dev ? j.in_limbs : 0;
        Build *b = bs.empty() ? nullptr : &bs.back();
        const bool link = b && b->first == j.first && b->count == j.count && (il == 0 || b->in_limbs == 0 || b->in_limbs == il) &&
                          (!dbl || b->idx.back() == j.add_idx);
        if (!link) {
            bs.push_back(Build{{}, {}, {}, j.first, j.count, 0});
            b = &bs.back();
            b->idx.push_back(j.add_idx);
        } else if (!dbl) {
            b->idx.push_back(j.add_idx);
        }
        if (dbl) b->idx.push_back(j.minus_idx);
        if (il) b->in_limbs = il;
        b->in.push_back(j.in_dev); b->out.push_back(j.out_dev);
    }
    std::vector<PrfChain> chains;
    for (const Build &b : bs)
        chains.push_back(PrfChain{b.idx.data(), static_cast<int>(b.out.size()), !dbl, b.first, b.count, b.in.data(), b.in_limbs ? b.in_limbs : 1, b.out.data()});
    return launch_prf_chains(env, iter, static_cast<int>(chains.size()), chains.data(), n, n_jobs);
}

hipError_t launch_prf_jobs(const LaunchEnv &env, uint32_t iter, bool dbl, int n_entries, const PrfJob *jobs, uint64_t n, uint32_t n_jobs)
{
    {
        // any number of entries: neighbours that share a prefix are linked across the whole list
        const hipError_t e = launch_jobs_as_chains(env, iter, dbl, n_entries, jobs, n, n_jobs);
        if (e != hipErrorNotSupported) return e;
    }
    if (env.codec && env.b > 64) return hipErrorNotSupported;          // (only the chained kernel and the list kernels carry the codec)
    if (n_entries > kMaxBatch) {
        // the job-table kernels hold kMaxBatch entries per launch: equal shares
        const int launches = (n_entries + kMaxBatch - 1) / kMaxBatch, per = (n_entries + launches - 1) / launches;
        for (int e0 = 0; e0 < n_entries; e0 += per) {
            const hipError_t e = launch_prf_jobs(env, iter, dbl, std::min(per, n_entries - e0), jobs + e0, n, n_jobs);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    // the job-table kernels below know only the one-limb layout: a uint32 (compact) launch that the chained path did not take must
    // fail, not run with the wrong element size (today abi.hip's check_u32 keeps every such condition off)
    if (env.elem32) return hipErrorInvalidValue;
    if (env.b <= 64) return launch_prf_jobs_small(env, iter, dbl, n_entries, jobs, n, n_jobs);
    if (lone_small_double_job(env, dbl, n_entries, jobs)) {
        constexpr int kLatThreads = 256;
        JobTable tb{};
        tb.add[0] = jobs[0].add_idx; tb.minus[0] = jobs[0].minus_idx; tb.first[0] = jobs[0].first; tb.count[0] = jobs[0].count;
        tb.in[0] = jobs[0].in_dev; tb.out[0] = jobs[0].out_dev; tb.in_limbs[0] = static_cast<uint8_t>(jobs[0].in_limbs); tb.n_in[0] = 1;
        tb.big_end[0] = 0; tb.small_end[0] = (jobs[0].count + kLatThreads - 1) / kLatThreads;
        uint64_t lo, hi;
        masks_of(env.b, &lo, &hi);
        const uint64_t cus = static_cast<uint64_t>(env.num_cus);
        const int grid = static_cast<int>(tb.small_end[0] < cus ? tb.small_end[0] : cus);
        hipLaunchKernelGGL((prf_wide_batch_kernel<true, kLatThreads, 3, JobTable>), dim3(grid), dim3(kLatThreads), 0, env.stream, env.rk, tb, 1, n, iter, lo,
                           hi, env.te0_dev);
        return hipGetLastError();
    }
    JobTable tb{};
    int nv = 0;
    uint64_t big[kMaxBatch];
    uint64_t n_big = 0;
    constexpr uint64_t kBigTile = static_cast<uint64_t>(kPrfThreads) * kBigEpl;
    for (int e = 0; e < n_entries; e++) {
        if (jobs[e].count == 0) continue;
        tb.add[nv] = jobs[e].add_idx; tb.minus[nv] = jobs[e].minus_idx;
        tb.first[nv] = jobs[e].first; tb.count[nv] = jobs[e].count;
        tb.in[nv] = jobs[e].in_dev; tb.out[nv] = jobs[e].out_dev; tb.in_limbs[nv] = static_cast<uint8_t>(jobs[e].in_limbs);
        if (jobs[e].n_in > 255 || (jobs[e].n_in > 1 && (jobs[e].in_limbs != 2 || !jobs[e].in_dev))) return hipErrorInvalidValue;
        tb.n_in[nv] = static_cast<uint8_t>(jobs[e].n_in ? jobs[e].n_in : 1); tb.in_stride[nv] = jobs[e].in_stride;
        tb.sum_out[nv] = jobs[e].n_in > 1 ? jobs[e].sum_out_dev : nullptr;
        big[nv] = jobs[e].count / kBigTile;
        n_big += big[nv++];
    }
    if (nv == 0) return hipSuccess;
    // whole rounds of big tiles only: the big tiles beyond a multiple of the grid become small ones (taken from the
    // last jobs), so that no workgroup is left with a 4096-element tile more than the others
    const uint64_t cus = static_cast<uint64_t>(env.num_cus);
    uint64_t excess = n_big % cus;
    for (int v = nv - 1; v >= 0 && excess; v--) {
        const uint64_t take = big[v] < excess ? big[v] : excess;
        big[v] -= take; excess -= take;
    }
    uint64_t be = 0, se = 0;
    for (int v = 0; v < nv; v++) {
        be += big[v];
        se += (tb.count[v] - big[v] * kBigTile + kPrfThreads - 1) / kPrfThreads;
        tb.big_end[v] = be; tb.small_end[v] = se;
    }
    const uint64_t tiles = be + se;
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    // measured on MI355X: 1024-thread workgroups beat 768 / 512 (2.71 vs 2.87 / 2.99 ms for ten
    // 1e7-element vectors) and raising the wave priority (s_setprio) costs ~1 %
    const int grid = static_cast<int>(tiles < cus ? tiles : cus);
#define JOBS_LAUNCH(DBL, MULTI)                                                                                              \
    hipLaunchKernelGGL((prf_wide_batch_kernel<DBL, kPrfThreads, MULTI, JobTable>), dim3(grid), dim3(kPrfThreads), 0, env.stream, env.rk, tb, \
                       nv, n, iter, lo, hi, env.te0_dev)
    bool summed = false;
    for (int v = 0; v < nv; v++) summed |= tb.n_in[v] > 1;
    if (dbl) { if (summed) JOBS_LAUNCH(true, 2); else if (nv > 1) JOBS_LAUNCH(true, 1); else JOBS_LAUNCH(true, 0); }
    else { if (summed) JOBS_LAUNCH(false, 2); else if (nv > 1) JOBS_LAUNCH(false, 1); else JOBS_LAUNCH(false, 0); }
#undef JOBS_LAUNCH
    return hipGetLastError();
}

// Chained launch (b > 64): see prf_chain_kernel.  Long chains are cut where the per-launch tables end (the stream at a
// cut is computed by both pieces); short launches are cut further so that every wave of the chip gets an item.
static hipError_t launch_small_chains(const LaunchEnv &env, uint32_t iter, int n_chains, const PrfChain *chains, uint64_t n, uint32_t n_jobs);

#ifndef FLASHE_CHAIN_QUARTER_DEFAULT
#define FLASHE_CHAIN_QUARTER_DEFAULT 1   // quarter tiles for the shortest chained launches (0: half tiles as before round 6; A/B builds)
#endif
hipError_t launch_prf_chains(const LaunchEnv &env, uint32_t iter, int n_chains, const PrfChain *chains, uint64_t n, uint32_t n_jobs)
{
    if (env.b <= 64) return launch_small_chains(env, iter, n_chains, chains, n, n_jobs);
    struct Piece { const PrfChain *ch; int l0, l1; uint64_t tiles; };
    std::vector<Piece> pieces;
    uint64_t total_tiles = 0;
    bool summed = false;
    for (int i = 0; i < n_chains; i++) {
        const PrfChain &c = chains[i];
        if (c.count == 0 || c.n_out == 0) continue;
        if (((c.first + c.count - 1) >> 32) != (c.first >> 32)) return hipErrorNotSupported;     // the CTR shortcuts need one counter window
        // a chain that also writes the sum of its outputs is never cut (a piece would only know its own share of the sum)
        if (c.sum_out_dev) { if (c.n_out > kMaxLinks) return hipErrorNotSupported; summed = true; }
        const uint64_t tiles = (c.first + c.count - (c.first & ~255ull) + 255) / 256;
        pieces.push_back(Piece{&c, 0, c.n_out, tiles});
        total_tiles += tiles;
    }
    if (pieces.empty()) return hipSuccess;
    const uint64_t waves = static_cast<uint64_t>(env.num_cus) * (kPrfThreads / 64);
    bool all_half = total_tiles < 2 * waves;
    // experiment knobs (tests/perf/sweep_chain.py), read per launch only when FLASHE_CHAIN_TUNE is set
    static const bool tune = FLASHE_TUNE_ENV("FLASHE_CHAIN_TUNE") != nullptr;
    int force_parts = 0, force_grid = 0, probe = 0;
    if (tune) {
        if (const char *e = FLASHE_TUNE_ENV("FLASHE_CHAIN_HALF")) all_half = atoi(e) != 0;
        if (const char *e = FLASHE_TUNE_ENV("FLASHE_CHAIN_PARTS")) force_parts = atoi(e);
        if (const char *e = FLASHE_TUNE_ENV("FLASHE_CHAIN_GRID")) force_grid = atoi(e);
        if (const char *e = FLASHE_TUNE_ENV("FLASHE_CHAIN_PROBE")) probe = atoi(e) == 1 ? 0x100 : atoi(e) == 2 ? 0x200 : 0;
    }
    // cut: (1) table limits, (2) parallelism of short launches (never below 4 outputs per piece: a cut costs one stream)
    // (3) SINGLE chains have no shared stream, a cut is free: cut until the launch has two whole tiles per wave and runs in whole tiles
    //     (full two-step counter shortcut, two pairs per lane) -- 61,706 x 100 masks: 100.5 -> 93.5 us, 1e6 x 12: 175.8 -> 163.4 us,
    //     255,570 x 50 (config 5): 182.4 -> 179 us (tests/perf/sweep_chain.py single)
    bool only_single = true;
    for (const Piece &pc : pieces) only_single &= pc.ch->single && !pc.ch->sum_out_dev;
    int single_parts = 0;
    if (only_single && all_half && !force_parts && !(tune && FLASHE_TUNE_ENV("FLASHE_CHAIN_HALF"))) {
        single_parts = static_cast<int>(std::min<uint64_t>((2 * waves + total_tiles - 1) / total_tiles, kMaxChains / pieces.size()));
        if (single_parts < 1) single_parts = 1;
        uint64_t cut_tiles = 0;
        for (const Piece &pc : pieces) cut_tiles += pc.tiles * static_cast<uint64_t>(std::min(single_parts, pc.l1));
        all_half = 2 * cut_tiles < waves;       // a workgroup with fewer tiles than waves halves them by itself (n_full = 0 in the kernel)
    }
    uint64_t cuttable_tiles = 0, fixed_items = 0;
    for (const Piece &pc : pieces) {
        if (pc.l1 >= 8 && !pc.ch->sum_out_dev) cuttable_tiles += pc.tiles;
        else fixed_items += 2 * pc.tiles;
    }
    // Round 6: launches that cannot give every wave a half tile even uncut run in QUARTER tiles (64 counters, two streams per step:
    // the kernel's quarter branch) -- four items per tile, a chain cut into as few pieces as fill the chip ONCE (every wave one item
    // of the same length; pieces of at least eight outputs).  Config 3's mask precompute (242 tiles x 101 + 2 streams): 3 pieces of
    // the hundred-client chain instead of 10-12, 0.111 -> see DESIGN 4.4.  Not with a fused codec (its instantiation keeps half tiles).
    bool quarter = !env.codec && !summed && !single_parts && all_half && 4 * total_tiles <= waves && !force_parts && FLASHE_CHAIN_QUARTER_DEFAULT;
    if (tune) {                                                            // (tests/perf/quarter_sweep.py: 1 / 0 force the mode on / off)
        if (const char *e = FLASHE_TUNE_ENV("FLASHE_CHAIN_QUARTER")) quarter = atoi(e) != 0 && !env.codec && !summed && !single_parts && all_half;
    }
    std::vector<Piece> cut;
    for (const Piece &pc : pieces) {
        int parts = (pc.l1 + kMaxLinks - 1) / kMaxLinks;
        if (single_parts) {
            parts = std::max(parts, std::min(single_parts, pc.l1));
        } else if (quarter && !pc.ch->sum_out_dev && pc.l1 >= 2) {
            // as many pieces as fill the chip ONCE with quarter-tile items: every wave then has at most one item, and a shorter one
            // -- while the chip is not full the extra stream a cut costs runs beside the others, not after them (tests/perf/
            // quarter_sweep.py: 61,706 x 100: 3-8 pieces 95-96 us, 1 piece 177; 61,706 x 10: 3 pieces 25.3, 1 piece 30.8;
            // 5,000 x 3: 3 pieces 12.3, 1 piece 18.7; 250,000 x 10 / x 100: 1 piece)
            uint64_t cut4 = 0, fixed4 = 0;
            for (const Piece &o : pieces) (o.l1 >= 2 && !o.ch->sum_out_dev ? cut4 : fixed4) += 4 * o.tiles;
            const uint64_t want = fixed4 < waves ? std::max<uint64_t>((waves - fixed4) / cut4, 1) : 1;
            parts = std::max<int>(parts, static_cast<int>(std::min<uint64_t>(want, static_cast<uint64_t>(pc.l1))));
            parts = std::min(parts, std::max(1, kMaxChains / static_cast<int>(pieces.size())));
            parts = std::max(parts, (pc.l1 + kMaxLinks - 1) / kMaxLinks);
        } else if (all_half && 2 * total_tiles < waves && !pc.ch->sum_out_dev) {
            // one half-tile item per wave of the chip, counting what the chains too short to be cut contribute anyway
            // (mask precompute of config 3: a chain of 100 clients beside the one-output decrypt chain)
            // (1.15 items per wave: 61,706 x 100 runs 117 us in 4 pieces, 107 in 8, 103-105 in 10-12, 109-115 in 16; 250,000 x 100 is best
            // in 2 -- tests/perf/precompute_shape.py, sweep_chain.py)
            const uint64_t aim = waves + waves * 15 / 100;
            const uint64_t want = fixed_items < aim && cuttable_tiles ? (aim - fixed_items + cuttable_tiles) / (2 * cuttable_tiles) : 1;
            const int cap = std::max(1, pc.l1 / 4);
            parts = std::max<int>(parts, static_cast<int>(std::min<uint64_t>(want, static_cast<uint64_t>(cap))));
            parts = std::min(parts, std::max(1, kMaxChains / static_cast<int>(pieces.size())));
            parts = std::max(parts, (pc.l1 + kMaxLinks - 1) / kMaxLinks);
        }
        if (force_parts > 0 && !pc.ch->sum_out_dev) parts = std::max(std::min(force_parts, pc.l1), (pc.l1 + kMaxLinks - 1) / kMaxLinks);
        for (int k = 0; k < parts; k++) {
            const int a = static_cast<int>(static_cast<int64_t>(pc.l1) * k / parts), b = static_cast<int>(static_cast<int64_t>(pc.l1) * (k + 1) / parts);
            if (b > a) cut.push_back(Piece{pc.ch, a, b, pc.tiles});
        }
    }
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    size_t at = 0;
    while (at < cut.size()) {
        ChainTable tb{};
        int nc = 0, links = 0, streams = 0;
        uint64_t wend = 0, tiles = 0;
        while (at < cut.size() && nc < kMaxChains && links + (cut[at].l1 - cut[at].l0) <= kMaxLinks) {
            const Piece &pc = cut[at++];
            const PrfChain &c = *pc.ch;
            const int len = pc.l1 - pc.l0, ns = len + (c.single ? 0 : 1);
            tb.first[nc] = c.first; tb.count[nc] = c.count;
            tb.link0[nc] = static_cast<uint16_t>(links); tb.sbase[nc] = static_cast<uint16_t>(streams);
            tb.len[nc] = static_cast<uint8_t>(len);
            tb.flags[nc] = static_cast<uint8_t>((c.single ? 1 : 0) | (c.in_limbs == 2 ? 2 : 0));
            tb.sum_out[nc] = c.sum_out_dev;
            for (int s = 0; s < ns; s++) tb.idx[streams + s] = c.idx[pc.l0 + s];
            for (int l = 0; l < len; l++) {
                tb.in[links + l] = c.in_dev ? c.in_dev[pc.l0 + l] : nullptr;
                tb.out[links + l] = c.out_dev[pc.l0 + l];
            }
            wend += pc.tiles * static_cast<uint64_t>(ns);
            tb.wend[nc] = wend;
            tiles += pc.tiles;
            links += len; streams += ns; nc++;
        }
        const uint64_t items = quarter ? 4 * tiles : all_half ? 2 * tiles : tiles, cus = static_cast<uint64_t>(env.num_cus);
        int grid = static_cast<int>(items < cus ? items : cus);
        if (force_grid > 0) grid = force_grid;
        Codec cq{};
        if (env.codec) {
            if (cut.size() != 1 || cut[0].l1 - cut[0].l0 != 1) return hipErrorInvalidValue;      // one job, one output
            cq = *env.codec;
        }
        if (env.codec && summed) return hipErrorInvalidValue;
        if (env.codec)
            hipLaunchKernelGGL((prf_chain_kernel<kPrfThreads, false, true>), dim3(grid), dim3(kPrfThreads), 0, env.stream, env.rk, tb, nc,
                               (all_half ? 1 : 0) | (quarter ? 4 : 0) | probe, iter, lo, hi, env.te0_dev, cq);
        else if (summed)
            hipLaunchKernelGGL((prf_chain_kernel<kPrfThreads, true, false>), dim3(grid), dim3(kPrfThreads), 0, env.stream, env.rk, tb, nc,
                               (all_half ? 1 : 0) | (quarter ? 4 : 0) | probe, iter, lo, hi, env.te0_dev, cq);
        else
            hipLaunchKernelGGL((prf_chain_kernel<kPrfThreads, false, false>), dim3(grid), dim3(kPrfThreads), 0, env.stream, env.rk, tb, nc,
                               (all_half ? 1 : 0) | (quarter ? 4 : 0) | probe, iter, lo, hi, env.te0_dev, cq);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// the per-launch constants of the b <= 64 chained kernels (n < 2^32)
static SmallParams small_params_of(const LaunchEnv &env, uint32_t iter, uint64_t n, uint32_t n_jobs)
{
    SmallParams p{};
    p.n = n; p.n_jobs = n_jobs; p.iter = iter; p.b = env.b; p.m = 128 / env.b; p.te0 = env.te0_dev;
    { static const int v = FLASHE_TUNE_ENV("FLASHE_SMALL_DIRECT") ? atoi(FLASHE_TUNE_ENV("FLASHE_SMALL_DIRECT")) : 1; p.no_direct = v == 0 ? 1 : v == 2 ? 2 : 0; }
    { static const int v = FLASHE_TUNE_ENV("FLASHE_SMALL_FIXED") ? atoi(FLASHE_TUNE_ENV("FLASHE_SMALL_FIXED")) : 1; p.no_fixed_width = v == 0; }
    p.m_magic = static_cast<uint32_t>(((1ull << 32) + p.m - 1) / p.m);
    uint64_t hi;
    masks_of(env.b, &p.mask_lo, &hi);
    unsigned __int128 top = 0;
    for (int t = 0; t < p.m; t++) top |= static_cast<unsigned __int128>(1) << (env.b * t + env.b - 1);
    p.top_lo = static_cast<uint64_t>(top); p.top_hi = static_cast<uint64_t>(top >> 64);
    const uint64_t mm = p.m, d = n / n_jobs, nb1 = (d + 1 + mm - 1) / mm, nb0 = d ? (d + mm - 1) / mm : 0;
    p.nb1_magic = nb1 > 1 && nb1 < (1ull << 32) ? static_cast<uint32_t>((1ull << 32) / nb1) : 0;
    p.nb0_magic = nb0 > 1 && nb0 < (1ull << 32) ? static_cast<uint32_t>((1ull << 32) / nb0) : 0;
    return p;
}

// The reduce fused with the decrypt of its result for b <= 64 (small_reduce_decrypt_kernel): elements [first, first + count) of an
// n-element vector; the C operand pointers, agg_out (may be null) and out address element `first`.  hipErrorNotSupported for what
// the kernel does not carry (the caller then reduces and decrypts in two launches).
hipError_t launch_small_reduce_decrypt(const LaunchEnv &env, uint32_t iter, uint32_t add_idx, bool has_minus, uint32_t minus_idx, uint64_t n,
                                       uint32_t n_jobs, uint64_t first, uint64_t count, int C, const uint64_t *const *ops, uint64_t *agg_out_dev,
                                       uint64_t *out_dev, int out_elem_bytes)
{
    if (env.b > 64 || env.codec || n >= (1ull << 32) || n == 0 || n_jobs == 0 || C < 1 || C > kMaxOps) return hipErrorNotSupported;
    if ((env.elem32 || out_elem_bytes == 4) && env.b > 32) return hipErrorInvalidValue;
    if (out_elem_bytes != 8 && !(out_elem_bytes == 4 && env.elem32)) return hipErrorInvalidValue;
    if (env.prf_backend != PRF_AUTO && env.prf_backend != PRF_TABLE) return hipErrorNotSupported;
    if (count == 0) return hipSuccess;
    SmallParams p = small_params_of(env, iter, n, n_jobs);
    p.swp_prio = 1;                           // (small_reduce_decrypt_kernel, one-limb layout: b = 64 -11.6 %, b = 40 +-0; the split kernel has one block per lane)
    const uint64_t bf = block_of(first, n, n_jobs, p.m), bc = block_of(first + count - 1, n, n_jobs, p.m) - bf + 1;
    PtrTable t;
    for (int c = 0; c < kMaxOps; c++) t.p[c] = c < C ? ops[c] : nullptr;
    const uint64_t tiles = (bc + 63) / 64, groups = (tiles + kSmallThreads / 64 - 1) / (kSmallThreads / 64), cus = static_cast<uint64_t>(env.num_cus);
    const int grid = static_cast<int>(groups < cus ? groups : cus);
    // operands per step (two elements per lane per step): FEWER streams at once stream faster at full occupancy -- ten 1e7-element
    // operands without the AES rounds: 0.180 ms in steps of 2, 0.188-0.193 in steps of 4-5, 0.225 in steps of 8; four elements per lane
    // per step changed nothing at m = 6 and wastes slots at m = 2 (tests/perf/small_reduce_decrypt.py; FLASHE_SMALL_REDUCE_CB = 1, 2, 4, 8)
    int cb = C < 2 ? 1 : 2;
    { static const int force = FLASHE_TUNE_ENV("FLASHE_SMALL_REDUCE_CB") ? atoi(FLASHE_TUNE_ENV("FLASHE_SMALL_REDUCE_CB")) : 0; if (force >= 1) cb = force; }
    static const int probe = FLASHE_TUNE_ENV("FLASHE_SMALL_REDUCE_PROBE") ? 0x100 : 0;
#define SRD_LAUNCH(CB)                                                                                                                      \
    hipLaunchKernelGGL((small_reduce_decrypt_kernel<CB, 2>), dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, p, add_idx, minus_idx, \
                       (has_minus ? 1 : 0) | probe, first, count, bf, bc, C, t, agg_out_dev, out_dev)
    static const bool split_off = FLASHE_TUNE_ENV("FLASHE_SMALL_REDUCE_SPLIT") && atoi(FLASHE_TUNE_ENV("FLASHE_SMALL_REDUCE_SPLIT")) == 0;
    if (env.b <= 32 && !split_off) {
        // 32-block tiles, the two streams in the two halves of the wave (see small_reduce_decrypt_split_kernel)
        const uint64_t tiles32 = (bc + 31) / 32, groups32 = (tiles32 + kSmallThreads / 64 - 1) / (kSmallThreads / 64);
        const int grid32 = static_cast<int>(groups32 < cus ? groups32 : cus);
        // 0x200: every operand and both outputs 16-byte aligned -- the compact layout's regular tiles take 16-byte accesses
        uintptr_t low_bits = reinterpret_cast<uintptr_t>(out_dev) | reinterpret_cast<uintptr_t>(agg_out_dev);
        for (int c = 0; c < C; c++) low_bits |= reinterpret_cast<uintptr_t>(ops[c]);
        static const bool quad_off = FLASHE_TUNE_ENV("FLASHE_SMALL_REDUCE_QUAD") && atoi(FLASHE_TUNE_ENV("FLASHE_SMALL_REDUCE_QUAD")) == 0;
        int quad = (low_bits & 15u) == 0 && !quad_off ? 0x200 : 0;
        if (quad && env.elem32) {
            // elements per lane and access of the regular tiles (32 m elements): the width among 4, 3, 2 that wastes the fewest lanes of
            // the wave's accesses (3 needs 32 m divisible by 3; ties go to the wider access)
            const uint32_t tile_elems = 32u * static_cast<uint32_t>(p.m);
            uint32_t best = 4;
            double best_util = 0.0;
            for (uint32_t e : {4u, 3u, 2u}) {
                if (tile_elems % e) continue;
                const uint32_t lanes = tile_elems / e;
                const double util = static_cast<double>(lanes) / (64.0 * ((lanes + 63u) / 64u));
                if (util > best_util + 1e-9) { best_util = util; best = e; }
            }
            static const int force_e = FLASHE_TUNE_ENV("FLASHE_SMALL_REDUCE_EPL") ? atoi(FLASHE_TUNE_ENV("FLASHE_SMALL_REDUCE_EPL")) : 0;
            if (force_e >= 2 && force_e <= 4 && tile_elems % static_cast<uint32_t>(force_e) == 0) best = static_cast<uint32_t>(force_e);
            quad |= static_cast<int>(best & 3u) << 10;            // (4 travels as 0)
        }
#define SRDS_LAUNCH(CB, IT, OT)                                                                                                               \
    hipLaunchKernelGGL((small_reduce_decrypt_split_kernel<CB, IT, OT>), dim3(grid32), dim3(kSmallThreads), 0, env.stream, env.rk, p, add_idx,  \
                       minus_idx, (has_minus ? 1 : 0) | probe | quad, first, count, bf, bc, C, t, agg_out_dev, out_dev)
#define SRDS_PICK(CB)                                                                                                                          \
    do {                                                                                                                                        \
        if (!env.elem32) SRDS_LAUNCH(CB, uint64_t, uint64_t);                                                                                   \
        else if (out_elem_bytes == 4) SRDS_LAUNCH(CB, uint32_t, uint32_t);                                                                      \
        else SRDS_LAUNCH(CB, uint32_t, uint64_t);                                                                                               \
    } while (0)
        switch (cb) {
        case 1: SRDS_PICK(1); break;
        case 2: case 3: SRDS_PICK(2); break;
        case 4: case 5: case 6: case 7: SRDS_PICK(4); break;
        default: SRDS_PICK(8); break;
        }
#undef SRDS_PICK
#undef SRDS_LAUNCH
        return hipGetLastError();
    }
    if (env.elem32) return hipErrorNotSupported;              // (FLASHE_SMALL_REDUCE_SPLIT=0 has no compact form)
    switch (cb) {
    case 1: SRD_LAUNCH(1); break;
    case 2: case 3: SRD_LAUNCH(2); break;
    case 4: case 5: case 6: case 7: SRD_LAUNCH(4); break;
    default: SRD_LAUNCH(8); break;
    }
#undef SRD_LAUNCH
    return hipGetLastError();
}

// b <= 64 form of launch_prf_chains.  hipErrorNotSupported (-> job-table kernel) for what this kernel does not carry: a fused
// codec, vectors of 2^32 elements or more.
static hipError_t launch_small_chains(const LaunchEnv &env, uint32_t iter, int n_chains, const PrfChain *chains, uint64_t n, uint32_t n_jobs)
{
    if (env.codec || n >= (1ull << 32) || n == 0) return hipErrorNotSupported;
    SmallParams p = small_params_of(env, iter, n, n_jobs);
    // the fused sum of a chain's outputs: the compact layout's compile-time-width kernels carry it (never cut, double mask, paired launch)
    bool summed = false;
    for (int i = 0; i < n_chains; i++) summed |= chains[i].sum_out_dev != nullptr;
    if (summed && !(env.elem32 && !p.no_fixed_width && fixed32_width(env.b) && n_chains == 1 && !chains[0].single &&
                    chains[0].n_out <= kMaxLinks))
        return hipErrorNotSupported;
    if (env.elem32) {
        if (env.b > 32) return hipErrorInvalidValue;
        p.no_direct = 1;                      // the 16-byte direct accesses of m <= 4 assume 8-byte elements: b = 32 walks its rows like b < 32
    }
    // rising wave priority inside the rounds of a block pair (device_common.h), where it was measured to pay: the direct output paths
    // (one-limb layout at m <= 4: -5 ... -7 %; compact layout at the compiled-in widths: 23 / 24 / 32 -8 %, 16 -2 %) -- not the long
    // chains of int_bits 20 compact (ten encrypts + their sum: +5 % in round 6's loop, +1.5 % in round 5's; the decrypt of ONE vector,
    // a chain of two streams: -9 %, 0.0616 -> 0.0562 ms) and not the staged walk of the one-limb layout at m >= 5 (+15 %)
    int longest = 0;
    for (int i = 0; i < n_chains; i++) longest = std::max(longest, chains[i].n_out);
#ifndef FLASHE_PRIO_B20
#define FLASHE_PRIO_B20 0      // (A/B builds: the rising wave priority at int_bits 20 in the compact layout whatever the chain length)
#endif
    p.swp_prio = env.elem32 ? (!p.no_fixed_width && fixed32_width(env.b) && (FLASHE_PRIO_B20 || env.b != 20 || longest <= 1)) : (p.m <= 4 && !p.no_direct);
    { static const int v = FLASHE_TUNE_ENV("FLASHE_SMALL_PRIO") ? atoi(FLASHE_TUNE_ENV("FLASHE_SMALL_PRIO")) : -1; if (v >= 0) p.swp_prio = v; }
    struct Piece { const PrfChain *ch; int l0, l1; uint64_t blk_first, blk_count; };
    std::vector<Piece> pieces;
    uint64_t total_blocks = 0;
    for (int i = 0; i < n_chains; i++) {
        const PrfChain &c = chains[i];
        if (c.count == 0 || c.n_out == 0) continue;
        if (c.in_dev && c.in_limbs != 1) return hipErrorInvalidValue;
        const uint64_t bf = block_of(c.first, n, n_jobs, p.m), bc = block_of(c.first + c.count - 1, n, n_jobs, p.m) - bf + 1;
        pieces.push_back(Piece{&c, 0, c.n_out, bf, bc});
        total_blocks += bc;
    }
    if (pieces.empty()) return hipSuccess;
    const uint64_t waves = static_cast<uint64_t>(env.num_cus) * (kSmallThreads / 64);
    // two blocks per lane (software pipelined) once every wave has work for several such tiles; short launches run one block per
    // lane and cut long chains so that more waves take part (a cut costs one stream)
    const bool pair = total_blocks >= 2 * 128 * waves;
    if (summed && !pair) return hipErrorNotSupported;           // (short launches cut their chains for parallelism: the separate reduce is the plan there)
    const uint64_t tile = pair ? 128 : 64;
    uint64_t total_tiles = 0;
    for (const Piece &pc : pieces) total_tiles += (pc.blk_count + tile - 1) / tile;
    std::vector<Piece> cut;
    for (const Piece &pc : pieces) {
        int parts = (pc.l1 + kMaxLinks - 1) / kMaxLinks;
        if (total_tiles < waves) {
            // an underfilled chip is latency bound: parallelism first, down to one output (= one stream pair) per piece
            const uint64_t want = waves / total_tiles;
            parts = std::max<int>(parts, static_cast<int>(std::min<uint64_t>(want, static_cast<uint64_t>(pc.l1))));
            parts = std::min(parts, std::max(1, kMaxChains / static_cast<int>(pieces.size())));
            parts = std::max(parts, (pc.l1 + kMaxLinks - 1) / kMaxLinks);
        }
        for (int k = 0; k < parts; k++) {
            const int a = static_cast<int>(static_cast<int64_t>(pc.l1) * k / parts), b = static_cast<int>(static_cast<int64_t>(pc.l1) * (k + 1) / parts);
            if (b > a) cut.push_back(Piece{pc.ch, a, b, pc.blk_first, pc.blk_count});
        }
    }
    size_t at = 0;
    while (at < cut.size()) {
        SmallChainTable tb{};
        int nc = 0, links = 0, streams = 0;
        uint64_t wend = 0, tiles = 0;
        while (at < cut.size() && nc < kMaxChains && links + (cut[at].l1 - cut[at].l0) <= kMaxLinks) {
            const Piece &pc = cut[at++];
            const PrfChain &c = *pc.ch;
            const int len = pc.l1 - pc.l0, ns = len + (c.single ? 0 : 1);
            tb.first[nc] = c.first; tb.count[nc] = c.count; tb.blk_first[nc] = pc.blk_first; tb.blk_count[nc] = pc.blk_count;
            tb.link0[nc] = static_cast<uint16_t>(links); tb.sbase[nc] = static_cast<uint16_t>(streams);
            tb.len[nc] = static_cast<uint8_t>(len); tb.flags[nc] = static_cast<uint8_t>(c.single ? 1 : 0);
            for (int q = 0; q < ns; q++) tb.idx[streams + q] = c.idx[pc.l0 + q];
            for (int l = 0; l < len; l++) {
                tb.in[links + l] = c.in_dev ? c.in_dev[pc.l0 + l] : nullptr;
                tb.out[links + l] = c.out_dev[pc.l0 + l];
            }
            tb.sum_out[nc] = c.sum_out_dev;
            const uint64_t t = (pc.blk_count + tile - 1) / tile;
            wend += t * static_cast<uint64_t>(ns);
            tb.wend[nc] = wend;
            tiles += t;
            links += len; streams += ns; nc++;
        }
        const uint64_t cus = static_cast<uint64_t>(env.num_cus);
        const int grid = static_cast<int>(tiles < cus ? tiles : cus);
        if (env.elem32) {
            // FLASHE_FIXED32_WIDTHS with their slot positions compiled in
            const int fixed = (pair && !p.no_fixed_width) ? env.b : 0;
            switch (fixed) {
#define FLASHE_FIXED32(B) case B: hipLaunchKernelGGL((prf_small_chain_kernel<true, uint32_t, B>), dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, tb, nc, p); break;
                FLASHE_FIXED32_WIDTHS(FLASHE_FIXED32)
#undef FLASHE_FIXED32
            default:
                if (pair) hipLaunchKernelGGL((prf_small_chain_kernel<true, uint32_t>), dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, tb, nc, p);
                else hipLaunchKernelGGL((prf_small_chain_kernel<false, uint32_t>), dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, tb, nc, p);
            }
        } else if (pair && env.b == 64 && !p.no_fixed_width && !p.no_direct)
            hipLaunchKernelGGL((prf_small_chain_kernel<true, uint64_t, 64>), dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, tb, nc, p);
        else if (pair) hipLaunchKernelGGL((prf_small_chain_kernel<true>), dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, tb, nc, p);
        else hipLaunchKernelGGL((prf_small_chain_kernel<false>), dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, tb, nc, p);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_aes_blocks(const LaunchEnv &env, uint32_t nblk, const uint32_t *in_words_dev, uint32_t *out_words_dev)
{
    hipLaunchKernelGGL(aes_blocks_kernel, dim3(1), dim3(kPrfThreads), 0, env.stream, env.rk, env.te0_dev, nblk,
                       in_words_dev, out_words_dev);
    return hipGetLastError();
}

// ---- the arbiter's reduce fused with the decrypt of its result, operands ANYWHERE in HBM (b > 64, one add / at most one minus prefix) ----
// out[k] = (sum_c ops[c][k] + term(add, first + k) - [DBL] term(minus, first + k)) mod 2^b for k < count.  The job-table form of this
// fusion (prf_wide_batch_kernel<..., 2>) wants the operands equally spaced in one allocation; separately allocated ciphertexts used to
// fall back to two launches (reduce, then decrypt: 0.61 ms for ten 1e7-element operands).  A plain grid-stride loop -- one element per
// lane per trip, the first kSumRegs operands requested before the element's AES pass and added after it, further operands summed up
// front -- measured the same as the tiled form on equally spaced operands (0.346 against 0.352 ms, tests/perf/ab_reduce_2wg.py), so
// this is the one-launch form for operand lists of any layout.
template <bool DBL>
__global__ __launch_bounds__(kPrfThreads) void reduce_decrypt_ptrs_kernel(const RoundKeys rk, const PtrTable ops, int C, uint64_t first, uint64_t count,
                                                                          uint32_t iter0, uint32_t add_idx, uint32_t minus_idx, uint64_t mask_lo,
                                                                          uint64_t mask_hi, const uint32_t *__restrict__ te0, uint64_t *agg_out,
                                                                          uint64_t *out)
{
    const uint32_t iter = iter0 + te0[kIterShiftWord];
    __shared__ uint32_t tab[kTabWords];
    fill_tables(tab, te0);
    const LaneRegs lr = lane_regs(tab);
    const u128 mask = (static_cast<u128>(mask_hi) << 64) | mask_lo;
    const uint32_t ctr_hi = static_cast<uint32_t>(first >> 32);                 // (host-checked: the range stays inside one 2^32 window)
    const CtrPrefix pre_a = scalar_prefix(ctr_prefix(rk, lr, iter, add_idx, ctr_hi));
    CtrPrefix pre_b{};
    if (DBL) pre_b = scalar_prefix(ctr_prefix(rk, lr, iter, minus_idx, ctr_hi));
    const uint64_t *const *tab_ops = ops.p;
    for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * kPrfThreads + threadIdx.x; k < count; k += static_cast<uint64_t>(gridDim.x) * kPrfThreads) {
        u64x2 held[kSumRegs];
#pragma unroll
        for (int c = 0; c < kSumRegs; c++) held[c] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(tab_ops[c < C ? c : 0] + 2 * k));
        u128 acc = 0;
        for (int c = kSumRegs; c < C; c++) acc += ld128_nt(tab_ops[c] + 2 * k);
        const CtrVar x = ctr_var(rk, lr, static_cast<uint32_t>(first + k));
        uint32_t s[DBL ? 2 : 1][4];
        ctr_round1(pre_a, x, s[0]);
        if (DBL) ctr_round1(pre_b, x, s[DBL ? 1 : 0]);
        aes256_rounds<DBL ? 2 : 1, 2>(rk, lr, s, true);
#pragma unroll
        for (int c = 0; c < kSumRegs; c++)
            if (c < C) acc += (static_cast<u128>(held[c][1]) << 64) | held[c][0];
        acc &= mask;
        if (agg_out) st128_nt(agg_out + 2 * k, acc);
        acc += words_to_u128(s[0]);
        if (DBL) acc -= words_to_u128(s[DBL ? 1 : 0]);
        st128_nt(out + 2 * k, acc & mask);
    }
}

hipError_t launch_reduce_decrypt_ptrs(const LaunchEnv &env, uint32_t iter, uint32_t add_idx, bool has_minus, uint32_t minus_idx, uint64_t first,
                                      uint64_t count, int C, const uint64_t *const *ops, uint64_t *agg_out_dev, uint64_t *out_dev)
{
    if (count == 0) return hipSuccess;
    if (env.b <= 64 || C < 1 || C > kMaxOps || ((first + count - 1) >> 32) != (first >> 32)) return hipErrorNotSupported;
    if (env.prf_backend != PRF_AUTO && env.prf_backend != PRF_TABLE) return hipErrorNotSupported;
    PtrTable t;
    for (int c = 0; c < kMaxOps; c++) t.p[c] = c < C ? ops[c] : nullptr;
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    const int grid = grid_for(env, count, kPrfThreads);
    if (has_minus)
        hipLaunchKernelGGL(reduce_decrypt_ptrs_kernel<true>, dim3(grid), dim3(kPrfThreads), 0, env.stream, env.rk, t, C, first, count, iter, add_idx,
                           minus_idx, lo, hi, env.te0_dev, agg_out_dev, out_dev);
    else
        hipLaunchKernelGGL(reduce_decrypt_ptrs_kernel<false>, dim3(grid), dim3(kPrfThreads), 0, env.stream, env.rk, t, C, first, count, iter, add_idx,
                           0u, lo, hi, env.te0_dev, agg_out_dev, out_dev);
    return hipGetLastError();
}

#ifdef FLASHE_TUNING
// ---- experiment (tuning build only): does the reduce fused with the decrypt gain from TWO workgroups per CU? ----
// The same simplified loop -- one element per lane per trip, all C operands requested before the two-block AES pass, result stored
// non-temporally -- in two shapes: THREADS = 1024 with the full 128-KiB tables (one workgroup per CU, the product's shape) and
// THREADS = 512 with HALF-size tables (64 KiB: two workgroups fit a CU).  The half-size form is a TIMING PROBE: tables 2 / 3 alias
// tables 0 / 1, so its results are wrong -- what a real two-table AES would add on top (one rotate per aliased lookup) is NOT in it,
// i.e. it measures an upper bound of what the split could buy.
template <int THREADS, bool HALF>
__global__ __launch_bounds__(THREADS) void reduce_decrypt_probe_kernel(const RoundKeys rk, const PtrTable ops, int C, uint64_t n, uint32_t iter0,
                                                                       uint32_t add_idx, uint32_t minus_idx, uint64_t mask_lo, uint64_t mask_hi,
                                                                       const uint32_t *__restrict__ te0, uint64_t *out)
{
    const uint32_t iter = iter0 + te0[kIterShiftWord];
    __shared__ uint32_t tab[HALF ? kTabWords / 2 : kTabWords];
    for (int e = threadIdx.x; e < (HALF ? 512 : 1024); e += THREADS) {
        const int t = e >> 8, x = e & 255;
        const uint32_t v = rotr32(te0[x], 8 * t);
        uint4 vv = make_uint4(v, v, v, v);
        uint4 *dst = reinterpret_cast<uint4 *>(tab + ((t >> 1) * 16384 + x * 64 + (t & 1) * 32));
#pragma unroll
        for (int q = 0; q < 8; q++) dst[q] = vv;
    }
    __syncthreads();
    LaneRegs lr = lane_regs(tab);
    if (HALF) lr.b = lr.a;
    const u128 mask = (static_cast<u128>(mask_hi) << 64) | mask_lo;
    const CtrPrefix pre_a = scalar_prefix(ctr_prefix(rk, lr, iter, add_idx, 0u)), pre_b = scalar_prefix(ctr_prefix(rk, lr, iter, minus_idx, 0u));
    const uint64_t *const *tab_ops = ops.p;
    for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * THREADS + threadIdx.x; k < n; k += static_cast<uint64_t>(gridDim.x) * THREADS) {
        u64x2 held[kSumRegs];
#pragma unroll
        for (int c = 0; c < kSumRegs; c++) held[c] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(tab_ops[c < C ? c : 0] + 2 * k));
        const CtrVar x = ctr_var(rk, lr, static_cast<uint32_t>(k));
        uint32_t s[2][4];
        ctr_round1(pre_a, x, s[0]);
        ctr_round1(pre_b, x, s[1]);
        aes256_rounds<2, 2>(rk, lr, s, true);
        u128 acc = 0;
#pragma unroll
        for (int c = 0; c < kSumRegs; c++)
            if (c < C) acc += (static_cast<u128>(held[c][1]) << 64) | held[c][0];
        acc += words_to_u128(s[0]);
        acc -= words_to_u128(s[1]);
        st128_nt(out + 2 * k, acc & mask);
    }
}

hipError_t launch_reduce_decrypt_probe(const LaunchEnv &env, int variant, uint32_t iter, uint32_t add_idx, uint32_t minus_idx, int C,
                                       const uint64_t *const *ops, uint64_t n, uint64_t *out_dev)
{
    if (C < 1 || C > kSumRegs || env.b <= 64) return hipErrorInvalidValue;
    PtrTable t;
    for (int c = 0; c < kMaxOps; c++) t.p[c] = c < C ? ops[c] : nullptr;
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    if (variant == 0)
        hipLaunchKernelGGL((reduce_decrypt_probe_kernel<1024, false>), dim3(env.num_cus), dim3(1024), 0, env.stream, env.rk, t, C, n, iter, add_idx,
                           minus_idx, lo, hi, env.te0_dev, out_dev);
    else if (variant == 1)
        hipLaunchKernelGGL((reduce_decrypt_probe_kernel<512, true>), dim3(2 * env.num_cus), dim3(512), 0, env.stream, env.rk, t, C, n, iter, add_idx,
                           minus_idx, lo, hi, env.te0_dev, out_dev);
    else if (variant == 2)          // the half-size tables with ONE 1024-thread workgroup per CU: separates "two workgroups" from "smaller tables"
        hipLaunchKernelGGL((reduce_decrypt_probe_kernel<1024, true>), dim3(env.num_cus), dim3(1024), 0, env.stream, env.rk, t, C, n, iter, add_idx,
                           minus_idx, lo, hi, env.te0_dev, out_dev);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
#endif

}  // namespace flashe
