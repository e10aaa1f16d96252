// Device-side core shared by the translation units of libflashe_hip.so (kernels.hip: the PRF kernels and their launchers;
// stream.hip: the HBM-bound kernels -- combine, reduce, packed reduce, bit-packing, the sparse passes; codec.hip: quantise / batch):
// the AES-256 T-table core with the CTR shortcuts, 128-bit helpers, the fused codec's device functions and a few launch helpers.
// Everything here is inline (device functions __forceinline__, host helpers static inline): no symbol is defined twice.
#pragma once
#include "kernels.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

// Timing probes and tuning knobs exist only in the -DFLASHE_TUNING build (make tuning -> libflashe_hip_tuning.so, what tests/perf/* load
// through FLASHE_LIB_NAME): the product library reads none of these variables and carries none of the early-exit probe branches, so
// no environment setting can make it return success without having computed what was asked.
#ifdef FLASHE_TUNING
#define FLASHE_TUNE_ENV(name) getenv(name)
#else
#define FLASHE_TUNE_ENV(name) (static_cast<const char *>(nullptr))
#endif

namespace flashe {

typedef unsigned __int128 u128;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------
// AES-256 core
// ------------------------------------------------------------------------------------------
constexpr int kTabWords = 32768;                 // 128 KiB: 4 tables x 256 entries x 32 copies
// Word 1024 of the device table buffer (right behind Te0..Te3) is the ITER SHIFT of the ctx: every PRF kernel adds it to the iter
// it was launched with.  It is 0 except while a captured graph is replayed for a later round (flashe_graph_launch_shifted): kernel
// arguments are frozen into a graph, the shift is read from memory at run time, so a replay never reuses a mask stream.
constexpr int kIterShiftWord = 1024;
constexpr int kPrfThreads = 1024;

// v_perm_b32 selectors: D = {0x00, lanereg.byte2, state.byte_k, lanereg.byte0}
// (selector bytes 0-3 pick from the second operand, 4-7 from the first, 0x0c = zero)
#define SEL_B0 0x0c020400u
#define SEL_B1 0x0c020500u
#define SEL_B2 0x0c020600u
#define SEL_B3 0x0c020700u

// LDS byte offset of a __shared__ object (address-space cast, folded at compile time).
__device__ __forceinline__ uint32_t lds_offset(uint32_t *shared_obj)
{
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_u32 *)shared_obj));
}

__device__ __forceinline__ uint32_t rotr32(uint32_t v, int r) { return (v >> r) | (v << ((32 - r) & 31)); }

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// Replicated T-tables.  Table t, entry x, copy k (k = lane & 31) at byte
//   (t >> 1) * 65536 + x * 256 + (t & 1) * 128 + k * 4
__device__ __forceinline__ void fill_tables(uint32_t *tab, const uint32_t *te0)
{
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) {
        const int t = e >> 8, x = e & 255;
        const uint32_t v = rotr32(te0[x], 8 * t);
        uint4 vv = make_uint4(v, v, v, v);
        uint4 *dst = reinterpret_cast<uint4 *>(tab + ((t >> 1) * 16384 + x * 64 + (t & 1) * 32));
#pragma unroll
        for (int q = 0; q < 8; q++) dst[q] = vv;
    }
    __syncthreads();
}

// a: tables 0/1 (low 64 KiB half), b: tables 2/3 (high half); base: the table object in LDS, so
// that every lookup is visibly a load from the array the prologue filled.
struct LaneRegs { uint32_t a, b; const lds_u8 *base; };

__device__ __forceinline__ LaneRegs lane_regs(uint32_t *tab)
{
    const uint32_t lane4 = (threadIdx.x & 31u) * 4u;
    return LaneRegs{lane4, lane4 | 0x00010000u, (const lds_u8 *)(lds_u32 *)tab};
}

#ifndef FLASHE_CTR2
#define FLASHE_CTR2 1   // wave-uniform part of rounds 1-2 through the scalar cache
#endif
// Wave priority inside the software-pipelined rounds (round 5): a wave RAISES its s_setprio as it advances through the rounds of a
// block pair (0 until round P1, then 1, 2, 3 from rounds P1 / P2 / P3), so the SIMD's arbiter serves the wave closest to the end of its
// dependent chain first instead of round-robin -- the waves of a SIMD drift apart and one wave's loads, stores and loop head fall under
// the others' lookups.  prf_chain_kernel<1024, SUM>: 1.440 -> 1.305 ms (-9.3 %), the decrypt of one vector 0.272 -> 0.250, b = 64
// -5 %; the VALU-bound compact kernels: 0 ... -3 % (tests/perf/ab_chain_libs.py, ab_compact_libs.py; 0 = off, 1 = falling: -4 %;
// thresholds 5/8/11, 3/6/9, 3/6/10, 6/9/12 within 1 % of 4/7/10, two levels only -6.7 %).  The span kernel keeps its own, FALLING,
// schedule: its waves meet at a counter every span (rising measured +5.6 % there).  `prio` is a per-launch choice (the last argument of
// aes256_rounds): on in the int_bits > 64 kernels; int_bits <= 64 (tests/perf/ab_compact_libs.py, 0 against 2): one-limb layout b = 64
// -5 % (its reduce + decrypt -11.6 %), 40: -7 %, 32: -5 %, but the staged walk of b <= 25: +15 %; compact layout 23 / 24 / 32: -8 %,
// 16: -2 %, 20: +1.5 % -- the launcher's table follows these (small_swp_prio).
#ifndef FLASHE_SWP_PRIO
#define FLASHE_SWP_PRIO 2
#endif
#ifndef FLASHE_SWP_P1
#define FLASHE_SWP_P1 4
#define FLASHE_SWP_P2 7
#define FLASHE_SWP_P3 10
#endif
#ifndef FLASHE_SWP_PRIO_HALF
#define FLASHE_SWP_PRIO_HALF 1  // prf_chain_kernel's half tiles (short launches, ragged ends): ten 1e6-element vectors -5 %, config 3's shape +-1 %
#endif
#ifndef FLASHE_DEEP_PRIO
#define FLASHE_DEEP_PRIO 1      // the rising schedule in small_reduce_decrypt_split_kernel (one block per lane): compact b = 20 -6.6 %, 23 -10 %, 16 -2 %
#endif
#ifndef FLASHE_SMALL_NP_PRIO
#define FLASHE_SMALL_NP_PRIO 1  // ... in the two-streams-per-step form of prf_small_chain_kernel and in prf_small_kernel (config 3 at b = 23: -2 %)
#endif
#ifndef FLASHE_EDGE_PRIO
#define FLASHE_EDGE_PRIO 1      // ... in sparse_edge_prf_kernel (the run edges of the sparse double mask): -3.5 %
#endif
#ifndef FLASHE_SWP_POST
#define FLASHE_SWP_POST -1      // >= 0: the priority a wave returns to after the rounds (measured: no difference)
#endif
#ifndef FLASHE_SWP
#define FLASHE_SWP 1   // two-block calls run software pipelined (measured 4.6 % faster than the compiler's own order)
#endif
#ifndef FLASHE_ADDR_BITOP
#define FLASHE_ADDR_BITOP 0   // 1 = lookup addresses by shift + v_bitop3 instead of v_perm_b32 (round 4 A/B builds: measured SLOWER in the kernels, see below)
#endif
template <int OFF>
__device__ __forceinline__ uint32_t lut(const lds_u8 *base, uint32_t w, uint32_t lanereg, uint32_t sel)
{
#if FLASHE_ADDR_BITOP
    // address = (byte k of w) << 8 | lane register.  v_perm_b32 builds it in one instruction, but every three-source VALU op except
    // v_bitop3_b32 issues at ~4.3 cycles per wave here and the rounds are bound by VALU issue as much as by the LDS; a shift that brings
    // byte k to bits 8..15 (a two-source op, 1.9 cycles; none for k = 1) and one v_bitop3 ((x & 0xff00) | lane register, 2.5 cycles)
    // cost 12 x 1.9 + 16 x 2.5 = 63 cycles per block-round instead of 69 (tools/ubench_lds.hip: 23.2 -> 25.1 lookups per clock per CU
    // with one block per lane, 24.1 -> 25.2 with two).  In the KERNELS the two builds alternated in one process say the opposite:
    // ten chained 1e7-element encrypts 1.575 ms against 1.429 with v_perm, b = 64 0.882 / 0.799, config 5 0.655 / 0.62 -- twelve more
    // instructions per block-round and a two-deep dependent chain in front of every lookup cost more than the cycles they save once the
    // software-pipelined rounds compete for issue slots.  Kept as a build option, off.  `sel` is a compile-time constant at every call site.
    const uint32_t x = sel == SEL_B1 ? w : sel == SEL_B0 ? w << 8 : sel == SEL_B2 ? w >> 8 : w >> 16;
    const uint32_t addr = __builtin_amdgcn_bitop3_b32(x, 0xff00u, lanereg, 0xea);
#else
    const uint32_t addr = __builtin_amdgcn_perm(w, lanereg, sel);
#endif
    return *reinterpret_cast<const lds_u32 *>(base + addr + OFF);
}

#define T0(w, sel) lut<0>(lr.base, w, lr.a, sel)
#define T1(w, sel) lut<128>(lr.base, w, lr.a, sel)
#define T2(w, sel) lut<0>(lr.base, w, lr.b, sel)
#define T3(w, sel) lut<128>(lr.base, w, lr.b, sel)

__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b)
{
    return (a & mask) | (b & ~mask);   // v_bfi_b32
}

// Rounds FIRST..13 and the final round on NB independent blocks (state = 4 big-endian column words,
// already carrying everything up to round FIRST - 1).
template <int FIRST>
__device__ __forceinline__ void aes256_rounds2_swp(const RoundKeys &rk, const LaneRegs lr, uint32_t (&s)[2][4], bool prio);

template <int NB, int FIRST>
__device__ __forceinline__ void aes256_rounds(const RoundKeys &rk, const LaneRegs lr, uint32_t (&s)[NB][4], bool prio = false)
{
    if constexpr (NB == 2 && FLASHE_SWP) {
        aes256_rounds2_swp<FIRST>(rk, lr, s, prio);
        return;
    }
#pragma unroll
    for (int r = FIRST; r < 14; r++) {
#pragma unroll
        for (int q = 0; q < NB; q++) {
            uint32_t t[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t x = xor3(T0(s[q][j], SEL_B3), T1(s[q][(j + 1) & 3], SEL_B2),
                                        T2(s[q][(j + 2) & 3], SEL_B1));
                t[j] = xor3(x, T3(s[q][(j + 3) & 3], SEL_B0), rk.w[4 * r + j]);
            }
            s[q][0] = t[0]; s[q][1] = t[1]; s[q][2] = t[2]; s[q][3] = t[3];
        }
    }
    // final round: SubBytes + ShiftRows + AddRoundKey.  S[x] sits in byte 3 of T2[x], byte 2 of
    // T3[x], byte 1 of T0[x] and byte 0 of T1[x].
#pragma unroll
    for (int q = 0; q < NB; q++) {
        uint32_t t[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t v = bfi(0xff000000u, T2(s[q][j], SEL_B3),
                               bfi(0x00ff0000u, T3(s[q][(j + 1) & 3], SEL_B2),
                               bfi(0x0000ff00u, T0(s[q][(j + 2) & 3], SEL_B1),
                                                T1(s[q][(j + 3) & 3], SEL_B0))));
            t[j] = v ^ rk.w[56 + j];
        }
        s[q][0] = t[0]; s[q][1] = t[1]; s[q][2] = t[2]; s[q][3] = t[3];
    }
}

// Two blocks, software pipelined: the 16 lookups of one block are always in flight while the other block is
// finished (column XORs) and its next 16 lookups are issued -- the LDS queue of the wave never drains.
struct Lk16 { uint32_t v[16]; };

__device__ __forceinline__ Lk16 issue_main(const LaneRegs lr, const uint32_t (&s)[4])
{
    Lk16 k;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        k.v[4 * j + 0] = T0(s[j], SEL_B3);
        k.v[4 * j + 1] = T1(s[(j + 1) & 3], SEL_B2);
        k.v[4 * j + 2] = T2(s[(j + 2) & 3], SEL_B1);
        k.v[4 * j + 3] = T3(s[(j + 3) & 3], SEL_B0);
    }
    return k;
}
__device__ __forceinline__ Lk16 issue_final(const LaneRegs lr, const uint32_t (&s)[4])
{
    Lk16 k;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        k.v[4 * j + 0] = T2(s[j], SEL_B3);
        k.v[4 * j + 1] = T3(s[(j + 1) & 3], SEL_B2);
        k.v[4 * j + 2] = T0(s[(j + 2) & 3], SEL_B1);
        k.v[4 * j + 3] = T1(s[(j + 3) & 3], SEL_B0);
    }
    return k;
}
__device__ __forceinline__ void finish_main(const RoundKeys &rk, int r, const Lk16 &k, uint32_t (&s)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++)
        s[j] = xor3(xor3(k.v[4 * j], k.v[4 * j + 1], k.v[4 * j + 2]), k.v[4 * j + 3], rk.w[4 * r + j]);
}
__device__ __forceinline__ void finish_final(const RoundKeys &rk, const Lk16 &k, uint32_t (&s)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++)
        s[j] = bfi(0xff000000u, k.v[4 * j], bfi(0x00ff0000u, k.v[4 * j + 1], bfi(0x0000ff00u, k.v[4 * j + 2], k.v[4 * j + 3]))) ^
               rk.w[56 + j];
}

template <int FIRST>
__device__ __forceinline__ void aes256_rounds2_swp(const RoundKeys &rk, const LaneRegs lr, uint32_t (&s)[2][4], bool prio)
{
    Lk16 ka = issue_main(lr, s[0]);
    __builtin_amdgcn_sched_barrier(0);
    Lk16 kb = issue_main(lr, s[1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = FIRST; r < 14; r++) {
        // (prio: wave-uniform, chosen per launch -- a scalar branch around each s_setprio)
#if FLASHE_SWP_PRIO == 1            // (falling with the progress through the rounds: the A/B alternative)
        if (prio && r == FIRST) __builtin_amdgcn_s_setprio(3);
        else if (prio && r == 5) __builtin_amdgcn_s_setprio(2);
        else if (prio && r == 8) __builtin_amdgcn_s_setprio(1);
        else if (prio && r == 11) __builtin_amdgcn_s_setprio(0);
#elif FLASHE_SWP_PRIO == 2
        if (prio && r == FIRST) __builtin_amdgcn_s_setprio(0);
        if (prio && r == FLASHE_SWP_P1) __builtin_amdgcn_s_setprio(1);
        if (prio && r == FLASHE_SWP_P2) __builtin_amdgcn_s_setprio(2);
        if (prio && r == FLASHE_SWP_P3) __builtin_amdgcn_s_setprio(3);
#endif
        finish_main(rk, r, ka, s[0]);
        ka = r < 13 ? issue_main(lr, s[0]) : issue_final(lr, s[0]);
        __builtin_amdgcn_sched_barrier(0);
        finish_main(rk, r, kb, s[1]);
        kb = r < 13 ? issue_main(lr, s[1]) : issue_final(lr, s[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    finish_final(rk, ka, s[0]);
    finish_final(rk, kb, s[1]);
#if FLASHE_SWP_PRIO == 2 && FLASHE_SWP_POST >= 0
    if (prio) __builtin_amdgcn_s_setprio(FLASHE_SWP_POST);
#endif
}

// One block per lane where there is no second one to pipeline against: all sixteen lookups of a round are issued as their state
// words become final, nothing of the next round moves up (left to itself the compiler interleaves the rounds with two or three
// lookups in flight per wave, and the LDS pipe idles: the span reduce with the PRF inside ran at 46 % of the lookup rate that way).
template <int FIRST>
__device__ __forceinline__ void aes256_rounds1_deep(const RoundKeys &rk, const LaneRegs lr, uint32_t (&s)[4], bool prio = false)
{
    Lk16 k = issue_main(lr, s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = FIRST; r < 14; r++) {
        if (prio && r == FIRST) __builtin_amdgcn_s_setprio(0);           // (the rising schedule of aes256_rounds2_swp)
        if (prio && r == FLASHE_SWP_P1) __builtin_amdgcn_s_setprio(1);
        if (prio && r == FLASHE_SWP_P2) __builtin_amdgcn_s_setprio(2);
        if (prio && r == FLASHE_SWP_P3) __builtin_amdgcn_s_setprio(3);
        finish_main(rk, r, k, s);
        k = r < 13 ? issue_main(lr, s) : issue_final(lr, s);
        __builtin_amdgcn_sched_barrier(0);
    }
    finish_final(rk, k, s);
}

// NB independent blocks; s holds the plaintext blocks.
template <int NB>
__device__ __forceinline__ void aes256_encrypt(const RoundKeys &rk, const LaneRegs lr, uint32_t (&s)[NB][4], bool prio = false)
{
#pragma unroll
    for (int q = 0; q < NB; q++) {
        s[q][0] ^= rk.w[0]; s[q][1] ^= rk.w[1]; s[q][2] ^= rk.w[2]; s[q][3] ^= rk.w[3];
    }
    aes256_rounds<NB, 1>(rk, lr, s, prio);
}

// PRF blocks are iter | idx | counter: within one launch only the low counter word varies between
// lanes (when the launch does not straddle a 2^32 counter boundary), so 12 of the 16 first-round
// lookups see lane-invariant bytes.  CtrPrefix folds them, the round-0 and the round-1 keys into four
// words per prefix, computed once per kernel; round 1 then costs 4 lookups instead of 16.
struct CtrPrefix { uint32_t u[4]; };

__device__ __forceinline__ CtrPrefix ctr_prefix(const RoundKeys &rk, const LaneRegs lr, uint32_t iter, uint32_t idx, uint32_t ctr_hi)
{
    const uint32_t s0 = iter ^ rk.w[0], s1 = idx ^ rk.w[1], s2 = ctr_hi ^ rk.w[2];
    CtrPrefix c;
    c.u[0] = xor3(T0(s0, SEL_B3), T1(s1, SEL_B2), T2(s2, SEL_B1)) ^ rk.w[4];   // + T3[b0(s3)]
    c.u[1] = xor3(T0(s1, SEL_B3), T1(s2, SEL_B2), T3(s0, SEL_B0)) ^ rk.w[5];   // + T2[b1(s3)]
    c.u[2] = xor3(T0(s2, SEL_B3), T2(s0, SEL_B1), T3(s1, SEL_B0)) ^ rk.w[6];   // + T1[b2(s3)]
    c.u[3] = xor3(T1(s0, SEL_B2), T2(s1, SEL_B1), T3(s2, SEL_B0)) ^ rk.w[7];   // + T0[b3(s3)]
    return c;
}

// The lane-dependent quarter of round 1: the four lookups on the low counter word.  They do not
// depend on the prefix, so the add and the minus block of one element share them.
struct CtrVar { uint32_t v[4]; };

__device__ __forceinline__ CtrVar ctr_var(const RoundKeys &rk, const LaneRegs lr, uint32_t ctr_lo)
{
    const uint32_t s3 = ctr_lo ^ rk.w[3];
    return CtrVar{{T3(s3, SEL_B0), T2(s3, SEL_B1), T1(s3, SEL_B2), T0(s3, SEL_B3)}};
}

// State after round 1 for the block with prefix c.
__device__ __forceinline__ void ctr_round1(const CtrPrefix &c, const CtrVar &x, uint32_t (&s)[4])
{
    s[0] = c.u[0] ^ x.v[0]; s[1] = c.u[1] ^ x.v[1]; s[2] = c.u[2] ^ x.v[2]; s[3] = c.u[3] ^ x.v[3];
}

// Second step of the CTR shortcut.  With 64 consecutive counters per wave (and a wave base that is a
// multiple of 64) bytes 1..3 of the low counter word are wave-uniform, so after round 1 only state
// column 0 differs between lanes, and in round 2 every output column has ONE lane-dependent lookup (a
// byte of column 0) and three wave-uniform ones.  The uniform part goes through the scalar cache (te4 =
// Te0|Te1|Te2|Te3 in global memory, SGPR indices) instead of the LDS: per block 4.5 LDS lookups for
// rounds 1-2 instead of 18.
struct CtrUniform { uint32_t u[4]; };

__device__ __forceinline__ CtrPrefix scalar_prefix(const CtrPrefix &c)
{
    return CtrPrefix{{static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(c.u[0])), static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(c.u[1])),
                      static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(c.u[2])), static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(c.u[3]))}};
}

// x3 = (wave's counter base) ^ rk.w[3]; only its bytes 1..3 are used.
__device__ __forceinline__ CtrUniform ctr_uniform(const RoundKeys &rk, const uint32_t *__restrict__ te4, const CtrPrefix &c, uint32_t x3)
{
    const uint32_t S1 = c.u[1] ^ te4[512 + ((x3 >> 8) & 0xffu)];
    const uint32_t S2 = c.u[2] ^ te4[256 + ((x3 >> 16) & 0xffu)];
    const uint32_t S3 = c.u[3] ^ te4[x3 >> 24];
    CtrUniform r;
    r.u[0] = te4[256 + ((S1 >> 16) & 0xffu)] ^ te4[512 + ((S2 >> 8) & 0xffu)] ^ te4[768 + (S3 & 0xffu)] ^ rk.w[8];
    r.u[1] = te4[S1 >> 24] ^ te4[256 + ((S2 >> 16) & 0xffu)] ^ te4[512 + ((S3 >> 8) & 0xffu)] ^ rk.w[9];
    r.u[2] = te4[S2 >> 24] ^ te4[256 + ((S3 >> 16) & 0xffu)] ^ te4[768 + (S1 & 0xffu)] ^ rk.w[10];
    r.u[3] = te4[S3 >> 24] ^ te4[512 + ((S1 >> 8) & 0xffu)] ^ te4[768 + (S2 & 0xffu)] ^ rk.w[11];
    return r;
}

// State after round 2: v0 = T3[b0(ctr_lo ^ rk.w[3])] (shared by the blocks of one element), c0 = c.u[0].
__device__ __forceinline__ void ctr_round2(const LaneRegs lr, uint32_t c0, uint32_t v0, const CtrUniform &U, uint32_t (&s)[4])
{
    const uint32_t s0 = c0 ^ v0;
    s[0] = U.u[0] ^ T0(s0, SEL_B3);
    s[1] = U.u[1] ^ T3(s0, SEL_B0);
    s[2] = U.u[2] ^ T2(s0, SEL_B1);
    s[3] = U.u[3] ^ T1(s0, SEL_B2);
}

__device__ __forceinline__ u128 words_to_u128(const uint32_t (&s)[4])
{
    const uint64_t hi = (static_cast<uint64_t>(s[0]) << 32) | s[1];
    const uint64_t lo = (static_cast<uint64_t>(s[2]) << 32) | s[3];
    return (static_cast<u128>(hi) << 64) | lo;
}

__device__ __forceinline__ void set_block(uint32_t (&s)[4], uint32_t iter, uint32_t idx, uint64_t ctr)
{
    s[0] = iter; s[1] = idx; s[2] = static_cast<uint32_t>(ctr >> 32); s[3] = static_cast<uint32_t>(ctr);
}

__device__ __forceinline__ u128 ld128(const uint64_t *p)
{
    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(p);
    return (static_cast<u128>(v.y) << 64) | v.x;
}
__device__ __forceinline__ void st128(uint64_t *p, u128 v)
{
    *reinterpret_cast<ulonglong2 *>(p) = make_ulonglong2(static_cast<uint64_t>(v), static_cast<uint64_t>(v >> 64));
}

// A device pointer that was itself LOADED from the LDS (per-client tables of list / value pointers) is a generic pointer to the
// compiler: it is dereferenced with flat_load / flat_store, which count on the LDS counter as well as on the memory counter -- every
// later wait for an LDS lookup then waits for that memory access too.  These casts say what such pointers are.
#define FLASHE_GLOBAL(T, p) (reinterpret_cast<__attribute__((address_space(1))) T *>(reinterpret_cast<uintptr_t>(p)))
__device__ __forceinline__ uint32_t ld32_g(const uint32_t *p) { return *FLASHE_GLOBAL(const uint32_t, p); }
__device__ __forceinline__ uint64_t ld64_g(const uint64_t *p) { return *FLASHE_GLOBAL(const uint64_t, p); }
__device__ __forceinline__ uint64_t ld64_nt_g(const uint64_t *p) { return __builtin_nontemporal_load(FLASHE_GLOBAL(const uint64_t, p)); }
__device__ __forceinline__ u128 ld128_g(const uint64_t *p)
{
    const u64x2 v = *FLASHE_GLOBAL(const u64x2, p);
    return (static_cast<u128>(v[1]) << 64) | v[0];
}
__device__ __forceinline__ u128 ld128_nt_g(const uint64_t *p)
{
    const u64x2 v = __builtin_nontemporal_load(FLASHE_GLOBAL(const u64x2, p));
    return (static_cast<u128>(v[1]) << 64) | v[0];
}
__device__ __forceinline__ void st128_nt_g(uint64_t *p, u128 v)
{
    u64x2 r;
    r[0] = static_cast<uint64_t>(v); r[1] = static_cast<uint64_t>(v >> 64);
    __builtin_nontemporal_store(r, FLASHE_GLOBAL(u64x2, p));
}

// streaming (read-once / write-once) forms: keep such traffic out of the caches
__device__ __forceinline__ u128 ld128_nt(const uint64_t *p)
{
    const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(p));
    return (static_cast<u128>(v[1]) << 64) | v[0];
}
__device__ __forceinline__ void st128_nt(uint64_t *p, u128 v)
{
    u64x2 r;
    r[0] = static_cast<uint64_t>(v); r[1] = static_cast<uint64_t>(v >> 64);
    __builtin_nontemporal_store(r, reinterpret_cast<u64x2 *>(p));
}


// correctly rounded (nearest-even) u128 -> double, as Python's int -> float
__device__ __forceinline__ double u128_to_double(u128 v)
{
    const uint64_t hi = static_cast<uint64_t>(v >> 64), lo = static_cast<uint64_t>(v);
    if (hi == 0) return static_cast<double>(lo);
    const int sh = 64 - __clzll(static_cast<long long>(hi));       // 1..64 bits above the low limb
    uint64_t m = static_cast<uint64_t>(v >> sh);                    // top 64 significant bits
    const u128 dropped = v & ((static_cast<u128>(1) << sh) - 1);
    if (dropped) m |= 1;                                            // sticky: 64 > 53 + 2 keeps rounding exact
    return ldexp(static_cast<double>(m), sh);
}

// _static_quantize_padding_asymmetric (jzf_quantize.py:55-67) on one value, in the array's own float type; the arithmetic must
// round exactly like numpy's: no contraction into FMAs
template <typename T>
__device__ __forceinline__ uint64_t quantize_one(T v, T alpha, T scale, T den, double u)
{
#pragma clang fp contract(off)
    v = v < -alpha ? -alpha : (v > alpha ? alpha : v);
    v = v + alpha;
    v = v * scale;
    v = v / den;
    return static_cast<uint64_t>(static_cast<int64_t>(floor(static_cast<double>(v) + u)));
}

// the layer of a flattened model that holds flat element `key`: the last table entry with start <= key
__device__ __forceinline__ const CodecLayer *codec_layer_of(const Codec &c, uint64_t key)
{
    int lo = 0, hi = c.n_layers - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (c.layers[mid].start <= key) lo = mid; else hi = mid - 1;
    }
    return c.layers + lo;
}

__device__ __forceinline__ uint64_t codec_quantize(const Codec &c, uint64_t k)
{
    if (c.layers != nullptr) {
        const uint64_t key = c.k0 + k;
        const CodecLayer *L = codec_layer_of(c, key);
        const uint64_t r = key - L->start;
        // (the layer's pointer comes out of a table in memory: said to be a global one, its loads do not count on the LDS counter
        // the AES lookups of the same kernel wait on)
        return L->x_is_f64 ? quantize_one<double>(*FLASHE_GLOBAL(const double, static_cast<const double *>(L->x) + r), L->p0, L->p1, L->p2, c.u[k])
                           : quantize_one<float>(*FLASHE_GLOBAL(const float, static_cast<const float *>(L->x) + r), static_cast<float>(L->p0),
                                                 static_cast<float>(L->p1), static_cast<float>(L->p2), c.u[k]);
    }
    return c.x_is_f64 ? quantize_one<double>(static_cast<const double *>(c.x)[k], c.alpha, c.scale, c.den, c.u[k])
                      : quantize_one<float>(static_cast<const float *>(c.x)[k], static_cast<float>(c.alpha), static_cast<float>(c.scale),
                                            static_cast<float>(c.den), c.u[k]);
}

// _static_unquantize_padding_asymmetric (jzf_quantize.py:102-107); k = the element's index in the launch (selects the layer)
__device__ __forceinline__ double codec_unquantize(const Codec &c, uint64_t k, u128 v)
{
#pragma clang fp contract(off)
    if (c.layers != nullptr) {
        const CodecLayer *L = codec_layer_of(c, c.k0 + k);
        return u128_to_double(v) * L->p1 / L->p2 - L->p0;
    }
    return u128_to_double(v) * c.two_a / c.uden - c.ac;
}


static inline void masks_of(int b, uint64_t *lo, uint64_t *hi)
{
    if (b >= 128) { *lo = ~0ull; *hi = ~0ull; }
    else if (b > 64) { *lo = ~0ull; *hi = (1ull << (b - 64)) - 1; }
    else if (b == 64) { *lo = ~0ull; *hi = 0; }
    else { *lo = (1ull << b) - 1; *hi = 0; }
}

static inline int grid_for(const LaunchEnv &env, uint64_t work_items, int threads)
{
    uint64_t blocks = (work_items + threads - 1) / threads;
    if (blocks < 1) blocks = 1;
    if (blocks > static_cast<uint64_t>(env.num_cus)) blocks = env.num_cus;
    return static_cast<int>(blocks);
}


constexpr int kStreamThreads = 256;

static inline int stream_grid(const LaunchEnv &env, uint64_t items)
{
    uint64_t blocks = (items + kStreamThreads - 1) / kStreamThreads;
    const uint64_t cap = static_cast<uint64_t>(env.num_cus) * 8;     // 8 x 256-thread blocks per CU
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return static_cast<int>(blocks);
}

// the compact layout of the *_u32_dev entry points <-> the ABI's one-limb vectors
}  // namespace flashe
