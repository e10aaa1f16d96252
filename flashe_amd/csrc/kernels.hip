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
#ifndef FLASHE_SWP
#define FLASHE_SWP 1   // two-block calls run software pipelined (measured 4.6 % faster than the compiler's own order)
#endif
template <int OFF>
__device__ __forceinline__ uint32_t lut(const lds_u8 *base, uint32_t w, uint32_t lanereg, uint32_t sel)
{
    const uint32_t addr = __builtin_amdgcn_perm(w, lanereg, sel);
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
__device__ __forceinline__ void aes256_rounds2_swp(const RoundKeys &rk, const LaneRegs lr, uint32_t (&s)[2][4]);

template <int NB, int FIRST>
__device__ __forceinline__ void aes256_rounds(const RoundKeys &rk, const LaneRegs lr, uint32_t (&s)[NB][4])
{
    if constexpr (NB == 2 && FLASHE_SWP) {
        aes256_rounds2_swp<FIRST>(rk, lr, s);
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
__device__ __forceinline__ void aes256_rounds2_swp(const RoundKeys &rk, const LaneRegs lr, uint32_t (&s)[2][4])
{
    Lk16 ka = issue_main(lr, s[0]);
    __builtin_amdgcn_sched_barrier(0);
    Lk16 kb = issue_main(lr, s[1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = FIRST; r < 14; r++) {
        finish_main(rk, r, ka, s[0]);
        ka = r < 13 ? issue_main(lr, s[0]) : issue_final(lr, s[0]);
        __builtin_amdgcn_sched_barrier(0);
        finish_main(rk, r, kb, s[1]);
        kb = r < 13 ? issue_main(lr, s[1]) : issue_final(lr, s[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    finish_final(rk, ka, s[0]);
    finish_final(rk, kb, s[1]);
}

// NB independent blocks; s holds the plaintext blocks.
template <int NB>
__device__ __forceinline__ void aes256_encrypt(const RoundKeys &rk, const LaneRegs lr, uint32_t (&s)[NB][4])
{
#pragma unroll
    for (int q = 0; q < NB; q++) {
        s[q][0] ^= rk.w[0]; s[q][1] ^= rk.w[1]; s[q][2] ^= rk.w[2]; s[q][3] ^= rk.w[3];
    }
    aes256_rounds<NB, 1>(rk, lr, s);
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
        return L->x_is_f64 ? quantize_one<double>(static_cast<const double *>(L->x)[r], L->p0, L->p1, L->p2, c.u[k])
                           : quantize_one<float>(static_cast<const float *>(L->x)[r], static_cast<float>(L->p0), static_cast<float>(L->p1),
                                                 static_cast<float>(L->p2), c.u[k]);
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
                aes256_rounds<DBL ? 2 : 1, 3>(rk, lr, s);
            } else if (ctr_fast) {
                const CtrVar x = ctr_var(rk, lr, static_cast<uint32_t>(j));
                ctr_round1(pre_a, x, s[0]);
                if (DBL) ctr_round1(pre_b, x, s[DBL ? 1 : 0]);
                aes256_rounds<DBL ? 2 : 1, 2>(rk, lr, s);
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
    all_half &= 1;
    const uint32_t wave = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t Ng = uniform64(d_cend[n_chains - 1]);
    const uint64_t n_full = all_half ? 0 : Ng - Ng % WAVES;
    const uint64_t n_items = n_full + 2 * (Ng - n_full);
    int cur = 0;
    uint64_t cbeg = 0;                                                     // local index of chain cur's first tile
    for (uint64_t q = wave; q < n_items; q += WAVES) {
        const bool whole = q < n_full;
        const uint64_t L = whole ? q : n_full + ((q - n_full) >> 1);
        const uint32_t half = whole ? 0u : static_cast<uint32_t>((q - n_full) & 1u);
        while (L >= uniform64(d_cend[cur])) cbeg = uniform64(d_cend[cur++]);
        const uint64_t first = tb.first[cur], end = first + tb.count[cur];
        const uint64_t tj = (first & ~255ull) + 256u * (uniform64(d_tlo[cur]) + (L - cbeg)) + 128u * half;   // first counter of the item
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
                        aes256_rounds<2, 3>(rk, lr, s);
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
        } else if (tj < end && tj + 128u > first) {
            // ---- 128 elements: one pair per lane; the four counter-dependent lookups of round 1 are shared by all streams ----
            const uint64_t j0 = tj + lane, j1 = j0 + 64u, k0 = j0 - first, k1 = j1 - first;
            const bool a0 = j0 >= first && j0 < end, a1 = j1 >= first && j1 < end;
            const CtrVar xv0 = ctr_var(rk, lr, static_cast<uint32_t>(j0)), xv1 = ctr_var(rk, lr, static_cast<uint32_t>(j1));
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
                ctr_round1(pre, xv0, s[0]);
                ctr_round1(pre, xv1, s[1]);
                aes256_rounds<2, 2>(rk, lr, s);
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
            aes256_rounds<DBL ? 2 : 1, 2>(rk, lr, s);
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
template <class ET>
__device__ __forceinline__ void small_walk(uint32_t *row0, uint32_t lane, bool valid, int cnt, uint64_t j0, u128 D, const ET *in,
                                           ET *out, uint64_t first, uint64_t range_end, const SmallParams &p)
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
                    __builtin_nontemporal_store(static_cast<ET>((pt + val) & p.mask_lo), out + (j - first));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    } else if (valid) {
        for (int tt = 0; tt < cnt; tt++) {
            const uint64_t j = j0 + tt;
            if (j < first || j >= range_end) continue;
            const uint64_t val = extract64(D, p.b * tt);
            out[j - first] = static_cast<ET>(((in ? static_cast<uint64_t>(in[j - first]) : 0ull) + val) & p.mask_lo);
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

template <bool PAIR, class ET = uint64_t>
__global__ __launch_bounds__(kSmallThreads) void prf_small_chain_kernel(const RoundKeys rk, const SmallChainTable tb, int n_chains, const SmallParams p)
{
    constexpr uint32_t WAVES = kSmallThreads / 64, TILE = PAIR ? 128u : 64u;
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
        u128 prevA = 0, prevB = 0;
        if (PAIR) {
            // two blocks per lane on the same prefix, one stream per step
            for (int c = 0; c < n_streams; c++) {
                const CtrPrefix pre = load_prefix(pre_lds, sbase + c);
                const int link = single ? c : c - 1;
                WalkPt ptA{}, ptB{};
                if (link >= 0 && fastA) ptA = small_walk32_load(reinterpret_cast<const ET *>(tb.in[link0 + link]), e0A, first, lane, m32);
                if (link >= 0 && fastB) ptB = small_walk32_load(reinterpret_cast<const ET *>(tb.in[link0 + link]), e0B, first, lane, m32);
                uint32_t s[2][4];
                ctr_round1(pre, xA, s[0]);
                ctr_round1(pre, xB, s[1]);
                aes256_rounds<2, 2>(rk, lr, s);
                const u128 SA = words_to_u128(s[0]), SB = words_to_u128(s[1]);
                if (link >= 0) {
                    const uint64_t *in = tb.in[link0 + link];
                    uint64_t *out = tb.out[link0 + link];
                    const ET *ein = reinterpret_cast<const ET *>(in);
                    ET *eout = reinterpret_cast<ET *>(out);
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
                aes256_rounds<2, 2>(rk, lr, s);
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
        aes256_rounds<2, 2>(rk, lr, s);
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
        aes256_rounds<1, 2>(rk, lr, s);
        // row word order = little-endian words of the 128-bit block value (word 0 = bits 0..31)
        *reinterpret_cast<uint4 *>(row0 + 4 * lane) = drop ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(s[0][3], s[0][2], s[0][1], s[0][0]);
        __builtin_amdgcn_wave_barrier();
        const uint32_t valid_mask = static_cast<uint32_t>(__ballot(valid));
        uint32_t partial_mask = static_cast<uint32_t>(__ballot(valid && cnt < p.m));
        const int n_valid = __popc(valid_mask);
        const uint32_t j0_lo = static_cast<uint32_t>(j0), j0_hi = static_cast<uint32_t>(j0 >> 32);
        int lane_base = 0;
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
    std::vector<Piece> cut;
    for (const Piece &pc : pieces) {
        int parts = (pc.l1 + kMaxLinks - 1) / kMaxLinks;
        if (single_parts) {
            parts = std::max(parts, std::min(single_parts, pc.l1));
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
        const uint64_t items = all_half ? 2 * tiles : tiles, cus = static_cast<uint64_t>(env.num_cus);
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
                               (all_half ? 1 : 0) | probe, iter, lo, hi, env.te0_dev, cq);
        else if (summed)
            hipLaunchKernelGGL((prf_chain_kernel<kPrfThreads, true, false>), dim3(grid), dim3(kPrfThreads), 0, env.stream, env.rk, tb, nc,
                               (all_half ? 1 : 0) | probe, iter, lo, hi, env.te0_dev, cq);
        else
            hipLaunchKernelGGL((prf_chain_kernel<kPrfThreads, false, false>), dim3(grid), dim3(kPrfThreads), 0, env.stream, env.rk, tb, nc,
                               (all_half ? 1 : 0) | probe, iter, lo, hi, env.te0_dev, cq);
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
    const SmallParams p = small_params_of(env, iter, n, n_jobs);
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
#define SRDS_LAUNCH(CB, IT, OT)                                                                                                               \
    hipLaunchKernelGGL((small_reduce_decrypt_split_kernel<CB, IT, OT>), dim3(grid32), dim3(kSmallThreads), 0, env.stream, env.rk, p, add_idx,  \
                       minus_idx, (has_minus ? 1 : 0) | probe, first, count, bf, bc, C, t, agg_out_dev, out_dev)
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
    for (int i = 0; i < n_chains; i++) if (chains[i].sum_out_dev) return hipErrorNotSupported;      // the fused sum lives in the wide kernel
    SmallParams p = small_params_of(env, iter, n, n_jobs);
    if (env.elem32) {
        if (env.b > 32) return hipErrorInvalidValue;
        p.no_direct = 1;                      // the 16-byte direct accesses of m <= 4 assume 8-byte elements: b = 32 walks its rows like b < 32
    }
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
            const uint64_t t = (pc.blk_count + tile - 1) / tile;
            wend += t * static_cast<uint64_t>(ns);
            tb.wend[nc] = wend;
            tiles += t;
            links += len; streams += ns; nc++;
        }
        const uint64_t cus = static_cast<uint64_t>(env.num_cus);
        const int grid = static_cast<int>(tiles < cus ? tiles : cus);
        if (env.elem32) {
            if (pair) hipLaunchKernelGGL((prf_small_chain_kernel<true, uint32_t>), dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, tb, nc, p);
            else hipLaunchKernelGGL((prf_small_chain_kernel<false, uint32_t>), dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, tb, nc, p);
        } else if (pair) hipLaunchKernelGGL((prf_small_chain_kernel<true>), dim3(grid), dim3(kSmallThreads), 0, env.stream, env.rk, tb, nc, p);
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

// ------------------------------------------------------------------------------------------
// Streaming kernels (HBM-bound)
// ------------------------------------------------------------------------------------------
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
__global__ __launch_bounds__(kStreamThreads) void widen_u32_kernel(uint64_t n, const uint32_t *in, uint64_t *out)
{
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n; j += static_cast<uint64_t>(gridDim.x) * kStreamThreads)
        out[j] = in[j];
}
__global__ __launch_bounds__(kStreamThreads) void narrow_u32_kernel(uint64_t n, const uint64_t *in, uint32_t *out)
{
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n; j += static_cast<uint64_t>(gridDim.x) * kStreamThreads)
        out[j] = static_cast<uint32_t>(in[j]);
}
hipError_t launch_widen_u32(const LaunchEnv &env, uint64_t n, const uint32_t *in_dev, uint64_t *out_dev)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(widen_u32_kernel, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, n, in_dev, out_dev);
    return hipGetLastError();
}
hipError_t launch_narrow_u32(const LaunchEnv &env, uint64_t n, const uint64_t *in_dev, uint32_t *out_dev)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(narrow_u32_kernel, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, n, in_dev, out_dev);
    return hipGetLastError();
}

// out = (in + add - minus) & mask.  L = 2: one 16-B element per lane-iteration.
__global__ __launch_bounds__(kStreamThreads) void combine_wide_kernel(uint64_t n, const uint64_t *in, int in_limbs,
                                                                      const uint64_t *add, const uint64_t *minus,
                                                                      uint64_t *out, uint64_t mask_lo, uint64_t mask_hi)
{
    const u128 mask = (static_cast<u128>(mask_hi) << 64) | mask_lo;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        u128 v = in_limbs == 2 ? ld128_nt(in + 2 * j) : static_cast<u128>(__builtin_nontemporal_load(in + j));
        if (add) v += ld128_nt(add + 2 * j);
        if (minus) v -= ld128_nt(minus + 2 * j);
        st128_nt(out + 2 * j, v & mask);
    }
}

__global__ __launch_bounds__(kStreamThreads) void combine_small_kernel(uint64_t n, const uint64_t *in, const uint64_t *add,
                                                                       const uint64_t *minus, uint64_t *out, uint64_t mask)
{
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        uint64_t v = in[j];
        if (add) v += add[j];
        if (minus) v -= minus[j];
        out[j] = v & mask;
    }
}

// Several combines of equal length in one launch (the online encrypts of the clients a process hosts when their masks were
// precomputed: a hundred LeNet-sized vectors are launch-bound one by one).  blockIdx.y = vector.
constexpr int kMaxCombine = 64;
struct CombineTable {
    const uint64_t *in[kMaxCombine], *add[kMaxCombine], *minus[kMaxCombine];
    uint64_t *out[kMaxCombine];
};

template <bool WIDE>
__global__ __launch_bounds__(kStreamThreads) void combine_batch_kernel(uint64_t n, const CombineTable tb, int in_limbs, uint64_t mask_lo, uint64_t mask_hi)
{
    const int v = blockIdx.y;
    const uint64_t *in = tb.in[v], *add = tb.add[v], *minus = tb.minus[v];
    uint64_t *out = tb.out[v];
    const u128 mask = (static_cast<u128>(mask_hi) << 64) | mask_lo;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        if (WIDE) {
            u128 x = in_limbs == 2 ? ld128_nt(in + 2 * j) : static_cast<u128>(__builtin_nontemporal_load(in + j));
            if (add) x += ld128_nt(add + 2 * j);
            if (minus) x -= ld128_nt(minus + 2 * j);
            st128_nt(out + 2 * j, x & mask);
        } else {
            uint64_t x = in[j];
            if (add) x += add[j];
            if (minus) x -= minus[j];
            out[j] = x & mask_lo;
        }
    }
}

hipError_t launch_combine_batch(const LaunchEnv &env, uint64_t n, int n_vec, const uint64_t *const *in_dev, int in_limbs,
                                const uint64_t *const *add_dev, const uint64_t *const *minus_dev, uint64_t *const *out_dev)
{
    if (n == 0 || n_vec == 0) return hipSuccess;
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    for (int v0 = 0; v0 < n_vec; v0 += kMaxCombine) {
        const int nv = std::min(kMaxCombine, n_vec - v0);
        CombineTable tb{};
        for (int v = 0; v < nv; v++) {
            tb.in[v] = in_dev[v0 + v]; tb.add[v] = add_dev ? add_dev[v0 + v] : nullptr;
            tb.minus[v] = minus_dev ? minus_dev[v0 + v] : nullptr; tb.out[v] = out_dev[v0 + v];
        }
        // enough blocks per vector to fill the chip together, at most 8 x 256 threads per CU in all
        uint64_t bx = (n + kStreamThreads - 1) / kStreamThreads;
        const uint64_t cap = std::max<uint64_t>(1, static_cast<uint64_t>(env.num_cus) * 8 / nv);
        if (bx > cap) bx = cap;
        const dim3 grid(static_cast<unsigned>(bx), static_cast<unsigned>(nv));
        if (env.b > 64) hipLaunchKernelGGL(combine_batch_kernel<true>, grid, dim3(kStreamThreads), 0, env.stream, n, tb, in_limbs, lo, hi);
        else hipLaunchKernelGGL(combine_batch_kernel<false>, grid, dim3(kStreamThreads), 0, env.stream, n, tb, in_limbs, lo, hi);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_combine(const LaunchEnv &env, uint64_t n, const uint64_t *in_dev, int in_limbs,
                          const uint64_t *add_dev, const uint64_t *minus_dev, uint64_t *out_dev)
{
    if (n == 0) return hipSuccess;
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    const int grid = stream_grid(env, n);
    if (env.b > 64)
        hipLaunchKernelGGL(combine_wide_kernel, dim3(grid), dim3(kStreamThreads), 0, env.stream, n, in_dev, in_limbs,
                           add_dev, minus_dev, out_dev, lo, hi);
    else
        hipLaunchKernelGGL(combine_small_kernel, dim3(grid), dim3(kStreamThreads), 0, env.stream, n, in_dev, add_dev,
                           minus_dev, out_dev, lo);
    return hipGetLastError();
}

// C-way element-wise mod-add.  WIDE: 128-bit elements (carry between the two limbs);
// otherwise each limb is its own element and a 16-B slot simply carries two of them.
template <bool WIDE>
__global__ __launch_bounds__(kStreamThreads) void aggregate_elem_kernel(int C, const PtrTable ops, uint64_t n_limbs,
                                                                        uint64_t *out, uint64_t mask_lo, uint64_t mask_hi)
{
    const uint64_t *const *tab = ops.p;
    const uint64_t n_slots = n_limbs / 2;
    for (uint64_t s = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; s < n_slots;
         s += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        uint64_t a0 = 0, a1 = 0;
        u128 acc = 0;
#pragma unroll 4
        for (int c = 0; c < C; c++) {
            // every operand byte is read exactly once: stream past the caches
            const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(tab[c] + 2 * s));
            if (WIDE) acc += (static_cast<u128>(v[1]) << 64) | v[0];
            else { a0 += v[0]; a1 += v[1]; }
        }
        if (WIDE) { a0 = static_cast<uint64_t>(acc); a1 = static_cast<uint64_t>(acc >> 64); }
        u64x2 r;
        r[0] = a0 & mask_lo; r[1] = a1 & (WIDE ? mask_hi : mask_lo);
        __builtin_nontemporal_store(r, reinterpret_cast<u64x2 *>(out + 2 * s));
    }
    if (!WIDE && (n_limbs & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t a = 0;
        for (int c = 0; c < C; c++) a += tab[c][n_limbs - 1];
        out[n_limbs - 1] = a & mask_lo;
    }
}

// one-limb vectors whose operands are only 8-byte aligned (a sub-range that starts at an odd element): one element per lane
__global__ __launch_bounds__(kStreamThreads) void aggregate_elem8_kernel(int C, const PtrTable ops, uint64_t n, uint64_t *out, uint64_t mask_lo)
{
    const uint64_t *const *tab = ops.p;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n; j += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        uint64_t a = 0;
#pragma unroll 4
        for (int c = 0; c < C; c++) a += __builtin_nontemporal_load(tab[c] + j);
        __builtin_nontemporal_store(a & mask_lo, out + j);
    }
}

// compact layout (int_bits <= 32, uint32 elements): four elements per lane in 16-byte accesses; 32-bit sums wrap mod 2^32, which 2^b divides
template <bool VEC>
__global__ __launch_bounds__(kStreamThreads) void aggregate_elem_u32_kernel(int C, const PtrTable ops, uint64_t n, uint32_t *out, uint32_t mask)
{
    const uint64_t *const *tab = ops.p;
    if (VEC) {
        const uint64_t n4 = n / 4;
        for (uint64_t s = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; s < n4; s += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 4
            for (int c = 0; c < C; c++) {
                const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(tab[c]) + s);
                a0 += static_cast<uint32_t>(v[0]); a1 += static_cast<uint32_t>(v[0] >> 32);
                a2 += static_cast<uint32_t>(v[1]); a3 += static_cast<uint32_t>(v[1] >> 32);
            }
            u64x2 r;
            r[0] = (a0 & mask) | (static_cast<uint64_t>(a1 & mask) << 32);
            r[1] = (a2 & mask) | (static_cast<uint64_t>(a3 & mask) << 32);
            __builtin_nontemporal_store(r, reinterpret_cast<u64x2 *>(out) + s);
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
            const uint64_t j = n4 * 4 + threadIdx.x;
            uint32_t a = 0;
            for (int c = 0; c < C; c++) a += reinterpret_cast<const uint32_t *>(tab[c])[j];
            out[j] = a & mask;
        }
    } else {
        for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n; j += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
            uint32_t a = 0;
#pragma unroll 4
            for (int c = 0; c < C; c++) a += __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(tab[c]) + j);
            __builtin_nontemporal_store(a & mask, out + j);
        }
    }
}

static inline PtrTable make_table(int C, const uint64_t *const *ops)
{
    PtrTable t;
    for (int c = 0; c < kMaxOps; c++) t.p[c] = c < C ? ops[c] : nullptr;
    return t;
}

hipError_t launch_aggregate_elem_u32(const LaunchEnv &env, int C, const uint32_t *const *ops, uint64_t n, uint32_t *out_dev)
{
    if (n == 0) return hipSuccess;
    if (C > kMaxOps || env.b > 32) return hipErrorInvalidValue;
    const PtrTable tab_dev = make_table(C, reinterpret_cast<const uint64_t *const *>(ops));
    const uint32_t mask = env.b == 32 ? 0xffffffffu : ((1u << env.b) - 1u);
    bool a16 = (reinterpret_cast<uintptr_t>(out_dev) & 15u) == 0;
    for (int c = 0; c < C; c++) a16 = a16 && (reinterpret_cast<uintptr_t>(ops[c]) & 15u) == 0;
    const int bpc = C >= 3 ? 2 : 8;
    int grid = stream_grid(env, a16 ? n / 4 + 1 : n);
    if (grid > env.num_cus * bpc) grid = env.num_cus * bpc;
    if (a16) hipLaunchKernelGGL(aggregate_elem_u32_kernel<true>, dim3(grid), dim3(kStreamThreads), 0, env.stream, C, tab_dev, n, out_dev, mask);
    else hipLaunchKernelGGL(aggregate_elem_u32_kernel<false>, dim3(grid), dim3(kStreamThreads), 0, env.stream, C, tab_dev, n, out_dev, mask);
    return hipGetLastError();
}

hipError_t launch_aggregate_elem(const LaunchEnv &env, int C, const uint64_t *const *ops, uint64_t n, uint64_t *out_dev)
{
    if (n == 0) return hipSuccess;
    if (C > kMaxOps) return hipErrorInvalidValue;
    const PtrTable tab_dev = make_table(C, ops);
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    const uint64_t n_limbs = env.b > 64 ? 2 * n : n;
    bool a16 = (reinterpret_cast<uintptr_t>(out_dev) & 15u) == 0;
    for (int c = 0; c < C; c++) a16 = a16 && (reinterpret_cast<uintptr_t>(ops[c]) & 15u) == 0;
    if (!a16) {
        if (env.b > 64) return hipErrorInvalidValue;
        hipLaunchKernelGGL(aggregate_elem8_kernel, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, C, tab_dev, n, out_dev, lo);
        return hipGetLastError();
    }
    // a lane has C 16-byte loads in flight per slot: with many operands FEWER resident waves stream faster (measured at
    // C = 10, n = 1e7: 8 / 4 / 2 / 1 blocks per CU -> 5.5 / 5.9 / 6.0 / 4.4 TB/s; two operands want 4-8)
    const int bpc = C >= 3 ? 2 : 8;
    int grid = stream_grid(env, n_limbs / 2 + 1);
    if (grid > env.num_cus * bpc) grid = env.num_cus * bpc;
    if (env.b > 64)
        hipLaunchKernelGGL(aggregate_elem_kernel<true>, dim3(grid), dim3(kStreamThreads), 0, env.stream, C, tab_dev, n_limbs,
                           out_dev, lo, hi);
    else
        hipLaunchKernelGGL(aggregate_elem_kernel<false>, dim3(grid), dim3(kStreamThreads), 0, env.stream, C, tab_dev, n_limbs,
                           out_dev, lo, hi);
    return hipGetLastError();
}

// ---- packed aggregate: C-way add of n_limbs-limb integers with full carry propagation ----
// Stage 1 (this kernel): per 16-B slot the C-way column sums (lo, hi = overflow count), the
// fold hi -> next limb, and a (generate, propagate) carry scan inside the 256-slot block via
// wave ballots + the integer-add trick; the block is resolved with carry-in 0 and publishes
// (G, P).  Stage 2 (packed_fixup_kernel): look-back over the block summaries and ripple the
// (rare) +1 into blocks whose carry-in is 1.
constexpr int kPackedThreads = 256;
uint64_t packed_num_blocks(uint64_t n_limbs) { return ((n_limbs + 1) / 2 + kPackedThreads - 1) / kPackedThreads; }

__device__ __forceinline__ void column_sums(int C, const PtrTable &ops, uint64_t slot, uint64_t n_limbs,
                                            uint64_t &lo0, uint64_t &hi0, uint64_t &lo1, uint64_t &hi1)
{
    const uint64_t *const *tab = ops.p;
    u128 a0 = 0, a1 = 0;
    const bool full = 2 * slot + 1 < n_limbs;
    if (full) {
#pragma unroll 4
        for (int c = 0; c < C; c++) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(tab[c] + 2 * slot);
            a0 += v.x; a1 += v.y;
        }
    } else {
        for (int c = 0; c < C; c++) a0 += tab[c][2 * slot];
    }
    lo0 = static_cast<uint64_t>(a0); hi0 = static_cast<uint64_t>(a0 >> 64);
    lo1 = static_cast<uint64_t>(a1); hi1 = static_cast<uint64_t>(a1 >> 64);
}

__global__ __launch_bounds__(kPackedThreads) void aggregate_packed_kernel(int C, const PtrTable tab, uint64_t n_limbs,
                                                                          uint64_t top_mask, uint64_t *out, uint32_t *summaries)
{
    __shared__ uint64_t sh_hi[kPackedThreads];
    __shared__ uint32_t sh_zc[kPackedThreads];
    __shared__ uint32_t sh_wg[kPackedThreads / 64], sh_wp[kPackedThreads / 64];
    const uint64_t n_slots = (n_limbs + 1) / 2;
    const uint64_t slot = static_cast<uint64_t>(blockIdx.x) * kPackedThreads + threadIdx.x;
    const bool live = slot < n_slots;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    uint64_t lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
    if (live) column_sums(C, tab, slot, n_limbs, lo0, hi0, lo1, hi1);
    // limb 2s+1: z1 = lo1 + hi0
    const uint64_t zl1 = lo1 + hi0;
    const uint32_t zc1 = zl1 < lo1;
    sh_hi[tid] = hi1;
    sh_zc[tid] = zc1;
    __syncthreads();
    uint64_t hi_prev = 0; uint32_t zc_prev = 0;
    if (tid > 0) { hi_prev = sh_hi[tid - 1]; zc_prev = sh_zc[tid - 1]; }
    else if (slot > 0 && live) {
        uint64_t pl0, ph0, pl1, ph1;
        column_sums(C, tab, slot - 1, n_limbs, pl0, ph0, pl1, ph1);
        const uint64_t pz = pl1 + ph0;
        hi_prev = ph1; zc_prev = pz < pl1;
    }
    // limb 2s: z0 = lo0 + hi_prev
    const uint64_t zl0 = lo0 + hi_prev;
    const uint32_t zc0 = zl0 < lo0;
    // X + Y with Y = z-carry of the previous limb
    const uint64_t s0 = zl0 + zc_prev;
    const uint32_t g0 = s0 < zl0, p0 = s0 == ~0ull;
    const uint64_t s1 = zl1 + zc0;
    const uint32_t g1 = s1 < zl1, p1 = s1 == ~0ull;
    const uint32_t Gt = live ? (g1 | (p1 & g0)) : 0u;
    const uint32_t Pt = live ? (p1 & p0) : 1u;      // dead lanes are transparent
    const uint64_t Gm = __ballot(Gt), Pm = __ballot(Pt);
    const uint64_t a = Gm | Pm, bb = Gm;
    const uint64_t sum0 = a + bb;
    if (lane == 0) { sh_wg[wave] = sum0 < a; sh_wp[wave] = Pm == ~0ull; }
    __syncthreads();
    uint32_t cin = 0;
    for (int v = 0; v < wave; v++) cin = sh_wg[v] | (sh_wp[v] & cin);
    const uint64_t cv = (a + bb + cin) ^ Pm;
    const uint32_t ct = (cv >> lane) & 1u;
    const uint64_t r0 = s0 + ct;
    const uint32_t k0 = g0 | (p0 & ct);
    const uint64_t r1 = s1 + k0;
    if (live) {
        const uint64_t i0 = 2 * slot, i1 = i0 + 1;
        if (i1 < n_limbs) {
            *reinterpret_cast<ulonglong2 *>(out + i0) =
                make_ulonglong2(r0, i1 == n_limbs - 1 ? (r1 & top_mask) : r1);
        } else {
            out[i0] = r0 & top_mask;
        }
    }
    if (tid == 0) {
        uint32_t c = 0, pall = 1;
        for (int v = 0; v < kPackedThreads / 64; v++) { c = sh_wg[v] | (sh_wp[v] & c); pall &= sh_wp[v]; }
        summaries[blockIdx.x] = c | (pall << 1);
    }
}

__global__ void packed_fixup_kernel(uint64_t n_blocks, uint64_t n_limbs, uint64_t top_mask, const uint32_t *summaries, uint64_t *out)
{
    const uint64_t B = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x + 1;
    if (B >= n_blocks) return;
    uint32_t cin = 0;
    for (uint64_t v = B; v-- > 0;) {
        const uint32_t s = summaries[v];
        if (s & 1u) { cin = 1; break; }
        if (!(s & 2u)) break;
    }
    if (!cin) return;
    const uint64_t first = B * kPackedThreads * 2;
    uint64_t last = first + kPackedThreads * 2;
    if (last > n_limbs) last = n_limbs;
    for (uint64_t i = first; i < last; i++) {
        uint64_t v = out[i] + 1;
        if (i == n_limbs - 1) v &= top_mask;
        out[i] = v;
        if (v != 0) break;
    }
}

hipError_t launch_aggregate_packed(const LaunchEnv &env, int C, const uint64_t *const *ops, uint64_t n_limbs,
                                   uint64_t total_bits, uint64_t *out_dev, uint32_t *summaries_dev)
{
    if (n_limbs == 0) return hipSuccess;
    if (C > kMaxOps) return hipErrorInvalidValue;
    const PtrTable tab_dev = make_table(C, ops);
    const unsigned top = static_cast<unsigned>(total_bits % 64);
    const uint64_t top_mask = top ? ((1ull << top) - 1) : ~0ull;
    const uint64_t nb = packed_num_blocks(n_limbs);
    hipLaunchKernelGGL(aggregate_packed_kernel, dim3(static_cast<unsigned>(nb)), dim3(kPackedThreads), 0, env.stream, C, tab_dev,
                       n_limbs, top_mask, out_dev, summaries_dev);
    if (nb > 1) {
        const unsigned fb = static_cast<unsigned>((nb - 1 + 255) / 256);
        hipLaunchKernelGGL(packed_fixup_kernel, dim3(fb), dim3(256), 0, env.stream, nb, n_limbs, top_mask, summaries_dev, out_dev);
    }
    return hipGetLastError();
}

// ---- slice helpers for a packed reduce that is cut across GPUs (flashe_amd/dist.py run_packed) ----
// probe: x holds a slice sum as n_limbs - 1 body limbs plus one carry limb on top.
// info[0] = x[0], info[1] = 1 iff body limbs [1, n_limbs - 1) are all ~0, info[2] = x[n_limbs - 1].
__global__ __launch_bounds__(kStreamThreads) void packed_probe_kernel(uint64_t n_limbs, const uint64_t *x, uint64_t *info)
{
    bool ones = true;
    for (uint64_t i = 1 + static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; i + 1 < n_limbs;
         i += static_cast<uint64_t>(gridDim.x) * kStreamThreads)
        ones &= __builtin_nontemporal_load(x + i) == ~0ull;
    if (!__all(ones) && (threadIdx.x & 63) == 0) info[1] = 0;     // every writer stores the same value
    if (blockIdx.x == 0 && threadIdx.x == 0) { info[0] = x[0]; info[2] = x[n_limbs - 1]; }
}
__global__ void packed_probe_init_kernel(uint64_t *info) { info[0] = 0; info[1] = 1; info[2] = 0; }

// x = (x + cin) mod 2^total_bits in place, one workgroup: the ripple stops at the first limb that is
// not all ones, which is limb 0 or 1 for anything but adversarial data.
constexpr int kRippleThreads = 1024;
// infos != null: the carry-in is derived on the device from the (low limb, body-all-ones, carry-out) triples of the limb
// slices below this one (ranks 0 .. n_below - 1 of a packed reduce cut across GPUs): a slice passes its carry-in on when all its
// body limbs are ones and the low limb overflows, on top of its own carry-out.
// stride: words from one slice's triple to the next more significant one's (3 = slice 0 first; -3 = the lowest slice's triple is the
// LAST of the gathered ones and infos points at it: slices numbered from the most significant end, as element slices of a packed
// vector are -- element 0 is the most significant, jzf_weights.py:59-62).
__global__ __launch_bounds__(kRippleThreads) void packed_add_carry_kernel(uint64_t n_limbs, uint64_t top_mask, uint64_t cin, uint64_t *x,
                                                                          const uint64_t *__restrict__ infos, int n_below, int stride)
{
    __shared__ int first_stop;
    const int tid = threadIdx.x;
    if (infos) {
        uint64_t carry = 0;
        for (int g = 0; g < n_below; g++) {
            const uint64_t *t = infos + static_cast<int64_t>(g) * stride;
            const uint64_t low = t[0], ones = t[1], cout = t[2];
            carry = cout + ((ones && low + carry < low) ? 1ull : 0ull);
        }
        cin = carry;
    }
    const uint64_t x0 = x[0];
    __syncthreads();
    const uint64_t s0 = x0 + cin;
    if (tid == 0) x[0] = s0;
    if (s0 < x0) {
        for (uint64_t base = 1; base < n_limbs; base += kRippleThreads) {
            const uint64_t i = base + tid;
            const uint64_t v = i < n_limbs ? x[i] : 0;
            if (tid == 0) first_stop = kRippleThreads;
            __syncthreads();
            if (v != ~0ull) atomicMin(&first_stop, tid);
            __syncthreads();
            const int f = first_stop;
            if (tid < f) x[i] = 0;
            else if (tid == f && i < n_limbs) x[i] = v + 1;
            __syncthreads();
            if (f < kRippleThreads) break;
        }
    }
    __syncthreads();
    if (tid == 0) x[n_limbs - 1] &= top_mask;
}

hipError_t launch_packed_probe(const LaunchEnv &env, uint64_t n_limbs, const uint64_t *x_dev, uint64_t *info_dev)
{
    hipLaunchKernelGGL(packed_probe_init_kernel, dim3(1), dim3(1), 0, env.stream, info_dev);
    if (n_limbs == 0) return hipGetLastError();
    hipLaunchKernelGGL(packed_probe_kernel, dim3(stream_grid(env, n_limbs)), dim3(kStreamThreads), 0, env.stream, n_limbs, x_dev, info_dev);
    return hipGetLastError();
}

hipError_t launch_packed_add_carry(const LaunchEnv &env, uint64_t n_limbs, uint64_t total_bits, uint64_t cin, uint64_t *x_dev)
{
    if (n_limbs == 0) return hipSuccess;
    const unsigned top = static_cast<unsigned>(total_bits % 64);
    const uint64_t top_mask = top ? ((1ull << top) - 1) : ~0ull;
    hipLaunchKernelGGL(packed_add_carry_kernel, dim3(1), dim3(kRippleThreads), 0, env.stream, n_limbs, top_mask, cin, x_dev,
                       static_cast<const uint64_t *>(nullptr), 0, 3);
    return hipGetLastError();
}

hipError_t launch_packed_resolve_carry(const LaunchEnv &env, uint64_t n_limbs, uint64_t total_bits, const uint64_t *infos_dev, int n_below,
                                       uint64_t *x_dev, int stride_words)
{
    if (n_limbs == 0) return hipSuccess;
    const unsigned top = static_cast<unsigned>(total_bits % 64);
    const uint64_t top_mask = top ? ((1ull << top) - 1) : ~0ull;
    hipLaunchKernelGGL(packed_add_carry_kernel, dim3(1), dim3(kRippleThreads), 0, env.stream, n_limbs, top_mask, 0ull, x_dev, infos_dev, n_below,
                       stride_words);
    return hipGetLastError();
}

// ---- bit-packing codec ----
// pack: one output limb per lane, gathering every element that overlaps bits [64w, 64w + 64).
__global__ __launch_bounds__(kStreamThreads) void pack_kernel(uint64_t n, int b, int L, const uint64_t *in, uint64_t *out,
                                                              uint64_t n_limbs, uint64_t mask_lo, uint64_t mask_hi)
{
    const u128 mask = (static_cast<u128>(mask_hi) << 64) | mask_lo;
    for (uint64_t w = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; w < n_limbs;
         w += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        const uint64_t bit0 = 64 * w;
        uint64_t e = bit0 / b;
        uint64_t e_last = (bit0 + 63) / b;
        if (e_last >= n) e_last = n - 1;
        uint64_t acc = 0;
        for (; e <= e_last; e++) {
            const uint64_t j = n - 1 - e;
            const u128 v = (L == 2 ? ld128(in + 2 * j) : static_cast<u128>(in[j])) & mask;
            const uint64_t pos = e * b;
            if (pos >= bit0) acc |= static_cast<uint64_t>(v) << (pos - bit0);
            else acc |= static_cast<uint64_t>(v >> (bit0 - pos));
        }
        out[w] = acc;
    }
}

__global__ __launch_bounds__(kStreamThreads) void unpack_kernel(uint64_t n, int b, int L, const uint64_t *in, uint64_t *out,
                                                                uint64_t n_limbs, uint64_t mask_lo, uint64_t mask_hi)
{
    const u128 mask = (static_cast<u128>(mask_hi) << 64) | mask_lo;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        const uint64_t pos = (n - 1 - j) * b;
        const uint64_t w = pos / 64;
        const unsigned s = static_cast<unsigned>(pos % 64);
        const u128 l0 = in[w];
        const u128 l1 = w + 1 < n_limbs ? in[w + 1] : 0;
        const u128 l2 = w + 2 < n_limbs ? in[w + 2] : 0;
        u128 v = l0 >> s;
        if (s) { v |= l1 << (64 - s); v |= l2 << (128 - s); }
        else v |= l1 << 64;
        v &= mask;
        if (L == 2) st128(out + 2 * j, v);
        else out[j] = static_cast<uint64_t>(v);
    }
}

// b = 128: the packed integer is the element order reversed (element 0 most significant) -- one 16-byte move per lane,
// the same kernel packs and unpacks.
__global__ __launch_bounds__(kStreamThreads) void reverse128_kernel(uint64_t n, const uint64_t *in, uint64_t *out)
{
    for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; e < n;
         e += static_cast<uint64_t>(gridDim.x) * kStreamThreads)
        st128_nt(out + 2 * e, ld128_nt(in + 2 * (n - 1 - e)));
}

hipError_t launch_pack(const LaunchEnv &env, uint64_t n, const uint64_t *in_dev, uint64_t *out_dev)
{
    if (n == 0) return hipSuccess;
    if (env.b == 128) {
        hipLaunchKernelGGL(reverse128_kernel, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, n, in_dev, out_dev);
        return hipGetLastError();
    }
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    const uint64_t n_limbs = (n * static_cast<uint64_t>(env.b) + 63) / 64;
    hipLaunchKernelGGL(pack_kernel, dim3(stream_grid(env, n_limbs)), dim3(kStreamThreads), 0, env.stream, n, env.b,
                       env.b > 64 ? 2 : 1, in_dev, out_dev, n_limbs, lo, hi);
    return hipGetLastError();
}

hipError_t launch_unpack(const LaunchEnv &env, uint64_t n, const uint64_t *in_dev, uint64_t *out_dev)
{
    if (n == 0) return hipSuccess;
    if (env.b == 128) {
        hipLaunchKernelGGL(reverse128_kernel, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, n, in_dev, out_dev);
        return hipGetLastError();
    }
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    const uint64_t n_limbs = (n * static_cast<uint64_t>(env.b) + 63) / 64;
    hipLaunchKernelGGL(unpack_kernel, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, n, env.b,
                       env.b > 64 ? 2 : 1, in_dev, out_dev, n_limbs, lo, hi);
    return hipGetLastError();
}

// ---- sparse helpers ----
__global__ __launch_bounds__(kStreamThreads) void fill_kernel(uint64_t n, int L, uint64_t lo, uint64_t hi, uint64_t *out)
{
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        if (L == 2) *reinterpret_cast<ulonglong2 *>(out + 2 * j) = make_ulonglong2(lo, hi);
        else out[j] = lo;
    }
}

// out[loc[q]] = vals[q]  or  out[loc[q]] = (out[loc[q]] + vals[q]) mod 2^b.  loc must hold
// distinct positions within one launch (the reference's location lists are sets).
// sub (a constant, < 2^b) is subtracted from every value first: the sparse reduce adds vals[q] - zero.
__global__ __launch_bounds__(kStreamThreads) void scatter_kernel(uint64_t total, uint64_t k, int L, const uint32_t *loc, const uint64_t *vals,
                                                                 uint64_t *out, bool accumulate, uint64_t mask_lo, uint64_t mask_hi,
                                                                 uint64_t sub_lo, uint64_t sub_hi, uint32_t *err_flag)
{
    const u128 mask = (static_cast<u128>(mask_hi) << 64) | mask_lo;
    const u128 sub = (static_cast<u128>(sub_hi) << 64) | sub_lo;
    for (uint64_t q = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; q < k;
         q += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        const uint64_t p = loc[q];
        if (p >= total) { *err_flag = 1; continue; }       // never write outside the dense vector: skip and report
        if (L == 2) {
            u128 v = ld128(vals + 2 * q) - sub;
            if (accumulate) v += ld128(out + 2 * p);
            st128(out + 2 * p, v & mask);
        } else {
            uint64_t v = vals[q] - sub_lo;
            if (accumulate) v += out[p];
            out[p] = v & mask_lo;
        }
    }
}

// Sparse reduce over SORTED location lists, two launches for any number of clients.
// The dense vector is cut into spans of kSpan positions, one workgroup each.  Kernel A: start[s][c] = first entry of
// client c at or beyond position s * kSpan -- one thread per ENTRY: entry q opens every span between the span of entry
// q - 1 and its own (most entries open none), one coalesced pass over the location lists instead of a binary search per
// (span, client); it also reports lists that are not strictly increasing or reach beyond the vector.  Kernel B: the
// workgroup (1,024 threads: the gathers are the slow part, measured against 256 / 512 threads and 1 K ... 8 K spans) clears a
// span-sized accumulator in LDS, adds every client's entries that fall into its span (128-bit add =
// two LDS atomics; the low one returns the old value, which tells the lane exactly whether ITS add wrapped -- integer
// adds commute, so the sum does not depend on the order), then writes src + accumulator (or src - accumulator, base
// instead of src when there is none) for the WHOLE span: the dense output is written exactly once, coalesced.
#ifndef FLASHE_SPAN
#define FLASHE_SPAN 4096
#endif
constexpr int kSpan = FLASHE_SPAN;  // positions per span: 64 KiB of 128-bit accumulators, two workgroups per CU (8,192 = 128 KiB, one workgroup per CU,
                                    // twice as long slices per client: aggregate 0.171 against 0.169 ms, fused decrypt 0.442 against 0.420 -- config 5)
constexpr int kSpanThreads = 1024;
constexpr int kSpanBatch = 2;       // entries whose loads a lane keeps in flight at once (config 5, aggregate / fused decrypt: 8: 0.256 / 0.484 ms,
                                    // 4: 0.171 / 0.416, 2: 0.163 / 0.407, 1: 0.167 / 0.407 -- fewer gathers in flight stream faster here too)
struct ScatterTable {
    const uint32_t *loc[kMaxScatter];
    const uint64_t *vals[kMaxScatter];
    uint64_t k[kMaxScatter], sub_lo[kMaxScatter], sub_hi[kMaxScatter];
};

constexpr int kBoundsPerThread = 8;
__global__ __launch_bounds__(kStreamThreads) void span_bounds_kernel(const ScatterTable tb, int C, uint64_t n_spans, uint64_t total, uint32_t *start,
                                                                     uint32_t *err_flag)
{
    const int c = blockIdx.y;
    const uint64_t k = tb.k[c];
    const uint32_t *loc = tb.loc[c];
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int i = 0; i < kBoundsPerThread; i++) {
        const uint64_t q = (static_cast<uint64_t>(blockIdx.x) * kBoundsPerThread + i) * kStreamThreads + threadIdx.x;
        // q == k closes the list (the spans behind the last entry); lanes beyond it open nothing
        const bool live = q <= k;
        const uint64_t prev = live && q ? loc[q - 1] : 0, cur = live && q < k ? loc[q] : 0;
        if (live && q < k && (cur >= total || (q && cur <= prev))) *err_flag = 1;
        const uint64_t s_last = q < k ? std::min<uint64_t>(cur / kSpan, n_spans) : n_spans;
        const uint64_t s_first = !live ? s_last + 1 : q ? std::min<uint64_t>(prev / kSpan, n_spans) + 1 : 0;
        const bool is_long = s_first + 16 <= s_last;
        if (!is_long)
            for (uint64_t sp = s_first; sp <= s_last; sp++) start[sp * C + c] = static_cast<uint32_t>(q);
        // a long run of empty spans (a short list over a long vector, an empty client): the wave fills it together instead of one
        // lane storing span after span
        uint64_t pending = __ballot(is_long);
        while (pending) {
            const int src = __ffsll(static_cast<unsigned long long>(pending)) - 1;
            const uint64_t a = __shfl(s_first, src, 64), b = __shfl(s_last, src, 64);
            const uint32_t qq = static_cast<uint32_t>(__shfl(q, src, 64));
            for (uint64_t x = a + lane; x <= b; x += 64u) start[x * C + c] = qq;
            pending &= pending - 1;
        }
    }
}

// One batch of a span's entries, gathered into registers: flat entry f belongs to the client c with prefix[c] <= f < prefix[c + 1].
struct SpanBatch { uint32_t r[kSpanBatch]; int own[kSpanBatch]; u128 v[kSpanBatch]; };

template <int THREADS>
__device__ __forceinline__ void span_gather(SpanBatch &g, uint32_t f0, uint32_t n_entries, const uint32_t *prefix, const uint32_t *begin,
                                            const uint32_t *const *s_loc, const uint64_t *const *s_vals, int L, uint32_t p0)
{
#pragma unroll
    for (int e = 0; e < kSpanBatch; e++) {
        const uint32_t fe = f0 + e * THREADS;
        const uint32_t f = fe < n_entries ? fe : f0;     // surplus slots re-read the first entry (f0 < n_entries) and are not added
        int c = 0;
#pragma unroll
        for (int step = 32; step; step >>= 1)
            if (prefix[c + step] <= f) c += step;
        g.own[e] = c;
        const uint64_t q = static_cast<uint64_t>(begin[c]) + (f - prefix[c]);
        g.r[e] = s_loc[c][q] - p0;
        g.v[e] = L == 2 ? ld128(s_vals[c] + 2 * q) : static_cast<u128>(s_vals[c][q]);
    }
}

template <int THREADS>
__device__ __forceinline__ void span_add(const SpanBatch &g, uint32_t f0, uint32_t n_entries, unsigned long long *acc, const uint64_t *s_sub, int L,
                                         uint64_t span_len, uint32_t *err_flag)
{
#pragma unroll
    for (int e = 0; e < kSpanBatch; e++) {
        if (f0 + e * THREADS >= n_entries) break;
        if (g.r[e] >= span_len) { *err_flag = 1; continue; }      // position >= total, or a list that is not strictly increasing
        const u128 w = g.v[e] - ((static_cast<u128>(s_sub[2 * g.own[e] + 1]) << 64) | s_sub[2 * g.own[e]]);
        const unsigned long long wlo = static_cast<unsigned long long>(w), whi = static_cast<unsigned long long>(w >> 64);
        if (L == 2) {
            const unsigned long long old = atomicAdd(&acc[2 * g.r[e]], wlo);
            atomicAdd(&acc[2 * g.r[e] + 1], whi + (old + wlo < old ? 1ull : 0ull));
        } else {
            atomicAdd(&acc[g.r[e]], wlo);
        }
    }
}

template <int SPAN, int THREADS>
__global__ __launch_bounds__(THREADS) void span_reduce_kernel(const ScatterTable tb, int C, int L, uint64_t total, const uint32_t *start,
                                                              uint64_t base_lo, uint64_t base_hi, uint64_t mask_lo, uint64_t mask_hi,
                                                              const uint64_t *src, bool negate, uint64_t *out, uint32_t *err_flag)
{
    __shared__ unsigned long long acc[2 * SPAN];
    __shared__ uint32_t s_begin[kMaxScatter], s_prefix[2 * kMaxScatter + 2];
    __shared__ const uint32_t *s_loc[kMaxScatter];
    __shared__ const uint64_t *s_vals[kMaxScatter];
    __shared__ uint64_t s_sub[2 * kMaxScatter];
    const uint64_t span = blockIdx.x, p0 = span * SPAN;
    const uint64_t span_len = total - p0 < SPAN ? total - p0 : SPAN;
    const int tid = threadIdx.x;
    for (int i = tid; i < (L == 2 ? 2 : 1) * SPAN; i += THREADS) acc[i] = 0;
    // this span's slice [begin, begin + count) of every client's list (clamped: a malformed list was reported by kernel A and
    // must not turn into reads outside the lists) and the running total of the counts, by a shuffle scan in the first wave:
    // the entries of ALL clients are then walked as one flat index space -- consecutive lanes read consecutive entries
    if (tid < 64) {
        uint32_t cnt = 0;
        if (tid < C) {
            const uint32_t kc = static_cast<uint32_t>(tb.k[tid]);
            const uint32_t b0 = min(start[span * C + tid], kc), b1 = min(start[(span + 1) * C + tid], kc);
            cnt = b1 > b0 ? b1 - b0 : 0;
            s_begin[tid] = b0;
            s_loc[tid] = tb.loc[tid]; s_vals[tid] = tb.vals[tid];
            s_sub[2 * tid] = tb.sub_lo[tid]; s_sub[2 * tid + 1] = tb.sub_hi[tid];
        }
        uint32_t run = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(run, d, 64);
            if (tid >= d) run += up;
        }
        if (tid == 0) s_prefix[0] = 0;
        s_prefix[tid + 1] = tid < C ? run : 0xffffffffu;        // sentinels: the owner search needs no bounds
        s_prefix[tid + 65] = 0xffffffffu;
    }
    __syncthreads();
    const uint32_t n_entries = s_prefix[C];
    for (uint32_t f0 = tid; f0 < n_entries; f0 += kSpanBatch * THREADS) {
        SpanBatch g;
        span_gather<THREADS>(g, f0, n_entries, s_prefix, s_begin, s_loc, s_vals, L, static_cast<uint32_t>(p0));
        span_add<THREADS>(g, f0, n_entries, acc, s_sub, L, span_len, err_flag);
    }
    __syncthreads();
    const u128 mask = (static_cast<u128>(mask_hi) << 64) | mask_lo;
    const u128 base = (static_cast<u128>(base_hi) << 64) | base_lo;
    for (uint64_t r = tid; r < span_len; r += THREADS) {
        if (L == 2) {
            const u128 a = (static_cast<u128>(acc[2 * r + 1]) << 64) | acc[2 * r];
            const u128 from = src ? ld128_nt(src + 2 * (p0 + r)) : base;
            st128_nt(out + 2 * (p0 + r), (negate ? from - a : from + a) & mask);
        } else {
            const uint64_t from = src ? __builtin_nontemporal_load(src + p0 + r) : base_lo;
            __builtin_nontemporal_store((negate ? from - acc[r] : from + acc[r]) & mask_lo, out + p0 + r);
        }
    }
}

uint64_t span_count(uint64_t total) { return (total + kSpan - 1) / kSpan; }

// out[p] = from[p] +/- sum over clients c and entries q with loc[c][q] == p of (vals[c][q] - sub[c])   (mod 2^b), every p < total,
// from = src_dev when given (may be out_dev), the constant base otherwise; loc[c] strictly increasing.
// start_dev: (span_count(total) + 1) * C words of scratch.
hipError_t launch_span_bounds(const LaunchEnv &env, int C, const uint32_t *const *loc_dev, const uint64_t *k, uint64_t total, uint32_t *start_dev)
{
    if (C > kMaxScatter || C < 1) return hipErrorInvalidValue;
    if (total == 0) return hipSuccess;
    ScatterTable tb{};
    uint64_t kmax = 0;
    for (int c = 0; c < C; c++) {
        if (k[c] >= (1ull << 32)) return hipErrorInvalidValue;
        tb.loc[c] = loc_dev[c]; tb.k[c] = k[c];
        kmax = std::max(kmax, k[c]);
    }
    hipLaunchKernelGGL(span_bounds_kernel, dim3(static_cast<unsigned>(kmax / (kStreamThreads * kBoundsPerThread) + 1), C), dim3(kStreamThreads), 0,
                       env.stream, tb, C, span_count(total), total, start_dev, env.err_flag);
    return hipGetLastError();
}

hipError_t launch_span_reduce(const LaunchEnv &env, int C, const uint32_t *const *loc_dev, const uint64_t *const *vals_dev,
                              const uint64_t *k, const uint64_t *sub, uint64_t base_lo, uint64_t base_hi, uint64_t total,
                              uint32_t *start_dev, const uint64_t *src_dev, bool negate, uint64_t *out_dev, bool bounds_ready)
{
    if (C > kMaxScatter || C < 1) return hipErrorInvalidValue;
    if (total == 0) return hipSuccess;
    const int L = env.b > 64 ? 2 : 1;
    ScatterTable tb{};
    uint64_t kmax = 0;
    for (int c = 0; c < C; c++) {
        if (k[c] >= (1ull << 32)) return hipErrorInvalidValue;
        tb.loc[c] = loc_dev[c]; tb.vals[c] = vals_dev[c]; tb.k[c] = k[c];
        tb.sub_lo[c] = sub ? sub[static_cast<size_t>(L) * c] : 0;
        tb.sub_hi[c] = sub && L == 2 ? sub[2 * c + 1] : 0;
        kmax = std::max(kmax, k[c]);
    }
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    const uint64_t n_spans = span_count(total);
    if (!bounds_ready)
        hipLaunchKernelGGL(span_bounds_kernel, dim3(static_cast<unsigned>(kmax / (kStreamThreads * kBoundsPerThread) + 1), C), dim3(kStreamThreads), 0,
                           env.stream, tb, C, n_spans, total, start_dev, env.err_flag);
    hipLaunchKernelGGL((span_reduce_kernel<kSpan, kSpanThreads>), dim3(static_cast<unsigned>(n_spans)), dim3(kSpanThreads), 0, env.stream, tb, C, L, total,
                       start_dev, base_lo, base_hi, lo, hi, src_dev, negate, out_dev, env.err_flag);
    return hipGetLastError();
}

// ---- sparse + double mask: the masks of the run EDGES, computed only where they are needed -------------------------------------
// set_idx_list's sparse branch of the double mask (jzf_flashe.py:388-426) runs a per-position analysis of the clients' one-hot
// location vectors: position p of client c needs term(c + 1, p) on the ADD side unless client c + 1 holds p too (the masks of
// neighbouring clients telescope), and term(c, p) on the MINUS side unless client c - 1 holds p -- _static_prepare_decrypt_spar
// (:155-225) then evaluates the PRF at exactly the selected dense positions (a block without a selected slot costs no AES).  Here
// the work items are the clients' own list entries: entry q of client c looks its position up in the two neighbouring (sorted)
// lists and computes at most two AES blocks with the counter of the DENSE position (one chunk, begin = 0: counter = p / m, slot
// p % m), writing compact (add, minus) values that the span reduce then scatters -- sum_c k_c block pairs instead of
// (C + 1) x total blocks and no per-list one-hot of `total` bytes.
struct EdgeTable {
    const uint32_t *loc[kMaxScatter + 2];     // entry e + 1 = client c0 + e; entries 0 and nc + 1 = the neighbours outside the group (or null)
    uint64_t k[kMaxScatter + 2];
    uint64_t *va[kMaxScatter], *vm[kMaxScatter];
    uint64_t end[kMaxScatter];                // running total of the group's entries
};

__device__ __forceinline__ bool sorted_contains(const uint32_t *__restrict__ a, uint64_t n, uint32_t x)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo < n && a[lo] == x;
}

__global__ __launch_bounds__(kPrfThreads) void sparse_edge_prf_kernel(const RoundKeys rk, const EdgeTable tb, int nc, uint32_t c0, uint32_t iter0, int b,
                                                                      uint64_t mask_lo, uint64_t mask_hi, const uint32_t *__restrict__ te0)
{
    const uint32_t iter = iter0 + te0[kIterShiftWord];
    __shared__ uint32_t tab[kTabWords];
    fill_tables(tab, te0);
    const LaneRegs lr = lane_regs(tab);
    const u128 mask = (static_cast<u128>(mask_hi) << 64) | mask_lo;
    const uint32_t m = b > 64 ? 1u : 128u / static_cast<uint32_t>(b);
    const uint64_t n_items = tb.end[nc - 1];
    for (uint64_t f = static_cast<uint64_t>(blockIdx.x) * kPrfThreads + threadIdx.x; f < n_items; f += static_cast<uint64_t>(gridDim.x) * kPrfThreads) {
        int e = 0;
#pragma unroll
        for (int step = 32; step; step >>= 1)
            if (e + step < nc && tb.end[e + step - 1] <= f) e += step;
        const uint64_t q = f - (e ? tb.end[e - 1] : 0);
        const uint32_t p = tb.loc[e + 1][q];
        const bool in_prev = tb.loc[e] && sorted_contains(tb.loc[e], tb.k[e], p);
        const bool in_next = tb.loc[e + 2] && sorted_contains(tb.loc[e + 2], tb.k[e + 2], p);
        const uint32_t c = c0 + static_cast<uint32_t>(e);
        const uint64_t ctr = p / m;
        uint32_t s[2][4];
        set_block(s[0], iter, c + 1u, ctr);            // add side: list (prefix) c + 1
        set_block(s[1], iter, c, ctr);                 // minus side: list c
        aes256_encrypt<2>(rk, lr, s);
        const int sh = static_cast<int>(static_cast<uint32_t>(b) * (p - static_cast<uint32_t>(ctr) * m));
        const u128 A = in_next ? static_cast<u128>(0) : (words_to_u128(s[0]) >> sh) & mask;
        const u128 M = in_prev ? static_cast<u128>(0) : (words_to_u128(s[1]) >> sh) & mask;
        if (b > 64) { st128(tb.va[e] + 2 * q, A); st128(tb.vm[e] + 2 * q, M); }
        else { tb.va[e][q] = static_cast<uint64_t>(A); tb.vm[e][q] = static_cast<uint64_t>(M); }
    }
}

hipError_t launch_sparse_edge_prf(const LaunchEnv &env, uint32_t iter, int nc, uint32_t c0, const uint32_t *const *loc_with_neighbours,
                                  const uint64_t *k_with_neighbours, uint64_t *const *va_dev, uint64_t *const *vm_dev)
{
    if (nc < 1 || nc > kMaxScatter) return hipErrorInvalidValue;
    EdgeTable tb{};
    uint64_t total = 0;
    for (int e = 0; e < nc + 2; e++) { tb.loc[e] = loc_with_neighbours[e]; tb.k[e] = k_with_neighbours[e]; }
    for (int e = 0; e < nc; e++) { tb.va[e] = va_dev[e]; tb.vm[e] = vm_dev[e]; total += tb.k[e + 1]; tb.end[e] = total; }
    if (total == 0) return hipSuccess;
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    hipLaunchKernelGGL(sparse_edge_prf_kernel, dim3(grid_for(env, total, kPrfThreads)), dim3(kPrfThreads), 0, env.stream, env.rk, tb, nc, c0, iter, env.b, lo,
                       hi, env.te0_dev);
    return hipGetLastError();
}

// ---- Arbiter.dynamic_masking's cost model on the device (jzf_flashe_block.py:92-112) -----------------------------------------------
// canceled_out_pairs = sum over consecutive clients (c, c + 1) of the positions both hold (the reference ANDs one-hot vectors of
// `total` entries): every list entry of client c looks its position up in client c + 1's sorted list -- the neighbour lookup of
// sparse_edge_prf_kernel without the AES -- and the hits are counted (wave ballot, one atomic per wave).
__global__ __launch_bounds__(kStreamThreads) void shared_positions_kernel(const EdgeTable tb, int nc, unsigned long long *count)
{
    const uint64_t n_items = tb.end[nc - 1];
    unsigned long long mine = 0;
    for (uint64_t f = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; f < n_items; f += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        int e = 0;
#pragma unroll
        for (int step = 32; step; step >>= 1)
            if (e + step < nc && tb.end[e + step - 1] <= f) e += step;
        const uint64_t q = f - (e ? tb.end[e - 1] : 0);
        const uint32_t p = tb.loc[e + 1][q];
        if (tb.loc[e + 2] && sorted_contains(tb.loc[e + 2], tb.k[e + 2], p)) mine++;
    }
    for (int off = 32; off; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
}

hipError_t launch_shared_positions(const LaunchEnv &env, int nc, const uint32_t *const *loc_with_next, const uint64_t *k_with_next,
                                   unsigned long long *count_dev)
{
    if (nc < 1 || nc > kMaxScatter) return hipErrorInvalidValue;
    EdgeTable tb{};
    uint64_t total = 0;
    for (int e = 0; e < nc; e++) { tb.loc[e + 1] = loc_with_next[e]; tb.k[e + 1] = k_with_next[e]; total += k_with_next[e]; tb.end[e] = total; }
    tb.loc[nc + 1] = loc_with_next[nc]; tb.k[nc + 1] = k_with_next[nc];
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(shared_positions_kernel, dim3(stream_grid(env, total)), dim3(kStreamThreads), 0, env.stream, tb, nc, count_dev);
    return hipGetLastError();
}

// out[p] = (out[p] + (sel[p] ? stream[p] : 0)) mod 2^b
__global__ __launch_bounds__(kStreamThreads) void sel_accumulate_kernel(uint64_t n, int L, const uint8_t *sel, const uint64_t *stream,
                                                                        uint64_t *out, uint64_t mask_lo, uint64_t mask_hi)
{
    const u128 mask = (static_cast<u128>(mask_hi) << 64) | mask_lo;
    for (uint64_t p = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; p < n;
         p += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        if (!sel[p]) continue;
        if (L == 2) st128(out + 2 * p, (ld128(out + 2 * p) + ld128(stream + 2 * p)) & mask);
        else out[p] = (out[p] + stream[p]) & mask_lo;
    }
}

hipError_t launch_fill(const LaunchEnv &env, uint64_t n, uint64_t lo, uint64_t hi, uint64_t *out_dev)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, n, env.b > 64 ? 2 : 1, lo, hi, out_dev);
    return hipGetLastError();
}

hipError_t launch_scatter(const LaunchEnv &env, uint64_t total, uint64_t k, const uint32_t *loc_dev, const uint64_t *vals_dev,
                          uint64_t *out_dev, bool accumulate, uint64_t sub_lo, uint64_t sub_hi)
{
    if (k == 0) return hipSuccess;
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    hipLaunchKernelGGL(scatter_kernel, dim3(stream_grid(env, k)), dim3(kStreamThreads), 0, env.stream, total, k, env.b > 64 ? 2 : 1,
                       loc_dev, vals_dev, out_dev, accumulate, lo, hi, sub_lo, sub_hi, env.err_flag);
    return hipGetLastError();
}

// One list entry of _static_prepare_decrypt_spar: the whole-vector stream for prefix iter|list_idx
// (one chunk, begin = 0) must already be in stream_dev; selected positions are accumulated.
hipError_t launch_sel_accumulate(const LaunchEnv &env, uint64_t total, const uint8_t *sel_dev, const uint64_t *stream_dev,
                                 uint64_t *out_dev)
{
    if (total == 0) return hipSuccess;
    uint64_t lo, hi;
    masks_of(env.b, &lo, &hi);
    hipLaunchKernelGGL(sel_accumulate_kernel, dim3(stream_grid(env, total)), dim3(kStreamThreads), 0, env.stream, total,
                       env.b > 64 ? 2 : 1, sel_dev, stream_dev, out_dev, lo, hi);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Quantise / batch codec either side of the cipher (streaming, HBM-bound)
// ------------------------------------------------------------------------------------------
// float arithmetic below must round exactly like numpy's: no contraction into FMAs
template <typename T>
__global__ __launch_bounds__(kStreamThreads) void quantize_kernel(uint64_t n, const T *x, T alpha, T scale, T den,
                                                                  const double *u, uint64_t *q)
{
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads)
        q[j] = quantize_one<T>(x[j], alpha, scale, den, u[j]);
}

__global__ __launch_bounds__(kStreamThreads) void unquantize_kernel(uint64_t n, const uint64_t *v, int v_limbs, double ac,
                                                                    double two_a, double den, double *out)
{
#pragma clang fp contract(off)
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        const u128 x = v_limbs == 2 ? ld128(v + 2 * j) : static_cast<u128>(v[j]);
        const double d = u128_to_double(x);
        out[j] = d * two_a / den - ac;
    }
}

// one batch (bs consecutive values, first most significant) per lane
__global__ __launch_bounds__(kStreamThreads) void batch_kernel(uint64_t n, uint64_t nb, const uint64_t *vals, int L, int bs,
                                                               int field_bits, uint64_t *out)
{
    for (uint64_t b = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; b < nb;
         b += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        u128 t = 0;
        for (int i = 0; i < bs; i++) {
            const uint64_t j = b * bs + i;
            t = (field_bits >= 128 ? 0 : t << field_bits) + (j < n ? vals[j] : 0ull);
        }
        if (L == 2) st128(out + 2 * b, t);
        else out[b] = static_cast<uint64_t>(t);
    }
}

__global__ __launch_bounds__(kStreamThreads) void unbatch_kernel(uint64_t nb, const uint64_t *in, int L, int bs, int field_bits,
                                                                 uint64_t *out)
{
    const u128 mk = field_bits >= 128 ? ~static_cast<u128>(0) : ((static_cast<u128>(1) << field_bits) - 1);
    for (uint64_t b = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; b < nb;
         b += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        u128 item = L == 2 ? ld128(in + 2 * b) : static_cast<u128>(in[b]);
        for (int i = 0; i < bs; i++) {
            out[b * bs + (bs - 1 - i)] = static_cast<uint64_t>(item & mk);
            item = field_bits >= 128 ? 0 : item >> field_bits;
        }
    }
}

Codec codec_quantize_front(const void *x_dev, bool is_f64, double alpha, int bits, const double *u_dev)
{
    Codec c{};
    c.x = x_dev; c.u = u_dev; c.alpha = alpha; c.scale = static_cast<double>((1ull << bits) - 1); c.den = 2 * alpha; c.x_is_f64 = is_f64 ? 1 : 0;
    return c;
}

void codec_unquantize_back(Codec *c, double alpha, int bits, int num_clients, double *out_dev)
{
    c->fout = out_dev;
    c->ac = alpha * static_cast<double>(num_clients);
    c->two_a = 2 * c->ac;
    c->uden = static_cast<double>(((1ull << bits) - 1) * static_cast<uint64_t>(num_clients));
}

CodecLayer codec_layer_front(uint64_t start, const void *x_dev, bool is_f64, double alpha, int bits)
{
    const Codec c = codec_quantize_front(x_dev, is_f64, alpha, bits, nullptr);
    return CodecLayer{start, x_dev, c.alpha, c.scale, c.den, c.x_is_f64, 0};
}

CodecLayer codec_layer_back(uint64_t start, double alpha, int bits, int num_clients)
{
    Codec c{};
    codec_unquantize_back(&c, alpha, bits, num_clients, nullptr);
    return CodecLayer{start, nullptr, c.ac, c.two_a, c.uden, 0, 0};
}

// x <- x + shift (normalize: shift = -mean, unnormalize: shift = +mean; a - b and a + (-b) round identically)
template <typename T, bool WIDE>
__global__ __launch_bounds__(kStreamThreads) void shift_kernel(uint64_t n, T *x, double shift)
{
#pragma clang fp contract(off)
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads)
        x[j] = WIDE ? static_cast<T>(static_cast<double>(x[j]) + shift) : x[j] + static_cast<T>(shift);
}

hipError_t launch_shift(const LaunchEnv &env, uint64_t n, void *x_dev, bool is_f64, double shift, bool wide)
{
    if (n == 0) return hipSuccess;
    const dim3 g(stream_grid(env, n)), t(kStreamThreads);
    if (is_f64) hipLaunchKernelGGL((shift_kernel<double, false>), g, t, 0, env.stream, n, static_cast<double *>(x_dev), shift);
    else if (wide) hipLaunchKernelGGL((shift_kernel<float, true>), g, t, 0, env.stream, n, static_cast<float *>(x_dev), shift);
    else hipLaunchKernelGGL((shift_kernel<float, false>), g, t, 0, env.stream, n, static_cast<float *>(x_dev), shift);
    return hipGetLastError();
}

// part[block] = sum over the block's elements of (x - center)^POW in float64: per-thread partial, wave shuffle tree, one LDS hop
template <typename T, int POW>
__global__ __launch_bounds__(kStreamThreads) void moment_kernel(uint64_t n, const T *x, double center, double *part)
{
    __shared__ double ws[kStreamThreads / 64];
    double acc = 0;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        const double d = static_cast<double>(x[j]) - center;
        acc += POW == 1 ? d : d * d;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < kStreamThreads / 64; w++) t += ws[w];
        part[blockIdx.x] = t;
    }
}

int moments_grid(const LaunchEnv &env, uint64_t n) { return stream_grid(env, n); }

hipError_t launch_moment(const LaunchEnv &env, uint64_t n, const void *x_dev, bool is_f64, double center, int pow, double *part_dev)
{
    if (n == 0) return hipSuccess;
    const dim3 g(stream_grid(env, n)), t(kStreamThreads);
    if (is_f64 && pow == 1) hipLaunchKernelGGL((moment_kernel<double, 1>), g, t, 0, env.stream, n, static_cast<const double *>(x_dev), center, part_dev);
    else if (is_f64) hipLaunchKernelGGL((moment_kernel<double, 2>), g, t, 0, env.stream, n, static_cast<const double *>(x_dev), center, part_dev);
    else if (pow == 1) hipLaunchKernelGGL((moment_kernel<float, 1>), g, t, 0, env.stream, n, static_cast<const float *>(x_dev), center, part_dev);
    else hipLaunchKernelGGL((moment_kernel<float, 2>), g, t, 0, env.stream, n, static_cast<const float *>(x_dev), center, part_dev);
    return hipGetLastError();
}

hipError_t launch_quantize(const LaunchEnv &env, uint64_t n, const void *x_dev, bool is_f64, double alpha, int bits,
                           const double *u_dev, uint64_t *q_dev)
{
    if (n == 0) return hipSuccess;
    const double scale = static_cast<double>((1ull << bits) - 1);
    if (is_f64)
        hipLaunchKernelGGL(quantize_kernel<double>, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, n,
                           static_cast<const double *>(x_dev), alpha, scale, 2 * alpha, u_dev, q_dev);
    else
        hipLaunchKernelGGL(quantize_kernel<float>, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, n,
                           static_cast<const float *>(x_dev), static_cast<float>(alpha), static_cast<float>(scale),
                           static_cast<float>(2 * alpha), u_dev, q_dev);
    return hipGetLastError();
}

hipError_t launch_unquantize(const LaunchEnv &env, uint64_t n, const uint64_t *v_dev, int v_limbs, double alpha, int bits,
                             int num_clients, double *out_dev)
{
    if (n == 0) return hipSuccess;
    const double ac = alpha * static_cast<double>(num_clients);
    const double den = static_cast<double>(((1ull << bits) - 1) * static_cast<uint64_t>(num_clients));
    hipLaunchKernelGGL(unquantize_kernel, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, n, v_dev, v_limbs, ac,
                       2 * ac, den, out_dev);
    return hipGetLastError();
}

hipError_t launch_batch(const LaunchEnv &env, uint64_t n, const uint64_t *vals_dev, int field_bits, uint64_t *out_dev)
{
    const int bs = env.b / field_bits;
    const uint64_t nb = (n + bs - 1) / bs;
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(batch_kernel, dim3(stream_grid(env, nb)), dim3(kStreamThreads), 0, env.stream, n, nb, vals_dev,
                       env.b > 64 ? 2 : 1, bs, field_bits, out_dev);
    return hipGetLastError();
}

hipError_t launch_unbatch(const LaunchEnv &env, uint64_t nb, const uint64_t *in_dev, int field_bits, uint64_t *out_dev)
{
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(unbatch_kernel, dim3(stream_grid(env, nb)), dim3(kStreamThreads), 0, env.stream, nb, in_dev,
                       env.b > 64 ? 2 : 1, env.b / field_bits, field_bits, out_dev);
    return hipGetLastError();
}

}  // namespace flashe
