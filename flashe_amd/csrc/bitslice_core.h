// Bit-sliced AES-256 pipeline around the generated round functions (aes_bitslice_gen.h): plane
// construction for PRF input blocks, the 14 rounds, and planes -> per-block 128-bit integers.
// Device code (included by kernels.hip); tests/host_bitslice_check.cpp compiles the same text with
// g++ against software models of v_bitop3_b32 / v_perm_b32.
#pragma once
#include <stdint.h>

namespace flashe {
namespace bs {

typedef unsigned __int128 u128;

// One sub-byte butterfly stage of the transpose: swap the J x J off-diagonal bit blocks.
template <int J>
__device__ __forceinline__ void transpose_stage(uint32_t (&a)[32])
{
    constexpr uint32_t m = J == 4 ? 0x0f0f0f0fu : (J == 2 ? 0x33333333u : 0x55555555u);
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if (k & J) continue;
        const uint32_t x = a[k], y = a[k + J];
        a[k] = (x & m) | ((y << J) & ~m);
        a[k + J] = ((x >> J) & m) | (y & ~m);
    }
}

// In-place 32x32 bit-matrix transpose: afterwards bit i of a[p] = former bit p of a[i].
__device__ __forceinline__ void transpose32(uint32_t (&a)[32])
{
#pragma unroll
    for (int k = 0; k < 16; k++) {           // j = 16
        const uint32_t x = a[k], y = a[k + 16];
        a[k] = __builtin_amdgcn_perm(y, x, 0x05040100u);        // (x & 0xffff) | (y << 16)
        a[k + 16] = __builtin_amdgcn_perm(y, x, 0x07060302u);   // (x >> 16) | (y & 0xffff0000)
    }
#pragma unroll
    for (int k0 = 0; k0 < 32; k0 += 16)      // j = 8
#pragma unroll
        for (int k1 = 0; k1 < 8; k1++) {
            const int k = k0 + k1;
            const uint32_t x = a[k], y = a[k + 8];
            a[k] = __builtin_amdgcn_perm(y, x, 0x06020400u);     // bytes: x0, y0, x2, y2
            a[k + 8] = __builtin_amdgcn_perm(y, x, 0x07030501u); // bytes: x1, y1, x3, y3
        }
    transpose_stage<4>(a);
    transpose_stage<2>(a);
    transpose_stage<1>(a);
}

// Planes of 32 PRF input blocks per lane: block = iter(4B) | idx(4B) | counter(8B), big-endian.
// c0 = counter of slot 0 of this lane; slot q uses c0 + 64 * (q mod EPL).  With NSTREAM == 2 blocks
// 0..15 carry prefix idx_a and blocks 16..31 prefix idx_b over the SAME 16 counters.
// [t_first, t_last] = counter range of the whole wave-pass (decides whether bits 32..63 are uniform).
template <int NSTREAM>
__device__ __forceinline__ void load_planes(uint32_t (&s)[128], uint32_t iter, uint32_t idx_a, uint32_t idx_b,
                                            uint64_t c0, uint64_t t_first, uint64_t t_last)
{
    constexpr int EPL = 32 / NSTREAM;
    uint32_t lo[32], hi[32];
#pragma unroll
    for (int q = 0; q < 32; q++) {
        const uint64_t c = c0 + static_cast<uint64_t>(q & (EPL - 1)) * 64;
        lo[q] = static_cast<uint32_t>(c);
        hi[q] = static_cast<uint32_t>(c >> 32);
    }
    transpose32(lo);                                  // lo[i] = plane of counter bit i
    if ((t_first >> 32) == (t_last >> 32)) {          // wave-uniform: the usual case
        const uint32_t h = static_cast<uint32_t>(t_first >> 32);
#pragma unroll
        for (int i = 0; i < 32; i++) hi[i] = (h >> i) & 1u ? 0xffffffffu : 0u;
    } else {
        transpose32(hi);
    }
#pragma unroll
    for (int i = 0; i < 32; i++) {
        // counter bit i lives in state byte 15 - i/8, bit i%8; counter bit 32 + i in byte 11 - i/8
        s[8 * (15 - i / 8) + (i & 7)] = lo[i];
        s[8 * (11 - i / 8) + (i & 7)] = hi[i];
        // iter bit i: byte 3 - i/8; idx bit i: byte 7 - i/8
        s[8 * (3 - i / 8) + (i & 7)] = (iter >> i) & 1u ? 0xffffffffu : 0u;
        const uint32_t pa = (idx_a >> i) & 1u ? (NSTREAM == 2 ? 0x0000ffffu : 0xffffffffu) : 0u;
        const uint32_t pb = (NSTREAM == 2 && ((idx_b >> i) & 1u)) ? 0xffff0000u : 0u;
        s[8 * (7 - i / 8) + (i & 7)] = pa | pb;
    }
}

// AES-256 on the planes in place.  rk = the 60 expanded key words (big-endian columns); a round's key
// planes are expanded on the fly from its 4 words (scalar bit-field extracts, no memory traffic).
__device__ __forceinline__ void encrypt_planes(uint32_t (&s)[128], const uint32_t *__restrict__ rk)
{
    uint32_t o[128];
    {
        const uint32_t kw[4] = {rk[0], rk[1], rk[2], rk[3]};
#pragma unroll
        for (int i = 0; i < 128; i++) {
            const int B = i / 8, k = i % 8;
            s[i] ^= 0u - ((kw[B / 4] >> (24 - 8 * (B % 4) + k)) & 1u);
        }
    }
#pragma unroll 1
    for (int r = 1; r < 13; r += 2) {
        const uint32_t k0[4] = {rk[4 * r], rk[4 * r + 1], rk[4 * r + 2], rk[4 * r + 3]};
        round_main(s, k0, o);
        const uint32_t k1[4] = {rk[4 * r + 4], rk[4 * r + 5], rk[4 * r + 6], rk[4 * r + 7]};
        round_main(o, k1, s);
    }
    const uint32_t k13[4] = {rk[52], rk[53], rk[54], rk[55]};
    round_main(s, k13, o);
    const uint32_t k14[4] = {rk[56], rk[57], rk[58], rk[59]};
    round_final(o, k14, s);
}

// S[q] = AES output of block q as a big-endian 128-bit integer: 32-bit word w (0 = least significant)
// bit i = bit i%8 of state byte 15 - 4w - i/8.
__device__ __forceinline__ void planes_to_blocks(const uint32_t (&s)[128], u128 (&S)[32])
{
#pragma unroll
    for (int q = 0; q < 32; q++) S[q] = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        uint32_t t[32];
#pragma unroll
        for (int i = 0; i < 32; i++) t[i] = s[8 * (15 - 4 * w - i / 8) + (i & 7)];
        transpose32(t);
#pragma unroll
        for (int q = 0; q < 32; q++) S[q] |= static_cast<u128>(t[q]) << (32 * w);
    }
}

}  // namespace bs
}  // namespace flashe
