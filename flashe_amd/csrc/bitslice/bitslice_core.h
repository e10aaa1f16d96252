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

// ------------------------------------------------------------------------------------------
// Packed variant: 64 plane registers per lane, register 8*Bp + k = bit k of state byte Bp (low 16 bits)
// and of state byte Bp + 8 (high 16 bits) for 16 blocks.  Same work per instruction, half the
// registers, so 3-4 waves fit per SIMD.
// ------------------------------------------------------------------------------------------
// Planes of 16 PRF input blocks per lane.  With NSTREAM == 2 blocks 0..7 carry prefix idx_a and blocks
// 8..15 prefix idx_b over the same 8 counters c0 + 64 * q.
template <int NSTREAM>
__device__ __forceinline__ void load_planes_p(uint32_t (&s)[64], uint32_t iter, uint32_t idx_a, uint32_t idx_b, uint64_t c0)
{
    constexpr int EPL = 16 / NSTREAM;
    uint32_t t[32];
#pragma unroll
    for (int p = 0; p < 16; p++) {
        const uint64_t c = c0 + static_cast<uint64_t>(p & (EPL - 1)) * 64;
        t[p] = static_cast<uint32_t>(c);
        t[16 + p] = static_cast<uint32_t>(c >> 32);
    }
    transpose32(t);     // t[i]: low half = plane of counter bit i, high half = plane of counter bit 32 + i
    constexpr uint32_t mask_a = NSTREAM == 2 ? 0x00ffu : 0xffffu, mask_b = NSTREAM == 2 ? 0xff00u : 0u;
#pragma unroll
    for (int B = 0; B < 8; B++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            // low half: state byte B (bytes 0-3 = iter, 4-7 = idx, big-endian)
            uint32_t low;
            if (B < 4) low = (iter >> ((3 - B) * 8 + k)) & 1u ? 0xffffu : 0u;
            else low = ((idx_a >> ((7 - B) * 8 + k)) & 1u ? mask_a : 0u) | ((idx_b >> ((7 - B) * 8 + k)) & 1u ? mask_b : 0u);
            // high half: state byte B + 8 = counter byte, i.e. counter bit (7 - B) * 8 + k
            const int cb = (7 - B) * 8 + k;
            s[8 * B + k] = cb < 32 ? ((t[cb] << 16) | low) : ((t[cb - 32] & 0xffff0000u) | low);
        }
    }
}

// AES-256 on the packed planes in place.  rkp = 15 x 64 packed key planes.
__device__ __forceinline__ void encrypt_planes_p(uint32_t (&s)[64], const uint32_t *__restrict__ rkp)
{
    uint32_t o[64];
#pragma unroll
    for (int i = 0; i < 64; i++) s[i] ^= rkp[i];
#pragma unroll 1
    for (int r = 1; r < 13; r += 2) {
        round_main_p(s, rkp + 64 * r, o);
        round_main_p(o, rkp + 64 * (r + 1), s);
    }
    round_main_p(s, rkp + 64 * 13, o);
    round_final_p(o, rkp + 64 * 14, s);
}

// S[p] = AES output of block p (p = 0..15) as a big-endian 128-bit integer.
__device__ __forceinline__ void planes_to_blocks_p(const uint32_t (&s)[64], u128 (&S)[16])
{
    uint32_t w1[32], w2[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
        w1[i] = s[8 * (3 - i / 8) + (i & 7)];     // low halves: bytes 0..3 (word 3), high halves: bytes 8..11 (word 1)
        w2[i] = s[8 * (7 - i / 8) + (i & 7)];     // low halves: bytes 4..7 (word 2), high halves: bytes 12..15 (word 0)
    }
    transpose32(w1);
    transpose32(w2);
#pragma unroll
    for (int p = 0; p < 16; p++)
        S[p] = (static_cast<u128>(w1[p]) << 96) | (static_cast<u128>(w2[p]) << 64) |
               (static_cast<u128>(w1[p + 16]) << 32) | static_cast<u128>(w2[p + 16]);
}

}  // namespace bs
}  // namespace flashe
