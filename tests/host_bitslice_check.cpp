// CPU check of the GENERATED bit-sliced AES code (flashe_amd/csrc/aes_bitslice_gen.h) and its
// pipeline (bitslice_core.h): the device text is compiled with g++ against software models of
// v_bitop3_b32 and v_perm_b32 and compared with libcrypto's AES-256 on PRF input blocks.
// Build + run: see tests/test_bitslice_host.py.
#include <openssl/evp.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

static inline uint32_t emu_bitop3(uint32_t a, uint32_t b, uint32_t c, int tt)
{
    uint32_t r = 0;
    for (int bit = 0; bit < 32; bit++) {
        const int idx = (((a >> bit) & 1) << 2) | (((b >> bit) & 1) << 1) | ((c >> bit) & 1);   // src0 is the high index bit
        r |= static_cast<uint32_t>((tt >> idx) & 1) << bit;
    }
    return r;
}
static inline uint32_t emu_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
    const uint64_t data = (static_cast<uint64_t>(s0) << 32) | s1;   // selector 0-3 -> s1, 4-7 -> s0
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t sb = (sel >> (8 * i)) & 0xff;
        uint32_t byte = sb <= 7 ? static_cast<uint32_t>((data >> (8 * sb)) & 0xff) : (sb == 12 ? 0u : 0xffu);
        r |= byte << (8 * i);
    }
    return r;
}
#define __device__
#define __forceinline__ inline
#define __builtin_amdgcn_bitop3_b32(a, b, c, tt) emu_bitop3((a), (b), (c), (tt))
#define __builtin_amdgcn_perm(a, b, s) emu_perm((a), (b), (s))
#define __builtin_amdgcn_sched_barrier(x)
#define __builtin_amdgcn_alignbit(a, b, s) static_cast<uint32_t>(((static_cast<uint64_t>(a) << 32) | (b)) >> (s))
#include "aes_bitslice_gen.h"
#include "bitslice_core.h"

using flashe::bs::u128;

static void expand_key_words(const uint8_t key[32], uint32_t *w)
{
    // FIPS-197 key expansion via libcrypto is not exposed; do it by hand with an S-box from AES itself
    static uint8_t sbox[256];
    {   // S-box by encrypting with libcrypto is awkward; derive it from its definition
        auto mul = [](uint8_t a, uint8_t b) { uint8_t p = 0; for (int i = 0; i < 8; i++) { if (b & 1) p ^= a; uint8_t h = a & 0x80; a <<= 1; if (h) a ^= 0x1b; b >>= 1; } return p; };
        for (int x = 0; x < 256; x++) {
            uint8_t inv = 0;
            if (x) for (int y = 1; y < 256; y++) if (mul(x, y) == 1) { inv = y; break; }
            uint8_t s = inv, r = inv;
            for (int k = 0; k < 4; k++) { r = (r << 1) | (r >> 7); s ^= r; }
            sbox[x] = s ^ 0x63;
        }
    }
    for (int i = 0; i < 8; i++) w[i] = (key[4 * i] << 24) | (key[4 * i + 1] << 16) | (key[4 * i + 2] << 8) | key[4 * i + 3];
    uint32_t rcon = 0x01000000u;
    auto sub = [&](uint32_t v) { return (uint32_t(sbox[v >> 24]) << 24) | (uint32_t(sbox[(v >> 16) & 255]) << 16) | (uint32_t(sbox[(v >> 8) & 255]) << 8) | sbox[v & 255]; };
    for (int i = 8; i < 60; i++) {
        uint32_t t = w[i - 1];
        if (i % 8 == 0) { t = sub((t << 8) | (t >> 24)) ^ rcon; rcon = (rcon << 1) ^ ((rcon & 0x80000000u) ? 0x1b000000u : 0); }
        else if (i % 8 == 4) t = sub(t);
        w[i] = w[i - 8] ^ t;
    }
}

template <int NSTREAM>
static int check(EVP_CIPHER_CTX *ctx, const uint32_t *rkp, uint32_t iter, uint32_t idx_a, uint32_t idx_b, uint64_t t_first)
{
    constexpr int EPL = 32 / NSTREAM;
    int bad = 0;
    for (int lane = 0; lane < 64; lane += 21) {
        uint32_t s[128];
        flashe::bs::load_planes<NSTREAM>(s, iter, idx_a, idx_b, t_first + lane, t_first, t_first + 64ull * EPL - 1);
        flashe::bs::encrypt_planes(s, rkp);
        u128 S[32];
        flashe::bs::planes_to_blocks(s, S);
        for (int q = 0; q < 32; q++) {
            const uint64_t ctr = t_first + lane + 64ull * (q & (EPL - 1));
            const uint32_t idx = (NSTREAM == 2 && q >= 16) ? idx_b : idx_a;
            uint8_t in[16], out[32];
            for (int i = 0; i < 4; i++) { in[i] = iter >> (24 - 8 * i); in[4 + i] = idx >> (24 - 8 * i); }
            for (int i = 0; i < 8; i++) in[8 + i] = ctr >> (56 - 8 * i);
            int len = 0;
            EVP_EncryptUpdate(ctx, out, &len, in, 16);
            u128 want = 0;
            for (int i = 0; i < 16; i++) want = (want << 8) | out[i];
            if (want != S[q]) bad++;
        }
    }
    return bad;
}

template <int NSTREAM>
static int check_packed(EVP_CIPHER_CTX *ctx, const uint32_t *rkw, uint32_t iter, uint32_t idx_a, uint32_t idx_b, uint64_t t_first)
{
    constexpr int EPL = 16 / NSTREAM;
    static uint32_t rkp[15 * 64];
    for (int r = 0; r < 15; r++)
        for (int B = 0; B < 8; B++)
            for (int k = 0; k < 8; k++) {
                auto bit = [&](int byte) { return (rkw[4 * r + byte / 4] >> (24 - 8 * (byte % 4) + k)) & 1u; };
                rkp[64 * r + 8 * B + k] = (bit(B) ? 0xffffu : 0u) | (bit(B + 8) ? 0xffff0000u : 0u);
            }
    int bad = 0;
    for (int lane = 0; lane < 64; lane += 21) {
        uint32_t s[64];
        flashe::bs::load_planes_p<NSTREAM>(s, iter, idx_a, idx_b, t_first + lane);
        flashe::bs::encrypt_planes_p(s, rkp);
        u128 S[16];
        flashe::bs::planes_to_blocks_p(s, S);
        for (int q = 0; q < 16; q++) {
            const uint64_t ctr = t_first + lane + 64ull * (q & (EPL - 1));
            const uint32_t idx = (NSTREAM == 2 && q >= 8) ? idx_b : idx_a;
            uint8_t in[16], out[32];
            for (int i = 0; i < 4; i++) { in[i] = iter >> (24 - 8 * i); in[4 + i] = idx >> (24 - 8 * i); }
            for (int i = 0; i < 8; i++) in[8 + i] = ctr >> (56 - 8 * i);
            int len = 0;
            EVP_EncryptUpdate(ctx, out, &len, in, 16);
            u128 want = 0;
            for (int i = 0; i < 16; i++) want = (want << 8) | out[i];
            if (want != S[q]) bad++;
        }
    }
    return bad;
}

int main()
{
    uint8_t key[32];
    for (int i = 0; i < 32; i++) key[i] = i;
    static uint32_t rkp[60];
    expand_key_words(key, rkp);
    EVP_CIPHER_CTX *ctx = EVP_CIPHER_CTX_new();
    EVP_EncryptInit_ex(ctx, EVP_aes_256_ecb(), nullptr, key, nullptr);
    EVP_CIPHER_CTX_set_padding(ctx, 0);
    // transpose32 self-check
    uint32_t a[32], b[32];
    for (int i = 0; i < 32; i++) a[i] = b[i] = 0x9e3779b9u * (i + 1) ^ (i << 13);
    flashe::bs::transpose32(b);
    int bad = 0;
    for (int i = 0; i < 32; i++) for (int p = 0; p < 32; p++) if (((b[p] >> i) & 1) != ((a[i] >> p) & 1)) bad++;
    if (bad) { printf("transpose32 FAILED (%d)\n", bad); return 1; }
    bad += check<2>(ctx, rkp, 77, 5, 6, 0);
    bad += check<2>(ctx, rkp, 0xffffffffu, 0xfffffffeu, 0xffffffffu, 1537);
    bad += check<2>(ctx, rkp, 3, 0, 1, (1ull << 32) - 500);          // crosses the 2^32 counter boundary
    bad += check<1>(ctx, rkp, 1, 9, 0, 123456789012ull);
    bad += check<1>(ctx, rkp, 0, 0, 0, 0);
    bad += check_packed<2>(ctx, rkp, 77, 5, 6, 0);
    bad += check_packed<2>(ctx, rkp, 0xffffffffu, 0xfffffffeu, 0xffffffffu, 1537);
    bad += check_packed<2>(ctx, rkp, 3, 0, 1, (1ull << 32) - 300);
    bad += check_packed<1>(ctx, rkp, 1, 9, 0, 123456789012ull);
    printf(bad ? "bitslice host check FAILED: %d mismatching blocks\n" : "bitslice host check OK\n", bad);
    return bad ? 1 : 0;
}
