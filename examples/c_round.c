/* One FLASHE round through the C ABI alone -- no Python: C clients encrypt their vectors (double mask), the arbiter adds the
 * ciphertexts mod 2^b, a client decrypts the aggregate, and the result must be the plain sum.  What a non-Python host (the cgo /
 * JNI side of a FATE deployment) would do with include/flashe.h; host vectors in, host vectors out.
 *
 *   gcc -O2 -std=c11 -Iinclude examples/c_round.c -Lflashe_amd -lflashe_hip -Wl,-rpath,$PWD/flashe_amd -o c_round && ./c_round
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>

#include "flashe.h"

#define CHECK(call)                                                                            \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != FLASHE_OK) {                                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, flashe_last_error(ctx));             \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

int main(void)
{
    enum { C = 4, N_JOBS = 16 };
    const uint64_t n = 100003;
    const uint32_t iter = 7;
    uint8_t key[32];
    for (int i = 0; i < 32; i++) key[i] = (uint8_t)i;
    flashe_ctx *ctx = NULL;
    if (flashe_ctx_create(&ctx, key, 128, 0, NULL) != FLASHE_OK) {
        fprintf(stderr, "flashe_ctx_create: %s\n", flashe_last_error(NULL));
        return 1;
    }
    CHECK(flashe_selftest(ctx));

    uint64_t *pt[C], *ct[C];
    uint64_t *sum = calloc(n, sizeof *sum), *agg = malloc(n * 16), *dec = malloc(n * 16);
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (int c = 0; c < C; c++) {
        pt[c] = malloc(n * sizeof **pt);
        ct[c] = malloc(n * 16);                                  /* 128-bit ciphertexts: two little-endian limbs per element */
        for (uint64_t j = 0; j < n; j++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;             /* xorshift: any 56-bit plaintexts */
            pt[c][j] = x >> 8;
            sum[j] += pt[c][j];
        }
        /* client c: FlasheCipher.encrypt, prefixes iter|c and iter|c+1 */
        CHECK(flashe_encrypt(ctx, iter, (uint32_t)c, FLASHE_SCHEME_DOUBLE, n, N_JOBS, pt[c], 1, ct[c]));
    }
    /* arbiter: reduce(lambda x, y: (x + y) % 2^b) */
    const uint64_t *ops[C];
    for (int c = 0; c < C; c++) ops[c] = ct[c];
    CHECK(flashe_aggregate_elem(ctx, C, ops, n, agg));
    /* any client: every client 0 .. C-1 uploaded, so the masks telescope to + term(C) - term(0) */
    uint32_t raw[C], add[C], minus[C];
    int runs = 0;
    for (int c = 0; c < C; c++) raw[c] = (uint32_t)c;
    CHECK(flashe_telescope(raw, C, add, minus, &runs));
    CHECK(flashe_decrypt(ctx, iter, add, runs, minus, runs, n, N_JOBS, agg, dec));

    uint64_t bad = 0;
    for (uint64_t j = 0; j < n; j++) bad += dec[2 * j] != sum[j] || dec[2 * j + 1] != 0;
    int same_as_plain = 0;
    for (uint64_t j = 0; j < n; j++) same_as_plain += ct[0][2 * j] == pt[0][j] && ct[0][2 * j + 1] == 0;
    printf("C_ROUND %s: n=%" PRIu64 " clients=%d runs=%d mismatches=%" PRIu64 " ciphertext_words_equal_to_plaintext=%d\n",
           bad == 0 && same_as_plain == 0 ? "OK" : "FAILED", n, C, runs, bad, same_as_plain);
    flashe_ctx_destroy(ctx);
    if (bad || same_as_plain) return 1;

    /* The same round at the width the reference's own jobs ship (int_bits = 20), device resident, in the compact layout: the vectors
     * are uint32 arrays in HBM, ONE launch encrypts all clients, ONE launch reduces and decrypts. */
    enum { B20 = 20 };
    if (flashe_ctx_create(&ctx, key, B20, 0, NULL) != FLASHE_OK) {
        fprintf(stderr, "flashe_ctx_create: %s\n", flashe_last_error(NULL));
        return 1;
    }
    uint32_t *h = malloc(n * sizeof *h), *want = calloc(n, sizeof *want), idx[C];
    const uint32_t *dpt[C];
    uint32_t *dct[C], *dout = NULL;
    for (int c = 0; c < C; c++) {
        void *p = NULL, *q = NULL;
        CHECK(flashe_dev_alloc(ctx, n * 4, &p));
        CHECK(flashe_dev_alloc(ctx, n * 4, &q));
        for (uint64_t j = 0; j < n; j++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            h[j] = (uint32_t)(x >> 48);                          /* 16-bit quantised values */
            want[j] = (want[j] + h[j]) & ((1u << B20) - 1);
        }
        CHECK(flashe_memcpy_h2d(ctx, p, h, n * 4));
        dpt[c] = p; dct[c] = q; idx[c] = (uint32_t)c;
    }
    { void *p = NULL; CHECK(flashe_dev_alloc(ctx, n * 4, &p)); dout = p; }
    CHECK(flashe_encrypt_batch_u32_dev(ctx, iter, FLASHE_SCHEME_DOUBLE, n, N_JOBS, C, idx, dpt, dct));
    const uint32_t addp[1] = {C}, minusp[1] = {0};
    CHECK(flashe_aggregate_decrypt_u32_dev(ctx, iter, addp, 1, minusp, 1, n, N_JOBS, 0, n, C, (const uint32_t *const *)dct, NULL, dout, 4));
    CHECK(flashe_memcpy_d2h(ctx, h, dout, n * 4));
    uint64_t bad32 = 0;
    for (uint64_t j = 0; j < n; j++) bad32 += h[j] != want[j];
    printf("C_ROUND_U32 %s: int_bits=%d n=%" PRIu64 " clients=%d mismatches=%" PRIu64 "\n", bad32 == 0 ? "OK" : "FAILED", B20, n, C, bad32);
    for (int c = 0; c < C; c++) { CHECK(flashe_dev_free(ctx, (void *)dpt[c])); CHECK(flashe_dev_free(ctx, dct[c])); }
    CHECK(flashe_dev_free(ctx, dout));
    flashe_ctx_destroy(ctx);
    return bad32 == 0 ? 0 : 1;
}
