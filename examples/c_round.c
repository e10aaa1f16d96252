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
    if (bad32) return 1;

    /* BASELINE config 3's shape from plain C: a LeNet-sized gradient (61,706 parameters), 100 clients, double mask with MASK
     * PRECOMPUTE (FlasheCipher.prepare_encrypt / prepare_decrypt, jzf_flashe.py:599-666).  The masks are computed in "idle time" and
     * stay inside the ctx; the online encrypt and decrypt then run no AES at all and consume the cache -- the consume-once state machine
     * of the reference (:483-486, :573-580) is the library's, not the caller's.  One ctx plays the parties in turn here; a deployment
     * holds one ctx per party.  Second half: client 37 drops out, so the decrypt needs two prefixes the precompute does not cover
     * (set_idx_list skips {C} / {0}, :372-386): they are computed online and merged in by the same call. */
    enum { C3 = 100, DROPPED = 37 };
    const uint64_t n3 = 61706;
    const uint32_t it3 = 11;
    if (flashe_ctx_create(&ctx, key, 128, 0, NULL) != FLASHE_OK) {
        fprintf(stderr, "flashe_ctx_create: %s\n", flashe_last_error(NULL));
        return 1;
    }
    uint64_t *sum_all = calloc(n3, sizeof *sum_all), *sum_drop = calloc(n3, sizeof *sum_drop);
    uint64_t *pt3[C3], *ct3[C3], *agg3 = malloc(n3 * 16), *dec3 = malloc(n3 * 16);
    const uint64_t *ops3[C3];
    for (int c = 0; c < C3; c++) {
        pt3[c] = malloc(n3 * sizeof **pt3);
        ct3[c] = malloc(n3 * 16);
        for (uint64_t j = 0; j < n3; j++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            pt3[c][j] = x >> 10;                                 /* 54-bit plaintexts: 100 of them sum below 2^64 */
            sum_all[j] += pt3[c][j];
            if (c != DROPPED) sum_drop[j] += pt3[c][j];
        }
        /* idle time of round it3 - 1: the masks of round it3 (the caller passes iter + 1) */
        CHECK(flashe_prepare_encrypt(ctx, it3, (uint32_t)c, FLASHE_SCHEME_DOUBLE, n3, N_JOBS));
        /* online: ct = pt + add - minus, no AES; the cache is consumed */
        CHECK(flashe_encrypt_prepared(ctx, n3, pt3[c], 1, ct3[c]));
        if (flashe_prepared_query(ctx, FLASHE_PREPARED_ENCRYPT, NULL, NULL, NULL) != 0) { fprintf(stderr, "cache not consumed\n"); return 1; }
        ops3[c] = ct3[c];
    }
    /* a consumed cache refuses a second encrypt; and the prepared encrypt equals the online one */
    uint64_t *again = malloc(n3 * 16);
    if (flashe_encrypt_prepared(ctx, n3, pt3[0], 1, again) != FLASHE_EINVAL) { fprintf(stderr, "a consumed cache must refuse a second encrypt\n"); return 1; }
    CHECK(flashe_encrypt(ctx, it3, 0, FLASHE_SCHEME_DOUBLE, n3, N_JOBS, pt3[0], 1, again));
    uint64_t differ = 0;
    for (uint64_t j = 0; j < 2 * n3; j++) differ += again[j] != ct3[0][j];
    /* nobody dropped: everything the decrypt needs was precomputed */
    CHECK(flashe_aggregate_elem(ctx, C3, ops3, n3, agg3));
    CHECK(flashe_prepare_decrypt(ctx, it3, C3, n3, N_JOBS));
    CHECK(flashe_decrypt_prepared(ctx, it3, NULL, 0, NULL, 0, n3, N_JOBS, agg3, dec3));
    uint64_t bad3 = 0;
    for (uint64_t j = 0; j < n3; j++) bad3 += dec3[2 * j] != sum_all[j] || dec3[2 * j + 1] != 0;
    /* client DROPPED missing: uploaded = {0 .. C3-1} \ {DROPPED} telescopes to +term(DROPPED) +term(C3) -term(0) -term(DROPPED + 1);
     * {C3} / {0} are in the cache, the other two go online */
    int m = 0;
    for (int c = 0; c < C3; c++) if (c != DROPPED) ops3[m++] = ct3[c];
    CHECK(flashe_aggregate_elem(ctx, m, ops3, n3, agg3));
    uint32_t raw3[C3], add3[C3], minus3[C3], xa[C3], xm[C3];
    int runs3 = 0, na = 0, nm = 0;
    m = 0;
    for (int c = 0; c < C3; c++) if (c != DROPPED) raw3[m++] = (uint32_t)c;
    CHECK(flashe_telescope(raw3, m, add3, minus3, &runs3));
    for (int r = 0; r < runs3; r++) { if (add3[r] != C3) xa[na++] = add3[r]; if (minus3[r] != 0) xm[nm++] = minus3[r]; }
    CHECK(flashe_prepare_decrypt(ctx, it3, C3, n3, N_JOBS));
    CHECK(flashe_decrypt_prepared(ctx, it3, xa, na, xm, nm, n3, N_JOBS, agg3, dec3));
    uint64_t bad3d = 0;
    for (uint64_t j = 0; j < n3; j++) bad3d += dec3[2 * j] != sum_drop[j] || dec3[2 * j + 1] != 0;
    printf("C_ROUND_PRECOMPUTE %s: n=%" PRIu64 " clients=%d mismatches=%" PRIu64 " dropout_runs=%d extra_prefixes=%d+%d dropout_mismatches=%" PRIu64
           " prepared_vs_online_differences=%" PRIu64 "\n",
           bad3 == 0 && bad3d == 0 && differ == 0 ? "OK" : "FAILED", n3, C3, bad3, runs3, na, nm, bad3d, differ);
    flashe_ctx_destroy(ctx);
    if (!(bad3 == 0 && bad3d == 0 && differ == 0)) return 1;

    /* BASELINE config 5's path from plain C, device resident: CS clients upload their top-k values (sorted positions + compact values,
     * what Client.sparsify leaves), single mask over the COMPACT positions.  One call encrypts all of them and writes the sum of their
     * expanded uploads (Arbiter.expand_to_dense + reduce: the plain `zero` everywhere, value - zero at the locations); the decrypt
     * rebuilds the dense minus-mask from the location lists and subtracts it in the same pass.  The span bounds of the round's lists are
     * computed once and shared by both. */
    enum { CS = 6 };
    const uint64_t total = 300007, ks = 3000, zero = 1ull << 31;
    const uint32_t its = 5;
    if (flashe_ctx_create(&ctx, key, 128, 0, NULL) != FLASHE_OK) {
        fprintf(stderr, "flashe_ctx_create: %s\n", flashe_last_error(NULL));
        return 1;
    }
    uint64_t *wants = malloc(total * sizeof *wants), *got = malloc(total * 16), *hv = malloc(ks * sizeof *hv);
    uint32_t *hl = malloc(ks * sizeof *hl), idxs[CS];
    const uint32_t *dloc[CS];
    const uint64_t *dval[CS];
    uint64_t *dcts[CS], kk[CS], zeros[2 * CS], *dagg = NULL, *ddec = NULL;
    for (uint64_t p = 0; p < total; p++) wants[p] = CS * zero;
    for (int c = 0; c < CS; c++) {
        uint32_t pos = (uint32_t)(c * 7);
        for (uint64_t q = 0; q < ks; q++) {                      /* strictly increasing positions, gaps of 1 .. 89 */
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            pos += 1 + (uint32_t)(x % 89);
            hl[q] = pos;
            hv[q] = x >> 12;
            wants[pos] += hv[q] - zero;
        }
        void *p1 = NULL, *p2 = NULL, *p3 = NULL;
        CHECK(flashe_dev_alloc(ctx, ks * 4, &p1)); CHECK(flashe_dev_alloc(ctx, ks * 8, &p2)); CHECK(flashe_dev_alloc(ctx, ks * 16, &p3));
        CHECK(flashe_memcpy_h2d(ctx, p1, hl, ks * 4)); CHECK(flashe_memcpy_h2d(ctx, p2, hv, ks * 8));
        dloc[c] = p1; dval[c] = p2; dcts[c] = p3; kk[c] = ks; idxs[c] = (uint32_t)c; zeros[2 * c] = zero; zeros[2 * c + 1] = 0;
    }
    CHECK(flashe_dev_alloc(ctx, total * 16, (void **)&dagg)); CHECK(flashe_dev_alloc(ctx, total * 16, (void **)&ddec));
    flashe_span_bounds *bounds = NULL;
    CHECK(flashe_span_bounds_create(ctx, total, CS, dloc, kk, &bounds));
    CHECK(flashe_sparse_encrypt_aggregate_dev(ctx, its, N_JOBS, total, CS, idxs, dloc, kk, dval, 1, zeros, bounds, dcts, dagg));
    CHECK(flashe_sparse_decrypt_bounds_dev(ctx, its, CS, dloc, kk, total, N_JOBS, bounds, dagg, ddec));
    CHECK(flashe_memcpy_d2h(ctx, got, ddec, total * 16));
    uint64_t bads = 0;
    for (uint64_t p = 0; p < total; p++) bads += got[2 * p] != wants[p] || got[2 * p + 1] != 0;
    /* and the ciphertexts are what a lone client's encrypt gives */
    uint64_t *lone = malloc(ks * 16), *both = malloc(ks * 16), differs = 0;
    CHECK(flashe_memcpy_d2h(ctx, hv, dval[CS - 1], ks * 8));
    CHECK(flashe_encrypt(ctx, its, CS - 1, FLASHE_SCHEME_SINGLE, ks, N_JOBS, hv, 1, lone));
    CHECK(flashe_memcpy_d2h(ctx, both, dcts[CS - 1], ks * 16));
    for (uint64_t j = 0; j < 2 * ks; j++) differs += lone[j] != both[j];
    printf("C_ROUND_SPARSE %s: total=%" PRIu64 " k=%" PRIu64 " clients=%d mismatches=%" PRIu64 " ciphertext_differences=%" PRIu64 "\n",
           bads == 0 && differs == 0 ? "OK" : "FAILED", total, ks, CS, bads, differs);
    flashe_span_bounds_destroy(bounds);
    flashe_ctx_destroy(ctx);
    return bads == 0 && differs == 0 ? 0 : 1;
}
