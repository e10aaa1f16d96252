/*
 * flashe.h -- C ABI of libflashe_hip.so, the MI355X (gfx950) FLASHE cipher engine.
 *
 * This is the drop-in boundary for ONE path of SamuelGong/FLASHE: the
 * federatedml/secureprotol FLASHE cipher (AES-256 PRF mask generation, per-element
 * modular encrypt / decrypt over big-integer vectors) plus the arbiter's ciphertext
 * mod-add reduce and the bit-packing codec either side of it.  Reference citations are
 * relative to the reference tree (federatedml/...).
 *
 * Conventions
 *   - plain C, no exceptions, no ownership transfer: the caller allocates every buffer.
 *   - every function returns 0 (FLASHE_OK) or a negative errno-style code;
 *     flashe_last_error(ctx) gives the text of the last failure on that ctx.
 *   - a ctx is bound to one HIP device and one HIP stream; it is not thread-safe, separate
 *     ctxs are independent.  *_dev functions take DEVICE pointers and are asynchronous on
 *     the ctx stream (flashe_sync to wait); the un-suffixed twins take HOST pointers and
 *     are synchronous (H2D + kernels + D2H).
 *   - element layout: an element of b = int_bits bits (1 <= b <= 128) is
 *     L = flashe_limbs(b) = ceil(b/64) little-endian uint64 limbs; vectors are [n][L].
 *     Inputs need not be reduced: everything is taken mod 2^b as the reference does
 *     (jzf_flashe.py:480-481).
 *   - there is NO CPU fallback: without a HIP device every ctx call fails with
 *     FLASHE_ENODEV.
 */
#ifndef FLASHE_H
#define FLASHE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLASHE_OK       0
#define FLASHE_EIO     (-5)    /* a HIP runtime call failed; see flashe_last_error */
#define FLASHE_ENOMEM  (-12)
#define FLASHE_ENODEV  (-19)   /* no usable HIP device */
#define FLASHE_EINVAL  (-22)

#define FLASHE_SCHEME_SINGLE 0 /* FlasheCipher(int_bits, mask="single") */
#define FLASHE_SCHEME_DOUBLE 1 /* FlasheCipher(int_bits)  (default "double") */

typedef struct flashe_ctx flashe_ctx;

/* ---- library / device ------------------------------------------------------------- */
/* ABI version of this header; flashe_abi_version() returns the one the library was built from.  Bumped whenever an exported
 * signature or struct layout changes or entry points are added.
 *   1  rounds 1-3 (entry points and flashe_prf_job fields were added without a bump: n_in / in_stride / sum_out_dev, the *_u32_dev,
 *      *_sum_dev, sparse_double_masks, mt19937 and sparsify_batch calls -- a round-1 binding of flashe_prf_job is NOT compatible)
 *   2  round 4: flashe_codec_layer + flashe_quantize_encrypt_model_dev / flashe_decrypt_unquantize_model_dev (the fused codec over a
 *      flattened model), flashe_batch_layer + flashe_quantize_batch_model_dev / flashe_unbatch_unquantize_model_dev (batched jobs);
 *      ctx-resident mask precompute (flashe_prepare_* / flashe_*_prepared[_dev] / flashe_prepared_query / _discard); flashe_span_bounds
 *      handles (+ flashe_sparse_*_bounds_dev); flashe_dynamic_masking_cost_dev; flashe_encrypt_batch_range_dev and
 *      flashe_packed_resolve_carry_strided_dev (element-sharded multi-GPU round); flashe_aggregate_elem_u32_dev; flashe_mt19937_plan;
 *      flashe_sparse_encrypt_aggregate_dev (the clients' sparse encrypts and the aggregate of their uploads in one pass);
 *      flashe_unquantize_model_dev (the model-wide codec back end without a decrypt: the sparse job's way back);
 *      flashe_sparse_span, flashe_sparse_encrypt_aggregate_range_dev, flashe_sparse_decrypt_range_dev (the sparse round by position ranges);
 *      timing probes and tuning knobs compiled out of libflashe_hip.so (-DFLASHE_TUNING build only)
 *   3  round 5: flashe_ctx_compact_layout (does this ctx run the *_u32_dev entry points?); the double-mask encrypt entry points
 *      refuse idx = 2^32 - 1 with FLASHE_EINVAL (the reference's OverflowError, jzf_flashe.py:352-353) instead of wrapping to
 *      prefix 0; flashe_prepared_discard releases the cached mask buffers; flashe_combine_batch_sum_dev (online encrypts with
 *      precomputed masks + their sum in one pass); flashe_encrypt_batch_sum_u32_dev (the compact layout's encrypts + their sum)
 *   4  round 6: flashe_device_peer_access and flashe_rccl_version (the preflight a rank of a multi-GPU launch runs before it creates
 *      its ctx); flashe_combine_batch_sum_decrypt_dev (online encrypts + their sum + the decrypt of the sum in one pass) */
#define FLASHE_ABI_VERSION 4
int flashe_abi_version(void);
int flashe_device_count(int *count);
/* Preflight of a multi-GPU launch (new): can `device` read and write `peer`'s memory directly (hipDeviceCanAccessPeer: what RCCL's
 * point-to-point transport over xGMI needs)?  *can_access = 1 / 0; device == peer gives 1.  Creates no ctx and no stream. */
int flashe_device_peer_access(int device, int peer, int *can_access);
int flashe_limbs(int int_bits);                 /* 1 or 2; 0 if int_bits is out of range */

/* ---- context ---------------------------------------------------------------------- */
/* Replaces FlasheCipher.__init__ + generate_prp_seed (jzf_flashe.py:230-260, :280-295):
 * key is the 32-byte AES-256 key, i.e. the low 256 bits of the PRP seed, big-endian
 * (jzf_aes.py:21-28).  stream: a hipStream_t to run on (e.g. the caller framework's
 * current stream) or NULL to let the ctx create its own non-blocking stream. */
int flashe_ctx_create(flashe_ctx **out, const uint8_t key[32], int int_bits, int device, void *stream);
int flashe_ctx_destroy(flashe_ctx *ctx);
int flashe_ctx_set_key(flashe_ctx *ctx, const uint8_t key[32]);
int flashe_ctx_int_bits(const flashe_ctx *ctx);
/* new: 1 if the compact uint32 entry points (flashe_encrypt_batch_u32_dev, flashe_aggregate_decrypt_u32_dev, ...) run on this ctx --
 * int_bits <= 32, the table PRF backend and the chained kernels (FLASHE_CHAIN != 0) -- 0 if they would answer FLASHE_EINVAL. */
int flashe_ctx_compact_layout(const flashe_ctx *ctx);
/* new: how many compute units the persistent launches of this ctx occupy (0 = the whole device; flashe_ctx_cu_count = what the
 * device has).  The PRF workgroups hold 128 KiB of LDS per CU, so a kernel from ANOTHER stream that needs more than the rest (RCCL's
 * transfer kernels do) runs beside a PRF launch only on CUs that launch leaves free: the multi-GPU schedules that hide the exchange
 * under the next chunk's encrypts set this to cu_count - 16 or so.  Results do not depend on it. */
int flashe_ctx_set_cu_limit(flashe_ctx *ctx, int cus);
int flashe_ctx_cu_count(const flashe_ctx *ctx);
/* Which implementation of the AES-256 PRF the fused kernels use (results are identical):
 * 0 = automatic, 1 = LDS T-table kernel, 2 = bit-sliced VALU kernel (b > 64, single add prefix with
 * at most one minus prefix; other shapes always use the table kernel).  Also settable at ctx creation
 * through the environment variable FLASHE_PRF_BACKEND=table|bitslice.  The bit-sliced kernels are measured
 * alternatives (2.5x the VALU work of the table kernel) and are NOT in libflashe_hip.so: they are built by
 * `make -C flashe_amd/csrc bitslice` into libflashe_hip_bitslice.so; the product library answers FLASHE_EINVAL. */
#define FLASHE_PRF_AUTO     0
#define FLASHE_PRF_TABLE    1
#define FLASHE_PRF_BITSLICE 2
int flashe_ctx_set_prf_backend(flashe_ctx *ctx, int backend);
/* Text of the last error on ctx (ctx == NULL: last flashe_ctx_create failure of this thread). */
const char *flashe_last_error(const flashe_ctx *ctx);
/* Known-answer self test on the device: FIPS-197 C.3 through the PRF kernel. */
int flashe_selftest(flashe_ctx *ctx);

/* ---- host-side logic of the path (no device needed) -------------------------------- */
/* chunks_idx(range(n), n_jobs) -- jzf_flashe.py:12-16.  begins has n_jobs + 1 entries. */
int flashe_chunks(uint64_t n, uint32_t n_jobs, uint64_t *begins);
/* set_idx_list(mode="decrypt") telescoping -- jzf_flashe.py:356-367.  raw is sorted in
 * place (as the reference sorts its argument); add_out / minus_out hold n_raw entries. */
int flashe_telescope(uint32_t *raw, int n_raw, uint32_t *add_out, uint32_t *minus_out, int *n_runs);
/* AES-256 of one 16-byte block on the HOST (key schedule check / small utilities) --
 * PsuedoRandomPermutation.get_permutation, jzf_aes_prp.py:24-30. */
int flashe_prp_block(const uint8_t key[32], const uint8_t in[16], uint8_t out[16]);

/* ---- device memory, stream, events -------------------------------------------------- */
/* Device blocks come from a per-device caching allocator (new): flashe_dev_free PARKS the block (a hipMalloc + hipFree pair of a
 * 160 MB block costs more than moving it over PCIe), flashe_dev_alloc hands a parked block of the same size class out again --
 * after a device-wide synchronisation has happened since it was parked, which keeps hipFree's guarantee that kernels of any
 * stream still reading the block finish first.  FLASHE_DEV_POOL_MB bounds the parked bytes per device (default 16384, 0 = off);
 * flashe_dev_trim gives every parked block of a device back (zeroed first). */
int flashe_dev_alloc(flashe_ctx *ctx, size_t bytes, void **dptr);
int flashe_dev_free(flashe_ctx *ctx, void *dptr);
int flashe_dev_trim(int device);
int flashe_dev_pool_stats(int device, uint64_t *parked_bytes, uint64_t *hits, uint64_t *misses);
/* new: page-locked (pinned) host memory, for callers that hand host vectors to the host-pointer calls repeatedly: DMA without
 * a staging copy, and a reused buffer spares the page faults of a fresh one.  Independent of any ctx. */
int flashe_host_alloc(size_t bytes, void **hptr);
int flashe_host_free(void *hptr);
int flashe_memcpy_h2d(flashe_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int flashe_memcpy_d2h(flashe_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int flashe_memcpy_d2d(flashe_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);
int flashe_memset_dev(flashe_ctx *ctx, void *dst_dev, int byte, size_t bytes);
int flashe_sync(flashe_ctx *ctx);
int flashe_event_create(flashe_ctx *ctx, void **event);
int flashe_event_destroy(flashe_ctx *ctx, void *event);
int flashe_event_record(flashe_ctx *ctx, void *event);               /* on the ctx stream */
int flashe_event_elapsed_ms(flashe_ctx *ctx, void *start, void *stop, float *ms); /* syncs on stop */
/* Make ctx's stream wait (on the device, not the host) for an event recorded by ANOTHER ctx of the
 * same device: the fork/join primitive for running e.g. the arbiter reduce on a second stream. */
int flashe_stream_wait_event(flashe_ctx *ctx, void *event);

/* Capture and replay (HIP graphs): the *_dev calls made on ctx between begin and end are recorded instead of
 * run; launch replays the whole sequence with one submission -- for launch-bound work such as a round over a
 * LeNet-sized model (six kernels of 10-60 us).  Pointers and scalar arguments are frozen into the graph: replay
 * with new DATA in the same buffers.  Host-pointer twins, syncs and event timing are not capturable, and
 * ctx-owned scratch must already have its size: run the sequence once normally before capturing it.
 * What a frozen argument block means for the cipher:
 *   - `iter` is frozen, so flashe_graph_launch REPEATS the captured round's mask streams: never feed it new plaintexts
 *     (ct1 - ct2 would equal pt1 - pt2).  For the NEXT rounds use flashe_graph_launch_shifted: every PRF kernel adds a
 *     device-resident iter shift to its frozen iter at run time, so replaying with iter_shift = r runs the captured
 *     sequence exactly as if every call in it had been made with iter + r (what round r after the captured one needs,
 *     prepare_encrypt's iter + 1 included).
 *   - the expanded key is frozen too: after flashe_ctx_set_key a graph captured earlier refuses to launch (FLASHE_EINVAL). */
typedef struct flashe_graph flashe_graph;
int flashe_graph_begin(flashe_ctx *ctx);
int flashe_graph_end(flashe_ctx *ctx, flashe_graph **graph);
int flashe_graph_launch(flashe_ctx *ctx, flashe_graph *graph);
int flashe_graph_launch_shifted(flashe_ctx *ctx, flashe_graph *graph, uint32_t iter_shift);    /* new */
int flashe_graph_destroy(flashe_graph *graph);

/* ---- PRF mask streams -------------------------------------------------------------- */
/* out[j] = sum_k term(iter, idx[k], j) mod 2^b, chunked like chunks_idx(range(n), n_jobs).
 * n_idx == 1 is _static_prepare_encrypt_single (jzf_flashe.py:19-45), i.e. one stream of
 * prepare_encrypt / prepare_decrypt (:599-666); n_idx > 1 is one half of
 * _static_prepare_decrypt / _static_prepare_decrypt_single (:85-152). */
int flashe_mask_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *idx, int n_idx,
                    uint64_t n, uint32_t n_jobs, uint64_t *out_dev);
int flashe_mask(flashe_ctx *ctx, uint32_t iter, const uint32_t *idx, int n_idx,
                uint64_t n, uint32_t n_jobs, uint64_t *out);

/* ---- encrypt / decrypt -------------------------------------------------------------- */
/* FlasheCipher.encrypt -- jzf_flashe.py:490-504 -> _multiprocessing_encrypt (:456-488,
 * double: ct = pt + term(iter, idx) - term(iter, idx+1)) or _multiprocessing_encrypt_single
 * (:431-454, single: ct = pt + term(iter, idx)), fused with the mask generation.
 * pt has pt_limbs limbs per element (1 = uint64 plaintext, zero-extended; or L). */
int flashe_encrypt_dev(flashe_ctx *ctx, uint32_t iter, uint32_t idx, int scheme,
                       uint64_t n, uint32_t n_jobs,
                       const uint64_t *pt_dev, int pt_limbs, uint64_t *ct_dev);
int flashe_encrypt(flashe_ctx *ctx, uint32_t iter, uint32_t idx, int scheme,
                   uint64_t n, uint32_t n_jobs,
                   const uint64_t *pt, int pt_limbs, uint64_t *ct);

/* n_vec independent encrypts of equal length (e.g. the clients a simulation or a multi-tenant service hosts on one
 * GPU, or the layers of one model) in as few launches as possible: ct[v] = FlasheCipher.encrypt with cipher index
 * idx[v], exactly as n_vec calls of flashe_encrypt_dev.  pt / ct are HOST arrays of n_vec device pointers.
 * int_bits > 64, double mask: runs of CONSECUTIVE cipher indices (idx[v + 1] == idx[v] + 1) share their PRF streams --
 * client c's minus stream term(iter, c + 1) is client c + 1's add stream (jzf_flashe.py:349-353) -- so a run of C clients
 * costs C + 1 AES blocks per element instead of 2 C; the ciphertexts are bit-identical to C separate calls. */
int flashe_encrypt_batch_dev(flashe_ctx *ctx, uint32_t iter, int scheme, uint64_t n, uint32_t n_jobs, int n_vec,
                             const uint32_t *idx, const uint64_t *const *pt_dev, int pt_limbs, uint64_t *const *ct_dev);

/* ---- mask precompute, resident in the ctx (FlasheCipher.prepare_encrypt / prepare_decrypt, jzf_flashe.py:599-666) ----
 * The reference computes the next round's masks in idle time and caches them in next_iter_encrypt_prepared / next_iter_decrypt_prepared;
 * the NEXT encrypt / decrypt then only adds vectors (no AES) and deletes the cache (:457, :483-486, :557-580).  Here the cache lives in
 * HBM inside the ctx -- a C caller needs no state machine of its own:
 *   flashe_prepare_encrypt(iter_next, idx, scheme, num_params): masks term(iter_next, idx) and, double mask, term(iter_next, idx + 1)
 *       of length num_params (:599-631; the caller passes iter + 1);
 *   flashe_prepare_decrypt(iter, num_clients, num_params): term(iter, num_clients) and term(iter, 0) -- nobody dropped (:633-666);
 *   flashe_encrypt_prepared[_dev](n, pt): ct = (pt + add - minus) mod 2^b from the cache, which is CONSUMED; FLASHE_EINVAL without a
 *       cache, or -- cache kept, like NumPy's broadcast error in the reference -- when n differs from num_params;
 *   flashe_decrypt_prepared[_dev](iter, extra add / minus prefixes, n, in): out = in + add - minus from the cache plus the prefixes the
 *       precompute does not cover (dropouts: what set_idx_list leaves after skipping {num_clients} / {0}, :372-386), computed online
 *       and merged in (:557-564); consumes the cache;
 *   flashe_prepared_query(which, &n, &add_dev, &minus_dev): 1 / 0 = a cache is / is not held; its length and device vectors (valid
 *       until consumed or re-prepared); flashe_prepared_discard(which) drops it (which: FLASHE_PREPARED_ENCRYPT | _DECRYPT) AND
 *       gives the mask blocks back to the device.  (A cache that is merely consumed keeps its blocks inside the ctx for the next
 *       round's masks -- up to four vectors of num_params elements per ctx -- until flashe_prepared_discard or flashe_ctx_destroy.)
 *       flashe_prepared_discard SYNCHRONISES the ctx stream before it frees (the only call of this group that does), and what it frees
 *       is what flashe_prepared_query handed out: pointers from an earlier query, and a graph captured around flashe_*_prepared_dev
 *       (its kernel arguments are those pointers), must not be used after a discard.  Inside a capture it only marks the cache invalid.
 * prepare_* are asynchronous on the ctx stream like every *_dev call. */
#define FLASHE_PREPARED_ENCRYPT 1
#define FLASHE_PREPARED_DECRYPT 2
int flashe_prepare_encrypt(flashe_ctx *ctx, uint32_t iter_next, uint32_t idx, int scheme, uint64_t num_params, uint32_t n_jobs);
int flashe_prepare_decrypt(flashe_ctx *ctx, uint32_t iter, uint32_t num_clients, uint64_t num_params, uint32_t n_jobs);
int flashe_prepared_query(flashe_ctx *ctx, int which, uint64_t *n, const uint64_t **add_dev, const uint64_t **minus_dev);
int flashe_prepared_discard(flashe_ctx *ctx, int which);
int flashe_encrypt_prepared_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *pt_dev, int pt_limbs, uint64_t *ct_dev);
int flashe_encrypt_prepared(flashe_ctx *ctx, uint64_t n, const uint64_t *pt, int pt_limbs, uint64_t *ct);
int flashe_decrypt_prepared_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add, const uint32_t *minus_idx,
                                int n_minus, uint64_t n, uint32_t n_jobs, const uint64_t *in_dev, uint64_t *out_dev);
int flashe_decrypt_prepared(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add, const uint32_t *minus_idx,
                            int n_minus, uint64_t n, uint32_t n_jobs, const uint64_t *in, uint64_t *out);

/* flashe_encrypt_batch_dev on ONE element slice [first, first + count) of the n-element vectors (new): the launch of a GPU that owns
 * that slice of EVERY client's vector -- element sharding, SURVEY.md section 8e (i): mask streams are position-indexed, so the slices
 * need no exchange for the element-wise aggregate.  pt_dev[v] / ct_dev[v] / sum_out_dev address element `first`; n and n_jobs still
 * describe the whole vector (the int_bits <= 64 chunking).  sum_out_dev (may be NULL): receives the slice of sum_v ct[v] mod 2^b,
 * from the same launch when flashe_encrypt_batch_sum_dev's conditions hold for the slice, from a reduce launch otherwise. */
int flashe_encrypt_batch_range_dev(flashe_ctx *ctx, uint32_t iter, int scheme, uint64_t n, uint32_t n_jobs, uint64_t first,
                                   uint64_t count, int n_vec, const uint32_t *idx, const uint64_t *const *pt_dev, int pt_limbs,
                                   uint64_t *const *ct_dev, uint64_t *sum_out_dev);

/* flashe_encrypt_batch_dev that ALSO writes the local partial aggregate sum_out[j] = sum_v ct[v][j] mod 2^b (new): what the
 * arbiter's reduce (jzf_aggregator.py:424-430) yields for the clients this GPU hosts -- SURVEY.md section 5: "each GPU encrypts and
 * locally mod-adds its share".  int_bits > 64, double mask, one run of consecutive cipher indices, a vector long enough to fill the
 * device: ONE launch (every ciphertext of an element passes through the lane's registers, the running sum costs one extra 16-byte
 * store per element and the C ciphertexts are never re-read).  Any other shape: the encrypts followed by flashe_aggregate_elem_dev.
 * The ciphertexts are written as by flashe_encrypt_batch_dev; sum_out_dev (n x L limbs) must not be one of them nor a plaintext
 * (FLASHE_EINVAL; the same holds for the _range and _u32 forms). */
int flashe_encrypt_batch_sum_dev(flashe_ctx *ctx, uint32_t iter, int scheme, uint64_t n, uint32_t n_jobs, int n_vec,
                                 const uint32_t *idx, const uint64_t *const *pt_dev, int pt_limbs, uint64_t *const *ct_dev,
                                 uint64_t *sum_out_dev);

/* General batched form (new): every entry is one piece of mask arithmetic on elements
 * [first, first + count) of an n-element vector,
 *     out[k] = in[k] + term(iter, add_idx, first + k) - [has_minus] term(iter, minus_idx, first + k)   mod 2^b,
 * with in = 0 when in_dev is NULL; in_dev / out_dev address element `first`.  It covers, with one
 * launch for many entries (b > 64; entry by entry otherwise):
 *   FlasheCipher.encrypt (double)     add = idx, minus = idx + 1, in = plaintext   jzf_flashe.py:349-353, :480-481
 *   FlasheCipher.encrypt (single)     add = idx, has_minus = 0                     :308-309, :450-451
 *   decrypt, nobody dropped           add = num_clients, minus = 0, in = aggregate :633-666, :570-571
 *   prepare_encrypt / prepare_decrypt the same with in_dev = NULL (the cached add - minus) :599-666
 * All entries of one call must agree on has_minus.
 * n_in > 1 fuses the arbiter's reduce (jzf_aggregator.py:424-430) into the entry: the input is
 * sum_{c < n_in} of the vectors at in_dev + c * in_stride elements (in_limbs = L required), reduced
 * mod 2^b; sum_out_dev, when not NULL, receives that sum (the ciphertext aggregate). n_in = 0 means 1. */
typedef struct flashe_prf_job {
    uint32_t add_idx;
    uint32_t minus_idx;
    int32_t has_minus;       /* 0 or 1 */
    int32_t in_limbs;        /* limbs per input element: 1 (uint64, zero-extended) or L; ignored when in_dev is NULL */
    uint64_t first, count;
    const uint64_t *in_dev;
    uint64_t *out_dev;
    uint32_t n_in;           /* number of input vectors summed (0 or 1: plain input) */
    uint32_t reserved;       /* must be 0 */
    uint64_t in_stride;      /* elements between consecutive input vectors when n_in > 1 */
    uint64_t *sum_out_dev;   /* optional, n_in > 1 only */
} flashe_prf_job;
int flashe_prf_jobs_dev(flashe_ctx *ctx, uint32_t iter, uint64_t n, uint32_t n_jobs,
                        int n_entries, const flashe_prf_job *entries);

/* FlasheCipher.decrypt -- jzf_flashe.py:584-594 -> _multiprocessing_decrypt (:537-582) /
 * _multiprocessing_decrypt_single (:506-535) with the prefix lists set_idx_list derived
 * (:356-386; single: :311-314 with n_add = 0):
 * out = in + sum_k term(iter, add_idx[k]) - sum_k term(iter, minus_idx[k])  mod 2^b.
 * The lists may have ANY length, as in the reference (single mask: one minus prefix per uploaded client; scattered
 * dropouts: one pair per run); beyond 96 entries per list the call chains launches that accumulate in place.
 * in_dev == out_dev is allowed.  The same holds for flashe_mask* and the range twins. */
int flashe_decrypt_dev(flashe_ctx *ctx, uint32_t iter,
                       const uint32_t *add_idx, int n_add, const uint32_t *minus_idx, int n_minus,
                       uint64_t n, uint32_t n_jobs, const uint64_t *in_dev, uint64_t *out_dev);
int flashe_decrypt(flashe_ctx *ctx, uint32_t iter,
                   const uint32_t *add_idx, int n_add, const uint32_t *minus_idx, int n_minus,
                   uint64_t n, uint32_t n_jobs, const uint64_t *in, uint64_t *out);

/* Range twins for element-sharded execution (one vector split over several GPUs): the call
 * covers global elements [first, first + count) of an n-element vector; the device pointers
 * address element `first` (i.e. are indexed by element - first).  n and n_jobs still describe the
 * WHOLE vector, because the PRF counters depend on chunks_idx(range(n), n_jobs)
 * (jzf_flashe.py:12-16, :34). */
int flashe_mask_range_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *idx, int n_idx,
                          uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count, uint64_t *out_dev);
int flashe_encrypt_range_dev(flashe_ctx *ctx, uint32_t iter, uint32_t idx, int scheme,
                             uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count,
                             const uint64_t *pt_dev, int pt_limbs, uint64_t *ct_dev);
int flashe_decrypt_range_dev(flashe_ctx *ctx, uint32_t iter,
                             const uint32_t *add_idx, int n_add, const uint32_t *minus_idx, int n_minus,
                             uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count,
                             const uint64_t *in_dev, uint64_t *out_dev);

/* Pre-computed-mask arithmetic: out = in + add - minus mod 2^b (add / minus may be NULL).
 * The online half of encrypt / decrypt when next_iter_{en,de}crypt_prepared is populated
 * (jzf_flashe.py:457,480-481, :557-571) and the sparse single-mask decrypt (:531-532). */
int flashe_combine_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *in_dev, int in_limbs,
                       const uint64_t *add_dev, const uint64_t *minus_dev, uint64_t *out_dev);
int flashe_combine(flashe_ctx *ctx, uint64_t n, const uint64_t *in, int in_limbs,
                   const uint64_t *add, const uint64_t *minus, uint64_t *out);
/* n_vec combines of equal length in as few launches as possible (new): the online encrypts of the clients one process hosts --
 * vector v is out[v] = in[v] + add[v] - minus[v].  in / add / minus / out are HOST arrays of n_vec device pointers; add_dev,
 * minus_dev or single entries of them may be NULL. */
int flashe_combine_batch_dev(flashe_ctx *ctx, uint64_t n, int n_vec, const uint64_t *const *in_dev, int in_limbs,
                             const uint64_t *const *add_dev, const uint64_t *const *minus_dev, uint64_t *const *out_dev);
/* The same combines AND sum_out_dev = sum_v out[v] mod 2^b written by the same pass (new): the online encrypts with precomputed masks
 * (jzf_flashe.py:457, :480-481) plus the arbiter's element-wise reduce of what they wrote (jzf_aggregator.py:424-430) -- the reduce
 * costs one more store instead of a launch that reads every ciphertext back; the twin of flashe_encrypt_batch_sum_dev for the
 * precompute path.  sum_out_dev must not be one of out_dev, in_dev, add_dev or minus_dev (FLASHE_EINVAL). */
int flashe_combine_batch_sum_dev(flashe_ctx *ctx, uint64_t n, int n_vec, const uint64_t *const *in_dev, int in_limbs,
                                 const uint64_t *const *add_dev, const uint64_t *const *minus_dev, uint64_t *const *out_dev,
                                 uint64_t *sum_out_dev);
/* ... and the DECRYPT of that sum with the decrypting party's precomputed masks from the same pass (new, ABI 4): dec_out_dev =
 * (sum_out + dec_add - dec_minus) mod 2^b, i.e. jzf_flashe.py:557-571 with next_iter_decrypt_prepared populated (:633-666) applied to
 * the reduce of jzf_aggregator.py:424-430 -- the workgroup that completes an element's sum still holds it.  dec_add_dev / dec_minus_dev
 * may be NULL; dec_out_dev must be a vector of its own (not the sum, an operand or an output: FLASHE_EINVAL). */
int flashe_combine_batch_sum_decrypt_dev(flashe_ctx *ctx, uint64_t n, int n_vec, const uint64_t *const *in_dev, int in_limbs,
                                         const uint64_t *const *add_dev, const uint64_t *const *minus_dev, uint64_t *const *out_dev,
                                         uint64_t *sum_out_dev, const uint64_t *dec_add_dev, const uint64_t *dec_minus_dev, uint64_t *dec_out_dev);

/* ---- arbiter reduce ----------------------------------------------------------------- */
/* Element-wise: out[j] = sum_c cts[c][j] mod 2^b -- jzf_aggregator.py:424-430.
 * cts is a HOST array of C pointers (device pointers for _dev).  Device vectors of two-limb elements must be
 * 16-byte aligned; one-limb vectors 8-byte (16-byte aligned operands take the faster 16-byte form). */
int flashe_aggregate_elem_dev(flashe_ctx *ctx, int C, const uint64_t *const *cts_dev,
                              uint64_t n, uint64_t *out_dev);
int flashe_aggregate_elem(flashe_ctx *ctx, int C, const uint64_t *const *cts,
                          uint64_t n, uint64_t *out);
/* The element-wise reduce fused with the decrypt of its result (new): agg = sum_c cts[c] mod 2^b
 * (stored to agg_out_dev unless NULL), out = agg + sum term(add) - sum term(minus) -- exactly
 * flashe_aggregate_elem_dev followed by flashe_decrypt_range_dev on elements [first, first + count)
 * of the n-element vector, in ONE pass over the ciphertexts (one launch, the aggregate never
 * re-read) for one add and at most one minus prefix -- int_bits > 64: ciphertexts equally spaced
 * in memory; int_bits <= 64: any C <= 64 operands (n < 2^32); any other shape runs the two calls.
 * cts_dev / agg_out_dev / out_dev address element `first`. */
int flashe_aggregate_decrypt_range_dev(flashe_ctx *ctx, uint32_t iter,
                                       const uint32_t *add_idx, int n_add, const uint32_t *minus_idx, int n_minus,
                                       uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count,
                                       int C, const uint64_t *const *cts_dev, uint64_t *agg_out_dev, uint64_t *out_dev);
/* Compact layout for int_bits <= 32 (new; no reference counterpart).  The ABI stores one element per uint64 limb whatever
 * int_bits is; at the widths the reference's own jobs ship (int_bits = 20, 23: jzf configs) that is
 * 8 bytes moved for 20 useful bits, and the kernels of those widths are bound by exactly those bytes.  These entry points take
 * and produce the same VALUES as uint32 arrays, so that the hot round moves half of them:
 *   flashe_encrypt_batch_u32_dev      = flashe_encrypt_batch_dev (jzf_flashe.py:456-488 per vector) on uint32 plaintexts and ciphertexts;
 *   flashe_aggregate_decrypt_u32_dev  = flashe_aggregate_decrypt_range_dev with ONE add and at most one minus prefix (the no-dropout
 *                                       decrypt and every single telescoped run, jzf_flashe.py:356-367) on up to 64 uint32 operands;
 *                                       agg_out_dev (may be NULL) and out_dev are uint32 (out_elem_bytes = 4) or uint64 (8) arrays;
 *   flashe_widen_u32_dev / flashe_narrow_u32_dev convert to and from the one-limb layout every other call uses.
 * ct[j] here == (uint32) of what the uint64 call writes; n < 2^32; table PRF; pointers address element `first`. */
int flashe_encrypt_batch_u32_dev(flashe_ctx *ctx, uint32_t iter, int scheme, uint64_t n, uint32_t n_jobs, int n_vec,
                                 const uint32_t *idx, const uint32_t *const *pt_dev, uint32_t *const *ct_dev);
/* new (round 5): flashe_encrypt_batch_u32_dev and sum_out_dev = sum_v ct_dev[v] mod 2^b from the same launch -- the compact twin of
 * flashe_encrypt_batch_sum_dev.  One launch for consecutive clients under the double mask at int_bits 16 / 20 / 23 / 24 / 32 when the vectors are
 * long enough for the paired kernel; every other shape runs the encrypts and then the reduce of what they wrote (same results). */
int flashe_encrypt_batch_sum_u32_dev(flashe_ctx *ctx, uint32_t iter, int scheme, uint64_t n, uint32_t n_jobs, int n_vec, const uint32_t *idx,
                                     const uint32_t *const *pt_dev, uint32_t *const *ct_dev, uint32_t *sum_out_dev);
int flashe_aggregate_decrypt_u32_dev(flashe_ctx *ctx, uint32_t iter,
                                     const uint32_t *add_idx, int n_add, const uint32_t *minus_idx, int n_minus,
                                     uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count,
                                     int C, const uint32_t *const *cts_dev, void *agg_out_dev, void *out_dev, int out_elem_bytes);
/* The arbiter's element-wise reduce (jzf_aggregator.py:424-430) on uint32 vectors: out[j] = sum_c cts[c][j] mod 2^b; any C >= 1
 * (partial sums accumulate in out_dev beyond 64 operands), out_dev may be one of the operands only when C <= 64. */
int flashe_aggregate_elem_u32_dev(flashe_ctx *ctx, int C, const uint32_t *const *cts_dev, uint64_t n, uint32_t *out_dev);
int flashe_widen_u32_dev(flashe_ctx *ctx, uint64_t n, const uint32_t *in_dev, uint64_t *out_dev);
int flashe_narrow_u32_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *in_dev, uint32_t *out_dev);
/* Packed: each operand is ONE integer of total_bits bits (n_limbs = ceil(total_bits/64)
 * little-endian limbs); out = sum mod 2^total_bits -- jzf_aggregator.py:406-419. */
int flashe_aggregate_packed_dev(flashe_ctx *ctx, int C, const uint64_t *const *packed_dev,
                                uint64_t n_limbs, uint64_t total_bits, uint64_t *out_dev);
int flashe_aggregate_packed(flashe_ctx *ctx, int C, const uint64_t *const *packed,
                            uint64_t n_limbs, uint64_t total_bits, uint64_t *out);

/* Slice helpers (new; no reference counterpart): the packed reduce above cut into limb slices
 * over several GPUs ("packed variant") needs each slice's carry-out and
 * the one way a carry-in can ripple through a whole slice.
 * probe: x_dev holds n_limbs - 1 body limbs plus one carry limb (the slice was summed one limb
 * wider than it is).  info_dev[0] = x[0]; info_dev[1] = 1 iff body limbs [1, n_limbs - 1) are
 * all ones; info_dev[2] = x[n_limbs - 1].  Asynchronous; info_dev is device memory (24 bytes).
 * add_carry: x = (x + carry_in) mod 2^total_bits in place, n_limbs = ceil(total_bits / 64). */
int flashe_packed_probe_dev(flashe_ctx *ctx, uint64_t n_limbs, const uint64_t *x_dev, uint64_t *info_dev);
int flashe_packed_add_carry_dev(flashe_ctx *ctx, uint64_t n_limbs, uint64_t total_bits,
                                uint64_t carry_in, uint64_t *x_dev);
/* add_carry with the carry-in derived ON THE DEVICE from the probe triples of the n_below slices underneath (infos_dev =
 * the all-gathered 3-word infos, slice 0 first): no host round trip between the exchange of the infos and the ripple. */
int flashe_packed_resolve_carry_dev(flashe_ctx *ctx, uint64_t n_limbs, uint64_t total_bits,
                                    const uint64_t *infos_dev, int n_below, uint64_t *x_dev);
/* The same with the triples `stride_words` apart (a non-zero multiple of 3; infos_dev points at the LOWEST slice's triple): -3 walks
 * gathered triples that run from the most significant slice down -- element slices of a packed vector, whose element 0 is the most
 * significant (jzf_weights.py:59-62), gathered in rank order. */
int flashe_packed_resolve_carry_strided_dev(flashe_ctx *ctx, uint64_t n_limbs, uint64_t total_bits, const uint64_t *infos_dev,
                                            int n_below, int stride_words, uint64_t *x_dev);              /* new */

/* ---- bit-packing codec ---------------------------------------------------------------- */
/* pack: P = sum_j x[j] << (b * (n-1-j)) as ceil(n*b/64) little-endian limbs --
 * _to_bytes / _to_bytes_old + compress(), jzf_weights.py:36-84, :155-195.
 * unpack: the inverse -- _from_bytes_old + reverse(), jzf_weights.py:87-95, :197-231. */
int flashe_pack_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *in_dev, uint64_t *out_dev);
int flashe_pack(flashe_ctx *ctx, uint64_t n, const uint64_t *in, uint64_t *out);
int flashe_unpack_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *in_dev, uint64_t *out_dev);
int flashe_unpack(flashe_ctx *ctx, uint64_t n, const uint64_t *in, uint64_t *out);

/* ---- sparse path --------------------------------------------------------------------- */
/* Location lists live in device memory in the *_dev variants, so they cannot be validated before launch: a location
 * >= total (the reference raises IndexError there) -- or, in the "sorted" variants, an entry that is not strictly
 * increasing -- is SKIPPED by the kernels, nothing is written outside the dense vector, and the next synchronising call
 * on the ctx (flashe_sync, flashe_memcpy_d2h) returns FLASHE_EINVAL once.  The host-pointer twins validate up front. */
/* Arbiter.expand_to_dense -- jzf_aggregator.py:150-165: out[loc[q]] = vals[q], every other
 * of the `total` positions = zero (L limbs, HOST pointer in both variants). */
int flashe_expand_to_dense_dev(flashe_ctx *ctx, uint64_t total, uint64_t k, const uint32_t *loc_dev,
                               const uint64_t *vals_dev, const uint64_t *zero, uint64_t *out_dev);
int flashe_expand_to_dense(flashe_ctx *ctx, uint64_t total, uint64_t k, const uint32_t *loc,
                           const uint64_t *vals, const uint64_t *zero, uint64_t *out);
/* The sparse job's arbiter step as one operation (new): out = sum_c expand_to_dense(total, loc[c], vals[c], zero_c)
 * mod 2^b -- Arbiter.expand_to_dense (jzf_aggregator.py:150-165, :382-384) for every client followed by the
 * element-wise reduce (:424-430) -- without materialising the C dense vectors: the sum of the zero values is
 * written everywhere, then each client's (vals[q] - zero_c) is added at loc[c][q].  loc / vals are HOST arrays of
 * C device pointers, k a HOST array, zeros a HOST array of C x L limbs; locations are distinct within a client.
 * sorted != 0 promises that every loc[c] is strictly increasing -- what Client.sparsify emits (jzf_aggregator.py:598,
 * :604) -- and selects the one-pass form: spans of the dense vector are accumulated in LDS and written once. */
int flashe_sparse_aggregate_dev(flashe_ctx *ctx, uint64_t total, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                const uint64_t *const *vals_dev, const uint64_t *zeros, int sorted, uint64_t *out_dev);
/* Sparse single-mask dense minus-mask -- set_idx_list_single sparse branch,
 * jzf_flashe.py:316-343: for client c the stream over COMPACT positions 0..k[c]-1 (prefix
 * iter|c, chunks_idx(range(k[c]), n_jobs)) scattered to loc[c][q] and summed over clients.
 * loc is a HOST array of C pointers, k a HOST array. */
int flashe_sparse_minus_mask_dev(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev,
                                 const uint64_t *k, uint64_t total, uint32_t n_jobs, uint64_t *out_dev);
/* The same for strictly increasing location lists (see flashe_sparse_aggregate_dev); the host-pointer twin picks it by
 * itself when the lists qualify. */
int flashe_sparse_minus_mask_sorted_dev(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev,
                                        const uint64_t *k, uint64_t total, uint32_t n_jobs, uint64_t *out_dev);
/* The single-mask sparse DECRYPT in the pass that builds the mask: out = (agg - minus-mask) mod 2^b, i.e. set_idx_list_single's
 * sparse branch (jzf_flashe.py:316-343) followed by _multiprocessing_decrypt_single's `value - minus` (:531-532) without the dense
 * mask ever reaching HBM.  Same arguments as flashe_sparse_minus_mask[_sorted]_dev plus the aggregate (total x L limbs, must not be out_dev).
 * new: fusion of two reference steps. */
int flashe_sparse_decrypt_dev(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                              uint64_t total, uint32_t n_jobs, int sorted, const uint64_t *agg_dev, uint64_t *out_dev);
/* Span bounds of a round's location lists, computed ONCE (new).  The LDS-staged sparse passes cut the dense vector into spans and
 * first find, for every span, where each client's (strictly increasing) list enters it -- a pass over all lists that the sparse
 * aggregate and the sparse decrypt of one round would otherwise both run on the same lists.  (The handle carries a table for both
 * span sizes in use: the one the ctx's hot passes read -- the passes with the PRF inside where the ctx has them -- is filled by create /
 * recompute, the plain reduce's by the first call that needs it, on THAT call's ctx stream; inside a graph capture that fill is part
 * of the graph.)  flashe_span_bounds_create computes the
 * table for C lists (any C) asynchronously on the ctx stream; the *_bounds_dev calls take it instead of recomputing; the handle is valid
 * for exactly these list pointers / lengths / total (checked) and until flashe_span_bounds_destroy.  The table describes the lists'
 * CONTENTS, which the check cannot see: the lists must not change while a handle built on them is in use, and after any in-place
 * rewrite (a job that reuses its upload buffers every round) flashe_span_bounds_recompute is MANDATORY before the next *_bounds_dev /
 * *_range_dev call -- a stale table stays memory-safe (slices are clamped, an entry outside its span raises the ctx's deferred error
 * flag) but the sums are wrong. */
typedef struct flashe_span_bounds flashe_span_bounds;
int flashe_span_bounds_create(flashe_ctx *ctx, uint64_t total, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                              flashe_span_bounds **out);
/* the same handle for the next round's lists (same total and C): recomputed in place, asynchronously, without an allocation */
int flashe_span_bounds_recompute(flashe_ctx *ctx, flashe_span_bounds *bounds, const uint32_t *const *loc_dev, const uint64_t *k);
void flashe_span_bounds_destroy(flashe_span_bounds *bounds);
int flashe_sparse_aggregate_bounds_dev(flashe_ctx *ctx, uint64_t total, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                       const uint64_t *const *vals_dev, const uint64_t *zeros, const flashe_span_bounds *bounds,
                                       uint64_t *out_dev);
int flashe_sparse_decrypt_bounds_dev(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                     uint64_t total, uint32_t n_jobs, const flashe_span_bounds *bounds, const uint64_t *agg_dev,
                                     uint64_t *out_dev);
/* The sparse twin of flashe_encrypt_batch_sum_dev (new): the C clients this device plays encrypt their compact uploads with the single
 * mask -- ct_dev[c] = flashe_encrypt_dev(iter, idx[c], SINGLE, k[c], n_jobs) of pt_dev[c] (jzf_flashe.py:471-478) -- AND the sum of
 * their expanded uploads, flashe_sparse_aggregate_dev(loc_dev, ct_dev, zeros) (jzf_aggregator.py:150-165, :419-430), is written to
 * agg_out_dev in the same pass.  int_bits > 64 on the table PRF: ONE persistent launch per 64 clients computes every entry's mask block
 * inside the LDS-staged span reduce (the ciphertexts are stored on the way, the compact values never travel twice); otherwise the two
 * calls it stands for.  loc_dev[c] strictly increasing; bounds: NULL or the handle of exactly these lists. */
int flashe_sparse_encrypt_aggregate_dev(flashe_ctx *ctx, uint32_t iter, uint32_t n_jobs, uint64_t total, int C, const uint32_t *idx,
                                        const uint32_t *const *loc_dev, const uint64_t *k, const uint64_t *const *pt_dev, int pt_limbs,
                                        const uint64_t *zeros, const flashe_span_bounds *bounds, uint64_t *const *ct_dev,
                                        uint64_t *agg_out_dev);
/* The sparse round sharded by POSITION ranges over several GPUs (new; SURVEY 8e (i) for the sparse path): GPU g owns the positions
 * [first, first + count) of the dense vector and runs every client's entries that fall into them -- counters and list indices stay
 * global (the full lists and their bounds handle are passed), nothing is exchanged for the aggregate.  first is a multiple of
 * flashe_sparse_span(), first + count one or the end of the vector; agg_out_dev / agg_dev / out_dev address position `first`; only the
 * ciphertexts of entries inside the range are written.  int_bits > 64 on the table PRF, strictly increasing lists, bounds required. */
int flashe_sparse_span(void);
int flashe_sparse_encrypt_aggregate_range_dev(flashe_ctx *ctx, uint32_t iter, uint32_t n_jobs, uint64_t total, int C, const uint32_t *idx,
                                              const uint32_t *const *loc_dev, const uint64_t *k, const uint64_t *const *pt_dev,
                                              int pt_limbs, const uint64_t *zeros, const flashe_span_bounds *bounds, uint64_t first,
                                              uint64_t count, uint64_t *const *ct_dev, uint64_t *agg_out_dev);
int flashe_sparse_decrypt_range_dev(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                    uint64_t total, uint32_t n_jobs, const flashe_span_bounds *bounds, uint64_t first, uint64_t count,
                                    const uint64_t *agg_dev, uint64_t *out_dev);
int flashe_sparse_minus_mask(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc,
                             const uint64_t *k, uint64_t total, uint32_t n_jobs, uint64_t *out);
/* Dense-position selected masks -- _static_prepare_decrypt_spar as ONE chunk
 * (jzf_flashe.py:155-225, begin = 0): out[p] = sum_i sel[i][p] * term(iter, i, p) mod 2^b,
 * sel[i] a 0/1 byte vector of length total.  sel is a HOST array of n_lists pointers. */
/* The sparse branch of set_idx_list for the DOUBLE mask as one operation (new) -- jzf_flashe.py:388-426 (per-position run analysis of
 * the clients' one-hot location vectors) + _static_prepare_decrypt_spar (:155-225, dense-position counters, one chunk, begin = 0):
 *   add_out[p]   = sum over clients c that hold p while client c + 1 does not (or c is the last) of term(iter, c + 1, p)
 *   minus_out[p] = sum over clients c that hold p while client c - 1 does not (or c is the first) of term(iter, c, p)      (mod 2^b)
 * straight from the clients' STRICTLY INCREASING location lists (what Client.sparsify emits): every list entry looks its position up
 * in the two neighbouring lists and computes at most two AES blocks, the span reduce scatters the compact values -- sum_c k_c block
 * pairs instead of (C + 1) x total blocks, no one-hot vectors of `total` bytes per list (the reference itself skips blocks without a
 * selected slot, :170, :201).  loc is a HOST array of C device pointers, k a HOST array; add_out / minus_out: total x L limbs each. */
int flashe_sparse_double_masks_dev(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                   uint64_t total, uint64_t *add_out_dev, uint64_t *minus_out_dev);
/* The arbiter's per-round choice between single and double masks for a sparse job (Arbiter.dynamic_masking,
 * jzf_flashe_block.py:92-112): single_cost = 2 sum_c len(mask_c); double_cost = 2 single_cost - 2 canceled, canceled = the positions
 * consecutive clients share (the reference ANDs one-hot vectors of `total` entries per pair).  Here every list entry of client c is
 * looked up in client c + 1's list on the device -- no one-hots, sum_c k_c binary searches.  loc_dev: HOST array of C device pointers
 * to STRICTLY INCREASING lists (what Client.sparsify emits), k: HOST array of their lengths.  The caller decides: "single" iff
 * single_cost <= double_cost (:106).  Synchronous. */
int flashe_dynamic_masking_cost_dev(flashe_ctx *ctx, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                    uint64_t *single_cost, uint64_t *double_cost);
int flashe_sparse_dense_mask_dev(flashe_ctx *ctx, uint32_t iter, int n_lists, const uint8_t *const *sel_dev,
                                 uint64_t total, uint64_t *out_dev);
int flashe_sparse_dense_mask(flashe_ctx *ctx, uint32_t iter, int n_lists, const uint8_t *const *sel,
                             uint64_t total, uint64_t *out);

/* ---- quantise / batch codec either side of the cipher ------------------ */
/* Fused forms (new): the quantiser is the step immediately before encrypt and after decrypt (QuantizingClient.quantize ->
 * JZFWeights.encrypted, decrypted -> unquantize; jzf_quantize.py:394-491, jzf_weights.py:334-338), so one launch does both and the
 * 8-byte integer never makes a round trip through HBM:
 *   quantize_encrypt: ct[j] = encrypt(_static_quantize_padding_asymmetric(x, alpha, element_bits)[j]), u_dev = the uniform draws;
 *   decrypt_unquantize: out[j] = _static_unquantize_padding_asymmetric(decrypt(in)[j], alpha, element_bits, num_clients) as float64.
 * Bit-identical to the two-call sequences.  Un-batched values only (one quantised value per ciphertext element). */
int flashe_quantize_encrypt_dev(flashe_ctx *ctx, uint32_t iter, uint32_t idx, int scheme, uint64_t n, uint32_t n_jobs,
                                const void *x_dev, int x_is_f64, double alpha, int element_bits, const double *u_dev,
                                uint64_t *ct_dev);
int flashe_decrypt_unquantize_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add,
                                  const uint32_t *minus_idx, int n_minus, uint64_t n, uint32_t n_jobs,
                                  const uint64_t *in_dev, double alpha, int element_bits, int num_clients, double *out_dev);
/* The same for a whole FLATTENED model in one launch -- what a reference job computes (Client.secure_aggregate,
 * jzf_aggregator.py:721-741: QuantizingClient.quantize layer by layer, each layer with its own alpha, then Client.flatten_weights
 * :625-650 and ONE cipher.encrypt over the concatenation; back: cipher.decrypt of the one vector, Client.unflatten_weights :652-671,
 * unquantize layer by layer :887-899).  The PRF counters -- and for int_bits <= 64 the chunks_idx(range(n), n_jobs) chunking --
 * therefore run across the layers: n is the length of the flattened vector, the call covers its elements [first, first + count)
 * (ct_dev / in_dev / u_dev / out_dev address element `first`), and `layers` (HOST array, ascending `start`, layers[0].start == 0)
 * says which alpha -- and, quantize_encrypt, which device array -- flat element j belongs to: the last entry with start <= j.
 * x_dev of a layer points to that layer's OWN first value (the layers need not be contiguous in HBM and may mix float32 and
 * float64); decrypt_unquantize ignores x_dev / x_is_f64.  u_dev[k] = the stochastic-rounding draw of flat element first + k: the
 * reference draws np.random.random(layer.shape) layer by layer in walking order (jzf_quantize.py:61 under :417-462), i.e. one
 * stretch of NumPy's stream in flat order.  Not capturable into a graph (the table is staged per call). */
typedef struct flashe_codec_layer {
    uint64_t start;        /* flat index of the layer's first value */
    const void *x_dev;     /* quantize_encrypt: the layer's float32 / float64 values */
    double alpha;          /* the layer's clipping threshold (QuantizingClient.alpha_list; 1.0 for the sparse job's 'zzz' layer) */
    int32_t x_is_f64;
    int32_t reserved;      /* 0 */
} flashe_codec_layer;
int flashe_quantize_encrypt_model_dev(flashe_ctx *ctx, uint32_t iter, uint32_t idx, int scheme, uint64_t n, uint32_t n_jobs,
                                      uint64_t first, uint64_t count, const flashe_codec_layer *layers, int n_layers,
                                      int element_bits, const double *u_dev, uint64_t *ct_dev);
int flashe_decrypt_unquantize_model_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add,
                                        const uint32_t *minus_idx, int n_minus, uint64_t n, uint32_t n_jobs, uint64_t first,
                                        uint64_t count, const uint64_t *in_dev, const flashe_codec_layer *layers, int n_layers,
                                        int element_bits, int num_clients, double *out_dev);
/* The same back end WITHOUT a decrypt (new): in_dev holds plaintext sums already -- the sparse job's way back, whose decrypt is the
 * sparse minus-mask pass (flashe_sparse_decrypt_dev), not a prefix list; element first + k of the flattened model -> out_dev[k]. */
int flashe_unquantize_model_dev(flashe_ctx *ctx, uint64_t n, uint64_t first, uint64_t count, const uint64_t *in_dev,
                                const flashe_codec_layer *layers, int n_layers, int element_bits, int num_clients, double *out_dev);
/* The BATCHED form of the same job (the paper's main configuration, "batch": true): QuantizingClient.quantize packs
 * batch_size = int_bits / field_bits quantised values (field_bits = element_bits + ceil(log2(num_clients))) into every ciphertext element,
 * first value most significant, EVERY LAYER padded with zeros to whole elements on its own (_static_batching_padding_asymmetric,
 * jzf_quantize.py:162-185 under :436-451); flatten_weights then concatenates the batched layers and the cipher encrypts that vector.
 * quantize_batch_model: one launch from the layers' float values (x_dev per layer) and the model's uniform draws (u_dev[j] = draw of the
 * j-th value in walking order) to the flattened batched plaintext out_dev[n_elems][L] (then flashe_encrypt_dev); unbatch_unquantize_model:
 * one launch from the decrypted flattened vector to the model's float64 values in walking order (_static_unbatching_padding_asymmetric
 * :234-251, the `[:size]` cut of QuantizingClient.unquantize :509-513, _static_unquantize_padding_asymmetric :102-107).
 * n_elems must equal sum over layers of ceil(size / batch_size).  layers is a HOST array. */
typedef struct flashe_batch_layer {
    uint64_t size;         /* values in the layer */
    const void *x_dev;     /* quantize_batch_model: the layer's float32 / float64 values; ignored on the way back */
    double alpha;
    int32_t x_is_f64;
    int32_t reserved;      /* 0 */
} flashe_batch_layer;
int flashe_quantize_batch_model_dev(flashe_ctx *ctx, const flashe_batch_layer *layers, int n_layers, int element_bits, int field_bits,
                                    const double *u_dev, uint64_t n_elems, uint64_t *out_dev);
int flashe_unbatch_unquantize_model_dev(flashe_ctx *ctx, const flashe_batch_layer *layers, int n_layers, int element_bits,
                                        int field_bits, int num_clients, const uint64_t *in_dev, uint64_t n_elems, double *out_dev);
/* QuantizingClient.normalize / unnormalize (jzf_quantize.py:542-564): x <- x + shift in place (normalize passes -mean).  wide: for
 * float32 arrays the addition runs in float64 and is rounded once -- NumPy's loop when the scalar is a float64 (np.mean / np.std
 * results), as opposed to a Python float; bit-exact either way.  mean_std: the per-layer statistics unnormalize records for the next
 * round's alpha, accumulated in float64 by a two-pass parallel reduction; NOT bit-identical to NumPy's summation order (the
 * reference sums a Python-float object array left to right): agreement is to ~1e-12 relative, tests use 1e-10.  Synchronous. */
int flashe_shift_dev(flashe_ctx *ctx, uint64_t n, void *x_dev, int x_is_f64, double shift, int wide);
int flashe_mean_std_dev(flashe_ctx *ctx, uint64_t n, const void *x_dev, int x_is_f64, double *mean, double *stddev);

/* np.random.random(n) ON THE DEVICE, bit for bit (new): the stochastic-rounding draws of _static_quantize_padding_asymmetric
 * (jzf_quantize.py:61, `np.random.random(value.shape)`) come from NumPy's global MT19937 generator; this writes the same n doubles
 * to u_dev -- mt19937_next_double: a = next >> 5, b = next >> 6, (a * 2^26 + b) / 2^53 -- from the generator state the caller hands
 * in (key[624] and pos exactly as np.random.get_state() returns them, HOST memory) and advances that state in place as NumPy would,
 * so that the caller can put it back (np.random.set_state) and host draws continue the same stream.  The stream is cut into
 * substreams of 65,536 doubles whose starting states are found by jump-ahead (the recurrence is linear over GF(2)); they run in
 * parallel, one workgroup each: 1e7 doubles in 0.5 ms (np.random.random: 27-35 ms on a host core), and no 8 B/element upload.
 * Synchronous.  FLASHE_MT_PARALLEL=0 keeps the one-workgroup walk. */
int flashe_mt19937_random_dev(flashe_ctx *ctx, uint32_t key[624], uint32_t *pos, uint64_t n, double *u_dev);
/* Host only: builds the twelve jump polynomials x^(2^17 2^j) mod phi and checks the first against the generator itself. */
int flashe_mt19937_jump_selfcheck(void);
/* Host only: the pass plan of flashe_mt19937_random_dev for n draws from stream position pos -- the largest number of substreams any
 * pass is cut into, the number the jump tables can start (a pass must never exceed it), and the number of passes. */
int flashe_mt19937_plan(uint32_t pos, uint64_t n, uint32_t *max_substreams, uint32_t *substreams_available, uint32_t *passes);

/* _static_quantize_padding_asymmetric -- federatedml/secureprotol/jzf_quantize.py:55-67:
 * q = floor(clip(x, -alpha, alpha) + alpha) * (2^element_bits - 1) / (2 alpha) + u), with numpy's
 * dtype rules (x_is_f64 == 0: x is float32 and every step before "+ u" is a float32 operation).
 * u holds the stochastic-rounding draws in [0, 1) (the reference uses np.random.random). */
int flashe_quantize_dev(flashe_ctx *ctx, uint64_t n, const void *x_dev, int x_is_f64, double alpha,
                        int element_bits, const double *u_dev, uint64_t *q_dev);
int flashe_quantize(flashe_ctx *ctx, uint64_t n, const void *x, int x_is_f64, double alpha,
                    int element_bits, const double *u, uint64_t *q);
/* _static_unquantize_padding_asymmetric -- jzf_quantize.py:102-107:
 * out = v * (2 alpha C) / ((2^element_bits - 1) C) - alpha C in float64, v converted like a Python int
 * (correctly rounded).  v has v_limbs (1 or 2) limbs per element. */
int flashe_unquantize_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *v_dev, int v_limbs, double alpha,
                          int element_bits, int num_clients, double *out_dev);
int flashe_unquantize(flashe_ctx *ctx, uint64_t n, const uint64_t *v, int v_limbs, double alpha,
                      int element_bits, int num_clients, double *out);
/* _static_batching_padding_asymmetric / _static_unbatching_padding_asymmetric -- jzf_quantize.py:162-185,
 * :234-251: batch_size = int_bits // field_bits values (uint64, zero-padded to a multiple) <-> one
 * int_bits-wide element, first value most significant; field_bits = element_bits + ceil(log2(num_clients)).
 * batch: out holds ceil(n / batch_size) elements; unbatch: out holds n_batches * batch_size values. */
int flashe_batch_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *vals_dev, int field_bits, uint64_t *out_dev);
int flashe_batch(flashe_ctx *ctx, uint64_t n, const uint64_t *vals, int field_bits, uint64_t *out);
int flashe_unbatch_dev(flashe_ctx *ctx, uint64_t n_batches, const uint64_t *in_dev, int field_bits, uint64_t *out_dev);
int flashe_unbatch(flashe_ctx *ctx, uint64_t n_batches, const uint64_t *in, int field_bits, uint64_t *out);

/* ---- top-s% sparsifier of the client ------------------------------------ */
/* Client.sparsify for ONE layer -- federatedml/framework/homo/procedure/jzf_aggregator.py:578-623: the k entries
 * of largest |x| (ranked BEFORE the residual is added, as the reference does; ties at the k-th magnitude go to the
 * higher index) are emitted in ascending index order -- loc[k] and vals[k] = x + residual -- and the residual is
 * updated in place (selected positions 0, the others x + residual).  residual may be NULL (treated as zeros, not
 * written).  x / residual / vals are float32 (x_is_f64 == 0) or float64; n < 2^32; k <= n with
 * k = max(1, floor(sparsity * n)) in the reference. */
int flashe_sparsify_dev(flashe_ctx *ctx, uint64_t n, uint64_t k, const void *x_dev, int x_is_f64, void *residual_dev,
                        uint32_t *loc_dev, void *vals_dev);
int flashe_sparsify(flashe_ctx *ctx, uint64_t n, uint64_t k, const void *x, int x_is_f64, void *residual,
                    uint32_t *loc, void *vals);
/* Client.sparsify for EVERY layer of a model in one set of launches (new): layer l is elements [off_l, off_l + n[l]) of the flat
 * vectors x / residual (layers back to back, off_l = n[0] + ... + n[l-1]); its k[l] selected entries go to [koff_l, koff_l + k[l])
 * of the flat outputs loc / vals (koff_l = k[0] + ... + k[l-1]), locations relative to the layer, ascending -- layer by layer exactly
 * what flashe_sparsify gives (same tie rule).  n and k are HOST arrays; 12 (float32) / 20 (float64) launches per MODEL instead of per
 * layer (a ResNet-50 has 161 layers).  The _dev form synchronises the ctx stream once (layer table upload); not inside a graph capture. */
int flashe_sparsify_batch_dev(flashe_ctx *ctx, int n_layers, const uint64_t *n, const uint64_t *k, const void *x_dev, int x_is_f64,
                              void *residual_dev, uint32_t *loc_dev, void *vals_dev);
int flashe_sparsify_batch(flashe_ctx *ctx, int n_layers, const uint64_t *n, const uint64_t *k, const void *x, int x_is_f64,
                          void *residual, uint32_t *loc, void *vals);

/* ---- multi-GPU exchange (RCCL over xGMI; one process per GPU) ----------------------------------------------- */
/* Replaces, inside one node, the arbiter's gather of client models + reduce in Python + broadcast of the aggregate
 * (jzf_aggregator.py:292-308, :404-430, :502-508): every GPU encrypts and locally reduces the clients it hosts, ONE
 * reduce-scatter mod 2^b combines the partial aggregates, every GPU decrypts the slice it owns (the *_range_dev twins)
 * and an all-gather returns the plaintext aggregate to all.  librccl.so is loaded on first use (dlopen), so programs
 * that never call these functions do not depend on it.  All transfers are enqueued on the ctx stream; a flashe_comm may
 * be used with any ctx of the device it was created on (e.g. a side ctx whose stream overlaps the exchange with the
 * AES-bound kernels).  Every rank must issue the same sequence of collective calls.  All "new". */
#define FLASHE_RCCL_ID_BYTES 128                      /* sizeof(ncclUniqueId) */
typedef struct flashe_comm flashe_comm;
int flashe_rccl_unique_id(uint8_t id[FLASHE_RCCL_ID_BYTES]);       /* rank 0 makes it, the launcher hands it to every rank */
int flashe_rccl_init(flashe_ctx *ctx, const uint8_t id[FLASHE_RCCL_ID_BYTES], int rank, int world, flashe_comm **out);
int flashe_rccl_destroy(flashe_comm *comm);
int flashe_rccl_rank(const flashe_comm *comm);
int flashe_rccl_world(const flashe_comm *comm);          /* as RCCL itself reports it (ncclCommCount) */
/* ncclGetVersion of the librccl.so this library loads (e.g. 22703), without creating a communicator: part of the preflight of a
 * multi-GPU launch.  FLASHE_ENODEV when librccl.so cannot be loaded. */
int flashe_rccl_version(int *version);
/* Piece p (bytes long, at send_dev + p * send_stride) goes to rank p; the piece from rank p lands at recv_dev + p * recv_stride.
 * Grouped ncclSend / ncclRecv: on xGMI every GPU pair has its own link, so the W - 1 transfers run concurrently. */
int flashe_rccl_all_to_all(flashe_ctx *ctx, flashe_comm *comm, const void *send_dev, size_t send_stride,
                           void *recv_dev, size_t recv_stride, size_t bytes);
int flashe_rccl_all_gather(flashe_ctx *ctx, flashe_comm *comm, const void *send_dev, void *recv_dev, size_t bytes);
/* The reduce-scatter mod 2^int_bits that RCCL cannot express (no 128-bit type; ncclSum does not wrap at 2^b):
 * partial_dev = world slices of slice_elems elements; after the call out_slice_dev = sum over ranks of slice [rank]
 * (+ extra_dev when given: one more local operand, e.g. the decrypt mask difference).  recv_dev = world x slice_elems
 * elements of scratch. */
int flashe_rccl_reduce_scatter_modadd(flashe_ctx *ctx, flashe_comm *comm, const uint64_t *partial_dev, uint64_t slice_elems,
                                      uint64_t *recv_dev, const uint64_t *extra_dev, uint64_t *out_slice_dev);
/* Host-value all-reduce, synchronous (timing: max over ranks; agreement: min): op 0 = max, 1 = min, 2 = sum. */
/* int_bits <= 64: buf[j] = (sum over ranks of buf[j]) mod 2^b on every rank, in place -- ncclAllReduce(ncclUint64, ncclSum) + mask
 * (the arbiter's reduce jzf_aggregator.py:424-430 across GPUs; wider moduli use flashe_rccl_reduce_scatter_modadd). */
int flashe_rccl_allreduce_modadd_u64(flashe_ctx *ctx, flashe_comm *comm, uint64_t *buf_dev, uint64_t count);
int flashe_rccl_allreduce_f64(flashe_ctx *ctx, flashe_comm *comm, double *value, int op);
int flashe_rccl_barrier(flashe_ctx *ctx, flashe_comm *comm);

#ifdef __cplusplus
}
#endif
#endif /* FLASHE_H */
