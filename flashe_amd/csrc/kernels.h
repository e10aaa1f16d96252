// Internal launch interface between the C ABI (abi.hip) and the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flashe {

// AES-256 expanded key, 60 big-endian words; passed to kernels by value (lands in SGPRs).
struct RoundKeys { uint32_t w[60]; };

// Optional fused codec of a SINGLE-job PRF launch (the quantiser is the step right before encrypt / after
// decrypt, jzf_quantize.py:55-67,102-107): front end = the plaintext of element k is the stochastic-rounded quantisation of a
// float instead of a stored integer; back end = the result is written as the unquantised float64 instead of the integer.
// Saves the 8-byte integer round trip through HBM on either side.  All-zero = off.
// One layer of a FLATTENED model (jzf_aggregator.py:625-650: a job quantises layer by layer, each with its own alpha, then encrypts
// the concatenation as one vector): device-resident table, sorted by start, empty layers left out.
struct CodecLayer {
    uint64_t start;            // flat index of the layer's first value
    const void *x;             // front end: the layer's own float32 / float64 values (x[0] = flat element `start`)
    double p0, p1, p2;         // front end: alpha, scale, den; back end: ac, two_a, uden
    int x_is_f64, pad_;
};
struct Codec {
    const void *x;             // front end: float32 / float64 values (null = off)
    const double *u;           //            one uniform draw in [0, 1) per value
    double alpha, scale, den;  //            q = floor((clip(x, -alpha, alpha) + alpha) * scale / den + u), computed in x's type
    int x_is_f64, n_layers;
    double *fout;              // back end: float64 output (null = off)
    double ac, two_a, uden;    //           out = float(value) * two_a / uden - ac
    // layer table (null = one layer described by the fields above): element k of the launch is flat element k0 + k of the model,
    // its parameters (and, front end, its value) come from the layer that holds it; u and fout stay indexed by k
    const CodecLayer *layers;
    uint64_t k0;
};

// Everything a launch needs that is not a per-call argument.
struct LaunchEnv {
    hipStream_t stream;
    int num_cus;
    const uint32_t *te0_dev;   // Te0 | Te1 | Te2 | Te3, 256 entries each, in device memory (4 KiB)
    const uint32_t *rkw_dev;   // the 60 expanded key words in device memory (scalar-loaded per round by the bit-sliced PRF)
    const uint32_t *rkp_dev;   // packed key planes for the 16-blocks-per-lane bit-sliced PRF: 15 x 64 words
    RoundKeys rk;
    int b;                     // int_bits
    int prf_backend;           // PRF_AUTO / PRF_TABLE / PRF_BITSLICE / PRF_HYBRID
    // PRF_HYBRID: the bit-sliced kernel takes `hybrid_bs_permille` / 1000 of the elements on a second
    // stream while the LDS-table kernel runs the rest -- they saturate different pipes (VALU vs LDS)
    hipStream_t stream2;
    hipEvent_t ev_fork, ev_join;
    int hybrid_bs_permille;
    const Codec *codec;        // host pointer, null except inside the fused quantise / unquantise entry points (single-job launches)
    uint32_t *err_flag;        // host-mapped word: set to 1 when a sparse kernel skips an out-of-range / out-of-order location
    int use_chain;             // 1 (default): jobs over the same range that share a prefix share the PRF stream (prf_chain_kernel)
    int elem32 = 0;            // 1 inside the *_u32_dev entry points (int_bits <= 32): plaintext / ciphertext vectors are uint32 arrays
};

enum { PRF_AUTO = 0, PRF_TABLE = 1, PRF_BITSLICE = 2, PRF_HYBRID = 3, PRF_BITSLICE16 = 4 };

// out = (in? + sum_{k<n_add} term(iter, add[k]) - sum_{k<n_minus} term(iter, minus[k])) mod 2^b.
// add/minus are HOST arrays (copied into the kernel argument block; at most kMaxIdx each).
constexpr int kMaxIdx = 96;
// The launch covers global elements [first, first + count) of an n-element vector; in_dev / out_dev
// are indexed by (element - first).  n and n_jobs define the chunking (b <= 64).
hipError_t launch_prf(const LaunchEnv &env, uint32_t iter,
                      const uint32_t *add, int n_add, const uint32_t *minus, int n_minus,
                      uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count,
                      const uint64_t *in_dev, int in_limbs, uint64_t *out_dev);

// Several independent encrypts of equal length in ONE launch: vector v is encrypted with prefix idx[v]
// (and idx[v] + 1 when `dbl`); at most kMaxBatch vectors per launch.
constexpr int kMaxBatch = 32;      // jobs per launch (general table)
constexpr int kMaxUniformBatch = 128;   // equal-length whole vectors per launch (compact table, b > 64)
hipError_t launch_prf_batch(const LaunchEnv &env, uint32_t iter, bool dbl, int n_vec, const uint32_t *idx,
                            const uint64_t *const *in_dev, int in_limbs, uint64_t *const *out_dev, uint64_t n,
                            uint32_t n_jobs);
// The same batch as ONE chain that also writes sum_v out[v] mod 2^b to sum_out_dev (the local partial aggregate).  Returns
// hipErrorNotSupported -- nothing launched -- unless the batch is one run of consecutive cipher indices of the double mask with
// int_bits > 64, at most kMaxUniformBatch vectors, long enough to fill the chip: the caller then encrypts and reduces separately.
hipError_t launch_prf_batch_sum(const LaunchEnv &env, uint32_t iter, int n_vec, const uint32_t *idx, const uint64_t *const *in_dev,
                                int in_limbs, uint64_t *const *out_dev, uint64_t *sum_out_dev, uint64_t n, uint32_t n_jobs, uint64_t first,
                                uint64_t count);
// launch_prf_batch on elements [first, first + count) of the n-element vectors (pointers address element `first`): what a GPU that owns
// an element slice of every client's vector runs
hipError_t launch_prf_batch_range(const LaunchEnv &env, uint32_t iter, bool dbl, int n_vec, const uint32_t *idx, const uint64_t *const *in_dev,
                                  int in_limbs, uint64_t *const *out_dev, uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count);

// General form of the batched launch: out[k] = in[k] + term(iter, add_idx, first + k)
// - [dbl] term(iter, minus_idx, first + k) for k < count; in_dev may be null (zeros); pointers address element `first`.
// n_in > 1 (in_limbs == 2 only): the input is the mod-2^b sum of n_in vectors, vector c at in_dev + c * in_stride
// limbs -- the arbiter's reduce fused into the decrypt of its result; sum_out_dev (optional) receives that sum.
struct PrfJob {
    uint32_t add_idx, minus_idx;
    uint64_t first, count;
    const uint64_t *in_dev;
    int in_limbs;
    uint64_t *out_dev;
    uint32_t n_in = 1;
    uint64_t in_stride = 0;
    uint64_t *sum_out_dev = nullptr;
};
hipError_t launch_prf_jobs(const LaunchEnv &env, uint32_t iter, bool dbl, int n_entries, const PrfJob *jobs, uint64_t n,
                           uint32_t n_jobs);

// Chained jobs (int_bits > 64): `n_out` outputs over elements [first, first + count) that share their PRF streams.
// Double mask (single == false): n_out + 1 prefixes, out[c] = in[c] + term(idx[c]) - term(idx[c + 1]) -- consecutive clients
// (idx[c + 1] = idx[c] + 1, jzf_flashe.py:349-353) cost n_out + 1 streams instead of 2 n_out; a plain (add, minus) job is a chain
// of one output.  single == true: n_out prefixes, out[c] = in[c] + term(idx[c]).  idx, in_dev, out_dev are HOST arrays; in_dev (or
// an entry of it) may be null = zeros; device pointers address element `first`.  Returns hipErrorNotSupported when a chain cannot
// take this path (a range that straddles a 2^32 counter boundary; int_bits <= 64: a vector of 2^32 elements or more, a fused codec):
// the caller then uses the job-table kernels.  n, n_jobs: the whole vector's length and chunking (int_bits <= 64 counters).
struct PrfChain {
    const uint32_t *idx;
    int n_out;
    bool single;
    uint64_t first, count;
    const uint64_t *const *in_dev;
    int in_limbs;
    uint64_t *const *out_dev;
    uint64_t *sum_out_dev = nullptr;   // optional (int_bits > 64, n_out <= kMaxLinks): receives sum_c out[c] mod 2^b, written by the same launch
};
hipError_t launch_prf_chains(const LaunchEnv &env, uint32_t iter, int n_chains, const PrfChain *chains, uint64_t n, uint32_t n_jobs);

hipError_t launch_combine(const LaunchEnv &env, uint64_t n, const uint64_t *in_dev, int in_limbs,
                          const uint64_t *add_dev, const uint64_t *minus_dev, uint64_t *out_dev);

// n_vec combines of equal length in as few launches as possible (host arrays of device pointers; add_dev / minus_dev or single
// entries of them may be null)
hipError_t launch_combine_batch(const LaunchEnv &env, uint64_t n, int n_vec, const uint64_t *const *in_dev, int in_limbs,
                                const uint64_t *const *add_dev, const uint64_t *const *minus_dev, uint64_t *const *out_dev);
// the same and sum_out = sum_v out[v] mod 2^b in one pass (the online encrypts with precomputed masks + the reduce of what they wrote)
hipError_t launch_combine_batch_sum(const LaunchEnv &env, uint64_t n, int n_vec, const uint64_t *const *in_dev, int in_limbs,
                                    const uint64_t *const *add_dev, const uint64_t *const *minus_dev, uint64_t *const *out_dev, uint64_t *sum_out_dev,
                                    const uint64_t *dec_add_dev = nullptr, const uint64_t *dec_minus_dev = nullptr, uint64_t *dec_out_dev = nullptr);

// Operand pointers of one reduce pass travel in the kernel argument block (scalar loads).
constexpr int kMaxOps = 64;
struct PtrTable { const uint64_t *p[kMaxOps]; };

// ops: HOST array of C <= kMaxOps device pointers.  out may alias an operand.
// b <= 64: out = (sum of the C operands + term(add) - [has_minus] term(minus)) mod 2^b on elements [first, first + count) in one pass
// (pointers address element `first`; agg_out_dev may be null).  hipErrorNotSupported: reduce and decrypt in two launches instead.
// (env.elem32: the operands are uint32 arrays; out_elem_bytes = 4: so are agg_out_dev and out_dev)
hipError_t launch_small_reduce_decrypt(const LaunchEnv &env, uint32_t iter, uint32_t add_idx, bool has_minus, uint32_t minus_idx, uint64_t n,
                                       uint32_t n_jobs, uint64_t first, uint64_t count, int C, const uint64_t *const *ops, uint64_t *agg_out_dev,
                                       uint64_t *out_dev, int out_elem_bytes = 8);
// uint32 <-> uint64 element arrays (zero extension / truncation), n elements
hipError_t launch_widen_u32(const LaunchEnv &env, uint64_t n, const uint32_t *in_dev, uint64_t *out_dev);
hipError_t launch_narrow_u32(const LaunchEnv &env, uint64_t n, const uint64_t *in_dev, uint32_t *out_dev);
hipError_t launch_aggregate_elem(const LaunchEnv &env, int C, const uint64_t *const *ops,
                                 uint64_t n, uint64_t *out_dev);
// the same on uint32 element arrays (compact layout, int_bits <= 32)
hipError_t launch_aggregate_elem_u32(const LaunchEnv &env, int C, const uint32_t *const *ops, uint64_t n, uint32_t *out_dev);

// summaries_dev: scratch of at least packed_num_blocks(n_limbs) uint32 words.
// out must NOT alias an operand (blocks re-read their left neighbour's inputs).
uint64_t packed_num_blocks(uint64_t n_limbs);
hipError_t launch_aggregate_packed(const LaunchEnv &env, int C, const uint64_t *const *ops,
                                   uint64_t n_limbs, uint64_t total_bits, uint64_t *out_dev,
                                   uint32_t *summaries_dev);

// helpers for a packed reduce cut into limb slices across GPUs: (low limb, body-all-ones, carry limb) of a
// slice sum, and the in-place carry-in ripple
hipError_t launch_packed_probe(const LaunchEnv &env, uint64_t n_limbs, const uint64_t *x_dev, uint64_t *info_dev);
hipError_t launch_packed_add_carry(const LaunchEnv &env, uint64_t n_limbs, uint64_t total_bits, uint64_t cin, uint64_t *x_dev);
// the same with the carry-in derived on the device from the probe triples of the n_below slices underneath (infos_dev: 3 words each)
// (stride_words: from one slice's triple to the next more significant one's; negative = the gathered triples run from the most
// significant slice down and infos_dev points at the lowest slice's)
hipError_t launch_packed_resolve_carry(const LaunchEnv &env, uint64_t n_limbs, uint64_t total_bits, const uint64_t *infos_dev, int n_below,
                                       uint64_t *x_dev, int stride_words = 3);

hipError_t launch_pack(const LaunchEnv &env, uint64_t n, const uint64_t *in_dev, uint64_t *out_dev);
hipError_t launch_unpack(const LaunchEnv &env, uint64_t n, const uint64_t *in_dev, uint64_t *out_dev);

hipError_t launch_fill(const LaunchEnv &env, uint64_t n, uint64_t lo, uint64_t hi, uint64_t *out_dev);
// locations >= total are skipped and reported through env.err_flag (the reference raises IndexError, jzf_aggregator.py:150-165)
hipError_t launch_scatter(const LaunchEnv &env, uint64_t total, uint64_t k, const uint32_t *loc_dev,
                          const uint64_t *vals_dev, uint64_t *out_dev, bool accumulate,
                          uint64_t sub_lo = 0, uint64_t sub_hi = 0);
// Sparse reduce over strictly increasing location lists (LDS-staged, the dense output is written once):
// out[p] = from[p] +/- sum_{c, q: loc[c][q] == p} (vals[c][q] - sub[c]) mod 2^b for every p < total, where from = src_dev when it
// is given (it may be out_dev: accumulate in place) and the constant base otherwise; negate subtracts the sum (a decrypt:
// aggregate - minus-mask in the pass that builds the mask).  At most kMaxScatter clients per call;
// start_dev = (span_count(total, span) + 1) * C words of scratch.
constexpr int kMaxScatter = 64;
// positions per span: the plain reduce keeps 64 KiB of accumulators per workgroup, the passes with the PRF inside (launch_span_prf)
// what the AES tables leave of a CU's LDS.  A table of span bounds serves the kernels of ONE of the two sizes (launch_span_bounds fills both in one pass).
#ifndef FLASHE_SPAN_FUSED
#define FLASHE_SPAN_FUSED 1752
#endif
constexpr int kSpanReduce = 4096, kSpanFused = FLASHE_SPAN_FUSED;
uint64_t span_count(uint64_t total, int span);
// words of a bounds table that is large enough for either span size
inline size_t span_table_words(uint64_t total, int C) { return static_cast<size_t>(span_count(total, kSpanFused) + 1) * static_cast<size_t>(C); }
// (bounds_ready: start_dev already holds the bounds of exactly these lists -- launch_span_bounds -- and the pass that computes them is skipped)
hipError_t launch_span_reduce(const LaunchEnv &env, int C, const uint32_t *const *loc_dev, const uint64_t *const *vals_dev,
                              const uint64_t *k, const uint64_t *sub, uint64_t base_lo, uint64_t base_hi, uint64_t total,
                              uint32_t *start_dev, const uint64_t *src_dev, bool negate, uint64_t *out_dev, bool bounds_ready = false);
// the first list entry of each of the C <= kMaxScatter clients in every span, at kSpanReduce and / or kSpanFused positions per span
// (either table may be null): (span_count(total, span) + 1) * C words each, one pass over the lists
hipError_t launch_span_bounds(const LaunchEnv &env, int C, const uint32_t *const *loc_dev, const uint64_t *k, uint64_t total, uint32_t *start_reduce_dev,
                              uint32_t *start_fused_dev);
// The span reduce with the PRF inside (int_bits > 64, table PRF; one persistent launch): entry q of client c contributes
// term(iter, idx[c], q) -- pt_dev null: out[p] = from[p] +/- the sum of the masks at p (sparse minus-mask / decrypt); pt_dev given:
// ct_dev[c][q] = (pt_dev[c][q] + mask) mod 2^b is stored (where ct_dev[c] is not null) and out[p] = from[p] + sum (ct - sub[c]);
// pt_limbs = 1: the plaintexts are uint64 arrays.  start_dev: bounds of exactly these lists at kSpanFused.  [first, first + count):
// the positions this launch covers (span-aligned, see stream.hip) -- src_dev / out_dev address position `first`; entries outside it
// are left alone (their ciphertexts are not written).
hipError_t launch_span_prf(const LaunchEnv &env, uint32_t iter, int C, const uint32_t *idx, const uint32_t *const *loc_dev, const uint64_t *k,
                           const uint64_t *const *pt_dev, int pt_limbs, uint64_t *const *ct_dev, const uint64_t *sub, uint64_t base_lo, uint64_t base_hi,
                           uint64_t total, const uint32_t *start_dev, const uint64_t *src_dev, bool negate, uint64_t *out_dev, uint64_t first, uint64_t count);
#ifdef FLASHE_TUNING
hipError_t span_prf_cycles(unsigned long long *out8, bool reset);      // phase cycle sums of span_prf_kernel's workgroup 0 (FLASHE_SPAN_PROBE=9)
#endif
// Sparse + double mask (jzf_flashe.py:388-426, :155-225): compact (add, minus) mask values of a group of nc <= kMaxScatter clients
// c0 .. c0 + nc - 1 at their own sorted locations -- entry q of client c gets term(c + 1, p) unless client c + 1 holds p, and
// term(c, p) unless client c - 1 holds p (dense-position counters, one chunk).  loc / k carry nc + 2 entries: the group's lists
// framed by the neighbouring clients' (null / 0 where there is none).
hipError_t launch_sparse_edge_prf(const LaunchEnv &env, uint32_t iter, int nc, uint32_t c0, const uint32_t *const *loc_with_neighbours,
                                  const uint64_t *k_with_neighbours, uint64_t *const *va_dev, uint64_t *const *vm_dev);
// Adds to *count_dev the number of list entries of clients 0 .. nc - 1 whose position the NEXT client's (sorted) list holds too.
// loc_with_next / k_with_next carry nc + 1 entries: the group's lists followed by the client after the group (null / 0 if none).
hipError_t launch_shared_positions(const LaunchEnv &env, int nc, const uint32_t *const *loc_with_next, const uint64_t *k_with_next,
                                   unsigned long long *count_dev);
// out[p] = (out[p] + (sel[p] ? stream[p] : 0)) mod 2^b
hipError_t launch_sel_accumulate(const LaunchEnv &env, uint64_t total, const uint8_t *sel_dev,
                                 const uint64_t *stream_dev, uint64_t *out_dev);

// Quantise / batch codec
// descriptors for LaunchEnv::codec
Codec codec_quantize_front(const void *x_dev, bool is_f64, double alpha, int bits, const double *u_dev);
void codec_unquantize_back(Codec *c, double alpha, int bits, int num_clients, double *out_dev);
// the same per layer of a flattened model, as entries of the device table Codec::layers points to
CodecLayer codec_layer_front(uint64_t start, const void *x_dev, bool is_f64, double alpha, int bits);
CodecLayer codec_layer_back(uint64_t start, double alpha, int bits, int num_clients);
// x <- x + shift in place (normalize / unnormalize, jzf_quantize.py:542-564).  f32 arrays: wide = the add runs in float64 and is
// rounded once (NumPy's loop for a float64 scalar operand), otherwise in float32
hipError_t launch_shift(const LaunchEnv &env, uint64_t n, void *x_dev, bool is_f64, double shift, bool wide);
// partial sums for mean / std: part[g] = sum over block g's elements of (x - center)^pow, pow = 1 or 2, float64; returns the grid size
int moments_grid(const LaunchEnv &env, uint64_t n);
hipError_t launch_moment(const LaunchEnv &env, uint64_t n, const void *x_dev, bool is_f64, double center, int pow, double *part_dev);
hipError_t launch_quantize(const LaunchEnv &env, uint64_t n, const void *x_dev, bool is_f64, double alpha, int bits,
                           const double *u_dev, uint64_t *q_dev);
// the codec back end over a flattened model without a PRF stream: out[k] = unquantise(v[k]) with the layer table of cq (k0 = first element)
hipError_t launch_unquantize_model(const LaunchEnv &env, uint64_t count, const uint64_t *v_dev, const Codec &cq, double *out_dev);
hipError_t launch_unquantize(const LaunchEnv &env, uint64_t n, const uint64_t *v_dev, int v_limbs, double alpha, int bits,
                             int num_clients, double *out_dev);
hipError_t launch_batch(const LaunchEnv &env, uint64_t n, const uint64_t *vals_dev, int field_bits, uint64_t *out_dev);
// The BATCHED codec over a whole model (jzf_quantize.py:436-451 under :417-462, then jzf_aggregator.py:723): every layer is quantised
// with its own alpha and batched ON ITS OWN (zero padded to whole elements), the batched layers lie back to back in the flattened
// vector.  Device table, one entry per non-empty layer, ascending elem_start.
struct BatchLayer {
    uint64_t elem_start;       // first batched element of the layer in the flattened vector
    uint64_t value_start;      // index of the layer's first value among all values of the model (its draw / its float output)
    uint64_t size;             // values in the layer
    const void *x;             // front end: the layer's float32 / float64 values
    double p0, p1, p2;         // front end: alpha, scale, den; back end: ac, two_a, uden
    int x_is_f64, pad_;
};
// out[e] = sum_t quantize(value bs * e_local + t) << (field_bits * (bs - 1 - t)) for the n_elems batched elements (bs = int_bits / field_bits)
hipError_t launch_quantize_batch_model(const LaunchEnv &env, const BatchLayer *layers_dev, int n_layers, int field_bits, const double *u_dev,
                                       uint64_t n_elems, uint64_t *out_dev);
// out[value_start + j] = unquantize(field t of element elem_start + j / bs), the inverse walk, for the n_values values of the model
hipError_t launch_unbatch_unquantize_model(const LaunchEnv &env, const BatchLayer *layers_dev, int n_layers, int field_bits, const uint64_t *in_dev,
                                           uint64_t n_values, double *out_dev);
BatchLayer batch_layer_front(uint64_t elem_start, uint64_t value_start, uint64_t size, const void *x_dev, bool is_f64, double alpha, int bits);
BatchLayer batch_layer_back(uint64_t elem_start, uint64_t value_start, uint64_t size, double alpha, int bits, int num_clients);
hipError_t launch_unbatch(const LaunchEnv &env, uint64_t nb, const uint64_t *in_dev, int field_bits, uint64_t *out_dev);

// Top-k of every layer of a model in one set of launches: the layers lie back to back in flat buffers.  The caller fills the host
// layer table with sparsify_batch_layout (-> number of 1024-element blocks), copies its sparsify_batch_desc_bytes(L) bytes to the START
// of a device workspace of sparsify_batch_workspace_bytes(L, blocks), then launches.
size_t sparsify_batch_workspace_bytes(int L, uint64_t n_blocks);
size_t sparsify_batch_desc_bytes(int L);
uint64_t sparsify_batch_layout(int L, const uint64_t *n, const uint64_t *k, void *desc_host);
hipError_t launch_sparsify_batch(const LaunchEnv &env, int L, uint64_t n_blocks, const void *x, bool is_f64, void *residual, uint32_t *loc, void *vals,
                                 void *ws);
// Top-k sparsifier; ws = device workspace of sparsify_workspace_bytes(n).
size_t sparsify_workspace_bytes(uint64_t n);
hipError_t launch_sparsify(const LaunchEnv &env, uint64_t n, uint64_t k, const void *x, bool is_f64, void *residual, uint32_t *loc,
                           void *vals, void *ws);

// Device KAT: encrypts `nblk` 16-byte blocks (big-endian words in) with the PRF core.
hipError_t launch_aes_blocks(const LaunchEnv &env, uint32_t nblk, const uint32_t *in_words_dev,
                             uint32_t *out_words_dev);

// b > 64, one add and at most one minus prefix, C <= kMaxOps operands ANYWHERE in HBM (pointers address element `first`): out = sum of the
// operands + term(add) - term(minus) on [first, first + count) in ONE launch; agg_out_dev (may be null) receives the sum.
// hipErrorNotSupported (nothing launched): b <= 64, too many operands, a range across a 2^32 counter boundary, another PRF backend.
hipError_t launch_reduce_decrypt_ptrs(const LaunchEnv &env, uint32_t iter, uint32_t add_idx, bool has_minus, uint32_t minus_idx, uint64_t first,
                                      uint64_t count, int C, const uint64_t *const *ops, uint64_t *agg_out_dev, uint64_t *out_dev);
#ifdef FLASHE_TUNING
// tuning build only: the two-workgroups-per-CU experiment on the reduce fused with the decrypt (kernels.hip); variant 0 = 1024 threads,
// full tables (correct results); 1 = 2 x 512 threads on half-size aliased tables (timing probe, wrong results); 2 = 1024 threads on them
hipError_t launch_reduce_decrypt_probe(const LaunchEnv &env, int variant, uint32_t iter, uint32_t add_idx, uint32_t minus_idx, int C,
                                       const uint64_t *const *ops, uint64_t n, uint64_t *out_dev);
#endif

}  // namespace flashe
