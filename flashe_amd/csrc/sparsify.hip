// Top-k sparsifier of the FLASHE client: Client.sparsify for one layer,
// federatedml/framework/homo/procedure/jzf_aggregator.py:578-623, as three streaming stages on the GPU:
//   1. radix select (8-bit digits, most significant first) of the k-th largest |x|  -> threshold key T and
//      how many elements equal to T still belong to the top k;
//   2. per-1024-element block counts of (key > T, key == T) + one exclusive scan over the blocks;
//   3. ordered compaction: selected indices / (x + residual) values in ascending index order, residual updated
//      in the same pass (selected positions reset to 0, the others keep x + residual).
// Ranking uses |x| BEFORE the residual is added, as the reference does; ties at T go to the higher index.
// HBM-bound: (digits + 2) reads of x, one read + write of the residual.
#include "kernels.h"

namespace flashe {

constexpr int kSpThreads = 1024;

struct SelectState {            // lives in device memory, updated by pick_digit_kernel
    unsigned long long prefix;  // key bits fixed so far (in place)
    unsigned long long mask;    // which key bits are fixed
    unsigned long long remaining;   // how many of the elements matching `prefix` are still to be taken
    unsigned long long total_eq;    // after the scan: number of elements with key == T
};

template <typename T> struct KeyOf;
template <> struct KeyOf<float> {
    typedef uint32_t type;
    static constexpr int bits = 32;
    __device__ static uint32_t get(float v) { return __float_as_uint(v) & 0x7fffffffu; }
};
template <> struct KeyOf<double> {
    typedef uint64_t type;
    static constexpr int bits = 64;
    __device__ static uint64_t get(double v) { return static_cast<uint64_t>(__double_as_longlong(v)) & 0x7fffffffffffffffull; }
};

// One histogram vote per lane with `on`.  The high digits of real weights are concentrated (a layer's magnitudes share a few
// exponents): sixty-four lanes adding 1 to the same LDS word are sixty-four serialised atomics.  The digits the wave's lanes share
// with its first few active lanes are therefore counted by ballot and added once; whatever is left votes lane by lane.
__device__ __forceinline__ void sp_vote(uint32_t *lh, bool on, uint32_t digit)
{
    const int lane = threadIdx.x & 63;
    unsigned long long active = __ballot(on);
#pragma unroll 1
    for (int it = 0; it < 4 && active; it++) {
        const int leader = static_cast<int>(__ffsll(static_cast<unsigned long long>(active))) - 1;
        const uint32_t d0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(digit), leader));
        const unsigned long long same = __ballot(on && digit == d0) & active;
        if (__popcll(same) < 4) break;                            // spread out: not worth another round
        if (lane == leader) atomicAdd(&lh[d0], static_cast<uint32_t>(__popcll(same)));
        active &= ~same;
    }
    if ((active >> lane) & 1ull) atomicAdd(&lh[digit], 1u);
}

// The digit of this pass from the layer's 256-bin histogram: the highest d with count(bins > d) < remaining <= count(bins >= d).
// 256 threads: suffix sums by a Hillis-Steele scan in LDS (a single thread walking the bins is 256 dependent loads, ~15 us).
__device__ __forceinline__ void sp_pick(SelectState *st, int shift, uint32_t *hist)
{
    __shared__ unsigned long long suf[256];
    __shared__ int s_d;
    const int t = threadIdx.x;                                    // 0 .. 255
    suf[t] = hist[255 - t];                                       // reversed: suf[t] will hold count(bins >= 255 - t)
    if (t == 0) s_d = st->remaining == 0 ? 255 : 0;             // (nothing left to take: the walk from the top stops at once)
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned long long a = t >= off ? suf[t - off] : 0ull;
        __syncthreads();
        suf[t] += a;
        __syncthreads();
    }
    const unsigned long long remaining = st->remaining;
    // bin d = 255 - t is the one iff count(bins > d) < remaining <= count(bins >= d); bin 0 takes what no higher bin covers
    const unsigned long long above = t ? suf[t - 1] : 0ull;
    if (above < remaining && (suf[t] >= remaining || t == 255)) s_d = 255 - t;     // exactly one thread (suffix sums are monotone)
    __syncthreads();
    const int d = s_d;
    if (t == 0) {
        const unsigned long long acc = d < 255 ? suf[254 - d] : 0ull;              // count(bins > d)
        st->remaining = remaining - acc;                                            // everything in the higher bins is taken whole
        st->prefix |= static_cast<unsigned long long>(d) << shift;
        st->mask |= 255ull << shift;
    }
    __syncthreads();
    hist[t] = 0;
}

// The histogram pass over ONE layer (launch_sparsify): a grid of at most two 1,024-lane workgroups per CU strides over the layer with
// four loads in flight per lane.  With a single layer every workgroup ends on the SAME 256 words of the global histogram, so fewer,
// larger workgroups beat the 256-lane trips of the model-wide pass below (25.5 M float32 values: 4 x 50 against 4 x 69 us).
template <typename T>
__global__ __launch_bounds__(kSpThreads) void sp_hist_kernel(uint64_t n, const T *x, const SelectState *st, int shift, uint32_t *hist)
{
    __shared__ uint32_t lh[256];
    if (threadIdx.x < 256) lh[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix, mask = st->mask;
    // four loads in flight per lane (one per step left the pass latency bound: 75 us for a 102 MB read)
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kSpThreads;
    for (uint64_t i0 = static_cast<uint64_t>(blockIdx.x) * kSpThreads + threadIdx.x; i0 < n; i0 += 4 * stride) {
        T v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = i0 + u * stride < n ? x[i0 + u * stride] : T(0);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const unsigned long long key = KeyOf<T>::get(v[u]);
            sp_vote(lh, i0 + u * stride < n && (key & mask) == prefix, static_cast<uint32_t>((key >> shift) & 255u));
        }
    }
    __syncthreads();
    if (threadIdx.x < 256 && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// ---- every layer of a model in one set of launches ---------------------------------------------------------------------------
// Client.sparsify walks the layers (jzf_aggregator.py:585-613): one top-k per layer.  Layer by layer that is 12 (float32) or 20
// (float64) launches each -- a ResNet-50 has 161 layers, most of them far smaller than what fills the chip, so the model costs ~15 ms
// of launches for 0.4 ms of work.  Here the layers lie back to back in ONE flat buffer, the 1024-element blocks of the stages never
// cross a layer boundary (block b belongs to the layer l with blk0[l] <= b < blk0[l] + nb[l]), every layer has its own select state
// and histogram, and each stage is ONE launch over all layers: 12 / 20 launches per model.
struct SpLayer {                // per layer, in device memory
    uint64_t off, n, k, koff;   // first element in the flat buffers, elements, entries to select, first entry in the flat outputs
    uint32_t blk0, nb;          // first 1024-element block, number of blocks
};

__device__ __forceinline__ int spb_layer_of(const SpLayer *ly, int L, uint32_t b)
{
    int lo = 0, hi = L - 1;
    while (lo < hi) {           // the last layer whose first block is <= b
        const int mid = (lo + hi + 1) >> 1;
        if (ly[mid].blk0 <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// block b -> its layer, once per call (every stage then reads one word instead of searching the layer table)
__global__ void spb_map_kernel(int L, const SpLayer *ly, uint32_t n_blocks, uint32_t *blk_layer)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_blocks) blk_layer[b] = static_cast<uint32_t>(spb_layer_of(ly, L, b));
}

__global__ void spb_init_kernel(const SpLayer *ly, SelectState *st, uint32_t *hist)
{
    const int l = blockIdx.x;
    if (threadIdx.x == 0) { st[l].prefix = 0; st[l].mask = 0; st[l].remaining = ly[l].k; st[l].total_eq = 0; }
    hist[l * 256 + threadIdx.x] = 0;
}

// ---- the streaming passes of the batch form (round 5): a lane owns FOUR CONSECUTIVE elements of a 1024-element block, one 16-byte load
// per block (two for float64), a workgroup of 256 lanes takes kSpPer consecutive blocks with every load in flight before the first is
// used.  (Round 3-4: one element per lane, one load per trip: every pass over a ResNet-50-sized model ran at a third to a half of
// what its bytes need -- 161 / 57 / 138 us for passes whose reads take 25 / 25 / 75 us.)
constexpr int kSpWg = 256;              // lanes per workgroup of the streaming passes = 1024 elements per block / 4 per lane
constexpr uint32_t kSpPer = 4;          // consecutive blocks per workgroup trip

// elements i .. i + 3 of a layer of n elements at p (i a multiple of 4); the vector form never reads beyond the layer
template <typename T>
__device__ __forceinline__ void sp_load4(const T *__restrict__ p, uint64_t i, uint64_t n, T (&v)[4])
{
    if (i + 4 <= n) {
        if constexpr (sizeof(T) == 4) {
            typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            const f32x4_a4 q = *reinterpret_cast<const f32x4_a4 *>(p + i);
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
        } else {
            typedef double f64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
            const f64x2_a8 q0 = *reinterpret_cast<const f64x2_a8 *>(p + i), q1 = *reinterpret_cast<const f64x2_a8 *>(p + i + 2);
            v[0] = q0[0]; v[1] = q0[1]; v[2] = q1[0]; v[3] = q1[1];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = i + j < n ? p[i + j] : T(0);
    }
}
template <typename T>
__device__ __forceinline__ void sp_store4(T *__restrict__ p, uint64_t i, uint64_t n, const T (&v)[4])
{
    if (i + 4 <= n) {
        if constexpr (sizeof(T) == 4) {
            typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
            f32x4_a4 q;
            q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3];
            *reinterpret_cast<f32x4_a4 *>(p + i) = q;
        } else {
            typedef double f64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
            f64x2_a8 q0, q1;
            q0[0] = v[0]; q0[1] = v[1]; q1[0] = v[2]; q1[1] = v[3];
            *reinterpret_cast<f64x2_a8 *>(p + i) = q0; *reinterpret_cast<f64x2_a8 *>(p + i + 2) = q1;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) if (i + j < n) p[i + j] = v[j];
    }
}

// inclusive prefix sum over the 64 lanes (row shifts, then the two row broadcasts of the GFX9 DPP set)
__device__ __forceinline__ uint32_t sp_wave_scan(uint32_t v)
{
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);
    return v;
}

// The histogram pass: persistent workgroups walk the blocks in trips of kSpPer and keep ONE LDS histogram while the layer stays the
// same (it goes to the layer's global histogram when the layer changes and at the end: a few hundred global atomics per workgroup
// instead of 256 per eight blocks -- thousands of workgroups adding to the same 256 words queued up behind each other).
template <typename T>
__global__ __launch_bounds__(kSpWg) void spb_hist_kernel(const uint32_t *blk_layer, const SpLayer *ly, uint32_t n_blocks, const T *x,
                                                           const SelectState *st, int shift, uint32_t *hist)
{
    __shared__ uint32_t lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    int cur = -1;
    const uint32_t n_trips = (n_blocks + kSpPer - 1) / kSpPer;
    for (uint32_t trip = blockIdx.x; trip < n_trips; trip += gridDim.x) {
        int lay[kSpPer];
        uint64_t base[kSpPer], nl[kSpPer];
        T v[kSpPer][4];
#pragma unroll
        for (uint32_t s = 0; s < kSpPer; s++) {
            const uint32_t b = trip * kSpPer + s;
            lay[s] = b < n_blocks ? __builtin_amdgcn_readfirstlane(static_cast<int>(blk_layer[b])) : -1;
            const int l = lay[s] < 0 ? 0 : lay[s];
            base[s] = static_cast<uint64_t>(b - ly[l].blk0) * kSpThreads + 4u * threadIdx.x;
            nl[s] = lay[s] < 0 ? 0 : ly[l].n;
            sp_load4(x + ly[l].off, base[s], nl[s], v[s]);
        }
#pragma unroll
        for (uint32_t s = 0; s < kSpPer; s++) {
            const int l = lay[s];
            if (l < 0) break;
            if (l != cur) {
                __syncthreads();                                  // every vote of the layer that ends here is in
                if (cur >= 0 && lh[threadIdx.x]) atomicAdd(&hist[cur * 256 + threadIdx.x], lh[threadIdx.x]);
                lh[threadIdx.x] = 0;
                cur = l;
                __syncthreads();
            }
            const unsigned long long prefix = st[l].prefix, mask = st[l].mask;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned long long key = static_cast<unsigned long long>(KeyOf<T>::get(v[s][j]));
                sp_vote(lh, base[s] + j < nl[s] && (key & mask) == prefix, static_cast<uint32_t>((key >> shift) & 255u));
            }
        }
    }
    __syncthreads();
    if (cur >= 0 && lh[threadIdx.x]) atomicAdd(&hist[cur * 256 + threadIdx.x], lh[threadIdx.x]);
}

__global__ void spb_pick_digit_kernel(SelectState *st_all, int shift, uint32_t *hist_all)
{
    sp_pick(st_all + blockIdx.x, shift, hist_all + blockIdx.x * 256);
}

template <typename T>
__global__ __launch_bounds__(kSpWg) void spb_count_kernel(const uint32_t *blk_layer, const SpLayer *ly, uint32_t n_blocks, const T *x, const SelectState *st,
                                                            uint32_t *blk_gt, uint32_t *blk_eq)
{
    __shared__ uint32_t c[kSpPer][2];
    if (threadIdx.x < 2 * kSpPer) c[threadIdx.x >> 1][threadIdx.x & 1] = 0;
    int lay[kSpPer];
    uint64_t base[kSpPer], nl[kSpPer];
    T v[kSpPer][4];
#pragma unroll
    for (uint32_t s = 0; s < kSpPer; s++) {
        const uint32_t b = blockIdx.x * kSpPer + s;
        lay[s] = b < n_blocks ? __builtin_amdgcn_readfirstlane(static_cast<int>(blk_layer[b])) : -1;
        const int l = lay[s] < 0 ? 0 : lay[s];
        base[s] = static_cast<uint64_t>(b - ly[l].blk0) * kSpThreads + 4u * threadIdx.x;
        nl[s] = lay[s] < 0 ? 0 : ly[l].n;
        sp_load4(x + ly[l].off, base[s], nl[s], v[s]);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t s = 0; s < kSpPer; s++) {
        if (lay[s] < 0) break;
        const unsigned long long thr = st[lay[s]].prefix;
        uint32_t cg = 0, ce = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned long long key = static_cast<unsigned long long>(KeyOf<T>::get(v[s][j]));
            const bool live = base[s] + j < nl[s];
            cg += live && key > thr; ce += live && key == thr;
        }
        const uint32_t tg = sp_wave_scan(cg), te = sp_wave_scan(ce);           // lane 63 holds the wave's totals
        if ((threadIdx.x & 63) == 63) { atomicAdd(&c[s][0], tg); atomicAdd(&c[s][1], te); }
    }
    __syncthreads();
    if (threadIdx.x < kSpPer && blockIdx.x * kSpPer + threadIdx.x < n_blocks) {
        blk_gt[blockIdx.x * kSpPer + threadIdx.x] = c[threadIdx.x][0];
        blk_eq[blockIdx.x * kSpPer + threadIdx.x] = c[threadIdx.x][1];
    }
}

// per layer: exclusive scans over its blocks' counts (workgroup l scans layer l).  Round 5: chunk by chunk of 1,024 blocks, one block
// per lane -- coalesced reads and writes, a DPP scan per wave, the sixteen wave totals through the LDS, a running carry -- instead of a
// contiguous run of blocks per lane (strided, uncoalesced accesses and 2 x 25 dependent loads per lane for a 25 M-element layer: 70 us).
__global__ __launch_bounds__(kSpThreads) void spb_scan_kernel(const SpLayer *ly, uint32_t *blk_gt_all, uint32_t *blk_eq_all, SelectState *st_all)
{
    __shared__ uint32_t wt[2][kSpThreads / 64];
    const uint32_t n_blocks = ly[blockIdx.x].nb;
    uint32_t *blk_gt = blk_gt_all + ly[blockIdx.x].blk0, *blk_eq = blk_eq_all + ly[blockIdx.x].blk0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long carry_g = 0, carry_e = 0;
    for (uint32_t c0 = 0; c0 < n_blocks; c0 += kSpThreads) {
        const uint32_t b = c0 + threadIdx.x;
        const uint32_t g = b < n_blocks ? blk_gt[b] : 0u, e = b < n_blocks ? blk_eq[b] : 0u;
        const uint32_t ig = sp_wave_scan(g), ie = sp_wave_scan(e);
        if (lane == 63) { wt[0][wave] = ig; wt[1][wave] = ie; }
        __syncthreads();
        uint32_t og = 0, oe = 0, tg = 0, te = 0;
#pragma unroll
        for (int w = 0; w < kSpThreads / 64; w++) {
            const uint32_t a = wt[0][w], c = wt[1][w];
            if (w < wave) { og += a; oe += c; }
            tg += a; te += c;
        }
        if (b < n_blocks) {
            blk_gt[b] = static_cast<uint32_t>(carry_g + og + ig - g);            // offsets fit: n < 2^32
            blk_eq[b] = static_cast<uint32_t>(carry_e + oe + ie - e);
        }
        carry_g += tg; carry_e += te;
        __syncthreads();
    }
    if (threadIdx.x == 0) st_all[blockIdx.x].total_eq = carry_e;
}

template <typename T>
__global__ __launch_bounds__(kSpWg) void spb_write_kernel(const uint32_t *blk_layer, const SpLayer *ly, uint32_t n_blocks, const T *x_all, T *residual_all,
                                                            const SelectState *st_all, const uint32_t *blk_gt_off, const uint32_t *blk_eq_off,
                                                            uint32_t *loc_all, T *vals_all)
{
    constexpr int WV = kSpWg / 64;
    __shared__ uint32_t wg[kSpPer][WV], we[kSpPer][WV];
    const int wave = threadIdx.x >> 6;
    int lay[kSpPer];
    uint64_t base[kSpPer], nl[kSpPer];
    T xv[kSpPer][4], rv[kSpPer][4];
    uint32_t g0[kSpPer], e0[kSpPer];          // the blocks' offsets in their layers' outputs
    // every load of the workgroup's kSpPer blocks first (values and residuals), then block by block the ordered compaction
#pragma unroll
    for (uint32_t s = 0; s < kSpPer; s++) {
        const uint32_t b = blockIdx.x * kSpPer + s;
        lay[s] = b < n_blocks ? __builtin_amdgcn_readfirstlane(static_cast<int>(blk_layer[b])) : -1;
        const int l = lay[s] < 0 ? 0 : lay[s];
        base[s] = static_cast<uint64_t>(b - ly[l].blk0) * kSpThreads + 4u * threadIdx.x;
        nl[s] = lay[s] < 0 ? 0 : ly[l].n;
        sp_load4(x_all + ly[l].off, base[s], nl[s], xv[s]);
        if (residual_all) sp_load4(residual_all + ly[l].off, base[s], nl[s], rv[s]);
        else { rv[s][0] = rv[s][1] = rv[s][2] = rv[s][3] = T(0); }
        g0[s] = lay[s] < 0 ? 0u : blk_gt_off[b]; e0[s] = lay[s] < 0 ? 0u : blk_eq_off[b];
    }
    // per lane: how many of its four elements lie above / at the threshold, and the same summed over the lanes in front of it
    uint32_t pg[kSpPer], pe[kSpPer];          // exclusive prefix inside the wave
    uint32_t fg[kSpPer], fe[kSpPer];          // the lane's flags, bit j = element j
#pragma unroll
    for (uint32_t s = 0; s < kSpPer; s++) {
        const unsigned long long thr = st_all[lay[s] < 0 ? 0 : lay[s]].prefix;
        fg[s] = 0; fe[s] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned long long key = static_cast<unsigned long long>(KeyOf<T>::get(xv[s][j]));
            const bool live = base[s] + j < nl[s];
            fg[s] |= static_cast<uint32_t>(live && key > thr) << j;
            fe[s] |= static_cast<uint32_t>(live && key == thr) << j;
        }
        const uint32_t cg = __popc(fg[s]), ce = __popc(fe[s]);
        const uint32_t ig = sp_wave_scan(cg), ie = sp_wave_scan(ce);
        pg[s] = ig - cg; pe[s] = ie - ce;
        if ((threadIdx.x & 63) == 63) { wg[s][wave] = ig; we[s][wave] = ie; }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t s = 0; s < kSpPer; s++) {
        const int l = lay[s];
        if (l < 0 || base[s] >= nl[s]) continue;
        uint32_t og = 0, oe = 0;
#pragma unroll
        for (int w = 0; w < WV; w++) { if (w < wave) { og += wg[s][w]; oe += we[s][w]; } }
        unsigned long long gt_before = static_cast<unsigned long long>(g0[s]) + og + pg[s];
        unsigned long long eq_before = static_cast<unsigned long long>(e0[s]) + oe + pe[s];
        const unsigned long long skip = st_all[l].total_eq - st_all[l].remaining;      // the first `skip` ties (lowest indices) stay out
        const bool any_k = ly[l].k != 0;                                                // (a layer that keeps nothing only updates its residual)
        uint32_t *loc = loc_all + ly[l].koff;
        T *vals = vals_all + ly[l].koff;
        T out_r[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool is_gt = (fg[s] >> j) & 1u, is_eq = (fe[s] >> j) & 1u;
            const bool selected = any_k && (is_gt || (is_eq && eq_before >= skip));
            const T v = xv[s][j] + rv[s][j];
            if (selected) {
                const unsigned long long pos = gt_before + (eq_before > skip ? eq_before - skip : 0ull);
                loc[pos] = static_cast<uint32_t>(base[s] + j);
                vals[pos] = v;
            }
            out_r[j] = selected ? T(0) : v;
            gt_before += is_gt; eq_before += is_eq;
        }
        if (residual_all) sp_store4(residual_all + ly[l].off, base[s], nl[s], out_r);
    }
}

size_t sparsify_batch_workspace_bytes(int L, uint64_t n_blocks)
{
    return static_cast<size_t>(L) * (sizeof(SpLayer) + sizeof(SelectState) + 256 * 4) + 3 * n_blocks * 4 + 256;
}

size_t sparsify_batch_desc_bytes(int L) { return static_cast<size_t>(L) * sizeof(SpLayer); }

// Fills the host copy of the layer table (layers back to back); returns the number of 1024-element blocks.
uint64_t sparsify_batch_layout(int L, const uint64_t *n, const uint64_t *k, void *desc_host)
{
    SpLayer *ly = static_cast<SpLayer *>(desc_host);
    uint64_t off = 0, koff = 0, blk = 0;
    for (int l = 0; l < L; l++) {
        const uint64_t nb = (n[l] + kSpThreads - 1) / kSpThreads;
        ly[l] = SpLayer{off, n[l], k[l], koff, static_cast<uint32_t>(blk), static_cast<uint32_t>(nb)};
        off += n[l]; koff += k[l]; blk += nb;
    }
    return blk;
}

template <typename T>
static hipError_t sparsify_batch_impl(const LaunchEnv &env, int L, uint64_t n_blocks, const T *x, T *residual, uint32_t *loc, T *vals, void *ws, bool prepared = false,
                                      uint64_t n_single = 0)
{
    // workspace: SpLayer[L] (already uploaded) | SelectState[L] | hist[L][256] | blk_gt[n_blocks] | blk_eq[n_blocks] | blk_layer[n_blocks]
    const SpLayer *ly = static_cast<const SpLayer *>(ws);
    SelectState *st = reinterpret_cast<SelectState *>(const_cast<SpLayer *>(ly) + L);
    uint32_t *hist = reinterpret_cast<uint32_t *>(st + L);
    uint32_t *blk_gt = hist + static_cast<size_t>(L) * 256, *blk_eq = blk_gt + n_blocks, *blk_layer = blk_eq + n_blocks;
    const unsigned nb = static_cast<unsigned>(n_blocks);
    if (!prepared) {                     // (a single layer: launch_sparsify's one kernel has written the table, the map and the select state)
        hipLaunchKernelGGL(spb_map_kernel, dim3((nb + 255) / 256), dim3(256), 0, env.stream, L, ly, nb, blk_layer);
        hipLaunchKernelGGL(spb_init_kernel, dim3(L), dim3(256), 0, env.stream, ly, st, hist);
    }
    const unsigned trips = (nb + kSpPer - 1) / kSpPer;
    const unsigned hgrid = std::min<unsigned>(trips, static_cast<unsigned>(std::max(env.num_cus, 1)) * 8u);          // persistent: 8 x 256 lanes per CU (4 x 256 with eight blocks per trip measured 20 % slower)
    const unsigned sgrid = static_cast<unsigned>(std::min<uint64_t>(n_blocks, static_cast<uint64_t>(std::max(env.num_cus, 1)) * 2));
    for (int shift = KeyOf<T>::bits - 8; shift >= 0; shift -= 8) {
        if (prepared) hipLaunchKernelGGL(sp_hist_kernel<T>, dim3(sgrid), dim3(kSpThreads), 0, env.stream, n_single, x, st, shift, hist);
        else hipLaunchKernelGGL(spb_hist_kernel<T>, dim3(hgrid), dim3(kSpWg), 0, env.stream, blk_layer, ly, nb, x, st, shift, hist);
        hipLaunchKernelGGL(spb_pick_digit_kernel, dim3(L), dim3(256), 0, env.stream, st, shift, hist);
    }
    const unsigned pgrid = (nb + kSpPer - 1) / kSpPer;
    hipLaunchKernelGGL(spb_count_kernel<T>, dim3(pgrid), dim3(kSpWg), 0, env.stream, blk_layer, ly, nb, x, st, blk_gt, blk_eq);
    hipLaunchKernelGGL(spb_scan_kernel, dim3(L), dim3(kSpThreads), 0, env.stream, ly, blk_gt, blk_eq, st);
    hipLaunchKernelGGL(spb_write_kernel<T>, dim3(pgrid), dim3(kSpWg), 0, env.stream, blk_layer, ly, nb, x, residual, st, blk_gt, blk_eq, loc, vals);
    return hipGetLastError();
}

hipError_t launch_sparsify_batch(const LaunchEnv &env, int L, uint64_t n_blocks, const void *x, bool is_f64, void *residual, uint32_t *loc, void *vals,
                                 void *ws)
{
    if (L <= 0 || n_blocks == 0) return hipSuccess;
    return is_f64 ? sparsify_batch_impl<double>(env, L, n_blocks, static_cast<const double *>(x), static_cast<double *>(residual), loc,
                                                static_cast<double *>(vals), ws)
                  : sparsify_batch_impl<float>(env, L, n_blocks, static_cast<const float *>(x), static_cast<float *>(residual), loc,
                                               static_cast<float *>(vals), ws);
}

// One layer: the model-wide passes on a ONE-layer table that a kernel writes on the device -- table, block map, select state and
// histogram in one launch, no upload, no synchronisation (capturable).  Round 5: the single-layer kernels this replaces kept one element
// per lane in their count / write passes and a strided single-workgroup scan.
__global__ void spb_single_kernel(SpLayer *ly, uint64_t n, uint64_t k, uint32_t nb)
{
    SelectState *st = reinterpret_cast<SelectState *>(ly + 1);
    uint32_t *hist = reinterpret_cast<uint32_t *>(st + 1);
    uint32_t *blk_layer = hist + 256 + 2 * static_cast<size_t>(nb);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        ly[0] = SpLayer{0, n, k, 0, 0u, nb};
        st->prefix = 0; st->mask = 0; st->remaining = k; st->total_eq = 0;
    }
    if (t < 256) hist[t] = 0;
    for (uint32_t b = t; b < nb; b += gridDim.x * blockDim.x) blk_layer[b] = 0;
}

size_t sparsify_workspace_bytes(uint64_t n)
{
    const uint64_t nb = (n + kSpThreads - 1) / kSpThreads;
    return sparsify_batch_workspace_bytes(1, nb) + 64;
}

hipError_t launch_sparsify(const LaunchEnv &env, uint64_t n, uint64_t k, const void *x, bool is_f64, void *residual, uint32_t *loc,
                           void *vals, void *ws)
{
    if (n == 0 || k == 0) return hipSuccess;
    const uint64_t nb = (n + kSpThreads - 1) / kSpThreads;
    const unsigned g = static_cast<unsigned>(std::min<uint64_t>((nb + 255) / 256, 256));
    hipLaunchKernelGGL(spb_single_kernel, dim3(g), dim3(256), 0, env.stream, static_cast<SpLayer *>(ws), n, k, static_cast<uint32_t>(nb));
    return is_f64 ? sparsify_batch_impl<double>(env, 1, nb, static_cast<const double *>(x), static_cast<double *>(residual), loc, static_cast<double *>(vals), ws, true, n)
                  : sparsify_batch_impl<float>(env, 1, nb, static_cast<const float *>(x), static_cast<float *>(residual), loc, static_cast<float *>(vals), ws, true, n);
}

}  // namespace flashe
