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

__global__ void sp_init_kernel(SelectState *st, unsigned long long k, uint32_t *hist)
{
    if (threadIdx.x == 0) { st->prefix = 0; st->mask = 0; st->remaining = k; st->total_eq = 0; }
    hist[threadIdx.x] = 0;
}

__global__ void sp_pick_digit_kernel(SelectState *st, int shift, uint32_t *hist)
{
    sp_pick(st, shift, hist);
}

template <typename T>
__global__ __launch_bounds__(kSpThreads) void sp_count_kernel(uint64_t n, const T *x, const SelectState *st, uint32_t *blk_gt, uint32_t *blk_eq)
{
    __shared__ uint32_t c[2];
    if (threadIdx.x < 2) c[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kSpThreads + threadIdx.x;
    const unsigned long long thr = st->prefix;
    const unsigned long long key = i < n ? static_cast<unsigned long long>(KeyOf<T>::get(x[i])) : 0ull;
    const unsigned long long gt = __ballot(i < n && key > thr), eq = __ballot(i < n && key == thr);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&c[0], __popcll(gt)); atomicAdd(&c[1], __popcll(eq)); }
    __syncthreads();
    if (threadIdx.x == 0) { blk_gt[blockIdx.x] = c[0]; blk_eq[blockIdx.x] = c[1]; }
}

// exclusive scans over the per-block counts (one workgroup; n_blocks is a few ten thousand at most)
__global__ __launch_bounds__(kSpThreads) void sp_scan_kernel(uint32_t n_blocks, uint32_t *blk_gt, uint32_t *blk_eq, SelectState *st)
{
    __shared__ unsigned long long sums[2][kSpThreads];
    const uint32_t per = (n_blocks + kSpThreads - 1) / kSpThreads;
    const uint32_t b0 = threadIdx.x * per, b1 = min(b0 + per, n_blocks);
    unsigned long long g = 0, e = 0;
    for (uint32_t b = b0; b < b1; b++) { g += blk_gt[b]; e += blk_eq[b]; }
    sums[0][threadIdx.x] = g; sums[1][threadIdx.x] = e;
    __syncthreads();
    // inclusive Hillis-Steele scan over the 1024 per-thread totals (10 steps), then shift to exclusive
    for (int off = 1; off < kSpThreads; off <<= 1) {
        unsigned long long ag = 0, ae = 0;
        if (static_cast<int>(threadIdx.x) >= off) { ag = sums[0][threadIdx.x - off]; ae = sums[1][threadIdx.x - off]; }
        __syncthreads();
        sums[0][threadIdx.x] += ag; sums[1][threadIdx.x] += ae;
        __syncthreads();
    }
    if (threadIdx.x == kSpThreads - 1) st->total_eq = sums[1][threadIdx.x];
    const unsigned long long own_g = g, own_e = e;
    g = sums[0][threadIdx.x] - own_g; e = sums[1][threadIdx.x] - own_e;
    for (uint32_t b = b0; b < b1; b++) {
        const uint32_t tg = blk_gt[b], te = blk_eq[b];
        blk_gt[b] = static_cast<uint32_t>(g); blk_eq[b] = static_cast<uint32_t>(e);     // offsets fit: n < 2^32
        g += tg; e += te;
    }
}

template <typename T>
__global__ __launch_bounds__(kSpThreads) void sp_write_kernel(uint64_t n, const T *x, T *residual, const SelectState *st,
                                                              const uint32_t *blk_gt_off, const uint32_t *blk_eq_off,
                                                              uint32_t *loc, T *vals)
{
    __shared__ uint32_t wg[kSpThreads / 64], we[kSpThreads / 64];
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kSpThreads + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long thr = st->prefix;
    const unsigned long long skip = st->total_eq - st->remaining;      // the first `skip` ties (lowest indices) stay out
    const bool live = i < n;
    const T xv = live ? x[i] : T(0);
    const unsigned long long key = KeyOf<T>::get(xv);
    const bool is_gt = live && key > thr, is_eq = live && key == thr;
    const unsigned long long mg = __ballot(is_gt), me = __ballot(is_eq);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) { wg[wave] = __popcll(mg); we[wave] = __popcll(me); }
    __syncthreads();
    uint32_t og = 0, oe = 0;
    for (int w = 0; w < wave; w++) { og += wg[w]; oe += we[w]; }
    const unsigned long long gt_before = blk_gt_off[blockIdx.x] + og + __popcll(mg & below);
    const unsigned long long eq_before = blk_eq_off[blockIdx.x] + oe + __popcll(me & below);
    if (!live) return;
    const bool selected = is_gt || (is_eq && eq_before >= skip);
    const T v = xv + (residual ? residual[i] : T(0));
    if (selected) {
        const unsigned long long pos = gt_before + (eq_before > skip ? eq_before - skip : 0ull);
        loc[pos] = static_cast<uint32_t>(i);
        vals[pos] = v;
        if (residual) residual[i] = T(0);
    } else if (residual) {
        residual[i] = v;
    }
}

template <typename T>
static hipError_t sparsify_impl(const LaunchEnv &env, uint64_t n, uint64_t k, const T *x, T *residual, uint32_t *loc, T *vals, void *ws)
{
    // workspace: SelectState | hist[256] | blk_gt[nb] | blk_eq[nb]
    const uint32_t nb = static_cast<uint32_t>((n + kSpThreads - 1) / kSpThreads);
    SelectState *st = static_cast<SelectState *>(ws);
    uint32_t *hist = reinterpret_cast<uint32_t *>(st + 1);
    uint32_t *blk_gt = hist + 256, *blk_eq = blk_gt + nb;
    hipLaunchKernelGGL(sp_init_kernel, dim3(1), dim3(256), 0, env.stream, st, static_cast<unsigned long long>(k), hist);
    uint64_t hb = (n + kSpThreads - 1) / kSpThreads;
    const uint64_t cap = static_cast<uint64_t>(env.num_cus) * 2;
    if (hb > cap) hb = cap;
    for (int shift = KeyOf<T>::bits - 8; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(sp_hist_kernel<T>, dim3(static_cast<unsigned>(hb)), dim3(kSpThreads), 0, env.stream, n, x, st, shift, hist);
        hipLaunchKernelGGL(sp_pick_digit_kernel, dim3(1), dim3(256), 0, env.stream, st, shift, hist);
    }
    hipLaunchKernelGGL(sp_count_kernel<T>, dim3(nb), dim3(kSpThreads), 0, env.stream, n, x, st, blk_gt, blk_eq);
    hipLaunchKernelGGL(sp_scan_kernel, dim3(1), dim3(kSpThreads), 0, env.stream, nb, blk_gt, blk_eq, st);
    hipLaunchKernelGGL(sp_write_kernel<T>, dim3(nb), dim3(kSpThreads), 0, env.stream, n, x, residual, st, blk_gt, blk_eq, loc, vals);
    return hipGetLastError();
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

constexpr uint32_t kSpGroup = 8;        // consecutive blocks per workgroup of the histogram pass (one LDS histogram while the layer stays the same)

template <typename T>
__global__ __launch_bounds__(kSpThreads) void spb_hist_kernel(const uint32_t *blk_layer, const SpLayer *ly, uint32_t n_blocks, const T *x,
                                                                const SelectState *st, int shift, uint32_t *hist)
{
    __shared__ uint32_t lh[256];
    int cur = -1;
    for (uint32_t s = 0; s < kSpGroup; s++) {
        const uint32_t b = blockIdx.x * kSpGroup + s;
        if (b >= n_blocks) break;
        const int l = static_cast<int>(blk_layer[b]);
        if (l != cur) {
            __syncthreads();                                      // every vote of the layer that ends here is in
            if (cur >= 0 && threadIdx.x < 256 && lh[threadIdx.x]) atomicAdd(&hist[cur * 256 + threadIdx.x], lh[threadIdx.x]);
            __syncthreads();
            if (threadIdx.x < 256) lh[threadIdx.x] = 0;
            cur = l;
            __syncthreads();
        }
        const uint64_t i = static_cast<uint64_t>(b - ly[l].blk0) * kSpThreads + threadIdx.x;
        const bool live = i < ly[l].n;
        const unsigned long long key = live ? static_cast<unsigned long long>(KeyOf<T>::get(x[ly[l].off + i])) : 0ull;
        sp_vote(lh, live && (key & st[l].mask) == st[l].prefix, static_cast<uint32_t>((key >> shift) & 255u));
    }
    __syncthreads();
    if (cur >= 0 && threadIdx.x < 256 && lh[threadIdx.x]) atomicAdd(&hist[cur * 256 + threadIdx.x], lh[threadIdx.x]);
}

__global__ void spb_pick_digit_kernel(SelectState *st_all, int shift, uint32_t *hist_all)
{
    sp_pick(st_all + blockIdx.x, shift, hist_all + blockIdx.x * 256);
}

template <typename T>
__global__ __launch_bounds__(kSpThreads) void spb_count_kernel(const uint32_t *blk_layer, const SpLayer *ly, const T *x, const SelectState *st, uint32_t *blk_gt,
                                                                 uint32_t *blk_eq)
{
    __shared__ uint32_t c[2];
    if (threadIdx.x < 2) c[threadIdx.x] = 0;
    __syncthreads();
    const int l = static_cast<int>(blk_layer[blockIdx.x]);
    const uint64_t i = static_cast<uint64_t>(blockIdx.x - ly[l].blk0) * kSpThreads + threadIdx.x;
    const bool live = i < ly[l].n;
    const unsigned long long thr = st[l].prefix;
    const unsigned long long key = live ? static_cast<unsigned long long>(KeyOf<T>::get(x[ly[l].off + i])) : 0ull;
    const unsigned long long gt = __ballot(live && key > thr), eq = __ballot(live && key == thr);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&c[0], __popcll(gt)); atomicAdd(&c[1], __popcll(eq)); }
    __syncthreads();
    if (threadIdx.x == 0) { blk_gt[blockIdx.x] = c[0]; blk_eq[blockIdx.x] = c[1]; }
}

// per layer: exclusive scans over its blocks' counts (workgroup l scans layer l)
__global__ __launch_bounds__(kSpThreads) void spb_scan_kernel(const SpLayer *ly, uint32_t *blk_gt_all, uint32_t *blk_eq_all, SelectState *st_all)
{
    __shared__ unsigned long long sums[2][kSpThreads];
    const uint32_t n_blocks = ly[blockIdx.x].nb;
    uint32_t *blk_gt = blk_gt_all + ly[blockIdx.x].blk0, *blk_eq = blk_eq_all + ly[blockIdx.x].blk0;
    SelectState *st = st_all + blockIdx.x;
    const uint32_t per = (n_blocks + kSpThreads - 1) / kSpThreads;
    const uint32_t b0 = min(threadIdx.x * per, n_blocks), b1 = min(b0 + per, n_blocks);
    unsigned long long g = 0, e = 0;
    for (uint32_t b = b0; b < b1; b++) { g += blk_gt[b]; e += blk_eq[b]; }
    sums[0][threadIdx.x] = g; sums[1][threadIdx.x] = e;
    __syncthreads();
    for (int off = 1; off < kSpThreads; off <<= 1) {
        unsigned long long ag = 0, ae = 0;
        if (static_cast<int>(threadIdx.x) >= off) { ag = sums[0][threadIdx.x - off]; ae = sums[1][threadIdx.x - off]; }
        __syncthreads();
        sums[0][threadIdx.x] += ag; sums[1][threadIdx.x] += ae;
        __syncthreads();
    }
    if (threadIdx.x == kSpThreads - 1) st->total_eq = sums[1][threadIdx.x];
    const unsigned long long own_g = g, own_e = e;
    g = sums[0][threadIdx.x] - own_g; e = sums[1][threadIdx.x] - own_e;
    for (uint32_t b = b0; b < b1; b++) {
        const uint32_t tg = blk_gt[b], te = blk_eq[b];
        blk_gt[b] = static_cast<uint32_t>(g); blk_eq[b] = static_cast<uint32_t>(e);
        g += tg; e += te;
    }
}

template <typename T>
__global__ __launch_bounds__(kSpThreads) void spb_write_kernel(const uint32_t *blk_layer, const SpLayer *ly, const T *x_all, T *residual_all,
                                                                 const SelectState *st_all, const uint32_t *blk_gt_off, const uint32_t *blk_eq_off,
                                                                 uint32_t *loc_all, T *vals_all)
{
    __shared__ uint32_t wg[kSpThreads / 64], we[kSpThreads / 64];
    const int l = static_cast<int>(blk_layer[blockIdx.x]);
    const uint64_t i = static_cast<uint64_t>(blockIdx.x - ly[l].blk0) * kSpThreads + threadIdx.x;
    const T *x = x_all + ly[l].off;
    T *residual = residual_all ? residual_all + ly[l].off : nullptr;
    uint32_t *loc = loc_all + ly[l].koff;
    T *vals = vals_all + ly[l].koff;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long thr = st_all[l].prefix;
    const unsigned long long skip = st_all[l].total_eq - st_all[l].remaining;      // the first `skip` ties (lowest indices) stay out
    const bool live = i < ly[l].n;
    const T xv = live ? x[i] : T(0);
    const unsigned long long key = KeyOf<T>::get(xv);
    const bool is_gt = live && key > thr, is_eq = live && key == thr;
    const unsigned long long mg = __ballot(is_gt), me = __ballot(is_eq);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (lane == 0) { wg[wave] = __popcll(mg); we[wave] = __popcll(me); }
    __syncthreads();
    uint32_t og = 0, oe = 0;
    for (int w = 0; w < wave; w++) { og += wg[w]; oe += we[w]; }
    const unsigned long long gt_before = blk_gt_off[blockIdx.x] + og + __popcll(mg & below);
    const unsigned long long eq_before = blk_eq_off[blockIdx.x] + oe + __popcll(me & below);
    if (!live) return;
    const bool selected = ly[l].k != 0 && (is_gt || (is_eq && eq_before >= skip));       // (a layer that keeps nothing only updates its residual)
    const T v = xv + (residual ? residual[i] : T(0));
    if (selected) {
        const unsigned long long pos = gt_before + (eq_before > skip ? eq_before - skip : 0ull);
        loc[pos] = static_cast<uint32_t>(i);
        vals[pos] = v;
        if (residual) residual[i] = T(0);
    } else if (residual) {
        residual[i] = v;
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
static hipError_t sparsify_batch_impl(const LaunchEnv &env, int L, uint64_t n_blocks, const T *x, T *residual, uint32_t *loc, T *vals, void *ws)
{
    // workspace: SpLayer[L] (already uploaded) | SelectState[L] | hist[L][256] | blk_gt[n_blocks] | blk_eq[n_blocks] | blk_layer[n_blocks]
    const SpLayer *ly = static_cast<const SpLayer *>(ws);
    SelectState *st = reinterpret_cast<SelectState *>(const_cast<SpLayer *>(ly) + L);
    uint32_t *hist = reinterpret_cast<uint32_t *>(st + L);
    uint32_t *blk_gt = hist + static_cast<size_t>(L) * 256, *blk_eq = blk_gt + n_blocks, *blk_layer = blk_eq + n_blocks;
    const unsigned nb = static_cast<unsigned>(n_blocks);
    hipLaunchKernelGGL(spb_map_kernel, dim3((nb + 255) / 256), dim3(256), 0, env.stream, L, ly, nb, blk_layer);
    hipLaunchKernelGGL(spb_init_kernel, dim3(L), dim3(256), 0, env.stream, ly, st, hist);
    const unsigned hgrid = (nb + kSpGroup - 1) / kSpGroup;
    for (int shift = KeyOf<T>::bits - 8; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(spb_hist_kernel<T>, dim3(hgrid), dim3(kSpThreads), 0, env.stream, blk_layer, ly, nb, x, st, shift, hist);
        hipLaunchKernelGGL(spb_pick_digit_kernel, dim3(L), dim3(256), 0, env.stream, st, shift, hist);
    }
    hipLaunchKernelGGL(spb_count_kernel<T>, dim3(nb), dim3(kSpThreads), 0, env.stream, blk_layer, ly, x, st, blk_gt, blk_eq);
    hipLaunchKernelGGL(spb_scan_kernel, dim3(L), dim3(kSpThreads), 0, env.stream, ly, blk_gt, blk_eq, st);
    hipLaunchKernelGGL(spb_write_kernel<T>, dim3(nb), dim3(kSpThreads), 0, env.stream, blk_layer, ly, x, residual, st, blk_gt, blk_eq, loc, vals);
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

size_t sparsify_workspace_bytes(uint64_t n)
{
    const uint64_t nb = (n + kSpThreads - 1) / kSpThreads;
    return sizeof(SelectState) + 256 * 4 + 2 * nb * 4 + 64;
}

hipError_t launch_sparsify(const LaunchEnv &env, uint64_t n, uint64_t k, const void *x, bool is_f64, void *residual, uint32_t *loc,
                           void *vals, void *ws)
{
    if (n == 0 || k == 0) return hipSuccess;
    return is_f64 ? sparsify_impl<double>(env, n, k, static_cast<const double *>(x), static_cast<double *>(residual), loc,
                                          static_cast<double *>(vals), ws)
                  : sparsify_impl<float>(env, n, k, static_cast<const float *>(x), static_cast<float *>(residual), loc,
                                         static_cast<float *>(vals), ws);
}

}  // namespace flashe
