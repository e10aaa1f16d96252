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

template <typename T>
__global__ __launch_bounds__(kSpThreads) void sp_hist_kernel(uint64_t n, const T *x, const SelectState *st, int shift, uint32_t *hist)
{
    __shared__ uint32_t lh[256];
    if (threadIdx.x < 256) lh[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix, mask = st->mask;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kSpThreads + threadIdx.x; i < n;
         i += static_cast<uint64_t>(gridDim.x) * kSpThreads) {
        const unsigned long long key = KeyOf<T>::get(x[i]);
        if ((key & mask) == prefix) atomicAdd(&lh[(key >> shift) & 255u], 1u);
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
    if (threadIdx.x == 0) {
        unsigned long long remaining = st->remaining, acc = 0;
        int d = 255;
        for (; d > 0; d--) {
            if (acc + hist[d] >= remaining) break;
            acc += hist[d];
        }
        st->remaining = remaining - acc;                 // everything in the higher bins is taken whole
        st->prefix |= static_cast<unsigned long long>(d) << shift;
        st->mask |= 255ull << shift;
    }
    __syncthreads();
    hist[threadIdx.x] = 0;
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
