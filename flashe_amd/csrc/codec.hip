// gfx950 kernels of the quantise / batch codec either side of the cipher and the host-side descriptors of the fused codec.
#include "device_common.h"

namespace flashe {

// ------------------------------------------------------------------------------------------
// Quantise / batch codec either side of the cipher (streaming, HBM-bound)
// ------------------------------------------------------------------------------------------
// float arithmetic below must round exactly like numpy's: no contraction into FMAs
template <typename T>
__global__ __launch_bounds__(kStreamThreads) void quantize_kernel(uint64_t n, const T *x, T alpha, T scale, T den,
                                                                  const double *u, uint64_t *q)
{
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads)
        q[j] = quantize_one<T>(x[j], alpha, scale, den, u[j]);
}

__global__ __launch_bounds__(kStreamThreads) void unquantize_kernel(uint64_t n, const uint64_t *v, int v_limbs, double ac,
                                                                    double two_a, double den, double *out)
{
#pragma clang fp contract(off)
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        const u128 x = v_limbs == 2 ? ld128(v + 2 * j) : static_cast<u128>(v[j]);
        const double d = u128_to_double(x);
        out[j] = d * two_a / den - ac;
    }
}

// one batch (bs consecutive values, first most significant) per lane
__global__ __launch_bounds__(kStreamThreads) void batch_kernel(uint64_t n, uint64_t nb, const uint64_t *vals, int L, int bs,
                                                               int field_bits, uint64_t *out)
{
    for (uint64_t b = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; b < nb;
         b += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        u128 t = 0;
        for (int i = 0; i < bs; i++) {
            const uint64_t j = b * bs + i;
            t = (field_bits >= 128 ? 0 : t << field_bits) + (j < n ? vals[j] : 0ull);
        }
        if (L == 2) st128(out + 2 * b, t);
        else out[b] = static_cast<uint64_t>(t);
    }
}

__global__ __launch_bounds__(kStreamThreads) void unbatch_kernel(uint64_t nb, const uint64_t *in, int L, int bs, int field_bits,
                                                                 uint64_t *out)
{
    const u128 mk = field_bits >= 128 ? ~static_cast<u128>(0) : ((static_cast<u128>(1) << field_bits) - 1);
    for (uint64_t b = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; b < nb;
         b += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        u128 item = L == 2 ? ld128(in + 2 * b) : static_cast<u128>(in[b]);
        for (int i = 0; i < bs; i++) {
            out[b * bs + (bs - 1 - i)] = static_cast<uint64_t>(item & mk);
            item = field_bits >= 128 ? 0 : item >> field_bits;
        }
    }
}

// ---- the batched codec over a flattened model: quantise + batch in one launch, unbatch + unquantise in one launch ----
template <bool BY_VALUE>
__device__ __forceinline__ const BatchLayer *batch_layer_of(const BatchLayer *layers, int n_layers, uint64_t key)
{
    int lo = 0, hi = n_layers - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((BY_VALUE ? layers[mid].value_start : layers[mid].elem_start) <= key) lo = mid; else hi = mid - 1;
    }
    return layers + lo;
}

__global__ __launch_bounds__(kStreamThreads) void quantize_batch_model_kernel(const BatchLayer *__restrict__ layers, int n_layers, int L, int bs,
                                                                              int field_bits, const double *__restrict__ u, uint64_t n_elems,
                                                                              uint64_t *out)
{
    for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; e < n_elems; e += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        const BatchLayer *Ly = batch_layer_of<false>(layers, n_layers, e);
        const uint64_t j0 = (e - Ly->elem_start) * static_cast<uint64_t>(bs);
        u128 t = 0;
        for (int i = 0; i < bs; i++) {
            const uint64_t j = j0 + i;
            uint64_t v = 0;
            if (j < Ly->size) {
                const double draw = u[Ly->value_start + j];
                v = Ly->x_is_f64 ? quantize_one<double>(*FLASHE_GLOBAL(const double, static_cast<const double *>(Ly->x) + j), Ly->p0, Ly->p1, Ly->p2, draw)
                                 : quantize_one<float>(*FLASHE_GLOBAL(const float, static_cast<const float *>(Ly->x) + j), static_cast<float>(Ly->p0), static_cast<float>(Ly->p1),
                                                       static_cast<float>(Ly->p2), draw);
            }
            t = (field_bits >= 128 ? 0 : t << field_bits) + v;          // temp *= mod; temp += value (jzf_quantize.py:178-181)
        }
        if (L == 2) st128(out + 2 * e, t);
        else out[e] = static_cast<uint64_t>(t);
    }
}

__global__ __launch_bounds__(kStreamThreads) void unbatch_unquantize_model_kernel(const BatchLayer *__restrict__ layers, int n_layers, int L, int bs,
                                                                                  int field_bits, const uint64_t *__restrict__ in, uint64_t n_values,
                                                                                  double *out)
{
#pragma clang fp contract(off)
    const u128 mk = field_bits >= 128 ? ~static_cast<u128>(0) : ((static_cast<u128>(1) << field_bits) - 1);
    for (uint64_t g = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; g < n_values; g += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        const BatchLayer *Ly = batch_layer_of<true>(layers, n_layers, g);
        const uint64_t j = g - Ly->value_start;
        if (j >= Ly->size) continue;                                    // (cannot happen for a well-formed table)
        const uint64_t e = Ly->elem_start + j / static_cast<uint64_t>(bs);
        const int slot = static_cast<int>(j % static_cast<uint64_t>(bs));
        const u128 item = L == 2 ? ld128(in + 2 * e) : static_cast<u128>(in[e]);
        const int sh = field_bits * (bs - 1 - slot);                    // the first value of an element is its most significant field (:240-246)
        const u128 v = (sh >= 128 ? static_cast<u128>(0) : item >> sh) & mk;
        out[g] = u128_to_double(v) * Ly->p1 / Ly->p2 - Ly->p0;          // _static_unquantize_padding_asymmetric (:102-107)
    }
}

hipError_t launch_quantize_batch_model(const LaunchEnv &env, const BatchLayer *layers_dev, int n_layers, int field_bits, const double *u_dev,
                                       uint64_t n_elems, uint64_t *out_dev)
{
    if (n_elems == 0) return hipSuccess;
    if (field_bits < 1 || field_bits > env.b || n_layers < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(quantize_batch_model_kernel, dim3(stream_grid(env, n_elems)), dim3(kStreamThreads), 0, env.stream, layers_dev, n_layers,
                       env.b > 64 ? 2 : 1, env.b / field_bits, field_bits, u_dev, n_elems, out_dev);
    return hipGetLastError();
}

hipError_t launch_unbatch_unquantize_model(const LaunchEnv &env, const BatchLayer *layers_dev, int n_layers, int field_bits, const uint64_t *in_dev,
                                           uint64_t n_values, double *out_dev)
{
    if (n_values == 0) return hipSuccess;
    if (field_bits < 1 || field_bits > env.b || n_layers < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(unbatch_unquantize_model_kernel, dim3(stream_grid(env, n_values)), dim3(kStreamThreads), 0, env.stream, layers_dev, n_layers,
                       env.b > 64 ? 2 : 1, env.b / field_bits, field_bits, in_dev, n_values, out_dev);
    return hipGetLastError();
}

BatchLayer batch_layer_front(uint64_t elem_start, uint64_t value_start, uint64_t size, const void *x_dev, bool is_f64, double alpha, int bits)
{
    const Codec c = codec_quantize_front(x_dev, is_f64, alpha, bits, nullptr);
    return BatchLayer{elem_start, value_start, size, x_dev, c.alpha, c.scale, c.den, c.x_is_f64, 0};
}

BatchLayer batch_layer_back(uint64_t elem_start, uint64_t value_start, uint64_t size, double alpha, int bits, int num_clients)
{
    Codec c{};
    codec_unquantize_back(&c, alpha, bits, num_clients, nullptr);
    return BatchLayer{elem_start, value_start, size, nullptr, c.ac, c.two_a, c.uden, 0, 0};
}

Codec codec_quantize_front(const void *x_dev, bool is_f64, double alpha, int bits, const double *u_dev)
{
    Codec c{};
    c.x = x_dev; c.u = u_dev; c.alpha = alpha; c.scale = static_cast<double>((1ull << bits) - 1); c.den = 2 * alpha; c.x_is_f64 = is_f64 ? 1 : 0;
    return c;
}

void codec_unquantize_back(Codec *c, double alpha, int bits, int num_clients, double *out_dev)
{
    c->fout = out_dev;
    c->ac = alpha * static_cast<double>(num_clients);
    c->two_a = 2 * c->ac;
    c->uden = static_cast<double>(((1ull << bits) - 1) * static_cast<uint64_t>(num_clients));
}

CodecLayer codec_layer_front(uint64_t start, const void *x_dev, bool is_f64, double alpha, int bits)
{
    const Codec c = codec_quantize_front(x_dev, is_f64, alpha, bits, nullptr);
    return CodecLayer{start, x_dev, c.alpha, c.scale, c.den, c.x_is_f64, 0};
}

CodecLayer codec_layer_back(uint64_t start, double alpha, int bits, int num_clients)
{
    Codec c{};
    codec_unquantize_back(&c, alpha, bits, num_clients, nullptr);
    return CodecLayer{start, nullptr, c.ac, c.two_a, c.uden, 0, 0};
}

// x <- x + shift (normalize: shift = -mean, unnormalize: shift = +mean; a - b and a + (-b) round identically)
template <typename T, bool WIDE>
__global__ __launch_bounds__(kStreamThreads) void shift_kernel(uint64_t n, T *x, double shift)
{
#pragma clang fp contract(off)
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads)
        x[j] = WIDE ? static_cast<T>(static_cast<double>(x[j]) + shift) : x[j] + static_cast<T>(shift);
}

hipError_t launch_shift(const LaunchEnv &env, uint64_t n, void *x_dev, bool is_f64, double shift, bool wide)
{
    if (n == 0) return hipSuccess;
    const dim3 g(stream_grid(env, n)), t(kStreamThreads);
    if (is_f64) hipLaunchKernelGGL((shift_kernel<double, false>), g, t, 0, env.stream, n, static_cast<double *>(x_dev), shift);
    else if (wide) hipLaunchKernelGGL((shift_kernel<float, true>), g, t, 0, env.stream, n, static_cast<float *>(x_dev), shift);
    else hipLaunchKernelGGL((shift_kernel<float, false>), g, t, 0, env.stream, n, static_cast<float *>(x_dev), shift);
    return hipGetLastError();
}

// part[block] = sum over the block's elements of (x - center)^POW in float64: per-thread partial, wave shuffle tree, one LDS hop
template <typename T, int POW>
__global__ __launch_bounds__(kStreamThreads) void moment_kernel(uint64_t n, const T *x, double center, double *part)
{
    __shared__ double ws[kStreamThreads / 64];
    double acc = 0;
    for (uint64_t j = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; j < n;
         j += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        const double d = static_cast<double>(x[j]) - center;
        acc += POW == 1 ? d : d * d;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < kStreamThreads / 64; w++) t += ws[w];
        part[blockIdx.x] = t;
    }
}

int moments_grid(const LaunchEnv &env, uint64_t n) { return stream_grid(env, n); }

hipError_t launch_moment(const LaunchEnv &env, uint64_t n, const void *x_dev, bool is_f64, double center, int pow, double *part_dev)
{
    if (n == 0) return hipSuccess;
    const dim3 g(stream_grid(env, n)), t(kStreamThreads);
    if (is_f64 && pow == 1) hipLaunchKernelGGL((moment_kernel<double, 1>), g, t, 0, env.stream, n, static_cast<const double *>(x_dev), center, part_dev);
    else if (is_f64) hipLaunchKernelGGL((moment_kernel<double, 2>), g, t, 0, env.stream, n, static_cast<const double *>(x_dev), center, part_dev);
    else if (pow == 1) hipLaunchKernelGGL((moment_kernel<float, 1>), g, t, 0, env.stream, n, static_cast<const float *>(x_dev), center, part_dev);
    else hipLaunchKernelGGL((moment_kernel<float, 2>), g, t, 0, env.stream, n, static_cast<const float *>(x_dev), center, part_dev);
    return hipGetLastError();
}

hipError_t launch_quantize(const LaunchEnv &env, uint64_t n, const void *x_dev, bool is_f64, double alpha, int bits,
                           const double *u_dev, uint64_t *q_dev)
{
    if (n == 0) return hipSuccess;
    const double scale = static_cast<double>((1ull << bits) - 1);
    if (is_f64)
        hipLaunchKernelGGL(quantize_kernel<double>, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, n,
                           static_cast<const double *>(x_dev), alpha, scale, 2 * alpha, u_dev, q_dev);
    else
        hipLaunchKernelGGL(quantize_kernel<float>, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, n,
                           static_cast<const float *>(x_dev), static_cast<float>(alpha), static_cast<float>(scale),
                           static_cast<float>(2 * alpha), u_dev, q_dev);
    return hipGetLastError();
}

// the back end of a flattened model alone (the values are already plaintext sums, e.g. after the sparse decrypt): element k of
// [first, first + count) comes back as float64 with its layer's parameters
__global__ __launch_bounds__(kStreamThreads) void unquantize_model_kernel(uint64_t count, const uint64_t *v, int v_limbs, const Codec cq, double *out)
{
    for (uint64_t k = static_cast<uint64_t>(blockIdx.x) * kStreamThreads + threadIdx.x; k < count; k += static_cast<uint64_t>(gridDim.x) * kStreamThreads) {
        const u128 x = v_limbs == 2 ? ld128_nt(v + 2 * k) : static_cast<u128>(__builtin_nontemporal_load(v + k));
        __builtin_nontemporal_store(codec_unquantize(cq, k, x), out + k);
    }
}

hipError_t launch_unquantize_model(const LaunchEnv &env, uint64_t count, const uint64_t *v_dev, const Codec &cq, double *out_dev)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(unquantize_model_kernel, dim3(stream_grid(env, count)), dim3(kStreamThreads), 0, env.stream, count, v_dev, env.b > 64 ? 2 : 1, cq,
                       out_dev);
    return hipGetLastError();
}

hipError_t launch_unquantize(const LaunchEnv &env, uint64_t n, const uint64_t *v_dev, int v_limbs, double alpha, int bits,
                             int num_clients, double *out_dev)
{
    if (n == 0) return hipSuccess;
    const double ac = alpha * static_cast<double>(num_clients);
    const double den = static_cast<double>(((1ull << bits) - 1) * static_cast<uint64_t>(num_clients));
    hipLaunchKernelGGL(unquantize_kernel, dim3(stream_grid(env, n)), dim3(kStreamThreads), 0, env.stream, n, v_dev, v_limbs, ac,
                       2 * ac, den, out_dev);
    return hipGetLastError();
}

hipError_t launch_batch(const LaunchEnv &env, uint64_t n, const uint64_t *vals_dev, int field_bits, uint64_t *out_dev)
{
    const int bs = env.b / field_bits;
    const uint64_t nb = (n + bs - 1) / bs;
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(batch_kernel, dim3(stream_grid(env, nb)), dim3(kStreamThreads), 0, env.stream, n, nb, vals_dev,
                       env.b > 64 ? 2 : 1, bs, field_bits, out_dev);
    return hipGetLastError();
}

hipError_t launch_unbatch(const LaunchEnv &env, uint64_t nb, const uint64_t *in_dev, int field_bits, uint64_t *out_dev)
{
    if (nb == 0) return hipSuccess;
    hipLaunchKernelGGL(unbatch_kernel, dim3(stream_grid(env, nb)), dim3(kStreamThreads), 0, env.stream, nb, in_dev,
                       env.b > 64 ? 2 : 1, env.b / field_bits, field_bits, out_dev);
    return hipGetLastError();
}

}  // namespace flashe
