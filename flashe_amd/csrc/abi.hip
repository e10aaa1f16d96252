// C ABI of libflashe_hip.so (see include/flashe.h).  Host-side glue only: context, key
// schedule, device buffers, argument checks, and the host-pointer convenience wrappers.
#include "ctx.h"
#include "blockpool.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace flashe;

// ------------------------------------------------------------------------------------------
// Host AES-256 (key schedule + one-block encrypt).  GF(2^8) via log/antilog tables, generator 3.
// ------------------------------------------------------------------------------------------
namespace {

struct AesTables {
    uint8_t sbox[256];
    uint32_t te0[256];
    AesTables()
    {
        uint8_t alog[256], lg[256];
        uint8_t x = 1;
        for (int i = 0; i < 255; i++) {
            alog[i] = x; lg[x] = static_cast<uint8_t>(i);
            x = static_cast<uint8_t>(x ^ (x << 1) ^ ((x & 0x80) ? 0x1b : 0));   // x *= 3
        }
        alog[255] = alog[0];
        for (int v = 0; v < 256; v++) {
            const uint8_t inv = v ? alog[(255 - lg[v]) % 255] : 0;
            uint8_t s = inv, r = inv;
            for (int k = 0; k < 4; k++) { r = static_cast<uint8_t>((r << 1) | (r >> 7)); s ^= r; }
            sbox[v] = static_cast<uint8_t>(s ^ 0x63);
        }
        for (int v = 0; v < 256; v++) {
            const uint8_t s = sbox[v];
            const uint8_t s2 = static_cast<uint8_t>((s << 1) ^ ((s & 0x80) ? 0x1b : 0));
            const uint8_t s3 = static_cast<uint8_t>(s2 ^ s);
            te0[v] = (static_cast<uint32_t>(s2) << 24) | (static_cast<uint32_t>(s) << 16) |
                     (static_cast<uint32_t>(s) << 8) | s3;
        }
    }
};

const AesTables &aes_tables()
{
    static const AesTables t;
    return t;
}

uint32_t sub_word(uint32_t w)
{
    const uint8_t *sb = aes_tables().sbox;
    return (static_cast<uint32_t>(sb[w >> 24]) << 24) | (static_cast<uint32_t>(sb[(w >> 16) & 0xff]) << 16) |
           (static_cast<uint32_t>(sb[(w >> 8) & 0xff]) << 8) | sb[w & 0xff];
}

void expand_key(const uint8_t key[32], RoundKeys *rk)
{
    uint32_t *w = rk->w;
    for (int i = 0; i < 8; i++)
        w[i] = (static_cast<uint32_t>(key[4 * i]) << 24) | (static_cast<uint32_t>(key[4 * i + 1]) << 16) |
               (static_cast<uint32_t>(key[4 * i + 2]) << 8) | key[4 * i + 3];
    uint32_t rcon = 0x01000000u;
    for (int i = 8; i < 60; i++) {
        uint32_t t = w[i - 1];
        if (i % 8 == 0) {
            t = sub_word((t << 8) | (t >> 24)) ^ rcon;
            rcon = (rcon << 1) ^ ((rcon & 0x80000000u) ? 0x1b000000u : 0);
        } else if (i % 8 == 4) {
            t = sub_word(t);
        }
        w[i] = w[i - 8] ^ t;
    }
}

inline uint32_t ror(uint32_t v, int r) { return (v >> r) | (v << (32 - r)); }

void host_encrypt_block(const RoundKeys &rk, const uint8_t in[16], uint8_t out[16])
{
    const AesTables &T = aes_tables();
    uint32_t s[4], t[4];
    for (int i = 0; i < 4; i++)
        s[i] = ((static_cast<uint32_t>(in[4 * i]) << 24) | (static_cast<uint32_t>(in[4 * i + 1]) << 16) |
                (static_cast<uint32_t>(in[4 * i + 2]) << 8) | in[4 * i + 3]) ^ rk.w[i];
    for (int r = 1; r < 14; r++) {
        for (int j = 0; j < 4; j++)
            t[j] = T.te0[s[j] >> 24] ^ ror(T.te0[(s[(j + 1) & 3] >> 16) & 0xff], 8) ^
                   ror(T.te0[(s[(j + 2) & 3] >> 8) & 0xff], 16) ^ ror(T.te0[s[(j + 3) & 3] & 0xff], 24) ^ rk.w[4 * r + j];
        memcpy(s, t, sizeof s);
    }
    for (int j = 0; j < 4; j++) {
        t[j] = ((static_cast<uint32_t>(T.sbox[s[j] >> 24]) << 24) |
                (static_cast<uint32_t>(T.sbox[(s[(j + 1) & 3] >> 16) & 0xff]) << 16) |
                (static_cast<uint32_t>(T.sbox[(s[(j + 2) & 3] >> 8) & 0xff]) << 8) |
                T.sbox[s[(j + 3) & 3] & 0xff]) ^ rk.w[56 + j];
    }
    for (int i = 0; i < 4; i++) {
        out[4 * i] = static_cast<uint8_t>(t[i] >> 24); out[4 * i + 1] = static_cast<uint8_t>(t[i] >> 16);
        out[4 * i + 2] = static_cast<uint8_t>(t[i] >> 8); out[4 * i + 3] = static_cast<uint8_t>(t[i]);
    }
}

thread_local std::string g_create_error;

}  // namespace

// ------------------------------------------------------------------------------------------
// Context (struct flashe_ctx: ctx.h)
// ------------------------------------------------------------------------------------------
struct flashe_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint32_t key_epoch = 0;
};

namespace flashe_host {

int fail(flashe_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

}  // namespace flashe_host

using flashe_host::fail;

namespace {

int ensure(flashe_ctx *ctx, flashe_ctx::Buf &b, size_t bytes)
{
    if (bytes <= b.cap) return FLASHE_OK;
    if (ctx->capturing)
        return fail(ctx, FLASHE_EINVAL, "scratch memory cannot grow while a graph is being captured: run the sequence once before flashe_graph_begin");
    if (b.p) { HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream)); HIP_TRY(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t cap = std::max<size_t>(bytes, 4096);
    HIP_TRY(ctx, hipMalloc(&b.p, cap));
    b.cap = cap;
    return FLASHE_OK;
}

// Zero memory that held key material in a way the optimiser may not drop.
void wipe(void *p, size_t bytes)
{
    volatile uint8_t *q = static_cast<volatile uint8_t *>(p);
    for (size_t i = 0; i < bytes; i++) q[i] = 0;
}

// Device copy of the expanded key for the bit-sliced PRF (it scalar-loads 4 words per round).
int upload_key_words(flashe_ctx *ctx)
{
    HIP_TRY(ctx, hipMemcpyAsync(ctx->rkw_dev, ctx->env.rk.w, sizeof(ctx->env.rk.w), hipMemcpyHostToDevice, ctx->env.stream));
    // packed key planes of the 16-blocks-per-lane bit-sliced PRF: word 64*r + 8*B + k holds bit k of round-key
    // byte B in its low 16 bits and of byte B + 8 in its high 16 bits (byte b = byte b%4 of word 4r + b/4)
    std::vector<uint32_t> planes(15 * 64);
    for (int r = 0; r < 15; r++)
        for (int B = 0; B < 8; B++)
            for (int k = 0; k < 8; k++) {
                auto bit = [&](int byte) { return (ctx->env.rk.w[4 * r + byte / 4] >> (24 - 8 * (byte % 4) + k)) & 1u; };
                planes[64 * r + 8 * B + k] = (bit(B) ? 0xffffu : 0u) | (bit(B + 8) ? 0xffff0000u : 0u);
            }
    hipError_t e = hipMemcpyAsync(ctx->rkp_dev, planes.data(), planes.size() * 4, hipMemcpyHostToDevice, ctx->env.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->env.stream);
    wipe(planes.data(), planes.size() * 4);
    HIP_TRY(ctx, e);
    return FLASHE_OK;
}

// After a synchronisation point: did a sparse kernel meet (and skip) a location >= total since the last check?
int check_device_flag(flashe_ctx *ctx)
{
    if (ctx->err_flag_host && *static_cast<volatile uint32_t *>(ctx->err_flag_host)) {
        *static_cast<volatile uint32_t *>(ctx->err_flag_host) = 0;
        return fail(ctx, FLASHE_EINVAL, "a sparse location list held a position >= total, or a list passed as sorted was not "
                                        "strictly increasing (the offending entries were skipped, nothing was written out of bounds)");
    }
    return FLASHE_OK;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

size_t vec_bytes(const flashe_ctx *ctx, uint64_t n) { return static_cast<size_t>(n) * ctx->limbs * 8; }

// RAII temp device buffer for the host-pointer wrappers.  Staging blocks are kept by the ctx and reused: a hipMalloc + hipFree
// pair of a 160 MB block costs more than moving 160 MB over PCIe Gen5 on this platform (measured 7 ms against 2.9 ms,
// tests/perf/e2e_calls.py).  The blocks a ctx keeps are bounded by a byte budget (FLASHE_STAGING_POOL_MB, default 8 GiB of the
// 288 GB); beyond it the largest free block makes room, and a request that still does not fit is a plain allocation.
size_t pool_budget()
{
    static const size_t v = [] {
        const char *e = getenv("FLASHE_STAGING_POOL_MB");
        return (e && atoll(e) >= 0 ? static_cast<size_t>(atoll(e)) : static_cast<size_t>(8192)) << 20;
    }();
    return v;
}

struct HipBackend final : flashe_pool::Backend {
    int alloc(void **p, size_t bytes) override { const hipError_t e = hipMalloc(p, bytes); if (e != hipSuccess) (void)hipGetLastError(); return static_cast<int>(e); }
    int release(void *p) override { return static_cast<int>(hipFree(p)); }
    int sync_all() override { return static_cast<int>(hipDeviceSynchronize()); }
    void wipe(void *p, size_t bytes) override { (void)hipMemset(p, 0, bytes); }
};
HipBackend &hip_backend() { static HipBackend b; return b; }

struct Tmp {
    void *p = nullptr;
    flashe_ctx *owner = nullptr;
    int slot = -1;
    ~Tmp()
    {
        if (!p) return;
        if (slot >= 0) owner->staging->give_back(slot);
        else (void)hipFree(p);
    }
    hipError_t alloc(flashe_ctx *ctx, size_t bytes)
    {
        if (!ctx->staging) ctx->staging = new flashe_pool::StagingPool(&hip_backend(), pool_budget());
        owner = ctx;
        return static_cast<hipError_t>(ctx->staging->lease(bytes, &p, &slot));
    }
    template <class T> T *as() { return static_cast<T *>(p); }
};

}  // namespace

extern "C" {

int flashe_abi_version(void) { return FLASHE_ABI_VERSION; }

int flashe_device_count(int *count)
{
    if (!count) return FLASHE_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return FLASHE_ENODEV; }
    *count = n;
    return FLASHE_OK;
}

int flashe_device_peer_access(int device, int peer, int *can_access)
{
    if (!can_access) return FLASHE_EINVAL;
    *can_access = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return FLASHE_ENODEV;
    if (device < 0 || peer < 0 || device >= n || peer >= n) return FLASHE_EINVAL;
    if (device == peer) { *can_access = 1; return FLASHE_OK; }
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, device, peer) != hipSuccess) { (void)hipGetLastError(); return FLASHE_EIO; }
    *can_access = can ? 1 : 0;
    return FLASHE_OK;
}

int flashe_limbs(int int_bits) { return int_bits < 1 || int_bits > 128 ? 0 : (int_bits > 64 ? 2 : 1); }

int flashe_ctx_create(flashe_ctx **out, const uint8_t key[32], int int_bits, int device, void *stream)
{
    if (!out || !key) return fail(nullptr, FLASHE_EINVAL, "flashe_ctx_create: null argument");
    *out = nullptr;
    if (int_bits < 1 || int_bits > 128) return fail(nullptr, FLASHE_EINVAL, "int_bits must be in [1, 128], got %d", int_bits);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, FLASHE_ENODEV, "no HIP device available (this engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, FLASHE_ENODEV, "device %d out of range (0..%d)", device, ndev - 1);
    flashe_ctx *ctx = new (std::nothrow) flashe_ctx();
    if (!ctx) return fail(nullptr, FLASHE_ENOMEM, "out of host memory");
    ctx->device = device;
    ctx->int_bits = int_bits;
    ctx->limbs = int_bits > 64 ? 2 : 1;
    // every failure path releases what was created so far (and wipes the key) through flashe_ctx_destroy
    auto bail = [&](int code, const char *what, hipError_t e) {
        fail(nullptr, code, "%s: %s", what, hipGetErrorString(e));
        (void)flashe_ctx_destroy(ctx);
        return code;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(FLASHE_EIO, "hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail(FLASHE_EIO, "hipGetDeviceProperties", e);
    if (stream) { ctx->env.stream = static_cast<hipStream_t>(stream); }
    else {
        if ((e = hipStreamCreateWithFlags(&ctx->env.stream, hipStreamNonBlocking)) != hipSuccess)
            return bail(FLASHE_EIO, "hipStreamCreate", e);
        ctx->own_stream = true;
    }
    ctx->env.num_cus = ctx->device_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->env.b = int_bits;
    // Te0 followed by Te1..Te3 (byte rotations of Te0): 4 KiB; the kernels fill LDS from the first KiB and use the
    // whole table for wave-uniform lookups through the scalar cache.  Word 1024 = the iter shift (kernels.hip), 0 by default.
    uint32_t te4[1024 + 4] = {};
    for (int t = 0; t < 4; t++)
        for (int x = 0; x < 256; x++) te4[256 * t + x] = t ? ror(aes_tables().te0[x], 8 * t) : aes_tables().te0[x];
    if ((e = hipMalloc(&ctx->te0_dev, sizeof(te4))) != hipSuccess) return bail(FLASHE_ENOMEM, "hipMalloc(te0)", e);
    if ((e = hipMemcpy(ctx->te0_dev, te4, sizeof(te4), hipMemcpyHostToDevice)) != hipSuccess)
        return bail(FLASHE_EIO, "hipMemcpy(te0)", e);
    ctx->env.te0_dev = ctx->te0_dev;
    if ((e = hipMalloc(&ctx->rkw_dev, 256)) != hipSuccess) return bail(FLASHE_ENOMEM, "hipMalloc(rkw)", e);
    ctx->env.rkw_dev = ctx->rkw_dev;
    if ((e = hipMalloc(&ctx->rkp_dev, 15 * 64 * 4)) != hipSuccess) return bail(FLASHE_ENOMEM, "hipMalloc(rkp)", e);
    ctx->env.rkp_dev = ctx->rkp_dev;
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&ctx->err_flag_host), 64, hipHostMallocMapped)) != hipSuccess)
        return bail(FLASHE_ENOMEM, "hipHostMalloc(error flag)", e);
    *ctx->err_flag_host = 0;
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->env.err_flag), ctx->err_flag_host, 0)) != hipSuccess)
        return bail(FLASHE_EIO, "hipHostGetDevicePointer", e);
    expand_key(key, &ctx->env.rk);
    if (upload_key_words(ctx) != FLASHE_OK) {
        g_create_error = ctx->err;
        (void)flashe_ctx_destroy(ctx);
        return FLASHE_EIO;
    }
    ctx->env.prf_backend = PRF_AUTO;
    ctx->env.hybrid_bs_permille = 300;
    if ((e = hipStreamCreateWithFlags(&ctx->env.stream2, hipStreamNonBlocking)) != hipSuccess) return bail(FLASHE_EIO, "hipStreamCreate(2)", e);
    if ((e = hipEventCreateWithFlags(&ctx->env.ev_fork, hipEventDisableTiming)) != hipSuccess) return bail(FLASHE_EIO, "hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->env.ev_join, hipEventDisableTiming)) != hipSuccess) return bail(FLASHE_EIO, "hipEventCreate", e);
    if (const char *be = getenv("FLASHE_PRF_BACKEND")) {
        if (!strcmp(be, "table")) ctx->env.prf_backend = PRF_TABLE;
#ifdef FLASHE_WITH_BITSLICE
        else if (!strcmp(be, "bitslice")) ctx->env.prf_backend = PRF_BITSLICE;
        else if (!strcmp(be, "hybrid")) ctx->env.prf_backend = PRF_HYBRID;
        else if (!strcmp(be, "bitslice16")) ctx->env.prf_backend = PRF_BITSLICE16;
#endif
    }
#ifdef FLASHE_TUNING
    if (const char *pm = getenv("FLASHE_HYBRID_BS_PERMILLE")) ctx->env.hybrid_bs_permille = atoi(pm);
#endif
    ctx->env.use_chain = 1;
    if (const char *ch = getenv("FLASHE_CHAIN")) ctx->env.use_chain = atoi(ch) != 0;   // 0: every job computes both of its streams (A/B runs)
    *out = ctx;
    return FLASHE_OK;
}

int flashe_ctx_destroy(flashe_ctx *ctx)
{
    if (!ctx) return FLASHE_EINVAL;
    (void)hipSetDevice(ctx->device);
    if (ctx->env.stream) (void)hipStreamSynchronize(ctx->env.stream);
    for (flashe_ctx::Buf *b : {&ctx->summaries, &ctx->stream_tmp, &ctx->acc_tmp[0], &ctx->acc_tmp[1], &ctx->sp_ws, &ctx->bounds, &ctx->mt_ws, &ctx->codec_tab, &ctx->prep_enc.add,
                               &ctx->prep_enc.minus, &ctx->prep_dec.add, &ctx->prep_dec.minus})
        if (b->p) (void)hipFree(b->p);
    if (ctx->staging) { ctx->staging->destroy(); delete ctx->staging; ctx->staging = nullptr; }      // staging blocks held plaintexts and ciphertexts (wiped)
    if (ctx->te0_dev) (void)hipFree(ctx->te0_dev);
    // the expanded AES-256 key leaves neither HBM nor host memory behind
    if (ctx->rkw_dev) { (void)hipMemset(ctx->rkw_dev, 0, 256); (void)hipFree(ctx->rkw_dev); }
    if (ctx->rkp_dev) { (void)hipMemset(ctx->rkp_dev, 0, 15 * 64 * 4); (void)hipFree(ctx->rkp_dev); }
    wipe(&ctx->env.rk, sizeof(ctx->env.rk));
    if (ctx->err_flag_host) (void)hipHostFree(ctx->err_flag_host);
    if (ctx->env.stream2) { (void)hipStreamSynchronize(ctx->env.stream2); (void)hipStreamDestroy(ctx->env.stream2); }
    if (ctx->env.ev_fork) (void)hipEventDestroy(ctx->env.ev_fork);
    if (ctx->env.ev_join) (void)hipEventDestroy(ctx->env.ev_join);
    for (hipEvent_t &ev : ctx->ev_copy) if (ev) (void)hipEventDestroy(ev);
    if (ctx->own_stream && ctx->env.stream) (void)hipStreamDestroy(ctx->env.stream);
    delete ctx;
    return FLASHE_OK;
}

int flashe_ctx_set_key(flashe_ctx *ctx, const uint8_t key[32])
{
    if (!ctx || !key) return FLASHE_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream));
    wipe(&ctx->env.rk, sizeof(ctx->env.rk));
    expand_key(key, &ctx->env.rk);
    ctx->key_epoch++;             // graphs captured under the previous key refuse to replay
    return upload_key_words(ctx);
}

int flashe_ctx_set_prf_backend(flashe_ctx *ctx, int backend)
{
    if (!ctx) return FLASHE_EINVAL;
    if (backend < PRF_AUTO || backend > PRF_BITSLICE16)
        return fail(ctx, FLASHE_EINVAL, "unknown PRF backend %d", backend);
#ifndef FLASHE_WITH_BITSLICE
    if (backend != PRF_AUTO && backend != PRF_TABLE)
        return fail(ctx, FLASHE_EINVAL, "PRF backend %d (bit-sliced) is not in this library: build libflashe_hip_bitslice.so with `make -C "
                                        "flashe_amd/csrc bitslice` and select it with FLASHE_LIB_NAME", backend);
#endif
    ctx->env.prf_backend = backend;
    return FLASHE_OK;
}

// The persistent launches of this ctx fill `cus` compute units instead of the whole device (0 = all again).  A PRF workgroup
// holds 128 KiB of a CU's 160 KiB of LDS and most of its vector registers, so a kernel that needs more than the remainder --
// RCCL's transfer kernels: 36.8 KiB of LDS, 248-256 VGPRs per lane -- only ever starts on a CU that has none: without a few
// free CUs an exchange on another stream would wait for the whole PRF launch instead of running beside it.
int flashe_ctx_set_cu_limit(flashe_ctx *ctx, int cus)
{
    CHECK_CTX(ctx);
    if (cus < 0) return fail(ctx, FLASHE_EINVAL, "flashe_ctx_set_cu_limit: negative count");
    ctx->env.num_cus = cus == 0 || cus > ctx->device_cus ? ctx->device_cus : cus;
    return FLASHE_OK;
}
int flashe_ctx_cu_count(const flashe_ctx *ctx) { return ctx ? ctx->device_cus : FLASHE_EINVAL; }

int flashe_ctx_int_bits(const flashe_ctx *ctx) { return ctx ? ctx->int_bits : FLASHE_EINVAL; }

// the conditions of check_u32 below, as a question a binding can ask before it chooses the uint32 layout
int flashe_ctx_compact_layout(const flashe_ctx *ctx)
{
    if (!ctx) return FLASHE_EINVAL;
    return ctx->int_bits <= 32 && (ctx->env.prf_backend == PRF_AUTO || ctx->env.prf_backend == PRF_TABLE) && ctx->env.use_chain ? 1 : 0;
}

const char *flashe_last_error(const flashe_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int flashe_selftest(flashe_ctx *ctx)
{
    CHECK_CTX(ctx);
    // FIPS-197 C.3 uses its own key; test with the ctx key against the host implementation
    // and with the FIPS key through a temporary key schedule.
    static const uint8_t fips_key[32] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
                                         16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31};
    static const uint8_t fips_pt[16] = {0x00, 0x11, 0x22, 0x33, 0x44, 0x55, 0x66, 0x77, 0x88, 0x99, 0xaa, 0xbb, 0xcc, 0xdd, 0xee, 0xff};
    static const uint8_t fips_ct[16] = {0x8e, 0xa2, 0xb7, 0xca, 0x51, 0x67, 0x45, 0xbf, 0xea, 0xfc, 0x49, 0x90, 0x4b, 0x49, 0x60, 0x89};
    const int nblk = 2048 + 3;
    std::vector<uint32_t> in(4 * nblk), out(4 * nblk);
    for (int i = 0; i < nblk; i++) {
        for (int j = 0; j < 4; j++)
            in[4 * i + j] = i == 0 ? ((static_cast<uint32_t>(fips_pt[4 * j]) << 24) | (static_cast<uint32_t>(fips_pt[4 * j + 1]) << 16) |
                                      (static_cast<uint32_t>(fips_pt[4 * j + 2]) << 8) | fips_pt[4 * j + 3])
                                   : static_cast<uint32_t>(0x9e3779b9u * (4 * i + j + 1) ^ (i << 7));
    }
    Tmp din, dout;
    HIP_TRY(ctx, din.alloc(ctx, in.size() * 4));
    HIP_TRY(ctx, dout.alloc(ctx, out.size() * 4));
    LaunchEnv env = ctx->env;
    expand_key(fips_key, &env.rk);
    HIP_TRY(ctx, hipMemcpyAsync(din.p, in.data(), in.size() * 4, hipMemcpyHostToDevice, env.stream));
    HIP_TRY(ctx, launch_aes_blocks(env, nblk, din.as<uint32_t>(), dout.as<uint32_t>()));
    HIP_TRY(ctx, hipMemcpyAsync(out.data(), dout.p, out.size() * 4, hipMemcpyDeviceToHost, env.stream));
    HIP_TRY(ctx, hipStreamSynchronize(env.stream));
    for (int i = 0; i < nblk; i++) {
        uint8_t bi[16], bo[16];
        for (int j = 0; j < 4; j++) {
            bi[4 * j] = static_cast<uint8_t>(in[4 * i + j] >> 24); bi[4 * j + 1] = static_cast<uint8_t>(in[4 * i + j] >> 16);
            bi[4 * j + 2] = static_cast<uint8_t>(in[4 * i + j] >> 8); bi[4 * j + 3] = static_cast<uint8_t>(in[4 * i + j]);
        }
        host_encrypt_block(env.rk, bi, bo);
        if (i == 0 && memcmp(bo, fips_ct, 16) != 0) return fail(ctx, FLASHE_EIO, "host AES fails FIPS-197 C.3");
        for (int j = 0; j < 4; j++) {
            const uint32_t w = (static_cast<uint32_t>(bo[4 * j]) << 24) | (static_cast<uint32_t>(bo[4 * j + 1]) << 16) |
                               (static_cast<uint32_t>(bo[4 * j + 2]) << 8) | bo[4 * j + 3];
            if (w != out[4 * i + j])
                return fail(ctx, FLASHE_EIO, "device AES mismatch at block %d word %d: got %08x want %08x", i, j, out[4 * i + j], w);
        }
    }
    return FLASHE_OK;
}

// ---- host logic ----
int flashe_chunks(uint64_t n, uint32_t n_jobs, uint64_t *begins)
{
    if (!begins || n_jobs == 0) return FLASHE_EINVAL;
    const uint64_t d = n / n_jobs, r = n % n_jobs;
    for (uint64_t i = 0; i <= n_jobs; i++) begins[i] = i < r ? (d + 1) * i : (d + 1) * r + d * (i - r);
    return FLASHE_OK;
}

int flashe_telescope(uint32_t *raw, int n_raw, uint32_t *add_out, uint32_t *minus_out, int *n_runs)
{
    if (n_raw < 0 || !n_runs || (n_raw && (!raw || !add_out || !minus_out))) return FLASHE_EINVAL;
    std::sort(raw, raw + n_raw);
    int k = 0;
    for (int i = 0; i < n_raw; i++) {
        // a value equal to the open run's end (last + 1) extends it; anything else, duplicates
        // included, opens a new run
        if (k > 0 && raw[i] == add_out[k - 1]) { add_out[k - 1] = raw[i] + 1; continue; }
        add_out[k] = raw[i] + 1; minus_out[k] = raw[i]; k++;
    }
    *n_runs = k;
    return FLASHE_OK;
}

int flashe_prp_block(const uint8_t key[32], const uint8_t in[16], uint8_t out[16])
{
    if (!key || !in || !out) return FLASHE_EINVAL;
    RoundKeys rk;
    expand_key(key, &rk);
    host_encrypt_block(rk, in, out);
    return FLASHE_OK;
}

// ---- memory / stream / events ----
namespace {

// one cache per device (flashe_dev_alloc / flashe_dev_free run with the ctx's device current); FLASHE_DEV_POOL_MB = how many MiB of
// freed blocks a device keeps parked (default 16 GiB of the 288 GB; 0 = every free is a hipFree)
flashe_pool::DeviceCache *dev_cache(int device)
{
    constexpr int kMaxDev = 64;
    static flashe_pool::DeviceCache *caches[kMaxDev] = {};
    static std::mutex mu;
    if (device < 0 || device >= kMaxDev) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!caches[device]) {
        const char *e = getenv("FLASHE_DEV_POOL_MB");
        const size_t mb = e && atoll(e) >= 0 ? static_cast<size_t>(atoll(e)) : static_cast<size_t>(16384);
        caches[device] = new flashe_pool::DeviceCache(&hip_backend(), mb << 20);
    }
    return caches[device];
}

}  // namespace

int flashe_dev_alloc(flashe_ctx *ctx, size_t bytes, void **dptr)
{
    CHECK_CTX(ctx);
    if (!dptr) return fail(ctx, FLASHE_EINVAL, "flashe_dev_alloc: null out pointer");
    flashe_pool::DeviceCache *cache = ctx->capturing ? nullptr : dev_cache(ctx->device);
    if (!cache) { HIP_TRY(ctx, hipMalloc(dptr, bytes ? bytes : 16)); return FLASHE_OK; }
    const int rc = cache->alloc(bytes, dptr);
    if (rc) HIP_TRY(ctx, static_cast<hipError_t>(rc));
    return FLASHE_OK;
}
int flashe_dev_free(flashe_ctx *ctx, void *dptr)
{
    CHECK_CTX(ctx);
    if (!dptr) return FLASHE_OK;
    flashe_pool::DeviceCache *cache = dev_cache(ctx->device);
    if (!cache) { HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream)); HIP_TRY(ctx, hipFree(dptr)); return FLASHE_OK; }
    // parked, not freed: the block is handed out again only after a device-wide synchronisation (blockpool.h), which is the
    // guarantee hipFree gave -- kernels of any stream that still read it finish first
    const int rc = cache->release(dptr);
    if (rc) HIP_TRY(ctx, static_cast<hipError_t>(rc));
    return FLASHE_OK;
}
int flashe_dev_trim(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FLASHE_ENODEV;
    if (hipSetDevice(device) != hipSuccess) return FLASHE_EIO;
    if (flashe_pool::DeviceCache *cache = dev_cache(device)) cache->trim();
    return FLASHE_OK;
}
int flashe_dev_pool_stats(int device, uint64_t *parked_bytes, uint64_t *hits, uint64_t *misses)
{
    flashe_pool::DeviceCache *cache = dev_cache(device);
    if (!cache) return FLASHE_EINVAL;
    if (parked_bytes) *parked_bytes = cache->held_bytes();
    if (hits) *hits = cache->hits();
    if (misses) *misses = cache->misses();
    return FLASHE_OK;
}
// Page-locked host memory for callers that keep their vectors on the host: the DMA engines read and write it directly (no
// staging copy, no page pinning per transfer), and a buffer that is REUSED spares the page faults a fresh allocation pays on
// its first transfer.  Not tied to a ctx: a buffer may outlive the ctx it was used with.
int flashe_host_alloc(size_t bytes, void **hptr)
{
    if (!hptr) return FLASHE_EINVAL;
    *hptr = nullptr;
    const hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 16, hipHostMallocPortable);
    if (e != hipSuccess) { *hptr = nullptr; return fail(nullptr, FLASHE_ENOMEM, "hipHostMalloc(%zu bytes): %s", bytes, hipGetErrorString(e)); }
    return FLASHE_OK;
}
int flashe_host_free(void *hptr)
{
    if (hptr && hipHostFree(hptr) != hipSuccess) return fail(nullptr, FLASHE_EIO, "hipHostFree failed");
    return FLASHE_OK;
}
int flashe_memcpy_h2d(flashe_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    CHECK_CTX(ctx);
    if (bytes) { HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->env.stream)); HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream)); }
    return FLASHE_OK;
}
int flashe_memcpy_d2h(flashe_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    CHECK_CTX(ctx);
    if (bytes) { HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->env.stream)); HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream)); }
    return check_device_flag(ctx);
}
int flashe_memcpy_d2d(flashe_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    CHECK_CTX(ctx);
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->env.stream));
    return FLASHE_OK;
}
int flashe_memset_dev(flashe_ctx *ctx, void *dst, int byte, size_t bytes)
{
    CHECK_CTX(ctx);
    if (bytes) HIP_TRY(ctx, hipMemsetAsync(dst, byte, bytes, ctx->env.stream));
    return FLASHE_OK;
}
int flashe_sync(flashe_ctx *ctx)
{
    CHECK_CTX(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream));
    return check_device_flag(ctx);
}
int flashe_event_create(flashe_ctx *ctx, void **event)
{
    CHECK_CTX(ctx);
    if (!event) return fail(ctx, FLASHE_EINVAL, "null event pointer");
    hipEvent_t ev;
    HIP_TRY(ctx, hipEventCreate(&ev));
    *event = ev;
    return FLASHE_OK;
}
int flashe_event_destroy(flashe_ctx *ctx, void *event)
{
    CHECK_CTX(ctx);
    if (event) HIP_TRY(ctx, hipEventDestroy(static_cast<hipEvent_t>(event)));
    return FLASHE_OK;
}
int flashe_event_record(flashe_ctx *ctx, void *event)
{
    CHECK_CTX(ctx);
    HIP_TRY(ctx, hipEventRecord(static_cast<hipEvent_t>(event), ctx->env.stream));
    return FLASHE_OK;
}
int flashe_stream_wait_event(flashe_ctx *ctx, void *event)
{
    CHECK_CTX(ctx);
    if (!event) return fail(ctx, FLASHE_EINVAL, "null event");
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->env.stream, static_cast<hipEvent_t>(event), 0));
    return FLASHE_OK;
}

int flashe_graph_begin(flashe_ctx *ctx)
{
    CHECK_CTX(ctx);
    if (ctx->capturing) return fail(ctx, FLASHE_EINVAL, "a capture is already in progress on this context");
    HIP_TRY(ctx, hipStreamBeginCapture(ctx->env.stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    return FLASHE_OK;
}

int flashe_graph_end(flashe_ctx *ctx, flashe_graph **graph)
{
    CHECK_CTX(ctx);
    if (!ctx->capturing) return fail(ctx, FLASHE_EINVAL, "no capture in progress");
    ctx->capturing = false;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(ctx->env.stream, &g);
    if (e != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        return fail(ctx, FLASHE_EIO, "hipStreamEndCapture: %s (a call made during the capture was not capturable)", hipGetErrorString(e));
    }
    if (!graph) { (void)hipGraphDestroy(g); return fail(ctx, FLASHE_EINVAL, "null graph pointer"); }
    flashe_graph *out = new (std::nothrow) flashe_graph();
    if (!out) { (void)hipGraphDestroy(g); return fail(ctx, FLASHE_ENOMEM, "out of host memory"); }
    out->graph = g;
    out->key_epoch = ctx->key_epoch;
    e = hipGraphInstantiate(&out->exec, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        delete out;
        return fail(ctx, FLASHE_EIO, "hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    *graph = out;
    return FLASHE_OK;
}

int flashe_graph_launch_shifted(flashe_ctx *ctx, flashe_graph *graph, uint32_t iter_shift)
{
    CHECK_CTX(ctx);
    if (!graph || !graph->exec) return fail(ctx, FLASHE_EINVAL, "null graph");
    if (ctx->capturing) return fail(ctx, FLASHE_EINVAL, "cannot launch a graph while capturing");
    if (graph->key_epoch != ctx->key_epoch)
        return fail(ctx, FLASHE_EINVAL, "the graph was captured under a different key (flashe_ctx_set_key since): its kernels carry the old "
                                        "key schedule in their argument blocks -- capture it again");
    // kernel arguments (iter included) are frozen into the graph; the shift is a device word every PRF kernel adds to its
    // iter at run time, set and reset in stream order around the replay
    uint32_t *shift = ctx->te0_dev + 1024;
    if (iter_shift) HIP_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(shift), static_cast<int>(iter_shift), 1, ctx->env.stream));
    HIP_TRY(ctx, hipGraphLaunch(graph->exec, ctx->env.stream));
    if (iter_shift) HIP_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(shift), 0, 1, ctx->env.stream));
    return FLASHE_OK;
}

int flashe_graph_launch(flashe_ctx *ctx, flashe_graph *graph) { return flashe_graph_launch_shifted(ctx, graph, 0); }

int flashe_graph_destroy(flashe_graph *graph)
{
    if (!graph) return FLASHE_OK;
    if (graph->exec) (void)hipGraphExecDestroy(graph->exec);
    if (graph->graph) (void)hipGraphDestroy(graph->graph);
    delete graph;
    return FLASHE_OK;
}

int flashe_event_elapsed_ms(flashe_ctx *ctx, void *start, void *stop, float *ms)
{
    CHECK_CTX(ctx);
    if (!ms) return fail(ctx, FLASHE_EINVAL, "null ms pointer");
    HIP_TRY(ctx, hipEventSynchronize(static_cast<hipEvent_t>(stop)));
    HIP_TRY(ctx, hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    return FLASHE_OK;
}

// ---- PRF / encrypt / decrypt ----
// The double mask of client idx subtracts the stream of prefix idx + 1, and the reference builds that prefix with
// (self.idx + 1).to_bytes(4, 'big') (jzf_flashe.py:352-353): OverflowError for idx = 2^32 - 1.  The raw ABI refuses the same value
// instead of wrapping to prefix 0 ("ranges: idx + 1 < 2^32").
static int check_double_idx(flashe_ctx *ctx, int scheme, const uint32_t *idx, int n_idx)
{
    if (scheme != FLASHE_SCHEME_DOUBLE || !idx) return FLASHE_OK;
    for (int v = 0; v < n_idx; v++)
        if (idx[v] == 0xffffffffu)
            return fail(ctx, FLASHE_EINVAL, "double mask: idx + 1 = 2^32 does not fit the 4-byte prefix field (entry %d; the reference raises OverflowError, jzf_flashe.py:352-353)", v);
    return FLASHE_OK;
}

static int check_prf_args(flashe_ctx *ctx, int n_add, int n_minus, uint32_t n_jobs, const void *out, const void *in, int in_limbs)
{
    if (n_add < 0 || n_minus < 0) return fail(ctx, FLASHE_EINVAL, "negative prefix list length: add %d minus %d", n_add, n_minus);
    if (n_jobs == 0) return fail(ctx, FLASHE_EINVAL, "n_jobs must be >= 1");
    if (in && in_limbs != 1 && in_limbs != ctx->limbs) return fail(ctx, FLASHE_EINVAL, "in_limbs must be 1 or %d, got %d", ctx->limbs, in_limbs);
    if (in && ctx->limbs == 1 && in_limbs != 1) return fail(ctx, FLASHE_EINVAL, "in_limbs must be 1 for int_bits <= 64");
    if (ctx->limbs == 2 && (!aligned16(out) || (in && in_limbs == 2 && !aligned16(in))))
        return fail(ctx, FLASHE_EINVAL, "device vectors of 2-limb elements must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in)) & 7u)
        return fail(ctx, FLASHE_EINVAL, "device vectors must be 8-byte aligned");
    return FLASHE_OK;
}

// out = in + sum term(add[k]) - sum term(minus[k]) over prefix lists of ANY length (the reference sums whatever set_idx_list
// produced: one minus prefix per uploaded client in single-mask mode, one pair per run of a dropout pattern,
// jzf_flashe.py:126-150,311-314).  One launch holds kMaxIdx prefixes per list in its argument block; longer lists go in
// several launches that accumulate in place (first pass in -> out, later passes out -> out: the sums are additive mod 2^b and
// every lane reads its element before it writes it).
static hipError_t prf_lists(flashe_ctx *ctx, uint32_t iter, const uint32_t *add, int n_add, const uint32_t *minus, int n_minus,
                            uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count, const uint64_t *in_dev, int in_limbs,
                            uint64_t *out_dev)
{
    int a = 0, m = 0;
    bool first_pass = true;
    while (first_pass || a < n_add || m < n_minus) {
        const int na = std::min(kMaxIdx, n_add - a), nm = std::min(kMaxIdx, n_minus - m);
        const hipError_t e = launch_prf(ctx->env, iter, add + a, na, minus + m, nm, n, n_jobs, first, count,
                                        first_pass ? in_dev : out_dev, first_pass ? in_limbs : ctx->limbs, out_dev);
        if (e != hipSuccess) return e;
        a += na; m += nm;
        first_pass = false;
    }
    return hipSuccess;
}

int flashe_mask_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *idx, int n_idx, uint64_t n, uint32_t n_jobs, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (n && !out_dev) return fail(ctx, FLASHE_EINVAL, "null output");
    int rc = check_prf_args(ctx, n_idx, 0, n_jobs, out_dev, nullptr, 0);
    if (rc) return rc;
    if (n_idx == 0) { if (n) HIP_TRY(ctx, hipMemsetAsync(out_dev, 0, vec_bytes(ctx, n), ctx->env.stream)); return FLASHE_OK; }
    HIP_TRY(ctx, prf_lists(ctx, iter, idx, n_idx, nullptr, 0, n, n_jobs, 0, n, nullptr, 0, out_dev));
    return FLASHE_OK;
}

int flashe_encrypt_dev(flashe_ctx *ctx, uint32_t iter, uint32_t idx, int scheme, uint64_t n, uint32_t n_jobs,
                       const uint64_t *pt_dev, int pt_limbs, uint64_t *ct_dev)
{
    CHECK_CTX(ctx);
    if (scheme != FLASHE_SCHEME_SINGLE && scheme != FLASHE_SCHEME_DOUBLE) return fail(ctx, FLASHE_EINVAL, "unknown scheme %d", scheme);
    if (n && (!pt_dev || !ct_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    int rc = check_prf_args(ctx, 1, scheme, n_jobs, ct_dev, pt_dev, pt_limbs);
    if (rc) return rc;
    if ((rc = check_double_idx(ctx, scheme, &idx, 1))) return rc;
    const uint32_t add = idx, minus = idx + 1;
    HIP_TRY(ctx, launch_prf(ctx->env, iter, &add, 1, &minus, scheme == FLASHE_SCHEME_DOUBLE ? 1 : 0, n, n_jobs, 0, n, pt_dev, pt_limbs, ct_dev));
    return FLASHE_OK;
}

int flashe_encrypt_batch_dev(flashe_ctx *ctx, uint32_t iter, int scheme, uint64_t n, uint32_t n_jobs, int n_vec, const uint32_t *idx,
                             const uint64_t *const *pt_dev, int pt_limbs, uint64_t *const *ct_dev)
{
    CHECK_CTX(ctx);
    if (scheme != FLASHE_SCHEME_SINGLE && scheme != FLASHE_SCHEME_DOUBLE) return fail(ctx, FLASHE_EINVAL, "unknown scheme %d", scheme);
    if (n_vec < 0 || (n_vec && (!idx || !pt_dev || !ct_dev))) return fail(ctx, FLASHE_EINVAL, "bad batch arguments");
    if (int rc = check_double_idx(ctx, scheme, idx, n_vec)) return rc;
    for (int v = 0; v < n_vec; v++) {
        if (n && (!pt_dev[v] || !ct_dev[v])) return fail(ctx, FLASHE_EINVAL, "null vector %d", v);
        int rc = check_prf_args(ctx, 1, scheme, n_jobs, ct_dev[v], pt_dev[v], pt_limbs);
        if (rc) return rc;
    }
    if (ctx->env.prf_backend != PRF_AUTO && ctx->env.prf_backend != PRF_TABLE) {
        // the alternative PRF backends go vector by vector
        for (int v = 0; v < n_vec; v++) {
            int rc = flashe_encrypt_dev(ctx, iter, idx[v], scheme, n, n_jobs, pt_dev[v], pt_limbs, ct_dev[v]);
            if (rc) return rc;
        }
        return FLASHE_OK;
    }
    // equal shares when several launches are needed (b <= 64, 100 vectors: 4 x 25, not 32 + 32 + 32 + 4); 2-limb vectors
    // travel in the compact table, up to 128 per launch
    // (the chained launch holds 128 outputs for every int_bits; without it the b <= 64 job table holds kMaxBatch)
    const int cap = (ctx->limbs == 2 || ctx->env.use_chain) ? kMaxUniformBatch : kMaxBatch;
    const int per_launch = n_vec ? (n_vec + (n_vec + cap - 1) / cap - 1) / ((n_vec + cap - 1) / cap) : 1;
    for (int v0 = 0; v0 < n_vec; v0 += per_launch) {
        const int nv = std::min(per_launch, n_vec - v0);
        HIP_TRY(ctx, launch_prf_batch(ctx->env, iter, scheme == FLASHE_SCHEME_DOUBLE, nv, idx + v0, pt_dev + v0, pt_limbs, ct_dev + v0, n,
                                      n_jobs));
    }
    return FLASHE_OK;
}

// ---- compact layout for int_bits <= 32 (new; no reference counterpart) ----
// The ABI stores one element per uint64 limb whatever int_bits is.  At the widths the reference's own jobs ship (int_bits = 20, 23)
// that is 8 bytes moved for 20 useful bits, and the b <= 32 kernels are bound by exactly those bytes.  The *_u32_dev entry points
// take and produce the same VALUES as uint32 arrays: the hot round (batched encrypt, reduce fused with the decrypt of its result)
// moves half the bytes; flashe_widen_u32_dev / flashe_narrow_u32_dev convert at the edges.
static int check_range(flashe_ctx *ctx, uint64_t n, uint64_t first, uint64_t count);
static int check_u32(flashe_ctx *ctx, uint64_t n, uint32_t n_jobs)
{
    if (ctx->int_bits > 32) return fail(ctx, FLASHE_EINVAL, "the uint32 layout needs int_bits <= 32, this ctx has %d", ctx->int_bits);
    if (n >= (1ull << 32)) return fail(ctx, FLASHE_EINVAL, "the uint32 layout needs n < 2^32");
    if (n_jobs == 0) return fail(ctx, FLASHE_EINVAL, "n_jobs must be >= 1");
    if (ctx->env.prf_backend != PRF_AUTO && ctx->env.prf_backend != PRF_TABLE) return fail(ctx, FLASHE_EINVAL, "the uint32 layout runs on the table PRF only");
    if (!ctx->env.use_chain) return fail(ctx, FLASHE_EINVAL, "the uint32 layout needs the chained kernels (FLASHE_CHAIN=0 is set)");
    return FLASHE_OK;
}

int flashe_encrypt_batch_u32_dev(flashe_ctx *ctx, uint32_t iter, int scheme, uint64_t n, uint32_t n_jobs, int n_vec, const uint32_t *idx,
                                 const uint32_t *const *pt_dev, uint32_t *const *ct_dev)
{
    CHECK_CTX(ctx);
    if (scheme != FLASHE_SCHEME_SINGLE && scheme != FLASHE_SCHEME_DOUBLE) return fail(ctx, FLASHE_EINVAL, "unknown scheme %d", scheme);
    if (n_vec < 0 || (n_vec && (!idx || !pt_dev || !ct_dev))) return fail(ctx, FLASHE_EINVAL, "bad batch arguments");
    int rc = check_u32(ctx, n, n_jobs);
    if (rc || (rc = check_double_idx(ctx, scheme, idx, n_vec))) return rc;
    if (n == 0 || n_vec == 0) return FLASHE_OK;
    for (int v = 0; v < n_vec; v++) {
        if (!pt_dev[v] || !ct_dev[v]) return fail(ctx, FLASHE_EINVAL, "null vector %d", v);
        if ((reinterpret_cast<uintptr_t>(pt_dev[v]) | reinterpret_cast<uintptr_t>(ct_dev[v])) & 3u) return fail(ctx, FLASHE_EINVAL, "vector %d: not 4-byte aligned", v);
    }
    LaunchEnv env = ctx->env;
    env.elem32 = 1;
    const int cap = kMaxUniformBatch;
    const int per_launch = (n_vec + (n_vec + cap - 1) / cap - 1) / ((n_vec + cap - 1) / cap);
    for (int v0 = 0; v0 < n_vec; v0 += per_launch) {
        const int nv = std::min(per_launch, n_vec - v0);
        const hipError_t e = launch_prf_batch(env, iter, scheme == FLASHE_SCHEME_DOUBLE, nv, idx + v0, reinterpret_cast<const uint64_t *const *>(pt_dev + v0), 1,
                                              reinterpret_cast<uint64_t *const *>(ct_dev + v0), n, n_jobs);
        HIP_TRY(ctx, e);
    }
    return FLASHE_OK;
}

// flashe_encrypt_batch_u32_dev AND sum_out_dev = sum_v ct[v] mod 2^b written by the same launch: the compact twin of
// flashe_encrypt_batch_sum_dev ("each GPU encrypts and locally mod-adds its share").  One launch for a run of
// consecutive clients under the double mask at the compiled-in widths (int_bits 16 / 20 / 23 / 24 / 32) when the launch is long enough for the paired kernel (the lane that
// owns a block keeps the running sum of its elements in registers); every other shape: the encrypts, then the reduce of what they wrote.
int flashe_encrypt_batch_sum_u32_dev(flashe_ctx *ctx, uint32_t iter, int scheme, uint64_t n, uint32_t n_jobs, int n_vec, const uint32_t *idx,
                                     const uint32_t *const *pt_dev, uint32_t *const *ct_dev, uint32_t *sum_out_dev)
{
    CHECK_CTX(ctx);
    // (n_vec == 0 clears the sum: it needs the vector just the same -- a null pointer must come back as FLASHE_EINVAL, not as a HIP error
    // from the memset)
    if (n && !sum_out_dev) return fail(ctx, FLASHE_EINVAL, "flashe_encrypt_batch_sum_u32_dev: null sum_out_dev");
    if (reinterpret_cast<uintptr_t>(sum_out_dev) & 3u) return fail(ctx, FLASHE_EINVAL, "sum_out_dev must be 4-byte aligned");
    for (int v = 0; v < n_vec && ct_dev; v++)
        if (n && (ct_dev[v] == sum_out_dev || (pt_dev && pt_dev[v] == sum_out_dev)))
            return fail(ctx, FLASHE_EINVAL, "sum_out_dev must not be one of the plaintext or ciphertext vectors");
    if (scheme == FLASHE_SCHEME_DOUBLE && n && n_vec > 0 && n_vec <= kMaxUniformBatch && idx && pt_dev && ct_dev) {
        int rc = check_u32(ctx, n, n_jobs);
        if (rc || (rc = check_double_idx(ctx, scheme, idx, n_vec))) return rc;
        bool ok = true;
        for (int v = 0; v < n_vec && ok; v++)
            ok = pt_dev[v] && ct_dev[v] && !((reinterpret_cast<uintptr_t>(pt_dev[v]) | reinterpret_cast<uintptr_t>(ct_dev[v])) & 3u) && (v == 0 || idx[v] == idx[v - 1] + 1u);
        if (ok) {
            LaunchEnv env = ctx->env;
            env.elem32 = 1;
            std::vector<uint32_t> sidx(idx, idx + n_vec);
            sidx.push_back(idx[n_vec - 1] + 1u);
            PrfChain ch{sidx.data(), n_vec, false, 0, n, reinterpret_cast<const uint64_t *const *>(pt_dev), 1, reinterpret_cast<uint64_t *const *>(ct_dev)};
            ch.sum_out_dev = reinterpret_cast<uint64_t *>(sum_out_dev);
            const hipError_t e = launch_prf_chains(env, iter, 1, &ch, n, n_jobs);
            if (e == hipSuccess) return FLASHE_OK;
            if (e != hipErrorNotSupported) HIP_TRY(ctx, e);
        }
    }
    int rc = flashe_encrypt_batch_u32_dev(ctx, iter, scheme, n, n_jobs, n_vec, idx, pt_dev, ct_dev);
    if (rc || n == 0) return rc;
    if (n_vec == 0) { HIP_TRY(ctx, hipMemsetAsync(sum_out_dev, 0, n * 4, ctx->env.stream)); return FLASHE_OK; }
    return flashe_aggregate_elem_u32_dev(ctx, n_vec, ct_dev, n, sum_out_dev);
}

int flashe_aggregate_decrypt_u32_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add, const uint32_t *minus_idx, int n_minus,
                                     uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count, int C, const uint32_t *const *cts_dev,
                                     void *agg_out_dev, void *out_dev, int out_elem_bytes)
{
    CHECK_CTX(ctx);
    if (C < 1 || C > kMaxOps || !cts_dev || (count && !out_dev)) return fail(ctx, FLASHE_EINVAL, "aggregate_decrypt_u32: bad arguments (C = %d, at most %d)", C, kMaxOps);
    if (out_elem_bytes != 4 && out_elem_bytes != 8) return fail(ctx, FLASHE_EINVAL, "out_elem_bytes must be 4 or 8");
    if (n_add != 1 || n_minus < 0 || n_minus > 1 || !add_idx || (n_minus && !minus_idx))
        return fail(ctx, FLASHE_EINVAL, "aggregate_decrypt_u32: one add and at most one minus prefix (widen the vectors for prefix lists)");
    int rc = check_u32(ctx, n, n_jobs);
    if (rc) return rc;
    rc = check_range(ctx, n, first, count);
    if (rc) return rc;
    if (count == 0) return FLASHE_OK;
    for (int c = 0; c < C; c++)
        if (!cts_dev[c] || (reinterpret_cast<uintptr_t>(cts_dev[c]) & 3u)) return fail(ctx, FLASHE_EINVAL, "operand %d is null or not 4-byte aligned", c);
    const uintptr_t need = static_cast<uintptr_t>(out_elem_bytes - 1);
    if ((reinterpret_cast<uintptr_t>(out_dev) | reinterpret_cast<uintptr_t>(agg_out_dev)) & need) return fail(ctx, FLASHE_EINVAL, "outputs must be %d-byte aligned", out_elem_bytes);
    LaunchEnv env = ctx->env;
    env.elem32 = 1;
    if (C == 1 && !agg_out_dev && out_elem_bytes == 4 && n_minus == 1) {
        // ONE operand (the partial aggregate an encrypt launch wrote: flashe_encrypt_batch_sum_u32_dev): out = in + S(add) - S(minus) is a
        // chain of one output -- two blocks per lane, software pipelined, compile-time width where there is one -- instead of the reduce's
        // one-block-per-lane tiles (ten 1e7-element clients at int_bits 20: 0.082 -> 0.06 ms)
        const uint32_t sidx[2] = {add_idx[0], minus_idx[0]};
        const uint64_t *in1 = reinterpret_cast<const uint64_t *>(cts_dev[0]);
        uint64_t *out1 = static_cast<uint64_t *>(out_dev);
        PrfChain ch{sidx, 1, false, first, count, &in1, 1, &out1};
        const hipError_t e = launch_prf_chains(env, iter, 1, &ch, n, n_jobs);
        if (e == hipSuccess) return FLASHE_OK;
        if (e != hipErrorNotSupported) HIP_TRY(ctx, e);
    }
    HIP_TRY(ctx, launch_small_reduce_decrypt(env, iter, add_idx[0], n_minus == 1, n_minus ? minus_idx[0] : 0u, n, n_jobs, first, count, C,
                                             reinterpret_cast<const uint64_t *const *>(cts_dev), static_cast<uint64_t *>(agg_out_dev),
                                             static_cast<uint64_t *>(out_dev), out_elem_bytes));
    return FLASHE_OK;
}

int flashe_aggregate_elem_u32_dev(flashe_ctx *ctx, int C, const uint32_t *const *cts_dev, uint64_t n, uint32_t *out_dev)
{
    CHECK_CTX(ctx);
    if (ctx->int_bits > 32) return fail(ctx, FLASHE_EINVAL, "the uint32 layout needs int_bits <= 32, this ctx has %d", ctx->int_bits);
    if (C < 1 || !cts_dev || (n && !out_dev)) return fail(ctx, FLASHE_EINVAL, "aggregate_elem_u32: bad arguments");
    for (int c = 0; c < C; c++)
        if (n && (!cts_dev[c] || (reinterpret_cast<uintptr_t>(cts_dev[c]) & 3u))) return fail(ctx, FLASHE_EINVAL, "operand %d is null or not 4-byte aligned", c);
    if (reinterpret_cast<uintptr_t>(out_dev) & 3u) return fail(ctx, FLASHE_EINVAL, "out_dev must be 4-byte aligned");
    // more operands than one pass holds: partial sums accumulate in out (every pass reads out as one operand)
    int done = 0;
    while (done < C) {
        const int take = std::min(C - done, done ? kMaxOps - 1 : kMaxOps);
        std::vector<const uint32_t *> ops;
        if (done) ops.push_back(out_dev);
        for (int c = 0; c < take; c++) ops.push_back(cts_dev[done + c]);
        HIP_TRY(ctx, launch_aggregate_elem_u32(ctx->env, static_cast<int>(ops.size()), ops.data(), n, out_dev));
        done += take;
    }
    return FLASHE_OK;
}

int flashe_widen_u32_dev(flashe_ctx *ctx, uint64_t n, const uint32_t *in_dev, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (n && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "widen_u32: null vector");
    HIP_TRY(ctx, launch_widen_u32(ctx->env, n, in_dev, out_dev));
    return FLASHE_OK;
}

int flashe_narrow_u32_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *in_dev, uint32_t *out_dev)
{
    CHECK_CTX(ctx);
    if (n && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "narrow_u32: null vector");
    HIP_TRY(ctx, launch_narrow_u32(ctx->env, n, in_dev, out_dev));
    return FLASHE_OK;
}

int flashe_encrypt_batch_sum_dev(flashe_ctx *ctx, uint32_t iter, int scheme, uint64_t n, uint32_t n_jobs, int n_vec, const uint32_t *idx,
                                 const uint64_t *const *pt_dev, int pt_limbs, uint64_t *const *ct_dev, uint64_t *sum_out_dev)
{
    CHECK_CTX(ctx);
    if (!sum_out_dev && n && n_vec) return fail(ctx, FLASHE_EINVAL, "flashe_encrypt_batch_sum_dev: null sum_out_dev");
    if (n_vec > 0) { if (int rc = check_double_idx(ctx, scheme, idx, n_vec)) return rc; }
    if (n && n_vec && ((ctx->limbs == 2 && !aligned16(sum_out_dev)) || (reinterpret_cast<uintptr_t>(sum_out_dev) & 7u)))
        return fail(ctx, FLASHE_EINVAL, "sum_out_dev must be aligned like a ciphertext vector");
    if (n_vec > 0 && n) {
        for (int v = 0; v < n_vec; v++)
            if ((ct_dev && ct_dev[v] == sum_out_dev) || (pt_dev && pt_dev[v] == sum_out_dev))
                return fail(ctx, FLASHE_EINVAL, "sum_out_dev must not be one of the plaintext or ciphertext vectors");
    }
    if (scheme == FLASHE_SCHEME_DOUBLE && n && n_vec > 0 && idx && pt_dev && ct_dev) {
        // one launch: the chained encrypt keeps the running sum of its outputs in registers and stores it once
        bool ok = true;
        for (int v = 0; v < n_vec && ok; v++) ok = pt_dev[v] && ct_dev[v];
        if (ok) {
            for (int v = 0; v < n_vec; v++) {
                int rc = check_prf_args(ctx, 1, scheme, n_jobs, ct_dev[v], pt_dev[v], pt_limbs);
                if (rc) return rc;
            }
            const hipError_t e = launch_prf_batch_sum(ctx->env, iter, n_vec, idx, pt_dev, pt_limbs, ct_dev, sum_out_dev, n, n_jobs, 0, n);
            if (e == hipSuccess) return FLASHE_OK;
            if (e != hipErrorNotSupported) HIP_TRY(ctx, e);
        }
    }
    // every other shape: the encrypts, then the reduce of what they wrote
    int rc = flashe_encrypt_batch_dev(ctx, iter, scheme, n, n_jobs, n_vec, idx, pt_dev, pt_limbs, ct_dev);
    if (rc || n == 0) return rc;
    if (n_vec == 0) { HIP_TRY(ctx, hipMemsetAsync(sum_out_dev, 0, n * ctx->limbs * 8, ctx->env.stream)); return FLASHE_OK; }
    return flashe_aggregate_elem_dev(ctx, n_vec, ct_dev, n, sum_out_dev);
}

int flashe_prf_jobs_dev(flashe_ctx *ctx, uint32_t iter, uint64_t n, uint32_t n_jobs, int n_entries, const flashe_prf_job *entries)
{
    CHECK_CTX(ctx);
    if (n_entries < 0 || (n_entries && !entries)) return fail(ctx, FLASHE_EINVAL, "bad job list");
    if (n_entries == 0) return FLASHE_OK;
    const int dbl = entries[0].has_minus;
    for (int e = 0; e < n_entries; e++) {
        const flashe_prf_job &j = entries[e];
        if (j.has_minus != dbl || (dbl != 0 && dbl != 1)) return fail(ctx, FLASHE_EINVAL, "entry %d: has_minus must be 0 or 1 and equal across the call", e);
        if (j.first > n || j.count > n - j.first) return fail(ctx, FLASHE_EINVAL, "entry %d: range exceeds n", e);
        if (j.count && !j.out_dev) return fail(ctx, FLASHE_EINVAL, "entry %d: null output", e);
        if (j.reserved) return fail(ctx, FLASHE_EINVAL, "entry %d: reserved field must be 0", e);
        if (j.n_in > 1 && (ctx->limbs != 2 || !j.in_dev || j.in_limbs != 2 || j.n_in > 255 || !aligned16(j.in_dev + 2 * j.in_stride) ||
                           (j.sum_out_dev && !aligned16(j.sum_out_dev))))
            return fail(ctx, FLASHE_EINVAL, "entry %d: a summed input needs int_bits > 64, 2-limb 16-byte aligned vectors and n_in <= 255", e);
        int rc = check_prf_args(ctx, 1, dbl, n_jobs, j.out_dev, j.in_dev, j.in_dev ? j.in_limbs : 0);
        if (rc) return rc;
    }
    const bool one_launch = ctx->env.prf_backend == PRF_AUTO || ctx->env.prf_backend == PRF_TABLE || ctx->limbs == 1;
    if (!one_launch) {
        for (int e = 0; e < n_entries; e++) {
            const flashe_prf_job &j = entries[e];
            if (j.n_in > 1) return fail(ctx, FLASHE_EINVAL, "entry %d: summed inputs need the table PRF backend", e);
            HIP_TRY(ctx, launch_prf(ctx->env, iter, &j.add_idx, 1, &j.minus_idx, dbl, n, n_jobs, j.first, j.count, j.in_dev,
                                    j.in_dev ? j.in_limbs : 0, j.out_dev));
        }
        return FLASHE_OK;
    }
    std::vector<PrfJob> jobs(n_entries);
    for (int e = 0; e < n_entries; e++) {
        const flashe_prf_job &j = entries[e];
        jobs[e] = PrfJob{j.add_idx, j.minus_idx, j.first, j.count, j.in_dev, j.in_limbs, j.out_dev,
                         j.n_in ? j.n_in : 1u, j.in_stride * 2, j.n_in > 1 ? j.sum_out_dev : nullptr};
    }
    HIP_TRY(ctx, launch_prf_jobs(ctx->env, iter, dbl != 0, n_entries, jobs.data(), n, n_jobs));
    return FLASHE_OK;
}

int flashe_decrypt_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add, const uint32_t *minus_idx, int n_minus,
                       uint64_t n, uint32_t n_jobs, const uint64_t *in_dev, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (n && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    int rc = check_prf_args(ctx, n_add, n_minus, n_jobs, out_dev, in_dev, ctx->limbs);
    if (rc) return rc;
    if (n_add == 0 && n_minus == 0) {
        HIP_TRY(ctx, launch_combine(ctx->env, n, in_dev, ctx->limbs, nullptr, nullptr, out_dev));
        return FLASHE_OK;
    }
    HIP_TRY(ctx, prf_lists(ctx, iter, add_idx, n_add, minus_idx, n_minus, n, n_jobs, 0, n, in_dev, ctx->limbs, out_dev));
    return FLASHE_OK;
}

// ---- fused codec: quantise -> encrypt and decrypt -> unquantise in ONE launch each ----
static int check_codec_bits(flashe_ctx *ctx, int element_bits);
int flashe_quantize_encrypt_dev(flashe_ctx *ctx, uint32_t iter, uint32_t idx, int scheme, uint64_t n, uint32_t n_jobs, const void *x_dev,
                                int x_is_f64, double alpha, int element_bits, const double *u_dev, uint64_t *ct_dev)
{
    CHECK_CTX(ctx);
    if (scheme != FLASHE_SCHEME_SINGLE && scheme != FLASHE_SCHEME_DOUBLE) return fail(ctx, FLASHE_EINVAL, "unknown scheme %d", scheme);
    if (n && (!x_dev || !u_dev || !ct_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (!(alpha > 0)) return fail(ctx, FLASHE_EINVAL, "alpha must be positive");
    if (element_bits < 1 || element_bits > 62 || element_bits > ctx->int_bits)
        return fail(ctx, FLASHE_EINVAL, "element_bits must be in [1, min(62, int_bits)], got %d", element_bits);
    int rc = check_prf_args(ctx, 1, scheme, n_jobs, ct_dev, nullptr, 0);
    if (rc) return rc;
    const Codec cq = codec_quantize_front(x_dev, x_is_f64 != 0, alpha, element_bits, u_dev);
    LaunchEnv env = ctx->env;
    env.codec = &cq;
    if ((rc = check_double_idx(ctx, scheme, &idx, 1))) return rc;
    const uint32_t add = idx, minus = idx + 1;
    HIP_TRY(ctx, launch_prf(env, iter, &add, 1, &minus, scheme == FLASHE_SCHEME_DOUBLE ? 1 : 0, n, n_jobs, 0, n, nullptr, 0, ct_dev));
    return FLASHE_OK;
}

int flashe_decrypt_unquantize_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add, const uint32_t *minus_idx, int n_minus,
                                  uint64_t n, uint32_t n_jobs, const uint64_t *in_dev, double alpha, int element_bits, int num_clients,
                                  double *out_dev)
{
    CHECK_CTX(ctx);
    if (n && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (!(alpha > 0) || num_clients < 1) return fail(ctx, FLASHE_EINVAL, "alpha must be positive and num_clients >= 1");
    int rc = check_codec_bits(ctx, element_bits);
    if (rc) return rc;
    rc = check_prf_args(ctx, n_add, n_minus, n_jobs, in_dev, in_dev, ctx->limbs);
    if (rc) return rc;
    if (n == 0) return FLASHE_OK;
    Codec cq{};
    codec_unquantize_back(&cq, alpha, element_bits, num_clients, out_dev);
    if (n_add == 0 && n_minus == 0) {
        // nothing to unmask: reduce mod 2^b like every decrypt does, then unquantise (two launches; not a shape a round produces)
        rc = ensure(ctx, ctx->stream_tmp, vec_bytes(ctx, n));
        if (rc) return rc;
        uint64_t *tmp = static_cast<uint64_t *>(ctx->stream_tmp.p);
        HIP_TRY(ctx, launch_combine(ctx->env, n, in_dev, ctx->limbs, nullptr, nullptr, tmp));
        HIP_TRY(ctx, launch_unquantize(ctx->env, n, tmp, ctx->limbs, alpha, element_bits, num_clients, out_dev));
        return FLASHE_OK;
    }
    // lists longer than one launch holds: all but the last group accumulate into ctx scratch, the last launch writes the floats
    const uint64_t *src = in_dev;
    int a = 0, m = 0;
    if (n_add > kMaxIdx || n_minus > kMaxIdx) {
        rc = ensure(ctx, ctx->stream_tmp, vec_bytes(ctx, n));
        if (rc) return rc;
        uint64_t *tmp = static_cast<uint64_t *>(ctx->stream_tmp.p);
        while (n_add - a > kMaxIdx || n_minus - m > kMaxIdx) {
            const int na = std::min(kMaxIdx, n_add - a), nm = std::min(kMaxIdx, n_minus - m);
            HIP_TRY(ctx, launch_prf(ctx->env, iter, add_idx + a, na, minus_idx + m, nm, n, n_jobs, 0, n, src, ctx->limbs, tmp));
            a += na; m += nm; src = tmp;
        }
    }
    LaunchEnv env = ctx->env;
    env.codec = &cq;
    HIP_TRY(ctx, launch_prf(env, iter, add_idx + a, n_add - a, minus_idx + m, n_minus - m, n, n_jobs, 0, n, src, ctx->limbs,
                            const_cast<uint64_t *>(src)));
    return FLASHE_OK;
}

// ---- the same over a flattened model: one launch, per-layer parameters from a device table (jzf_aggregator.py:721-741, :887-899) ----
static int check_range(flashe_ctx *ctx, uint64_t n, uint64_t first, uint64_t count);
// validates the caller's table and stages the entries of the non-empty layers on the device (ctx->codec_tab)
static int stage_codec_layers(flashe_ctx *ctx, uint64_t n, const flashe_codec_layer *layers, int n_layers, bool front, int element_bits,
                              int num_clients, uint64_t first, uint64_t count, const CodecLayer **tab_dev, int *n_tab)
{
    if (n_layers < 1 || !layers) return fail(ctx, FLASHE_EINVAL, "the layer table needs at least one entry");
    if (layers[0].start != 0) return fail(ctx, FLASHE_EINVAL, "layers[0].start must be 0");
    if (ctx->capturing) return fail(ctx, FLASHE_EINVAL, "the model-wide codec calls stage their layer table per call and cannot be captured into a graph");
    std::vector<CodecLayer> tab;
    tab.reserve(static_cast<size_t>(n_layers));
    for (int l = 0; l < n_layers; l++) {
        const flashe_codec_layer &e = layers[l];
        const uint64_t end = l + 1 < n_layers ? layers[l + 1].start : n;
        if (e.start > end || end > n) return fail(ctx, FLASHE_EINVAL, "layer %d: starts must ascend and stay within n", l);
        if (e.reserved) return fail(ctx, FLASHE_EINVAL, "layer %d: reserved field must be 0", l);
        if (e.start == end) continue;                                   // an empty layer holds no element
        if (!(e.alpha > 0)) return fail(ctx, FLASHE_EINVAL, "layer %d: alpha must be positive", l);
        const bool touched = e.start < first + count && end > first;
        if (front && touched && !e.x_dev) return fail(ctx, FLASHE_EINVAL, "layer %d: null x_dev", l);
        if (front && (reinterpret_cast<uintptr_t>(e.x_dev) & (e.x_is_f64 ? 7u : 3u))) return fail(ctx, FLASHE_EINVAL, "layer %d: x_dev is misaligned", l);
        tab.push_back(front ? codec_layer_front(e.start, e.x_dev, e.x_is_f64 != 0, e.alpha, element_bits)
                            : codec_layer_back(e.start, e.alpha, element_bits, num_clients));
    }
    if (tab.empty()) { *tab_dev = nullptr; *n_tab = 0; return FLASHE_OK; }
    int rc = ensure(ctx, ctx->codec_tab, tab.size() * sizeof(CodecLayer));
    if (rc) return rc;
    // (pageable source: the copy has left `tab` when the call returns; stream order keeps an earlier launch's table intact until it ends)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->codec_tab.p, tab.data(), tab.size() * sizeof(CodecLayer), hipMemcpyHostToDevice, ctx->env.stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream));
    *tab_dev = static_cast<const CodecLayer *>(ctx->codec_tab.p);
    *n_tab = static_cast<int>(tab.size());
    return FLASHE_OK;
}

int flashe_quantize_encrypt_model_dev(flashe_ctx *ctx, uint32_t iter, uint32_t idx, int scheme, uint64_t n, uint32_t n_jobs, uint64_t first,
                                      uint64_t count, const flashe_codec_layer *layers, int n_layers, int element_bits, const double *u_dev,
                                      uint64_t *ct_dev)
{
    CHECK_CTX(ctx);
    if (scheme != FLASHE_SCHEME_SINGLE && scheme != FLASHE_SCHEME_DOUBLE) return fail(ctx, FLASHE_EINVAL, "unknown scheme %d", scheme);
    if (count && (!u_dev || !ct_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (element_bits < 1 || element_bits > 62 || element_bits > ctx->int_bits)
        return fail(ctx, FLASHE_EINVAL, "element_bits must be in [1, min(62, int_bits)], got %d", element_bits);
    int rc = check_prf_args(ctx, 1, scheme, n_jobs, ct_dev, nullptr, 0);
    if (rc || (rc = check_range(ctx, n, first, count))) return rc;
    if (reinterpret_cast<uintptr_t>(u_dev) & 7u) return fail(ctx, FLASHE_EINVAL, "u_dev must be 8-byte aligned");
    const CodecLayer *tab = nullptr;
    int n_tab = 0;
    rc = stage_codec_layers(ctx, n, layers, n_layers, true, element_bits, 1, first, count, &tab, &n_tab);
    if (rc || count == 0) return rc;
    Codec cq{};
    cq.x = tab;                     // (non-null = front end on; the values come through the table)
    cq.u = u_dev; cq.layers = tab; cq.n_layers = n_tab; cq.k0 = first;
    LaunchEnv env = ctx->env;
    env.codec = &cq;
    if ((rc = check_double_idx(ctx, scheme, &idx, 1))) return rc;
    const uint32_t add = idx, minus = idx + 1;
    HIP_TRY(ctx, launch_prf(env, iter, &add, 1, &minus, scheme == FLASHE_SCHEME_DOUBLE ? 1 : 0, n, n_jobs, first, count, nullptr, 0, ct_dev));
    return FLASHE_OK;
}

int flashe_decrypt_unquantize_model_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add, const uint32_t *minus_idx, int n_minus,
                                        uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count, const uint64_t *in_dev,
                                        const flashe_codec_layer *layers, int n_layers, int element_bits, int num_clients, double *out_dev)
{
    CHECK_CTX(ctx);
    if (count && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (num_clients < 1) return fail(ctx, FLASHE_EINVAL, "num_clients must be >= 1");
    if (n_add + n_minus == 0) return fail(ctx, FLASHE_EINVAL, "decrypt_unquantize_model: at least one prefix (a round always has one)");
    int rc = check_codec_bits(ctx, element_bits);
    if (rc) return rc;
    rc = check_prf_args(ctx, n_add, n_minus, n_jobs, in_dev, in_dev, ctx->limbs);
    if (rc || (rc = check_range(ctx, n, first, count))) return rc;
    if ((n_add && !add_idx) || (n_minus && !minus_idx)) return fail(ctx, FLASHE_EINVAL, "null prefix list");
    const CodecLayer *tab = nullptr;
    int n_tab = 0;
    rc = stage_codec_layers(ctx, n, layers, n_layers, false, element_bits, num_clients, first, count, &tab, &n_tab);
    if (rc || count == 0) return rc;
    Codec cq{};
    cq.fout = out_dev; cq.layers = tab; cq.n_layers = n_tab; cq.k0 = first;
    // lists longer than one launch holds: all but the last group accumulate into ctx scratch, the last launch writes the floats
    const uint64_t *src = in_dev;
    int a = 0, m = 0;
    if (n_add > kMaxIdx || n_minus > kMaxIdx) {
        rc = ensure(ctx, ctx->stream_tmp, vec_bytes(ctx, count));
        if (rc) return rc;
        uint64_t *tmp = static_cast<uint64_t *>(ctx->stream_tmp.p);
        while (n_add - a > kMaxIdx || n_minus - m > kMaxIdx) {
            const int na = std::min(kMaxIdx, n_add - a), nm = std::min(kMaxIdx, n_minus - m);
            HIP_TRY(ctx, launch_prf(ctx->env, iter, add_idx + a, na, minus_idx + m, nm, n, n_jobs, first, count, src, ctx->limbs, tmp));
            a += na; m += nm; src = tmp;
        }
    }
    LaunchEnv env = ctx->env;
    env.codec = &cq;
    HIP_TRY(ctx, launch_prf(env, iter, add_idx + a, n_add - a, minus_idx + m, n_minus - m, n, n_jobs, first, count, src, ctx->limbs,
                            const_cast<uint64_t *>(src)));
    return FLASHE_OK;
}

// unflatten_weights + QuantizingClient.unquantize (jzf_aggregator.py:652-671, jzf_quantize.py:493-540) of a flattened vector that is
// already decrypted -- the sparse job's way back, whose decrypt is the sparse minus-mask pass, not a prefix list
int flashe_unquantize_model_dev(flashe_ctx *ctx, uint64_t n, uint64_t first, uint64_t count, const uint64_t *in_dev,
                                const flashe_codec_layer *layers, int n_layers, int element_bits, int num_clients, double *out_dev)
{
    CHECK_CTX(ctx);
    if (count && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (num_clients < 1) return fail(ctx, FLASHE_EINVAL, "num_clients must be >= 1");
    int rc = check_codec_bits(ctx, element_bits);
    if (rc || (rc = check_range(ctx, n, first, count))) return rc;
    if ((ctx->limbs == 2 && !aligned16(in_dev)) || (reinterpret_cast<uintptr_t>(in_dev) & 7u) || (reinterpret_cast<uintptr_t>(out_dev) & 7u))
        return fail(ctx, FLASHE_EINVAL, "misaligned vector");
    const CodecLayer *tab = nullptr;
    int n_tab = 0;
    rc = stage_codec_layers(ctx, n, layers, n_layers, false, element_bits, num_clients, first, count, &tab, &n_tab);
    if (rc || count == 0) return rc;
    Codec cq{};
    cq.fout = out_dev; cq.layers = tab; cq.n_layers = n_tab; cq.k0 = first;
    HIP_TRY(ctx, launch_unquantize_model(ctx->env, count, in_dev, cq, out_dev));
    return FLASHE_OK;
}

// ---- the BATCHED codec over a flattened model (the paper's main job configuration, "batch": true) ----
static int stage_batch_layers(flashe_ctx *ctx, const flashe_batch_layer *layers, int n_layers, bool front, int element_bits, int field_bits,
                              int num_clients, uint64_t *n_elems, uint64_t *n_values, const BatchLayer **tab_dev, int *n_tab)
{
    if (n_layers < 1 || !layers) return fail(ctx, FLASHE_EINVAL, "the layer table needs at least one entry");
    if (ctx->capturing) return fail(ctx, FLASHE_EINVAL, "the model-wide codec calls stage their layer table per call and cannot be captured into a graph");
    if (element_bits < 1 || element_bits > 62 || field_bits < element_bits || field_bits > ctx->int_bits)
        return fail(ctx, FLASHE_EINVAL, "need 1 <= element_bits <= field_bits <= int_bits (element_bits <= 62)");
    const uint64_t bs = static_cast<uint64_t>(ctx->int_bits / field_bits);
    std::vector<BatchLayer> tab;
    uint64_t e = 0, v = 0;
    for (int l = 0; l < n_layers; l++) {
        const flashe_batch_layer &y = layers[l];
        if (y.reserved) return fail(ctx, FLASHE_EINVAL, "layer %d: reserved field must be 0", l);
        if (y.size == 0) continue;
        if (!(y.alpha > 0)) return fail(ctx, FLASHE_EINVAL, "layer %d: alpha must be positive", l);
        if (front && (!y.x_dev || (reinterpret_cast<uintptr_t>(y.x_dev) & (y.x_is_f64 ? 7u : 3u)))) return fail(ctx, FLASHE_EINVAL, "layer %d: null or misaligned x_dev", l);
        tab.push_back(front ? batch_layer_front(e, v, y.size, y.x_dev, y.x_is_f64 != 0, y.alpha, element_bits)
                            : batch_layer_back(e, v, y.size, y.alpha, element_bits, num_clients));
        e += (y.size + bs - 1) / bs;                       // every layer is padded to whole elements on its own (jzf_quantize.py:166-171)
        v += y.size;
    }
    *n_elems = e; *n_values = v;
    if (tab.empty()) { *tab_dev = nullptr; *n_tab = 0; return FLASHE_OK; }
    int rc = ensure(ctx, ctx->codec_tab, tab.size() * sizeof(BatchLayer));
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->codec_tab.p, tab.data(), tab.size() * sizeof(BatchLayer), hipMemcpyHostToDevice, ctx->env.stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream));
    *tab_dev = static_cast<const BatchLayer *>(ctx->codec_tab.p);
    *n_tab = static_cast<int>(tab.size());
    return FLASHE_OK;
}

int flashe_quantize_batch_model_dev(flashe_ctx *ctx, const flashe_batch_layer *layers, int n_layers, int element_bits, int field_bits,
                                    const double *u_dev, uint64_t n_elems, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    const BatchLayer *tab = nullptr;
    int n_tab = 0;
    uint64_t e = 0, v = 0;
    int rc = stage_batch_layers(ctx, layers, n_layers, true, element_bits, field_bits, 1, &e, &v, &tab, &n_tab);
    if (rc) return rc;
    if (e != n_elems) return fail(ctx, FLASHE_EINVAL, "the layers batch into %llu elements, n_elems says %llu", static_cast<unsigned long long>(e),
                                  static_cast<unsigned long long>(n_elems));
    if (n_elems && (!u_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if ((ctx->limbs == 2 && !aligned16(out_dev)) || (reinterpret_cast<uintptr_t>(out_dev) & 7u) || (reinterpret_cast<uintptr_t>(u_dev) & 7u))
        return fail(ctx, FLASHE_EINVAL, "misaligned vector");
    if (n_elems) HIP_TRY(ctx, launch_quantize_batch_model(ctx->env, tab, n_tab, field_bits, u_dev, n_elems, out_dev));
    return FLASHE_OK;
}

int flashe_unbatch_unquantize_model_dev(flashe_ctx *ctx, const flashe_batch_layer *layers, int n_layers, int element_bits, int field_bits,
                                        int num_clients, const uint64_t *in_dev, uint64_t n_elems, double *out_dev)
{
    CHECK_CTX(ctx);
    if (num_clients < 1) return fail(ctx, FLASHE_EINVAL, "num_clients must be >= 1");
    const BatchLayer *tab = nullptr;
    int n_tab = 0;
    uint64_t e = 0, v = 0;
    int rc = stage_batch_layers(ctx, layers, n_layers, false, element_bits, field_bits, num_clients, &e, &v, &tab, &n_tab);
    if (rc) return rc;
    if (e != n_elems) return fail(ctx, FLASHE_EINVAL, "the layers batch into %llu elements, n_elems says %llu", static_cast<unsigned long long>(e),
                                  static_cast<unsigned long long>(n_elems));
    if (v && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if ((ctx->limbs == 2 && !aligned16(in_dev)) || (reinterpret_cast<uintptr_t>(in_dev) & 7u) || (reinterpret_cast<uintptr_t>(out_dev) & 7u))
        return fail(ctx, FLASHE_EINVAL, "misaligned vector");
    if (v) HIP_TRY(ctx, launch_unbatch_unquantize_model(ctx->env, tab, n_tab, field_bits, in_dev, v, out_dev));
    return FLASHE_OK;
}

// ---- normalise / unnormalise (QuantizingClient.normalize / unnormalize, jzf_quantize.py:542-564) ----
int flashe_shift_dev(flashe_ctx *ctx, uint64_t n, void *x_dev, int x_is_f64, double shift, int wide)
{
    CHECK_CTX(ctx);
    if (n && !x_dev) return fail(ctx, FLASHE_EINVAL, "null vector");
    HIP_TRY(ctx, launch_shift(ctx->env, n, x_dev, x_is_f64 != 0, shift, wide != 0));
    return FLASHE_OK;
}

int flashe_mean_std_dev(flashe_ctx *ctx, uint64_t n, const void *x_dev, int x_is_f64, double *mean, double *stddev)
{
    CHECK_CTX(ctx);
    if (!mean || !stddev || (n && !x_dev)) return fail(ctx, FLASHE_EINVAL, "null argument");
    if (n == 0) return fail(ctx, FLASHE_EINVAL, "mean of an empty vector");
    const int grid = moments_grid(ctx->env, n);
    int rc = ensure(ctx, ctx->sp_ws, static_cast<size_t>(grid) * sizeof(double));
    if (rc) return rc;
    double *part = static_cast<double *>(ctx->sp_ws.p);
    std::vector<double> host(grid);
    auto total = [&](double center, int pow, double *out) -> int {
        HIP_TRY(ctx, launch_moment(ctx->env, n, x_dev, x_is_f64 != 0, center, pow, part));
        HIP_TRY(ctx, hipMemcpyAsync(host.data(), part, static_cast<size_t>(grid) * sizeof(double), hipMemcpyDeviceToHost, ctx->env.stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream));
        long double t = 0;
        for (double v : host) t += v;
        *out = static_cast<double>(t);
        return FLASHE_OK;
    };
    double s1 = 0, s2 = 0;
    if ((rc = total(0.0, 1, &s1))) return rc;
    const double mu = s1 / static_cast<double>(n);
    if ((rc = total(mu, 2, &s2))) return rc;
    *mean = mu;
    *stddev = sqrt(s2 / static_cast<double>(n));
    return FLASHE_OK;
}

// Range twins: operate on global elements [first, first + count) of an n-element vector.
static int check_range(flashe_ctx *ctx, uint64_t n, uint64_t first, uint64_t count)
{
    if (first > n || count > n - first) return fail(ctx, FLASHE_EINVAL, "range [%llu, +%llu) exceeds n = %llu",
                                                    static_cast<unsigned long long>(first), static_cast<unsigned long long>(count),
                                                    static_cast<unsigned long long>(n));
    return FLASHE_OK;
}

int flashe_mask_range_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *idx, int n_idx, uint64_t n, uint32_t n_jobs,
                          uint64_t first, uint64_t count, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (count && !out_dev) return fail(ctx, FLASHE_EINVAL, "null output");
    int rc = check_prf_args(ctx, n_idx, 0, n_jobs, out_dev, nullptr, 0);
    if (rc || (rc = check_range(ctx, n, first, count))) return rc;
    if (n_idx == 0) { if (count) HIP_TRY(ctx, hipMemsetAsync(out_dev, 0, vec_bytes(ctx, count), ctx->env.stream)); return FLASHE_OK; }
    HIP_TRY(ctx, prf_lists(ctx, iter, idx, n_idx, nullptr, 0, n, n_jobs, first, count, nullptr, 0, out_dev));
    return FLASHE_OK;
}

int flashe_encrypt_range_dev(flashe_ctx *ctx, uint32_t iter, uint32_t idx, int scheme, uint64_t n, uint32_t n_jobs,
                             uint64_t first, uint64_t count, const uint64_t *pt_dev, int pt_limbs, uint64_t *ct_dev)
{
    CHECK_CTX(ctx);
    if (scheme != FLASHE_SCHEME_SINGLE && scheme != FLASHE_SCHEME_DOUBLE) return fail(ctx, FLASHE_EINVAL, "unknown scheme %d", scheme);
    if (count && (!pt_dev || !ct_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    int rc = check_prf_args(ctx, 1, scheme, n_jobs, ct_dev, pt_dev, pt_limbs);
    if (rc || (rc = check_range(ctx, n, first, count))) return rc;
    if ((rc = check_double_idx(ctx, scheme, &idx, 1))) return rc;
    const uint32_t add = idx, minus = idx + 1;
    HIP_TRY(ctx, launch_prf(ctx->env, iter, &add, 1, &minus, scheme == FLASHE_SCHEME_DOUBLE ? 1 : 0, n, n_jobs, first, count,
                            pt_dev, pt_limbs, ct_dev));
    return FLASHE_OK;
}

// Every client's encrypt on ONE element slice of the vectors, optionally with the slice of their sum: the launch of a GPU that owns
// elements [first, first + count) of every client vector (element sharding); pointers address element `first`.
int flashe_encrypt_batch_range_dev(flashe_ctx *ctx, uint32_t iter, int scheme, uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count,
                                   int n_vec, const uint32_t *idx, const uint64_t *const *pt_dev, int pt_limbs, uint64_t *const *ct_dev,
                                   uint64_t *sum_out_dev)
{
    CHECK_CTX(ctx);
    if (scheme != FLASHE_SCHEME_SINGLE && scheme != FLASHE_SCHEME_DOUBLE) return fail(ctx, FLASHE_EINVAL, "unknown scheme %d", scheme);
    if (n_vec < 0 || (n_vec && (!idx || !pt_dev || !ct_dev))) return fail(ctx, FLASHE_EINVAL, "bad batch arguments");
    int rc = check_range(ctx, n, first, count);
    if (rc || (rc = check_double_idx(ctx, scheme, idx, n_vec))) return rc;
    if (n_jobs == 0) return fail(ctx, FLASHE_EINVAL, "n_jobs must be >= 1");
    if (ctx->env.prf_backend != PRF_AUTO && ctx->env.prf_backend != PRF_TABLE) return fail(ctx, FLASHE_EINVAL, "encrypt_batch_range runs on the table PRF only");
    for (int v = 0; v < n_vec; v++) {
        if (count && (!pt_dev[v] || !ct_dev[v])) return fail(ctx, FLASHE_EINVAL, "null vector %d", v);
        if (sum_out_dev && (ct_dev[v] == sum_out_dev || pt_dev[v] == sum_out_dev))
            return fail(ctx, FLASHE_EINVAL, "sum_out_dev must not be one of the plaintext or ciphertext vectors");
        rc = check_prf_args(ctx, 1, scheme, n_jobs, ct_dev[v], pt_dev[v], pt_limbs);
        if (rc) return rc;
    }
    if (sum_out_dev && ((ctx->limbs == 2 && !aligned16(sum_out_dev)) || (reinterpret_cast<uintptr_t>(sum_out_dev) & 7u)))
        return fail(ctx, FLASHE_EINVAL, "sum_out_dev must be aligned like a ciphertext vector");
    if (count == 0) return FLASHE_OK;
    if (n_vec == 0) { if (sum_out_dev) HIP_TRY(ctx, hipMemsetAsync(sum_out_dev, 0, vec_bytes(ctx, count), ctx->env.stream)); return FLASHE_OK; }
    if (sum_out_dev && scheme == FLASHE_SCHEME_DOUBLE) {
        const hipError_t e = launch_prf_batch_sum(ctx->env, iter, n_vec, idx, pt_dev, pt_limbs, ct_dev, sum_out_dev, n, n_jobs, first, count);
        if (e == hipSuccess) return FLASHE_OK;
        if (e != hipErrorNotSupported) HIP_TRY(ctx, e);
    }
    const int cap = (ctx->limbs == 2 || ctx->env.use_chain) ? kMaxUniformBatch : kMaxBatch;
    const int per_launch = (n_vec + (n_vec + cap - 1) / cap - 1) / ((n_vec + cap - 1) / cap);
    for (int v0 = 0; v0 < n_vec; v0 += per_launch) {
        const int nv = std::min(per_launch, n_vec - v0);
        HIP_TRY(ctx, launch_prf_batch_range(ctx->env, iter, scheme == FLASHE_SCHEME_DOUBLE, nv, idx + v0, pt_dev + v0, pt_limbs, ct_dev + v0, n, n_jobs,
                                            first, count));
    }
    if (sum_out_dev) return flashe_aggregate_elem_dev(ctx, n_vec, ct_dev, count, sum_out_dev);
    return FLASHE_OK;
}

int flashe_decrypt_range_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add, const uint32_t *minus_idx,
                             int n_minus, uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count,
                             const uint64_t *in_dev, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (count && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    int rc = check_prf_args(ctx, n_add, n_minus, n_jobs, out_dev, in_dev, ctx->limbs);
    if (rc || (rc = check_range(ctx, n, first, count))) return rc;
    if (n_add == 0 && n_minus == 0) {
        HIP_TRY(ctx, launch_combine(ctx->env, count, in_dev, ctx->limbs, nullptr, nullptr, out_dev));
        return FLASHE_OK;
    }
    HIP_TRY(ctx, prf_lists(ctx, iter, add_idx, n_add, minus_idx, n_minus, n, n_jobs, first, count, in_dev, ctx->limbs, out_dev));
    return FLASHE_OK;
}

int flashe_combine_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *in_dev, int in_limbs, const uint64_t *add_dev,
                       const uint64_t *minus_dev, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (n && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (in_limbs != 1 && in_limbs != ctx->limbs) return fail(ctx, FLASHE_EINVAL, "in_limbs must be 1 or %d", ctx->limbs);
    if (ctx->limbs == 2 && (!aligned16(out_dev) || !aligned16(add_dev) || !aligned16(minus_dev) || (in_limbs == 2 && !aligned16(in_dev))))
        return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    HIP_TRY(ctx, launch_combine(ctx->env, n, in_dev, in_limbs, add_dev, minus_dev, out_dev));
    return FLASHE_OK;
}

// ---- ctx-resident mask precompute (FlasheCipher.prepare_encrypt / prepare_decrypt, jzf_flashe.py:599-666; consumption :457,
// :483-486, :557-580) ----
static int prepare_masks(flashe_ctx *ctx, flashe_ctx::Prepared &pr, uint32_t iter, uint32_t add_idx, bool has_minus, uint32_t minus_idx,
                         uint64_t n, uint32_t n_jobs)
{
    if (n_jobs == 0) return fail(ctx, FLASHE_EINVAL, "n_jobs must be >= 1");
    pr.valid = false;
    int rc = ensure(ctx, pr.add, std::max<size_t>(vec_bytes(ctx, n), 16));
    if (rc == FLASHE_OK && has_minus) rc = ensure(ctx, pr.minus, std::max<size_t>(vec_bytes(ctx, n), 16));
    if (rc) return rc;
    if (n) {
        // both streams in ONE launch: two single-mask jobs with no input (a chain without subtraction)
        PrfJob jobs[2] = {PrfJob{add_idx, 0u, 0, n, nullptr, 0, static_cast<uint64_t *>(pr.add.p)},
                          PrfJob{minus_idx, 0u, 0, n, nullptr, 0, static_cast<uint64_t *>(pr.minus.p)}};
        HIP_TRY(ctx, launch_prf_jobs(ctx->env, iter, false, has_minus ? 2 : 1, jobs, n, n_jobs));
    }
    pr.n = n; pr.has_minus = has_minus; pr.iter = iter; pr.add_idx = add_idx; pr.minus_idx = minus_idx;
    pr.valid = true;
    return FLASHE_OK;
}

int flashe_prepare_encrypt(flashe_ctx *ctx, uint32_t iter_next, uint32_t idx, int scheme, uint64_t num_params, uint32_t n_jobs)
{
    CHECK_CTX(ctx);
    if (scheme != FLASHE_SCHEME_SINGLE && scheme != FLASHE_SCHEME_DOUBLE) return fail(ctx, FLASHE_EINVAL, "unknown scheme %d", scheme);
    // (iter_next is the caller's iter + 1 already: the reference's own range check of that sum, jzf_flashe.py:606, is the caller's)
    int rc = check_double_idx(ctx, scheme, &idx, 1);
    if (rc) return rc;
    return prepare_masks(ctx, ctx->prep_enc, iter_next, idx, scheme == FLASHE_SCHEME_DOUBLE, idx + 1u, num_params, n_jobs);
}

int flashe_prepare_decrypt(flashe_ctx *ctx, uint32_t iter, uint32_t num_clients, uint64_t num_params, uint32_t n_jobs)
{
    CHECK_CTX(ctx);
    return prepare_masks(ctx, ctx->prep_dec, iter, num_clients, true, 0u, num_params, n_jobs);
}

int flashe_prepared_query(flashe_ctx *ctx, int which, uint64_t *n, const uint64_t **add_dev, const uint64_t **minus_dev)
{
    if (!ctx) return FLASHE_EINVAL;
    if (which != FLASHE_PREPARED_ENCRYPT && which != FLASHE_PREPARED_DECRYPT) return fail(ctx, FLASHE_EINVAL, "which must be FLASHE_PREPARED_ENCRYPT or _DECRYPT");
    const flashe_ctx::Prepared &pr = which == FLASHE_PREPARED_ENCRYPT ? ctx->prep_enc : ctx->prep_dec;
    if (n) *n = pr.valid ? pr.n : 0;
    if (add_dev) *add_dev = pr.valid ? static_cast<const uint64_t *>(pr.add.p) : nullptr;
    if (minus_dev) *minus_dev = (pr.valid && pr.has_minus) ? static_cast<const uint64_t *>(pr.minus.p) : nullptr;
    return pr.valid ? 1 : 0;
}

int flashe_prepared_discard(flashe_ctx *ctx, int which)
{
    if (!ctx) return FLASHE_EINVAL;
    if (which & ~(FLASHE_PREPARED_ENCRYPT | FLASHE_PREPARED_DECRYPT)) return fail(ctx, FLASHE_EINVAL, "unknown cache %d", which);
    // A consumed cache keeps its blocks for the next round's masks (a job prepares every round); an explicit discard is the caller
    // saying it is done with precompute: the blocks (up to four model-sized vectors) go back to the device.
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    bool synced = false;
    for (flashe_ctx::Prepared *pr : {(which & FLASHE_PREPARED_ENCRYPT) ? &ctx->prep_enc : nullptr, (which & FLASHE_PREPARED_DECRYPT) ? &ctx->prep_dec : nullptr}) {
        if (!pr) continue;
        pr->valid = false;
        if (ctx->capturing) continue;                                     // (no frees inside a capture; the blocks stay with the ctx)
        for (flashe_ctx::Buf *b : {&pr->add, &pr->minus}) {
            if (!b->p) continue;
            if (!synced) { HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream)); synced = true; }
            HIP_TRY(ctx, hipFree(b->p));
            b->p = nullptr; b->cap = 0;
        }
    }
    return FLASHE_OK;
}

int flashe_encrypt_prepared_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *pt_dev, int pt_limbs, uint64_t *ct_dev)
{
    CHECK_CTX(ctx);
    flashe_ctx::Prepared &pr = ctx->prep_enc;
    if (!pr.valid) return fail(ctx, FLASHE_EINVAL, "no prepared encrypt masks: call flashe_prepare_encrypt first (they are consumed by one encrypt)");
    // (a length mismatch leaves the cache in place, as NumPy's broadcast error does in the reference, jzf_flashe.py:480)
    if (n != pr.n) return fail(ctx, FLASHE_EINVAL, "the prepared masks cover %llu elements, the plaintext has %llu",
                               static_cast<unsigned long long>(pr.n), static_cast<unsigned long long>(n));
    if (n && (!pt_dev || !ct_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    int rc = check_prf_args(ctx, 1, 0, 1, ct_dev, pt_dev, pt_limbs);
    if (rc) return rc;
    if (n) HIP_TRY(ctx, launch_combine(ctx->env, n, pt_dev, pt_limbs, static_cast<const uint64_t *>(pr.add.p),
                                       pr.has_minus ? static_cast<const uint64_t *>(pr.minus.p) : nullptr, ct_dev));
    pr.valid = false;                                                    // consumed (:483-486)
    return FLASHE_OK;
}

int flashe_decrypt_prepared_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add, const uint32_t *minus_idx, int n_minus,
                                uint64_t n, uint32_t n_jobs, const uint64_t *in_dev, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    flashe_ctx::Prepared &pr = ctx->prep_dec;
    if (!pr.valid) return fail(ctx, FLASHE_EINVAL, "no prepared decrypt masks: call flashe_prepare_decrypt first (they are consumed by one decrypt)");
    if (n != pr.n) return fail(ctx, FLASHE_EINVAL, "the prepared masks cover %llu elements, the aggregate has %llu",
                               static_cast<unsigned long long>(pr.n), static_cast<unsigned long long>(n));
    if (n && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    int rc = check_prf_args(ctx, n_add, n_minus, n_jobs, out_dev, in_dev, ctx->limbs);
    if (rc) return rc;
    if ((n_add && !add_idx) || (n_minus && !minus_idx)) return fail(ctx, FLASHE_EINVAL, "null prefix list");
    if (n) {
        // value + prepared add - prepared minus (:570-571), then the prefixes the precompute does not cover, merged in online (:557-564)
        HIP_TRY(ctx, launch_combine(ctx->env, n, in_dev, ctx->limbs, static_cast<const uint64_t *>(pr.add.p),
                                    static_cast<const uint64_t *>(pr.minus.p), out_dev));
        if (n_add || n_minus) HIP_TRY(ctx, prf_lists(ctx, iter, add_idx, n_add, minus_idx, n_minus, n, n_jobs, 0, n, out_dev, ctx->limbs, out_dev));
    }
    pr.valid = false;                                                    // consumed (:573-580)
    return FLASHE_OK;
}

int flashe_combine_batch_dev(flashe_ctx *ctx, uint64_t n, int n_vec, const uint64_t *const *in_dev, int in_limbs,
                             const uint64_t *const *add_dev, const uint64_t *const *minus_dev, uint64_t *const *out_dev)
{
    CHECK_CTX(ctx);
    if (n_vec < 0 || (n_vec && (!in_dev || !out_dev))) return fail(ctx, FLASHE_EINVAL, "bad batch arguments");
    if (in_limbs != 1 && in_limbs != ctx->limbs) return fail(ctx, FLASHE_EINVAL, "in_limbs must be 1 or %d", ctx->limbs);
    for (int v = 0; v < n_vec; v++) {
        if (n && (!in_dev[v] || !out_dev[v])) return fail(ctx, FLASHE_EINVAL, "null vector %d", v);
        const uint64_t *a = add_dev ? add_dev[v] : nullptr, *m = minus_dev ? minus_dev[v] : nullptr;
        if (ctx->limbs == 2 && (!aligned16(out_dev[v]) || !aligned16(a) || !aligned16(m) || (in_limbs == 2 && !aligned16(in_dev[v]))))
            return fail(ctx, FLASHE_EINVAL, "vector %d: device vectors must be 16-byte aligned", v);
    }
    HIP_TRY(ctx, launch_combine_batch(ctx->env, n, n_vec, in_dev, in_limbs, add_dev, minus_dev, out_dev));
    return FLASHE_OK;
}

// out[v] = in[v] + add[v] - minus[v] for every v AND sum_out = sum_v out[v] mod 2^b from the same pass: the online encrypts of the
// clients this process hosts (masks precomputed, jzf_flashe.py:457, :480-481) and the arbiter's element-wise reduce of what they wrote
// (jzf_aggregator.py:424-430) -- the precompute twin of flashe_encrypt_batch_sum_dev.
int flashe_combine_batch_sum_dev(flashe_ctx *ctx, uint64_t n, int n_vec, const uint64_t *const *in_dev, int in_limbs,
                                 const uint64_t *const *add_dev, const uint64_t *const *minus_dev, uint64_t *const *out_dev, uint64_t *sum_out_dev)
{
    CHECK_CTX(ctx);
    if (n_vec < 0 || (n_vec && (!in_dev || !out_dev))) return fail(ctx, FLASHE_EINVAL, "bad batch arguments");
    if (n && !sum_out_dev) return fail(ctx, FLASHE_EINVAL, "flashe_combine_batch_sum_dev: null sum_out_dev");
    if (in_limbs != 1 && in_limbs != ctx->limbs) return fail(ctx, FLASHE_EINVAL, "in_limbs must be 1 or %d", ctx->limbs);
    if ((ctx->limbs == 2 && !aligned16(sum_out_dev)) || (reinterpret_cast<uintptr_t>(sum_out_dev) & 7u))
        return fail(ctx, FLASHE_EINVAL, "sum_out_dev must be aligned like a ciphertext vector");
    for (int v = 0; v < n_vec; v++) {
        if (n && (!in_dev[v] || !out_dev[v])) return fail(ctx, FLASHE_EINVAL, "null vector %d", v);
        if (n && out_dev[v] == sum_out_dev) return fail(ctx, FLASHE_EINVAL, "sum_out_dev must not be one of the output vectors");
        const uint64_t *a = add_dev ? add_dev[v] : nullptr, *m = minus_dev ? minus_dev[v] : nullptr;
        // (more than 64 vectors run as several passes that fold the sum back in: a later pass would read an operand the sum overwrote)
        if (n && (in_dev[v] == sum_out_dev || a == sum_out_dev || m == sum_out_dev))
            return fail(ctx, FLASHE_EINVAL, "sum_out_dev must not be one of the operands (vector %d)", v);
        if (ctx->limbs == 2 && (!aligned16(out_dev[v]) || !aligned16(a) || !aligned16(m) || (in_limbs == 2 && !aligned16(in_dev[v]))))
            return fail(ctx, FLASHE_EINVAL, "vector %d: device vectors must be 16-byte aligned", v);
    }
    HIP_TRY(ctx, launch_combine_batch_sum(ctx->env, n, n_vec, in_dev, in_limbs, add_dev, minus_dev, out_dev, sum_out_dev));
    return FLASHE_OK;
}

// ... and dec_out = (sum_out + dec_add - dec_minus) mod 2^b from the same pass: the decrypt of the arbiter's reduce with the decrypting
// party's precomputed masks (jzf_flashe.py:557-571 with next_iter_decrypt_prepared populated, :633-666) -- the workgroup that completes
// an element's sum holds it in registers; one launch instead of the reduce's and the decrypt's (round 6, config 3).
int flashe_combine_batch_sum_decrypt_dev(flashe_ctx *ctx, uint64_t n, int n_vec, const uint64_t *const *in_dev, int in_limbs,
                                         const uint64_t *const *add_dev, const uint64_t *const *minus_dev, uint64_t *const *out_dev, uint64_t *sum_out_dev,
                                         const uint64_t *dec_add_dev, const uint64_t *dec_minus_dev, uint64_t *dec_out_dev)
{
    CHECK_CTX(ctx);
    if (n && !dec_out_dev) return fail(ctx, FLASHE_EINVAL, "flashe_combine_batch_sum_decrypt_dev: null dec_out_dev");
    if (n_vec < 0 || (n_vec && (!in_dev || !out_dev))) return fail(ctx, FLASHE_EINVAL, "bad batch arguments");
    if (n && !sum_out_dev) return fail(ctx, FLASHE_EINVAL, "flashe_combine_batch_sum_decrypt_dev: null sum_out_dev");
    if (in_limbs != 1 && in_limbs != ctx->limbs) return fail(ctx, FLASHE_EINVAL, "in_limbs must be 1 or %d", ctx->limbs);
    const uintptr_t need = ctx->limbs == 2 ? 15u : 7u;
    for (const void *q : {static_cast<const void *>(sum_out_dev), static_cast<const void *>(dec_add_dev), static_cast<const void *>(dec_minus_dev), static_cast<const void *>(dec_out_dev)})
        if (reinterpret_cast<uintptr_t>(q) & need) return fail(ctx, FLASHE_EINVAL, "sum / decrypt vectors must be aligned like a ciphertext vector");
    if (n && (dec_out_dev == sum_out_dev || dec_add_dev == sum_out_dev || dec_minus_dev == sum_out_dev))
        return fail(ctx, FLASHE_EINVAL, "dec_out_dev and the decrypt masks must not be sum_out_dev");
    for (int v = 0; v < n_vec; v++) {
        if (n && (!in_dev[v] || !out_dev[v])) return fail(ctx, FLASHE_EINVAL, "null vector %d", v);
        const uint64_t *a = add_dev ? add_dev[v] : nullptr, *m = minus_dev ? minus_dev[v] : nullptr;
        if (n && (out_dev[v] == sum_out_dev || out_dev[v] == dec_out_dev || out_dev[v] == dec_add_dev || out_dev[v] == dec_minus_dev))
            return fail(ctx, FLASHE_EINVAL, "output vector %d is also the sum, the decrypt result or a decrypt mask", v);
        if (n && (in_dev[v] == sum_out_dev || a == sum_out_dev || m == sum_out_dev || in_dev[v] == dec_out_dev || a == dec_out_dev || m == dec_out_dev))
            return fail(ctx, FLASHE_EINVAL, "sum_out_dev / dec_out_dev must not be one of the operands (vector %d)", v);
        if (ctx->limbs == 2 && (!aligned16(out_dev[v]) || !aligned16(a) || !aligned16(m) || (in_limbs == 2 && !aligned16(in_dev[v]))))
            return fail(ctx, FLASHE_EINVAL, "vector %d: device vectors must be 16-byte aligned", v);
    }
    HIP_TRY(ctx, launch_combine_batch_sum(ctx->env, n, n_vec, in_dev, in_limbs, add_dev, minus_dev, out_dev, sum_out_dev, dec_add_dev, dec_minus_dev, dec_out_dev));
    return FLASHE_OK;
}

// ---- arbiter reduce ----
int flashe_aggregate_elem_dev(flashe_ctx *ctx, int C, const uint64_t *const *cts_dev, uint64_t n, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (C < 1 || !cts_dev || (n && !out_dev)) return fail(ctx, FLASHE_EINVAL, "aggregate_elem: bad arguments (C = %d)", C);
    // two-limb vectors: 16-byte aligned; one-limb vectors: 8 bytes suffice (a sub-range may start at an odd element: slower 8-byte form)
    const uintptr_t need = ctx->limbs == 2 ? 15u : 7u;
    if (reinterpret_cast<uintptr_t>(out_dev) & need) return fail(ctx, FLASHE_EINVAL, "device vectors must be %d-byte aligned", static_cast<int>(need + 1));
    for (int c = 0; c < C; c++)
        if (!cts_dev[c] || (reinterpret_cast<uintptr_t>(cts_dev[c]) & need))
            return fail(ctx, FLASHE_EINVAL, "operand %d is null or not %d-byte aligned", c, static_cast<int>(need + 1));
    if (n == 0) return FLASHE_OK;
    // at most kMaxOps operands per pass; later passes fold the running sum (out) back in
    std::vector<const uint64_t *> ops;
    int done = 0;
    while (done < C) {
        ops.clear();
        if (done) ops.push_back(out_dev);
        while (done < C && static_cast<int>(ops.size()) < kMaxOps) ops.push_back(cts_dev[done++]);
        HIP_TRY(ctx, launch_aggregate_elem(ctx->env, static_cast<int>(ops.size()), ops.data(), n, out_dev));
    }
    return FLASHE_OK;
}

int flashe_aggregate_decrypt_range_dev(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add, const uint32_t *minus_idx,
                                       int n_minus, uint64_t n, uint32_t n_jobs, uint64_t first, uint64_t count, int C,
                                       const uint64_t *const *cts_dev, uint64_t *agg_out_dev, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (C < 1 || !cts_dev || (count && !out_dev)) return fail(ctx, FLASHE_EINVAL, "aggregate_decrypt: bad arguments (C = %d)", C);
    int rc = check_range(ctx, n, first, count);
    if (rc) return rc;
    if (count == 0) return FLASHE_OK;
    for (int c = 0; c < C; c++)
        if (!cts_dev[c]) return fail(ctx, FLASHE_EINVAL, "operand %d is null", c);
    if (ctx->limbs == 2 && n_add == 1 && n_minus <= 1 && C <= kMaxOps && aligned16(out_dev) && aligned16(agg_out_dev)) {
        // b > 64, one add and at most one minus prefix, up to 64 operands ANYWHERE in HBM: one launch (a grid-stride loop over the
        // elements; it measured 2-4 % faster than the tiled job-table form below even on equally spaced operands, tests/perf/ab_reduce_2wg.py)
        bool ok = true;
        for (int c = 0; c < C && ok; c++) ok = aligned16(cts_dev[c]);
        if (ok) {
            const hipError_t e = launch_reduce_decrypt_ptrs(ctx->env, iter, add_idx[0], n_minus == 1, n_minus ? minus_idx[0] : 0u, first, count, C, cts_dev,
                                                            agg_out_dev, out_dev);
            if (e == hipSuccess) return FLASHE_OK;
            if (e != hipErrorNotSupported) HIP_TRY(ctx, e);
        }
    }
    // more than 64 operands: one pass when the ciphertexts are equally spaced (ascending), one add and at most one minus prefix, b > 64
    // (a short vector does not fill the chip with 1024-element tiles: the streaming reduce kernel + a decrypt win there)
    bool strided = ctx->limbs == 2 && n_add == 1 && n_minus <= 1 && C <= 255 && count >= static_cast<uint64_t>(ctx->env.num_cus) * 1024 &&
                   (ctx->env.prf_backend == PRF_AUTO || ctx->env.prf_backend == PRF_TABLE) && aligned16(cts_dev[0]) && aligned16(out_dev) &&
                   aligned16(agg_out_dev);
    uint64_t stride = 0;
    if (strided && C > 1) {
        if (cts_dev[1] <= cts_dev[0]) strided = false;
        else {
            stride = static_cast<uint64_t>(cts_dev[1] - cts_dev[0]);
            for (int c = 2; c < C && strided; c++) strided = cts_dev[c] == cts_dev[0] + stride * c;
            strided = strided && stride % 2 == 0;
        }
    }
    if (strided) {
        const PrfJob job{add_idx[0], n_minus ? minus_idx[0] : 0u, first, count, cts_dev[0], 2, out_dev,
                         static_cast<uint32_t>(C), stride, agg_out_dev};
        HIP_TRY(ctx, launch_prf_jobs(ctx->env, iter, n_minus == 1, 1, &job, n, n_jobs));
        return FLASHE_OK;
    }
    if (ctx->limbs == 1 && n_add == 1 && n_minus <= 1) {
        // b <= 64: the same fusion in the small-modulus form (FLASHE_SMALL_FUSED_REDUCE=0: two launches, the A/B switch)
#ifdef FLASHE_TUNING
        static const bool off = getenv("FLASHE_SMALL_FUSED_REDUCE") && atoi(getenv("FLASHE_SMALL_FUSED_REDUCE")) == 0;
#else
        constexpr bool off = false;
#endif
        if (!off) {
            const hipError_t e = launch_small_reduce_decrypt(ctx->env, iter, add_idx[0], n_minus == 1, n_minus ? minus_idx[0] : 0u, n, n_jobs, first, count,
                                                             C, cts_dev, agg_out_dev, out_dev);
            if (e == hipSuccess) return FLASHE_OK;
            if (e != hipErrorNotSupported) HIP_TRY(ctx, e);
        }
    }
    uint64_t *agg = agg_out_dev;
    if (!agg) {
        rc = ensure(ctx, ctx->stream_tmp, vec_bytes(ctx, count));
        if (rc) return rc;
        agg = static_cast<uint64_t *>(ctx->stream_tmp.p);
    }
    rc = flashe_aggregate_elem_dev(ctx, C, cts_dev, count, agg);
    if (rc) return rc;
    return flashe_decrypt_range_dev(ctx, iter, add_idx, n_add, minus_idx, n_minus, n, n_jobs, first, count, agg, out_dev);
}

int flashe_aggregate_packed_dev(flashe_ctx *ctx, int C, const uint64_t *const *packed_dev, uint64_t n_limbs, uint64_t total_bits,
                                uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (C < 1 || !packed_dev || (n_limbs && !out_dev)) return fail(ctx, FLASHE_EINVAL, "aggregate_packed: bad arguments (C = %d)", C);
    if (n_limbs != (total_bits + 63) / 64) return fail(ctx, FLASHE_EINVAL, "n_limbs must equal ceil(total_bits / 64)");
    if (!aligned16(out_dev)) return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    for (int c = 0; c < C; c++)
        if (!packed_dev[c] || !aligned16(packed_dev[c])) return fail(ctx, FLASHE_EINVAL, "operand %d is null or not 16-byte aligned", c);
    for (int c = 0; c < C; c++)
        if (packed_dev[c] == out_dev) return fail(ctx, FLASHE_EINVAL, "aggregate_packed: out must not alias operand %d", c);
    if (n_limbs == 0) return FLASHE_OK;
    int rc = ensure(ctx, ctx->summaries, packed_num_blocks(n_limbs) * sizeof(uint32_t));
    if (rc) return rc;
    uint32_t *summ = static_cast<uint32_t *>(ctx->summaries.p);
    // at most kMaxOps operands per pass; partial sums ping-pong through ctx scratch so that a
    // pass never writes a buffer it reads
    std::vector<const uint64_t *> ops;
    const uint64_t *partial = nullptr;
    int done = 0, flip = 0;
    while (done < C) {
        ops.clear();
        if (partial) ops.push_back(partial);
        while (done < C && static_cast<int>(ops.size()) < kMaxOps) ops.push_back(packed_dev[done++]);
        uint64_t *dst = out_dev;
        if (done < C) {
            rc = ensure(ctx, ctx->acc_tmp[flip], static_cast<size_t>(n_limbs) * 8);
            if (rc) return rc;
            dst = static_cast<uint64_t *>(ctx->acc_tmp[flip].p);
            flip ^= 1;
        }
        HIP_TRY(ctx, launch_aggregate_packed(ctx->env, static_cast<int>(ops.size()), ops.data(), n_limbs, total_bits, dst, summ));
        partial = dst;
    }
    return FLASHE_OK;
}

int flashe_packed_probe_dev(flashe_ctx *ctx, uint64_t n_limbs, const uint64_t *x_dev, uint64_t *info_dev)
{
    CHECK_CTX(ctx);
    if (!info_dev || n_limbs < 2 || !x_dev) return fail(ctx, FLASHE_EINVAL, "packed_probe: needs a body limb and a carry limb (n_limbs = %llu)", (unsigned long long)n_limbs);
    HIP_TRY(ctx, launch_packed_probe(ctx->env, n_limbs, x_dev, info_dev));
    return FLASHE_OK;
}
int flashe_packed_add_carry_dev(flashe_ctx *ctx, uint64_t n_limbs, uint64_t total_bits, uint64_t carry_in, uint64_t *x_dev)
{
    CHECK_CTX(ctx);
    if (n_limbs != (total_bits + 63) / 64) return fail(ctx, FLASHE_EINVAL, "n_limbs must equal ceil(total_bits / 64)");
    if (n_limbs && !x_dev) return fail(ctx, FLASHE_EINVAL, "null vector");
    HIP_TRY(ctx, launch_packed_add_carry(ctx->env, n_limbs, total_bits, carry_in, x_dev));
    return FLASHE_OK;
}

int flashe_packed_resolve_carry_dev(flashe_ctx *ctx, uint64_t n_limbs, uint64_t total_bits, const uint64_t *infos_dev, int n_below,
                                    uint64_t *x_dev)
{
    CHECK_CTX(ctx);
    if (n_limbs != (total_bits + 63) / 64) return fail(ctx, FLASHE_EINVAL, "n_limbs must equal ceil(total_bits / 64)");
    if ((n_limbs && !x_dev) || n_below < 0 || (n_below && !infos_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    HIP_TRY(ctx, launch_packed_resolve_carry(ctx->env, n_limbs, total_bits, infos_dev, n_below, x_dev));
    return FLASHE_OK;
}

int flashe_packed_resolve_carry_strided_dev(flashe_ctx *ctx, uint64_t n_limbs, uint64_t total_bits, const uint64_t *infos_dev, int n_below,
                                            int stride_words, uint64_t *x_dev)
{
    CHECK_CTX(ctx);
    if (n_limbs != (total_bits + 63) / 64) return fail(ctx, FLASHE_EINVAL, "n_limbs must equal ceil(total_bits / 64)");
    if ((n_limbs && !x_dev) || n_below < 0 || (n_below && !infos_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (stride_words == 0 || stride_words % 3) return fail(ctx, FLASHE_EINVAL, "stride_words must be a non-zero multiple of 3");
    HIP_TRY(ctx, launch_packed_resolve_carry(ctx->env, n_limbs, total_bits, infos_dev, n_below, x_dev, stride_words));
    return FLASHE_OK;
}

// ---- codec ----
int flashe_pack_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *in_dev, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (n && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (ctx->limbs == 2 && !aligned16(in_dev)) return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    if (ctx->int_bits == 128 && !aligned16(out_dev)) return fail(ctx, FLASHE_EINVAL, "int_bits = 128: the packed buffer must be 16-byte aligned");
    HIP_TRY(ctx, launch_pack(ctx->env, n, in_dev, out_dev));
    return FLASHE_OK;
}
int flashe_unpack_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *in_dev, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (n && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (ctx->limbs == 2 && !aligned16(out_dev)) return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    if (ctx->int_bits == 128 && !aligned16(in_dev)) return fail(ctx, FLASHE_EINVAL, "int_bits = 128: the packed buffer must be 16-byte aligned");
    HIP_TRY(ctx, launch_unpack(ctx->env, n, in_dev, out_dev));
    return FLASHE_OK;
}

// ---- sparse ----
int flashe_expand_to_dense_dev(flashe_ctx *ctx, uint64_t total, uint64_t k, const uint32_t *loc_dev, const uint64_t *vals_dev,
                               const uint64_t *zero, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (!zero || (total && !out_dev) || (k && (!loc_dev || !vals_dev))) return fail(ctx, FLASHE_EINVAL, "null argument");
    if (ctx->limbs == 2 && (!aligned16(out_dev) || !aligned16(vals_dev))) return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    HIP_TRY(ctx, launch_fill(ctx->env, total, zero[0], ctx->limbs == 2 ? zero[1] : 0, out_dev));
    HIP_TRY(ctx, launch_scatter(ctx->env, total, k, loc_dev, vals_dev, out_dev, false));
    return FLASHE_OK;
}

// ---- span bounds of a round's location lists, computed once (new) ----
struct flashe_span_bounds {
    int device = 0, C = 0;
    uint64_t total = 0;
    std::vector<const uint32_t *> loc;
    std::vector<uint64_t> k;
    uint32_t *start = nullptr;           // per group of kMaxScatter clients: (span_count(total, kSpanReduce) + 1) * group words (the plain span reduce)
    size_t group_stride = 0;             // words between two groups' tables
    uint32_t *start_fused = nullptr;     // the same at kSpanFused positions per span (the passes with the PRF inside; null on a ctx that has none)
    size_t group_stride_fused = 0;
    // Round 5: create / recompute fill the table the ctx's hot sparse passes read (the fused one where there is one) and leave the other to
    // the first call that needs it -- a round that runs the fused passes never reads the plain reduce's table, and filling both cost the
    // bounds pass a third of its time (config 5: 0.032 -> 0.023 ms).
    mutable bool have_reduce = false;
    // the ctx whose stream the lazy fill ran on: a handle used from another ctx / stream has no ordering against that fill
    mutable const flashe_ctx *reduce_filled_by = nullptr;
};

// The plain span reduce's table of a handle, filled on first use ON THE CALLING CTX'S STREAM (same lists, same stream order).  Inside a
// graph capture the fill is only RECORDED -- it runs when (and each time) the graph is replayed -- so the flag is left alone there: a
// later eager call fills the table itself instead of reading what a never-replayed graph never wrote.  A second ctx (another stream)
// that shares the handle fills the table again on its own stream rather than racing the first one's fill.
static int ensure_reduce_table(flashe_ctx *ctx, const flashe_span_bounds *b)
{
    if (b->have_reduce && b->reduce_filled_by == ctx) return FLASHE_OK;
    for (int c0 = 0, g = 0; c0 < b->C; c0 += kMaxScatter, g++)
        HIP_TRY(ctx, launch_span_bounds(ctx->env, std::min(kMaxScatter, b->C - c0), b->loc.data() + c0, b->k.data() + c0, b->total,
                                        b->start + g * b->group_stride, nullptr));
    if (!ctx->capturing) { b->have_reduce = true; b->reduce_filled_by = ctx; }
    return FLASHE_OK;
}

// int_bits > 64 on the table PRF: the sparse single-mask passes run as launch_span_prf
static bool span_prf_ok(const flashe_ctx *ctx)
{
    return ctx->limbs == 2 && (ctx->env.prf_backend == PRF_AUTO || ctx->env.prf_backend == PRF_TABLE);
}

int flashe_span_bounds_create(flashe_ctx *ctx, uint64_t total, int C, const uint32_t *const *loc_dev, const uint64_t *k, flashe_span_bounds **out)
{
    CHECK_CTX(ctx);
    if (!out) return fail(ctx, FLASHE_EINVAL, "null output");
    *out = nullptr;
    if (C < 1 || !loc_dev || !k) return fail(ctx, FLASHE_EINVAL, "span_bounds_create: bad arguments");
    if (ctx->capturing) return fail(ctx, FLASHE_EINVAL, "span_bounds_create allocates: not capturable");
    for (int c = 0; c < C; c++) {
        if (k[c] && !loc_dev[c]) return fail(ctx, FLASHE_EINVAL, "client %d: null list", c);
        if (k[c] > total || k[c] >= (1ull << 32)) return fail(ctx, FLASHE_EINVAL, "client %d: more locations than positions", c);
    }
    auto *b = new flashe_span_bounds;
    b->device = ctx->device; b->C = C; b->total = total;
    b->loc.assign(loc_dev, loc_dev + C); b->k.assign(k, k + C);
    const int group = std::min(C, kMaxScatter), groups = (C + kMaxScatter - 1) / kMaxScatter;
    b->group_stride = (span_count(total, kSpanReduce) + 1) * static_cast<size_t>(group);
    b->group_stride_fused = span_prf_ok(ctx) ? (span_count(total, kSpanFused) + 1) * static_cast<size_t>(group) : 0;
    const hipError_t e = hipMalloc(&b->start, std::max<size_t>((b->group_stride + b->group_stride_fused) * groups * sizeof(uint32_t), 16));
    if (e != hipSuccess) { delete b; return fail(ctx, e == hipErrorOutOfMemory ? FLASHE_ENOMEM : FLASHE_EIO, "hipMalloc: %s", hipGetErrorString(e)); }
    if (b->group_stride_fused) b->start_fused = b->start + b->group_stride * groups;
    for (int g = 0; g < groups; g++) {
        const int c0 = g * kMaxScatter, nc = std::min(kMaxScatter, C - c0);
        const hipError_t le = launch_span_bounds(ctx->env, nc, loc_dev + c0, k + c0, total, b->start_fused ? nullptr : b->start + g * b->group_stride,
                                                 b->start_fused ? b->start_fused + g * b->group_stride_fused : nullptr);
        if (le != hipSuccess) { (void)hipFree(b->start); delete b; HIP_TRY(ctx, le); }
    }
    b->have_reduce = b->start_fused == nullptr;
    *out = b;
    return FLASHE_OK;
}

// the same handle for the NEXT round's lists (same total and C: a job's shape): recomputes the table in place, no allocation
int flashe_span_bounds_recompute(flashe_ctx *ctx, flashe_span_bounds *b, const uint32_t *const *loc_dev, const uint64_t *k)
{
    CHECK_CTX(ctx);
    if (!b || !loc_dev || !k) return fail(ctx, FLASHE_EINVAL, "span_bounds_recompute: bad arguments");
    if (b->device != ctx->device) return fail(ctx, FLASHE_EINVAL, "the handle belongs to another device");
    for (int c = 0; c < b->C; c++) {
        if (k[c] && !loc_dev[c]) return fail(ctx, FLASHE_EINVAL, "client %d: null list", c);
        if (k[c] > b->total || k[c] >= (1ull << 32)) return fail(ctx, FLASHE_EINVAL, "client %d: more locations than positions", c);
    }
    b->loc.assign(loc_dev, loc_dev + b->C); b->k.assign(k, k + b->C);
    for (int c0 = 0, g = 0; c0 < b->C; c0 += kMaxScatter, g++)
        HIP_TRY(ctx, launch_span_bounds(ctx->env, std::min(kMaxScatter, b->C - c0), loc_dev + c0, k + c0, b->total,
                                        b->start_fused ? nullptr : b->start + g * b->group_stride,
                                        b->start_fused ? b->start_fused + g * b->group_stride_fused : nullptr));
    b->have_reduce = b->start_fused == nullptr;
    return FLASHE_OK;
}

void flashe_span_bounds_destroy(flashe_span_bounds *b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->start) (void)hipFree(b->start);
    delete b;
}

// the handle must describe exactly the lists of this call
static int check_bounds(flashe_ctx *ctx, const flashe_span_bounds *b, uint64_t total, int C, const uint32_t *const *loc_dev, const uint64_t *k)
{
    if (!b) return FLASHE_OK;
    if (b->device != ctx->device || b->total != total || b->C != C) return fail(ctx, FLASHE_EINVAL, "the span bounds were computed for another set of lists (total / C differ)");
    for (int c = 0; c < C; c++)
        if (b->loc[c] != loc_dev[c] || b->k[c] != k[c]) return fail(ctx, FLASHE_EINVAL, "the span bounds were computed for another list of client %d", c);
    return FLASHE_OK;
}

static int sparse_aggregate_impl(flashe_ctx *ctx, uint64_t total, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                 const uint64_t *const *vals_dev, const uint64_t *zeros, int sorted, const flashe_span_bounds *bounds, uint64_t *out_dev);

int flashe_sparse_aggregate_dev(flashe_ctx *ctx, uint64_t total, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                const uint64_t *const *vals_dev, const uint64_t *zeros, int sorted, uint64_t *out_dev)
{
    return sparse_aggregate_impl(ctx, total, C, loc_dev, k, vals_dev, zeros, sorted, nullptr, out_dev);
}

int flashe_sparse_aggregate_bounds_dev(flashe_ctx *ctx, uint64_t total, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                       const uint64_t *const *vals_dev, const uint64_t *zeros, const flashe_span_bounds *bounds, uint64_t *out_dev)
{
    if (!bounds) return ctx ? fail(ctx, FLASHE_EINVAL, "null bounds handle") : FLASHE_EINVAL;
    return sparse_aggregate_impl(ctx, total, C, loc_dev, k, vals_dev, zeros, 1, bounds, out_dev);
}

static int sparse_aggregate_impl(flashe_ctx *ctx, uint64_t total, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                 const uint64_t *const *vals_dev, const uint64_t *zeros, int sorted, const flashe_span_bounds *bounds, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    { const int brc = C > 0 && loc_dev && k ? check_bounds(ctx, bounds, total, C, loc_dev, k) : FLASHE_OK; if (brc) return brc; }
    if (C < 0 || (C && (!loc_dev || !k || !vals_dev || !zeros)) || (total && !out_dev)) return fail(ctx, FLASHE_EINVAL, "bad arguments");
    if (ctx->limbs == 2 && !aligned16(out_dev)) return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    const int L = ctx->limbs;
    using u128 = unsigned __int128;
    const u128 mask = ctx->int_bits == 128 ? ~static_cast<u128>(0) : ((static_cast<u128>(1) << ctx->int_bits) - 1);
    u128 zsum = 0;
    for (int c = 0; c < C; c++) {
        const u128 z = (L == 2 ? static_cast<u128>(zeros[2 * c + 1]) << 64 : 0) | zeros[static_cast<size_t>(L) * c];
        if (z & ~mask) return fail(ctx, FLASHE_EINVAL, "zero value of client %d exceeds int_bits", c);
        if (k[c] && (!loc_dev[c] || !vals_dev[c] || (L == 2 && !aligned16(vals_dev[c])))) return fail(ctx, FLASHE_EINVAL, "client %d: null or misaligned vector", c);
        zsum = (zsum + z) & mask;
    }
    // sum_c expand_to_dense(c) = (sum_c zero_c everywhere) + per client (vals_c[q] - zero_c) at loc_c[q]; the clients go one
    // after the other because their location sets overlap
    if (sorted && C > 0) {
        // strictly increasing location lists: LDS-staged span reduce, the dense output is written exactly once
        int rc = bounds ? FLASHE_OK : ensure(ctx, ctx->bounds, span_table_words(total, std::min(C, kMaxScatter)) * sizeof(uint32_t));
        if (rc) return rc;
        for (int c0 = 0; c0 < C; c0 += kMaxScatter) {
            const int nc = std::min(kMaxScatter, C - c0);
            if (bounds) { const int rc2 = ensure_reduce_table(ctx, bounds); if (rc2) return rc2; }
            uint32_t *start = bounds ? bounds->start + (c0 / kMaxScatter) * bounds->group_stride : static_cast<uint32_t *>(ctx->bounds.p);
            HIP_TRY(ctx, launch_span_reduce(ctx->env, nc, loc_dev + c0, vals_dev + c0, k + c0, zeros + static_cast<size_t>(L) * c0,
                                            c0 ? 0 : static_cast<uint64_t>(zsum), c0 ? 0 : static_cast<uint64_t>(zsum >> 64), total,
                                            start, c0 != 0 ? out_dev : nullptr, false, out_dev, bounds != nullptr));
        }
        return FLASHE_OK;
    }
    HIP_TRY(ctx, launch_fill(ctx->env, total, static_cast<uint64_t>(zsum), static_cast<uint64_t>(zsum >> 64), out_dev));
    for (int c = 0; c < C; c++)
        HIP_TRY(ctx, launch_scatter(ctx->env, total, k[c], loc_dev[c], vals_dev[c], out_dev, true, zeros[static_cast<size_t>(L) * c],
                                    L == 2 ? zeros[2 * c + 1] : 0));
    return FLASHE_OK;
}

// The sparse twin of flashe_encrypt_batch_sum_dev: the C clients this device plays encrypt their compact uploads (single mask over the
// compact positions, jzf_flashe.py:471-478 on the values Client.sparsify kept) and the sum of the C expanded uploads -- what
// Arbiter.expand_to_dense + the reduce make of them (jzf_aggregator.py:150-165, :419-430) -- is written in the same pass.
static int sparse_encrypt_aggregate_impl(flashe_ctx *ctx, uint32_t iter, uint32_t n_jobs, uint64_t total, int C, const uint32_t *idx,
                                         const uint32_t *const *loc_dev, const uint64_t *k, const uint64_t *const *pt_dev, int pt_limbs,
                                         const uint64_t *zeros, const flashe_span_bounds *bounds, uint64_t *const *ct_dev, uint64_t *agg_out_dev,
                                         uint64_t first, uint64_t count, bool whole)
{
    CHECK_CTX(ctx);
    if (C < 1 || !idx || !loc_dev || !k || !pt_dev || !zeros || !ct_dev || (total && !agg_out_dev) || n_jobs == 0)
        return fail(ctx, FLASHE_EINVAL, "flashe_sparse_encrypt_aggregate_dev: bad arguments");
    { const int brc = check_bounds(ctx, bounds, total, C, loc_dev, k); if (brc) return brc; }
    if (ctx->limbs == 2 && !aligned16(agg_out_dev)) return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    const int L = ctx->limbs;
    using u128 = unsigned __int128;
    const u128 mask = ctx->int_bits == 128 ? ~static_cast<u128>(0) : ((static_cast<u128>(1) << ctx->int_bits) - 1);
    u128 zsum = 0;
    for (int c = 0; c < C; c++) {
        if (k[c] > total || k[c] >= (1ull << 32)) return fail(ctx, FLASHE_EINVAL, "client %d: more locations than positions", c);
        if (k[c] && (!loc_dev[c] || !pt_dev[c] || !ct_dev[c])) return fail(ctx, FLASHE_EINVAL, "client %d: null vector", c);
        if (ct_dev[c] == agg_out_dev && k[c]) return fail(ctx, FLASHE_EINVAL, "the aggregate must not be one of the ciphertext vectors");
        if (k[c]) { const int rc = check_prf_args(ctx, 1, FLASHE_SCHEME_SINGLE, n_jobs, ct_dev[c], pt_dev[c], pt_limbs); if (rc) return rc; }
        const u128 z = (L == 2 ? static_cast<u128>(zeros[2 * c + 1]) << 64 : 0) | zeros[static_cast<size_t>(L) * c];
        if (z & ~mask) return fail(ctx, FLASHE_EINVAL, "zero value of client %d exceeds int_bits", c);
        zsum = (zsum + z) & mask;
    }
    if (!whole && (!span_prf_ok(ctx) || !bounds || !bounds->start_fused))
        return fail(ctx, FLASHE_EINVAL, "the position-range form needs int_bits > 64 on the table PRF and a bounds handle");
    if (!span_prf_ok(ctx) || (bounds && !bounds->start_fused)) {
        // int_bits <= 64 / another PRF backend: the encrypts, then the sparse reduce of what they wrote
        for (int c = 0; c < C; c++) {
            if (!k[c]) continue;
            const int rc = flashe_encrypt_dev(ctx, iter, idx[c], FLASHE_SCHEME_SINGLE, k[c], n_jobs, pt_dev[c], pt_limbs, ct_dev[c]);
            if (rc) return rc;
        }
        return sparse_aggregate_impl(ctx, total, C, loc_dev, k, ct_dev, zeros, 1, bounds, agg_out_dev);
    }
    int rc = bounds ? FLASHE_OK : ensure(ctx, ctx->bounds, span_table_words(total, std::min(C, kMaxScatter)) * sizeof(uint32_t));
    if (rc) return rc;
    for (int c0 = 0; c0 < C; c0 += kMaxScatter) {
        const int nc = std::min(kMaxScatter, C - c0);
        uint32_t *start = bounds ? bounds->start_fused + (c0 / kMaxScatter) * bounds->group_stride_fused : static_cast<uint32_t *>(ctx->bounds.p);
        if (!bounds) HIP_TRY(ctx, launch_span_bounds(ctx->env, nc, loc_dev + c0, k + c0, total, nullptr, start));
        HIP_TRY(ctx, launch_span_prf(ctx->env, iter, nc, idx + c0, loc_dev + c0, k + c0, pt_dev + c0, pt_limbs, ct_dev + c0, zeros + 2 * static_cast<size_t>(c0),
                                     c0 ? 0 : static_cast<uint64_t>(zsum), c0 ? 0 : static_cast<uint64_t>(zsum >> 64), total, start,
                                     c0 != 0 ? agg_out_dev : nullptr, false, agg_out_dev, first, count));
    }
    return FLASHE_OK;
}

int flashe_sparse_encrypt_aggregate_dev(flashe_ctx *ctx, uint32_t iter, uint32_t n_jobs, uint64_t total, int C, const uint32_t *idx,
                                        const uint32_t *const *loc_dev, const uint64_t *k, const uint64_t *const *pt_dev, int pt_limbs,
                                        const uint64_t *zeros, const flashe_span_bounds *bounds, uint64_t *const *ct_dev, uint64_t *agg_out_dev)
{
    return sparse_encrypt_aggregate_impl(ctx, iter, n_jobs, total, C, idx, loc_dev, k, pt_dev, pt_limbs, zeros, bounds, ct_dev, agg_out_dev, 0, total, true);
}

// ---- the sparse round sharded by POSITION ranges (SURVEY 8e (i) for the sparse path): GPU g owns the positions [first, first + count) of
// the dense vector and runs every client's entries that fall into them -- counters and list indices stay global, nothing is exchanged
// for the aggregate.  Ranges start at multiples of flashe_sparse_span() and end at one or at the end of the vector. ----
int flashe_sparse_span(void) { return kSpanFused; }

static int check_span_range(flashe_ctx *ctx, uint64_t total, uint64_t first, uint64_t count)
{
    if (first > total || count > total - first) return fail(ctx, FLASHE_EINVAL, "position range outside the vector");
    if (first % kSpanFused || (first + count != total && (first + count) % kSpanFused))
        return fail(ctx, FLASHE_EINVAL, "position ranges start and end at multiples of flashe_sparse_span() = %d (or at the end of the vector)", kSpanFused);
    return FLASHE_OK;
}

int flashe_sparse_encrypt_aggregate_range_dev(flashe_ctx *ctx, uint32_t iter, uint32_t n_jobs, uint64_t total, int C, const uint32_t *idx,
                                              const uint32_t *const *loc_dev, const uint64_t *k, const uint64_t *const *pt_dev, int pt_limbs,
                                              const uint64_t *zeros, const flashe_span_bounds *bounds, uint64_t first, uint64_t count,
                                              uint64_t *const *ct_dev, uint64_t *agg_out_dev)
{
    CHECK_CTX(ctx);
    const int rc = check_span_range(ctx, total, first, count);
    if (rc) return rc;
    if (count == 0) return FLASHE_OK;
    return sparse_encrypt_aggregate_impl(ctx, iter, n_jobs, total, C, idx, loc_dev, k, pt_dev, pt_limbs, zeros, bounds, ct_dev, agg_out_dev, first, count, false);
}


// agg_dev == nullptr: out = the dense minus-mask.  agg_dev given: out = (agg - minus-mask) mod 2^b, the single-mask decrypt
// (jzf_flashe.py:531-532) in the pass that builds the mask.
static int sparse_minus_mask_impl(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                  uint64_t total, uint32_t n_jobs, bool sorted, const uint64_t *agg_dev, uint64_t *out_dev,
                                  const flashe_span_bounds *bounds = nullptr, uint64_t first = 0, uint64_t count = ~0ull)
{
    CHECK_CTX(ctx);
    const bool whole = count == ~0ull;                 // (a position range [first, first + count) is the *_range_dev form: the fused pass only)
    if (whole) count = total;
    if (C < 0 || (C && (!loc_dev || !k)) || (total && !out_dev) || n_jobs == 0) return fail(ctx, FLASHE_EINVAL, "bad arguments");
    { const int brc = C > 0 ? check_bounds(ctx, bounds, total, C, loc_dev, k) : FLASHE_OK; if (brc) return brc; }
    if (ctx->limbs == 2 && (!aligned16(out_dev) || !aligned16(agg_dev))) return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    uint64_t kmax = 0;
    for (int c = 0; c < C; c++) {
        if (k[c] > total) return fail(ctx, FLASHE_EINVAL, "client %d: more locations (%llu) than positions (%llu)", c,
                                      static_cast<unsigned long long>(k[c]), static_cast<unsigned long long>(total));
        kmax = std::max(kmax, k[c]);
    }
    const bool jobs_path = span_prf_ok(ctx);
    const bool fused = sorted && jobs_path && C > 0;      // the span reduce writes (or subtracts from agg) the whole vector itself
    if (!whole && !(fused && bounds)) return fail(ctx, FLASHE_EINVAL, "the position-range form needs int_bits > 64 on the table PRF, sorted lists and a bounds handle");
#ifdef FLASHE_TUNING
    const char *two_pass = getenv("FLASHE_SPAN_PRF");     // "0" = compact streams through HBM, then the plain span reduce (the round-2 .. 4 form; A/B runs)
    const bool prf_inside = fused && !(two_pass && two_pass[0] == '0' && !bounds);
#else
    const bool prf_inside = fused;
#endif
    if (prf_inside) {
        // strictly increasing lists: ONE persistent launch per group of clients generates the mask blocks inside the span reduce
        // (launch_span_prf) -- no compact streams in HBM, the AES rounds run under the dense read / write
        if (bounds && !bounds->start_fused) return fail(ctx, FLASHE_EINVAL, "the span bounds were computed by a ctx of another int_bits / PRF backend");
        int rc = bounds ? FLASHE_OK : ensure(ctx, ctx->bounds, span_table_words(total, std::min(C, kMaxScatter)) * sizeof(uint32_t));
        if (rc) return rc;
        for (int c0 = 0; c0 < C; c0 += kMaxScatter) {
            const int nc = std::min(kMaxScatter, C - c0);
            uint32_t idx[kMaxScatter];
            for (int e = 0; e < nc; e++) idx[e] = static_cast<uint32_t>(c0 + e);
            uint32_t *start = bounds ? bounds->start_fused + (c0 / kMaxScatter) * bounds->group_stride_fused : static_cast<uint32_t *>(ctx->bounds.p);
            if (!bounds) HIP_TRY(ctx, launch_span_bounds(ctx->env, nc, loc_dev + c0, k + c0, total, nullptr, start));
            HIP_TRY(ctx, launch_span_prf(ctx->env, iter, nc, idx, loc_dev + c0, k + c0, nullptr, 2, nullptr, nullptr, 0, 0, total, start,
                                         c0 != 0 ? out_dev : agg_dev, agg_dev != nullptr, out_dev, first, count));
        }
        return FLASHE_OK;
    }
    if (!fused && total) HIP_TRY(ctx, hipMemsetAsync(out_dev, 0, vec_bytes(ctx, total), ctx->env.stream));
    if (jobs_path) {
        // m = 1: the compact streams do not depend on their length, so every client's stream comes from one job-list
        // launch per group; unsorted lists are scattered one client at a time (location sets overlap between clients)
        const uint64_t kpad = (kmax + 1) & ~1ull;
        const int group = std::min(C, kMaxScatter);          // clients whose streams are held at once
        int rc = ensure(ctx, ctx->stream_tmp, vec_bytes(ctx, kpad) * static_cast<size_t>(std::max(group, 1)));
        if (rc) return rc;
        if (fused && (rc = ensure(ctx, ctx->bounds, span_table_words(total, group) * sizeof(uint32_t)))) return rc;
        uint64_t *tmp = static_cast<uint64_t *>(ctx->stream_tmp.p);
        for (int c0 = 0; c0 < C; c0 += group) {
            const int nc = std::min(group, C - c0);
            const uint64_t *streams[kMaxScatter];
            PrfJob jobs[kMaxScatter];
            for (int e = 0; e < nc; e++) {
                streams[e] = tmp + 2 * kpad * static_cast<uint64_t>(e);
                jobs[e] = PrfJob{static_cast<uint32_t>(c0 + e), 0u, 0, k[c0 + e], nullptr, 0, tmp + 2 * kpad * static_cast<uint64_t>(e)};
            }
            HIP_TRY(ctx, launch_prf_jobs(ctx->env, iter, false, nc, jobs, kmax, n_jobs));
            if (fused) {
                HIP_TRY(ctx, launch_span_reduce(ctx->env, nc, loc_dev + c0, streams, k + c0, nullptr, 0, 0, total, static_cast<uint32_t *>(ctx->bounds.p),
                                                c0 != 0 ? out_dev : agg_dev, agg_dev != nullptr, out_dev));
                continue;
            }
            for (int e = 0; e < nc; e++)           // unsorted lists
                HIP_TRY(ctx, launch_scatter(ctx->env, total, k[c0 + e], loc_dev[c0 + e], streams[e], out_dev, true));
        }
    } else {
        int rc = ensure(ctx, ctx->stream_tmp, vec_bytes(ctx, kmax));
        if (rc) return rc;
        uint64_t *tmp = static_cast<uint64_t *>(ctx->stream_tmp.p);
        for (int c = 0; c < C; c++) {
            if (!k[c]) continue;
            const uint32_t idx = static_cast<uint32_t>(c);
            HIP_TRY(ctx, launch_prf(ctx->env, iter, &idx, 1, nullptr, 0, k[c], n_jobs, 0, k[c], nullptr, 0, tmp));
            HIP_TRY(ctx, launch_scatter(ctx->env, total, k[c], loc_dev[c], tmp, out_dev, true));
        }
    }
    // the paths that built the mask itself: subtract it from the aggregate in place
    if (agg_dev && !fused && total) HIP_TRY(ctx, launch_combine(ctx->env, total, agg_dev, ctx->limbs, nullptr, out_dev, out_dev));
    return FLASHE_OK;
}

int flashe_sparse_minus_mask_dev(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                 uint64_t total, uint32_t n_jobs, uint64_t *out_dev)
{
    return sparse_minus_mask_impl(ctx, iter, C, loc_dev, k, total, n_jobs, false, nullptr, out_dev);
}

int flashe_sparse_minus_mask_sorted_dev(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev, const uint64_t *k,
                                        uint64_t total, uint32_t n_jobs, uint64_t *out_dev)
{
    return sparse_minus_mask_impl(ctx, iter, C, loc_dev, k, total, n_jobs, true, nullptr, out_dev);
}

int flashe_sparse_decrypt_dev(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev, const uint64_t *k, uint64_t total,
                              uint32_t n_jobs, int sorted, const uint64_t *agg_dev, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (total && (!agg_dev || agg_dev == out_dev)) return fail(ctx, FLASHE_EINVAL, "the aggregate must be given and must not be the output vector");
    return sparse_minus_mask_impl(ctx, iter, C, loc_dev, k, total, n_jobs, sorted != 0, agg_dev, out_dev);
}

int flashe_sparse_decrypt_bounds_dev(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev, const uint64_t *k, uint64_t total,
                                     uint32_t n_jobs, const flashe_span_bounds *bounds, const uint64_t *agg_dev, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (!bounds) return fail(ctx, FLASHE_EINVAL, "null bounds handle");
    if (total && (!agg_dev || agg_dev == out_dev)) return fail(ctx, FLASHE_EINVAL, "the aggregate must be given and must not be the output vector");
    if (!span_prf_ok(ctx) || std::min(C, kMaxScatter) != std::min(bounds->C, kMaxScatter))
        return fail(ctx, FLASHE_EINVAL, "sparse_decrypt_bounds needs int_bits > 64 on the table PRF (the span reduce is what consumes the bounds)");
    return sparse_minus_mask_impl(ctx, iter, C, loc_dev, k, total, n_jobs, true, agg_dev, out_dev, bounds);
}

int flashe_sparse_decrypt_range_dev(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev, const uint64_t *k, uint64_t total,
                                    uint32_t n_jobs, const flashe_span_bounds *bounds, uint64_t first, uint64_t count, const uint64_t *agg_dev,
                                    uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (!bounds) return fail(ctx, FLASHE_EINVAL, "null bounds handle");
    const int rc = check_span_range(ctx, total, first, count);
    if (rc) return rc;
    if (count == 0) return FLASHE_OK;
    if (!agg_dev || agg_dev == out_dev) return fail(ctx, FLASHE_EINVAL, "the aggregate must be given and must not be the output vector");
    if (std::min(C, kMaxScatter) != std::min(bounds->C, kMaxScatter)) return fail(ctx, FLASHE_EINVAL, "the bounds handle describes other lists");
    return sparse_minus_mask_impl(ctx, iter, C, loc_dev, k, total, n_jobs, true, agg_dev, out_dev, bounds, first, count);
}

int flashe_sparse_double_masks_dev(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc_dev, const uint64_t *k, uint64_t total,
                                   uint64_t *add_out_dev, uint64_t *minus_out_dev)
{
    CHECK_CTX(ctx);
    if (C < 0 || (C && (!loc_dev || !k)) || (total && (!add_out_dev || !minus_out_dev)) || add_out_dev == minus_out_dev)
        return fail(ctx, FLASHE_EINVAL, "flashe_sparse_double_masks_dev: bad arguments");
    if (ctx->limbs == 2 && (!aligned16(add_out_dev) || !aligned16(minus_out_dev))) return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    if (total >= (1ull << 32)) return fail(ctx, FLASHE_EINVAL, "total must be below 2^32 (32-bit locations)");
    uint64_t kmax = 0;
    for (int c = 0; c < C; c++) {
        if (k[c] > total) return fail(ctx, FLASHE_EINVAL, "client %d: more locations (%llu) than positions (%llu)", c,
                                      static_cast<unsigned long long>(k[c]), static_cast<unsigned long long>(total));
        if (k[c] && !loc_dev[c]) return fail(ctx, FLASHE_EINVAL, "client %d: null location list", c);
        kmax = std::max(kmax, k[c]);
    }
    if (total == 0) return FLASHE_OK;
    if (C == 0 || kmax == 0) {
        HIP_TRY(ctx, hipMemsetAsync(add_out_dev, 0, vec_bytes(ctx, total), ctx->env.stream));
        HIP_TRY(ctx, hipMemsetAsync(minus_out_dev, 0, vec_bytes(ctx, total), ctx->env.stream));
        return FLASHE_OK;
    }
    const uint64_t kpad = (kmax + 1) & ~1ull;
    const int group = std::min(C, kMaxScatter);
    int rc = ensure(ctx, ctx->stream_tmp, 2 * vec_bytes(ctx, kpad) * static_cast<size_t>(group));
    if (rc) return rc;
    if ((rc = ensure(ctx, ctx->bounds, span_table_words(total, group) * sizeof(uint32_t)))) return rc;
    uint64_t *tmp = static_cast<uint64_t *>(ctx->stream_tmp.p);
    const uint64_t stride = kpad * static_cast<uint64_t>(ctx->limbs);
    for (int c0 = 0; c0 < C; c0 += group) {
        const int nc = std::min(group, C - c0);
        const uint32_t *locs[kMaxScatter + 2];
        uint64_t ks[kMaxScatter + 2];
        uint64_t *va[kMaxScatter], *vm[kMaxScatter];
        for (int e = -1; e <= nc; e++) {
            const int c = c0 + e;
            const bool there = c >= 0 && c < C && k[c];
            locs[e + 1] = there ? loc_dev[c] : nullptr;
            ks[e + 1] = there ? k[c] : 0;
        }
        for (int e = 0; e < nc; e++) { va[e] = tmp + stride * static_cast<uint64_t>(2 * e); vm[e] = tmp + stride * static_cast<uint64_t>(2 * e + 1); }
        HIP_TRY(ctx, launch_sparse_edge_prf(ctx->env, iter, nc, static_cast<uint32_t>(c0), locs, ks, va, vm));
        // the compact values go to their dense positions through the span reduce (it also validates the lists: a position >= total or a
        // list that is not strictly increasing is skipped and reported by the next synchronising call)
        HIP_TRY(ctx, launch_span_reduce(ctx->env, nc, loc_dev + c0, va, k + c0, nullptr, 0, 0, total, static_cast<uint32_t *>(ctx->bounds.p),
                                        c0 ? add_out_dev : nullptr, false, add_out_dev));
        HIP_TRY(ctx, launch_span_reduce(ctx->env, nc, loc_dev + c0, vm, k + c0, nullptr, 0, 0, total, static_cast<uint32_t *>(ctx->bounds.p),
                                        c0 ? minus_out_dev : nullptr, false, minus_out_dev));
    }
    return FLASHE_OK;
}

// Arbiter.dynamic_masking's decision inputs (jzf_flashe_block.py:92-112) from device-resident, strictly increasing location lists:
// single_cost = 2 sum_c k_c; double_cost = 2 single_cost - 2 (positions shared by consecutive clients).  Synchronous.
int flashe_dynamic_masking_cost_dev(flashe_ctx *ctx, int C, const uint32_t *const *loc_dev, const uint64_t *k, uint64_t *single_cost,
                                    uint64_t *double_cost)
{
    CHECK_CTX(ctx);
    if (C < 0 || !single_cost || !double_cost || (C && (!loc_dev || !k))) return fail(ctx, FLASHE_EINVAL, "dynamic_masking_cost: bad arguments");
    if (ctx->capturing) return fail(ctx, FLASHE_EINVAL, "flashe_dynamic_masking_cost_dev is synchronous: not capturable");
    uint64_t entries = 0;
    for (int c = 0; c < C; c++) {
        if (k[c] && !loc_dev[c]) return fail(ctx, FLASHE_EINVAL, "list %d is null", c);
        if (k[c] >= (1ull << 32)) return fail(ctx, FLASHE_EINVAL, "list %d is too long", c);
        entries += k[c];
    }
    unsigned long long shared = 0;
    if (C > 1 && entries) {
        int rc = ensure(ctx, ctx->bounds, 4096);
        if (rc) return rc;
        unsigned long long *cnt = static_cast<unsigned long long *>(ctx->bounds.p);
        HIP_TRY(ctx, hipMemsetAsync(cnt, 0, sizeof *cnt, ctx->env.stream));
        for (int c0 = 0; c0 + 1 < C; c0 += kMaxScatter) {
            const int nc = std::min(kMaxScatter, C - 1 - c0);                  // clients c0 .. c0 + nc - 1 each look into their successor
            const uint32_t *loc[kMaxScatter + 1];
            uint64_t kk[kMaxScatter + 1];
            for (int e = 0; e <= nc; e++) { loc[e] = loc_dev[c0 + e]; kk[e] = k[c0 + e]; }
            HIP_TRY(ctx, launch_shared_positions(ctx->env, nc, loc, kk, cnt));
        }
        HIP_TRY(ctx, hipMemcpyAsync(&shared, cnt, sizeof shared, hipMemcpyDeviceToHost, ctx->env.stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream));
    }
    *single_cost = 2 * entries;
    *double_cost = 4 * entries - 2 * static_cast<uint64_t>(shared);
    return FLASHE_OK;
}

int flashe_sparse_dense_mask_dev(flashe_ctx *ctx, uint32_t iter, int n_lists, const uint8_t *const *sel_dev, uint64_t total,
                                 uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (n_lists < 0 || (n_lists && !sel_dev) || (total && !out_dev)) return fail(ctx, FLASHE_EINVAL, "bad arguments");
    if (ctx->limbs == 2 && !aligned16(out_dev)) return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    if (total) HIP_TRY(ctx, hipMemsetAsync(out_dev, 0, vec_bytes(ctx, total), ctx->env.stream));
    int rc = ensure(ctx, ctx->stream_tmp, vec_bytes(ctx, total));
    if (rc) return rc;
    uint64_t *tmp = static_cast<uint64_t *>(ctx->stream_tmp.p);
    for (int i = 0; i < n_lists; i++) {
        const uint32_t idx = static_cast<uint32_t>(i);
        // one chunk with begin = 0 (n_jobs = 1): counters are dense-position block indices
        HIP_TRY(ctx, launch_prf(ctx->env, iter, &idx, 1, nullptr, 0, total, 1, 0, total, nullptr, 0, tmp));
        HIP_TRY(ctx, launch_sel_accumulate(ctx->env, total, sel_dev[i], tmp, out_dev));
    }
    return FLASHE_OK;
}

// ---- quantise / batch codec ----
static int check_codec_bits(flashe_ctx *ctx, int element_bits)
{
    if (element_bits < 1 || element_bits > 62) return fail(ctx, FLASHE_EINVAL, "element_bits must be in [1, 62], got %d", element_bits);
    return FLASHE_OK;
}

int flashe_quantize_dev(flashe_ctx *ctx, uint64_t n, const void *x_dev, int x_is_f64, double alpha, int element_bits,
                        const double *u_dev, uint64_t *q_dev)
{
    CHECK_CTX(ctx);
    if (n && (!x_dev || !u_dev || !q_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (!(alpha > 0)) return fail(ctx, FLASHE_EINVAL, "alpha must be positive");
    int rc = check_codec_bits(ctx, element_bits);
    if (rc) return rc;
    HIP_TRY(ctx, launch_quantize(ctx->env, n, x_dev, x_is_f64 != 0, alpha, element_bits, u_dev, q_dev));
    return FLASHE_OK;
}

int flashe_unquantize_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *v_dev, int v_limbs, double alpha, int element_bits,
                          int num_clients, double *out_dev)
{
    CHECK_CTX(ctx);
    if (n && (!v_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (v_limbs != 1 && v_limbs != 2) return fail(ctx, FLASHE_EINVAL, "v_limbs must be 1 or 2");
    if (v_limbs == 2 && !aligned16(v_dev)) return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    if (num_clients < 1) return fail(ctx, FLASHE_EINVAL, "num_clients must be >= 1");
    int rc = check_codec_bits(ctx, element_bits);
    if (rc) return rc;
    HIP_TRY(ctx, launch_unquantize(ctx->env, n, v_dev, v_limbs, alpha, element_bits, num_clients, out_dev));
    return FLASHE_OK;
}

static int check_field_bits(flashe_ctx *ctx, int field_bits)
{
    if (field_bits < 1 || field_bits > 64 || field_bits > ctx->int_bits)
        return fail(ctx, FLASHE_EINVAL, "field_bits must be in [1, min(64, int_bits)], got %d", field_bits);
    return FLASHE_OK;
}

int flashe_batch_dev(flashe_ctx *ctx, uint64_t n, const uint64_t *vals_dev, int field_bits, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (n && (!vals_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    int rc = check_field_bits(ctx, field_bits);
    if (rc) return rc;
    if (ctx->limbs == 2 && !aligned16(out_dev)) return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    HIP_TRY(ctx, launch_batch(ctx->env, n, vals_dev, field_bits, out_dev));
    return FLASHE_OK;
}

int flashe_unbatch_dev(flashe_ctx *ctx, uint64_t n_batches, const uint64_t *in_dev, int field_bits, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (n_batches && (!in_dev || !out_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    int rc = check_field_bits(ctx, field_bits);
    if (rc) return rc;
    if (ctx->limbs == 2 && !aligned16(in_dev)) return fail(ctx, FLASHE_EINVAL, "device vectors must be 16-byte aligned");
    HIP_TRY(ctx, launch_unbatch(ctx->env, n_batches, in_dev, field_bits, out_dev));
    return FLASHE_OK;
}

// ---- sparsifier ----
int flashe_sparsify_dev(flashe_ctx *ctx, uint64_t n, uint64_t k, const void *x_dev, int x_is_f64, void *residual_dev, uint32_t *loc_dev,
                        void *vals_dev)
{
    CHECK_CTX(ctx);
    if (n >= (1ull << 32)) return fail(ctx, FLASHE_EINVAL, "sparsify: n must be < 2^32");
    if (k > n) return fail(ctx, FLASHE_EINVAL, "sparsify: k (%llu) > n (%llu)", static_cast<unsigned long long>(k),
                           static_cast<unsigned long long>(n));
    if (n && k && (!x_dev || !loc_dev || !vals_dev)) return fail(ctx, FLASHE_EINVAL, "null vector");
    int rc = ensure(ctx, ctx->sp_ws, sparsify_workspace_bytes(n));
    if (rc) return rc;
    HIP_TRY(ctx, launch_sparsify(ctx->env, n, k, x_dev, x_is_f64 != 0, residual_dev, loc_dev, vals_dev, ctx->sp_ws.p));
    return FLASHE_OK;
}

// Every layer of a model at once: layer l = elements [off_l, off_l + n[l]) of the flat vectors (layers back to back), its k[l] entries go
// to [koff_l, koff_l + k[l]) of the flat outputs, locations relative to the layer.  n and k are HOST arrays.
int flashe_sparsify_batch_dev(flashe_ctx *ctx, int n_layers, const uint64_t *n, const uint64_t *k, const void *x_dev, int x_is_f64, void *residual_dev,
                              uint32_t *loc_dev, void *vals_dev)
{
    CHECK_CTX(ctx);
    if (n_layers < 0 || (n_layers && (!n || !k))) return fail(ctx, FLASHE_EINVAL, "sparsify_batch: bad arguments");
    if (n_layers == 0) return FLASHE_OK;
    if (ctx->capturing) return fail(ctx, FLASHE_EINVAL, "sparsify_batch: not inside a graph capture (the layer table is uploaded synchronously)");
    uint64_t total = 0, total_k = 0;
    for (int l = 0; l < n_layers; l++) {
        if (n[l] >= (1ull << 32)) return fail(ctx, FLASHE_EINVAL, "sparsify_batch: layer %d: n must be < 2^32", l);
        if (k[l] > n[l]) return fail(ctx, FLASHE_EINVAL, "sparsify_batch: layer %d: k (%llu) > n (%llu)", l, static_cast<unsigned long long>(k[l]),
                                     static_cast<unsigned long long>(n[l]));
        total += n[l]; total_k += k[l];
    }
    if (total == 0 || total_k == 0) return FLASHE_OK;
    if (!x_dev || !loc_dev || !vals_dev) return fail(ctx, FLASHE_EINVAL, "null vector");
    std::vector<unsigned char> desc(sparsify_batch_desc_bytes(n_layers));
    const uint64_t blocks = sparsify_batch_layout(n_layers, n, k, desc.data());
    if (blocks >= (1ull << 32)) return fail(ctx, FLASHE_EINVAL, "sparsify_batch: too many elements");
    int rc = ensure(ctx, ctx->sp_ws, sparsify_batch_workspace_bytes(n_layers, blocks));
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_ws.p, desc.data(), desc.size(), hipMemcpyHostToDevice, ctx->env.stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream));                  // the table is on the device before `desc` goes away
    HIP_TRY(ctx, launch_sparsify_batch(ctx->env, n_layers, blocks, x_dev, x_is_f64 != 0, residual_dev, loc_dev, vals_dev, ctx->sp_ws.p));
    return FLASHE_OK;
}

// ------------------------------------------------------------------------------------------
// Host-pointer twins: H2D, the _dev call, D2H, synchronous.
// ------------------------------------------------------------------------------------------
#define H2D(dst, src, bytes) HIP_TRY(ctx, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, ctx->env.stream))
#define D2H(dst, src, bytes)                                                                         \
    do {                                                                                             \
        HIP_TRY(ctx, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, ctx->env.stream)); \
        HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream));                                         \
    } while (0)

// Pipelined form of the big twins (encrypt / decrypt / aggregate_elem): the vector is cut into chunks; chunk q's upload and kernel run
// on the ctx stream while chunk q - 1's result travels back on a second stream, so a call costs max(upload, download) instead of
// their sum.  It needs a PAGE-LOCKED destination: a download into pageable memory blocks the host until it is done (measured: an H2D
// and a D2H issued on two streams take 5.96 ms with pageable buffers, 3.46 ms pinned, tests/perf/pcie_probe.py), while an upload from
// pageable memory -- blocking as well -- runs at the pinned rate and overlaps with a download that is already in flight.  So the
// path is taken when the caller's output pointer is pinned (flashe_host_alloc / hipHostMalloc / hipHostRegister; the Python layer's
// result pool hands out such arrays) and the vector is large enough for chunks to matter.
namespace {

constexpr size_t kPipeMinBytes = 16u << 20;

bool host_pinned(const void *p)
{
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }      // plain malloc memory: an error, not a type
    return a.type == hipMemoryTypeHost;
}

// elements per chunk: FLASHE_TWIN_CHUNK_MB of output (default 32 MB), a multiple of 4096 elements (block- and tile-aligned for every b)
uint64_t pipe_chunk_elems(const flashe_ctx *ctx)
{
    const char *e = getenv("FLASHE_TWIN_CHUNK_MB");          // read per call: a tuning knob, and tests shrink it
    const long v = e ? atol(e) : 32;
    const size_t mb = static_cast<size_t>(v < 1 ? 1 : v);
    const uint64_t per = (mb << 20) / (static_cast<size_t>(ctx->limbs) * 8);
    return std::max<uint64_t>(4096, per & ~static_cast<uint64_t>(4095));
}

bool pipe_wanted(const flashe_ctx *ctx, uint64_t n, const void *out_host)
{
    const char *e = getenv("FLASHE_TWIN_PIPELINE");
    const bool off = e && atoi(e) == 0;
    return !off && !ctx->capturing && ctx->env.stream2 && vec_bytes(ctx, n) >= kPipeMinBytes && n > pipe_chunk_elems(ctx) && host_pinned(out_host);
}

// chunk results: device -> pinned host on the second stream, behind everything the ctx stream has been given so far
hipError_t pipe_copy_out(flashe_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes, int q)
{
    hipEvent_t &ev = ctx->ev_copy[q & 1];
    hipError_t e;
    if (!ev && (e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return e;
    if ((e = hipEventRecord(ev, ctx->env.stream)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(ctx->env.stream2, ev, 0)) != hipSuccess) return e;
    return hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->env.stream2);
}

// both streams idle: the staging blocks may be handed back (also on the error paths: a copy must not outlive its block's lease)
hipError_t pipe_drain(flashe_ctx *ctx)
{
    const hipError_t a = hipStreamSynchronize(ctx->env.stream), b = hipStreamSynchronize(ctx->env.stream2);
    return a != hipSuccess ? a : b;
}

}  // namespace

#define PIPE_TRY(expr)                                                       \
    do {                                                                     \
        hipError_t pe_ = (expr);                                             \
        if (pe_ != hipSuccess) { (void)pipe_drain(ctx); HIP_TRY(ctx, pe_); } \
    } while (0)

int flashe_mask(flashe_ctx *ctx, uint32_t iter, const uint32_t *idx, int n_idx, uint64_t n, uint32_t n_jobs, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (n == 0) return FLASHE_OK;
    if (!out) return fail(ctx, FLASHE_EINVAL, "null output");
    Tmp d;
    HIP_TRY(ctx, d.alloc(ctx, vec_bytes(ctx, n)));
    int rc = flashe_mask_dev(ctx, iter, idx, n_idx, n, n_jobs, d.as<uint64_t>());
    if (rc) return rc;
    D2H(out, d.p, vec_bytes(ctx, n));
    return FLASHE_OK;
}

int flashe_encrypt(flashe_ctx *ctx, uint32_t iter, uint32_t idx, int scheme, uint64_t n, uint32_t n_jobs, const uint64_t *pt,
                   int pt_limbs, uint64_t *ct)
{
    CHECK_CTX(ctx);
    if (int rc = check_double_idx(ctx, scheme, &idx, 1)) return rc;
    if (n == 0) return FLASHE_OK;
    if (!pt || !ct) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (pt_limbs != 1 && pt_limbs != ctx->limbs) return fail(ctx, FLASHE_EINVAL, "pt_limbs must be 1 or %d", ctx->limbs);
    Tmp dp, dc;
    HIP_TRY(ctx, dp.alloc(ctx, static_cast<size_t>(n) * pt_limbs * 8));
    HIP_TRY(ctx, dc.alloc(ctx, vec_bytes(ctx, n)));
    if (pipe_wanted(ctx, n, ct)) {
        const uint64_t ch = pipe_chunk_elems(ctx);
        int q = 0;
        for (uint64_t f = 0; f < n; f += ch, q++) {
            const uint64_t cnt = std::min(ch, n - f);
            PIPE_TRY(hipMemcpyAsync(dp.as<uint64_t>() + f * pt_limbs, pt + f * pt_limbs, static_cast<size_t>(cnt) * pt_limbs * 8, hipMemcpyHostToDevice,
                                    ctx->env.stream));
            const int rc = flashe_encrypt_range_dev(ctx, iter, idx, scheme, n, n_jobs, f, cnt, dp.as<uint64_t>() + f * pt_limbs, pt_limbs,
                                                    dc.as<uint64_t>() + f * ctx->limbs);
            if (rc) { (void)pipe_drain(ctx); return rc; }
            PIPE_TRY(pipe_copy_out(ctx, ct + f * ctx->limbs, dc.as<uint64_t>() + f * ctx->limbs, vec_bytes(ctx, cnt), q));
        }
        HIP_TRY(ctx, pipe_drain(ctx));
        return FLASHE_OK;
    }
    H2D(dp.p, pt, static_cast<size_t>(n) * pt_limbs * 8);
    int rc = flashe_encrypt_dev(ctx, iter, idx, scheme, n, n_jobs, dp.as<uint64_t>(), pt_limbs, dc.as<uint64_t>());
    if (rc) return rc;
    D2H(ct, dc.p, vec_bytes(ctx, n));
    return FLASHE_OK;
}

// host-pointer twins of the prepared calls (synchronous: H2D + one HBM-bound kernel + D2H; the masks never leave the device)
int flashe_encrypt_prepared(flashe_ctx *ctx, uint64_t n, const uint64_t *pt, int pt_limbs, uint64_t *ct)
{
    CHECK_CTX(ctx);
    if (n && (!pt || !ct)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (pt_limbs != 1 && pt_limbs != ctx->limbs) return fail(ctx, FLASHE_EINVAL, "pt_limbs must be 1 or %d", ctx->limbs);
    if (n == 0) return flashe_encrypt_prepared_dev(ctx, 0, nullptr, pt_limbs, nullptr);
    Tmp dp, dc;
    HIP_TRY(ctx, dp.alloc(ctx, static_cast<size_t>(n) * pt_limbs * 8));
    HIP_TRY(ctx, dc.alloc(ctx, vec_bytes(ctx, n)));
    H2D(dp.p, pt, static_cast<size_t>(n) * pt_limbs * 8);
    int rc = flashe_encrypt_prepared_dev(ctx, n, dp.as<uint64_t>(), pt_limbs, dc.as<uint64_t>());
    if (rc) return rc;
    D2H(ct, dc.p, vec_bytes(ctx, n));
    return FLASHE_OK;
}

int flashe_decrypt_prepared(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add, const uint32_t *minus_idx, int n_minus,
                            uint64_t n, uint32_t n_jobs, const uint64_t *in, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (n && (!in || !out)) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (n == 0) return flashe_decrypt_prepared_dev(ctx, iter, add_idx, n_add, minus_idx, n_minus, 0, n_jobs, nullptr, nullptr);
    Tmp di, dout;
    HIP_TRY(ctx, di.alloc(ctx, vec_bytes(ctx, n)));
    HIP_TRY(ctx, dout.alloc(ctx, vec_bytes(ctx, n)));
    H2D(di.p, in, vec_bytes(ctx, n));
    int rc = flashe_decrypt_prepared_dev(ctx, iter, add_idx, n_add, minus_idx, n_minus, n, n_jobs, di.as<uint64_t>(), dout.as<uint64_t>());
    if (rc) return rc;
    D2H(out, dout.p, vec_bytes(ctx, n));
    return FLASHE_OK;
}

int flashe_decrypt(flashe_ctx *ctx, uint32_t iter, const uint32_t *add_idx, int n_add, const uint32_t *minus_idx, int n_minus,
                   uint64_t n, uint32_t n_jobs, const uint64_t *in, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (n == 0) return FLASHE_OK;
    if (!in || !out) return fail(ctx, FLASHE_EINVAL, "null vector");
    Tmp di, dout;
    HIP_TRY(ctx, di.alloc(ctx, vec_bytes(ctx, n)));
    HIP_TRY(ctx, dout.alloc(ctx, vec_bytes(ctx, n)));
    if (pipe_wanted(ctx, n, out)) {
        const uint64_t ch = pipe_chunk_elems(ctx);
        int q = 0;
        for (uint64_t f = 0; f < n; f += ch, q++) {
            const uint64_t cnt = std::min(ch, n - f), off = f * ctx->limbs;
            PIPE_TRY(hipMemcpyAsync(di.as<uint64_t>() + off, in + off, vec_bytes(ctx, cnt), hipMemcpyHostToDevice, ctx->env.stream));
            const int rc = flashe_decrypt_range_dev(ctx, iter, add_idx, n_add, minus_idx, n_minus, n, n_jobs, f, cnt, di.as<uint64_t>() + off,
                                                    dout.as<uint64_t>() + off);
            if (rc) { (void)pipe_drain(ctx); return rc; }
            PIPE_TRY(pipe_copy_out(ctx, out + off, dout.as<uint64_t>() + off, vec_bytes(ctx, cnt), q));
        }
        HIP_TRY(ctx, pipe_drain(ctx));
        return FLASHE_OK;
    }
    H2D(di.p, in, vec_bytes(ctx, n));
    int rc = flashe_decrypt_dev(ctx, iter, add_idx, n_add, minus_idx, n_minus, n, n_jobs, di.as<uint64_t>(), dout.as<uint64_t>());
    if (rc) return rc;
    D2H(out, dout.p, vec_bytes(ctx, n));
    return FLASHE_OK;
}

int flashe_combine(flashe_ctx *ctx, uint64_t n, const uint64_t *in, int in_limbs, const uint64_t *add, const uint64_t *minus, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (n == 0) return FLASHE_OK;
    if (!in || !out) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (in_limbs != 1 && in_limbs != ctx->limbs) return fail(ctx, FLASHE_EINVAL, "in_limbs must be 1 or %d", ctx->limbs);
    Tmp di, da, dm, dout;
    HIP_TRY(ctx, di.alloc(ctx, static_cast<size_t>(n) * in_limbs * 8));
    HIP_TRY(ctx, dout.alloc(ctx, vec_bytes(ctx, n)));
    H2D(di.p, in, static_cast<size_t>(n) * in_limbs * 8);
    if (add) { HIP_TRY(ctx, da.alloc(ctx, vec_bytes(ctx, n))); H2D(da.p, add, vec_bytes(ctx, n)); }
    if (minus) { HIP_TRY(ctx, dm.alloc(ctx, vec_bytes(ctx, n))); H2D(dm.p, minus, vec_bytes(ctx, n)); }
    int rc = flashe_combine_dev(ctx, n, di.as<uint64_t>(), in_limbs, add ? da.as<uint64_t>() : nullptr,
                                minus ? dm.as<uint64_t>() : nullptr, dout.as<uint64_t>());
    if (rc) return rc;
    D2H(out, dout.p, vec_bytes(ctx, n));
    return FLASHE_OK;
}

int flashe_aggregate_elem(flashe_ctx *ctx, int C, const uint64_t *const *cts, uint64_t n, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (C < 1 || !cts) return fail(ctx, FLASHE_EINVAL, "aggregate_elem: bad arguments (C = %d)", C);
    if (n == 0) return FLASHE_OK;
    if (!out) return fail(ctx, FLASHE_EINVAL, "null output");
    const size_t vb = (vec_bytes(ctx, n) + 15) & ~static_cast<size_t>(15);
    Tmp all, dout;
    HIP_TRY(ctx, all.alloc(ctx, vb * C));
    HIP_TRY(ctx, dout.alloc(ctx, vb));
    std::vector<const uint64_t *> ptrs(C);
    for (int c = 0; c < C; c++) {
        if (!cts[c]) return fail(ctx, FLASHE_EINVAL, "operand %d is null", c);
        ptrs[c] = reinterpret_cast<const uint64_t *>(all.as<char>() + vb * c);
    }
    if (pipe_wanted(ctx, n, out)) {
        // upload bound (C vectors up, one down): the chunks hide the reduce kernels and the download under the uploads
        const uint64_t ch = pipe_chunk_elems(ctx);
        std::vector<const uint64_t *> part(C);
        int q = 0;
        for (uint64_t f = 0; f < n; f += ch, q++) {
            const uint64_t cnt = std::min(ch, n - f), off = f * ctx->limbs;
            for (int c = 0; c < C; c++) {
                part[c] = ptrs[c] + off;
                PIPE_TRY(hipMemcpyAsync(const_cast<uint64_t *>(part[c]), cts[c] + off, vec_bytes(ctx, cnt), hipMemcpyHostToDevice, ctx->env.stream));
            }
            const int rc = flashe_aggregate_elem_dev(ctx, C, part.data(), cnt, dout.as<uint64_t>() + off);
            if (rc) { (void)pipe_drain(ctx); return rc; }
            PIPE_TRY(pipe_copy_out(ctx, out + off, dout.as<uint64_t>() + off, vec_bytes(ctx, cnt), q));
        }
        HIP_TRY(ctx, pipe_drain(ctx));
        return FLASHE_OK;
    }
    for (int c = 0; c < C; c++) H2D(const_cast<uint64_t *>(ptrs[c]), cts[c], vec_bytes(ctx, n));
    int rc = flashe_aggregate_elem_dev(ctx, C, ptrs.data(), n, dout.as<uint64_t>());
    if (rc) return rc;
    D2H(out, dout.p, vec_bytes(ctx, n));
    return FLASHE_OK;
}

int flashe_aggregate_packed(flashe_ctx *ctx, int C, const uint64_t *const *packed, uint64_t n_limbs, uint64_t total_bits, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (C < 1 || !packed) return fail(ctx, FLASHE_EINVAL, "aggregate_packed: bad arguments (C = %d)", C);
    if (n_limbs != (total_bits + 63) / 64) return fail(ctx, FLASHE_EINVAL, "n_limbs must equal ceil(total_bits / 64)");
    if (n_limbs == 0) return FLASHE_OK;
    if (!out) return fail(ctx, FLASHE_EINVAL, "null output");
    const size_t vb = (static_cast<size_t>(n_limbs) * 8 + 15) & ~static_cast<size_t>(15);
    Tmp all, dout;
    HIP_TRY(ctx, all.alloc(ctx, vb * C));
    HIP_TRY(ctx, dout.alloc(ctx, vb));
    std::vector<const uint64_t *> ptrs(C);
    for (int c = 0; c < C; c++) {
        if (!packed[c]) return fail(ctx, FLASHE_EINVAL, "operand %d is null", c);
        ptrs[c] = reinterpret_cast<const uint64_t *>(all.as<char>() + vb * c);
        H2D(const_cast<uint64_t *>(ptrs[c]), packed[c], static_cast<size_t>(n_limbs) * 8);
    }
    int rc = flashe_aggregate_packed_dev(ctx, C, ptrs.data(), n_limbs, total_bits, dout.as<uint64_t>());
    if (rc) return rc;
    D2H(out, dout.p, static_cast<size_t>(n_limbs) * 8);
    return FLASHE_OK;
}

int flashe_pack(flashe_ctx *ctx, uint64_t n, const uint64_t *in, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (n == 0) return FLASHE_OK;
    if (!in || !out) return fail(ctx, FLASHE_EINVAL, "null vector");
    const size_t ob = static_cast<size_t>((n * ctx->int_bits + 63) / 64) * 8;
    Tmp di, dout;
    HIP_TRY(ctx, di.alloc(ctx, vec_bytes(ctx, n)));
    HIP_TRY(ctx, dout.alloc(ctx, ob));
    H2D(di.p, in, vec_bytes(ctx, n));
    int rc = flashe_pack_dev(ctx, n, di.as<uint64_t>(), dout.as<uint64_t>());
    if (rc) return rc;
    D2H(out, dout.p, ob);
    return FLASHE_OK;
}

int flashe_unpack(flashe_ctx *ctx, uint64_t n, const uint64_t *in, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (n == 0) return FLASHE_OK;
    if (!in || !out) return fail(ctx, FLASHE_EINVAL, "null vector");
    const size_t ib = static_cast<size_t>((n * ctx->int_bits + 63) / 64) * 8;
    Tmp di, dout;
    HIP_TRY(ctx, di.alloc(ctx, ib));
    HIP_TRY(ctx, dout.alloc(ctx, vec_bytes(ctx, n)));
    H2D(di.p, in, ib);
    int rc = flashe_unpack_dev(ctx, n, di.as<uint64_t>(), dout.as<uint64_t>());
    if (rc) return rc;
    D2H(out, dout.p, vec_bytes(ctx, n));
    return FLASHE_OK;
}

int flashe_expand_to_dense(flashe_ctx *ctx, uint64_t total, uint64_t k, const uint32_t *loc, const uint64_t *vals,
                           const uint64_t *zero, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (total == 0) return FLASHE_OK;
    if (!out || !zero || (k && (!loc || !vals))) return fail(ctx, FLASHE_EINVAL, "null argument");
    for (uint64_t q = 0; q < k; q++)
        if (loc[q] >= total) return fail(ctx, FLASHE_EINVAL, "location %llu out of range", static_cast<unsigned long long>(loc[q]));
    Tmp dl, dv, dout;
    HIP_TRY(ctx, dl.alloc(ctx, static_cast<size_t>(k) * 4));
    HIP_TRY(ctx, dv.alloc(ctx, vec_bytes(ctx, k)));
    HIP_TRY(ctx, dout.alloc(ctx, vec_bytes(ctx, total)));
    if (k) { H2D(dl.p, loc, static_cast<size_t>(k) * 4); H2D(dv.p, vals, vec_bytes(ctx, k)); }
    int rc = flashe_expand_to_dense_dev(ctx, total, k, dl.as<uint32_t>(), dv.as<uint64_t>(), zero, dout.as<uint64_t>());
    if (rc) return rc;
    D2H(out, dout.p, vec_bytes(ctx, total));
    return FLASHE_OK;
}

int flashe_sparse_minus_mask(flashe_ctx *ctx, uint32_t iter, int C, const uint32_t *const *loc, const uint64_t *k, uint64_t total,
                             uint32_t n_jobs, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (total == 0) return FLASHE_OK;
    if (C < 0 || (C && (!loc || !k)) || !out) return fail(ctx, FLASHE_EINVAL, "bad arguments");
    std::vector<Tmp> dl(C);
    std::vector<const uint32_t *> ptrs(C);
    bool sorted = true;            // the reference's lists are (jzf_aggregator.py:598); then the one-pass span reduce applies
    for (int c = 0; c < C; c++) {
        for (uint64_t q = 0; q < k[c]; q++) {
            if (loc[c][q] >= total) return fail(ctx, FLASHE_EINVAL, "client %d location out of range", c);
            if (q && loc[c][q] <= loc[c][q - 1]) sorted = false;
        }
        HIP_TRY(ctx, dl[c].alloc(ctx, static_cast<size_t>(k[c]) * 4));
        if (k[c]) H2D(dl[c].p, loc[c], static_cast<size_t>(k[c]) * 4);
        ptrs[c] = dl[c].as<uint32_t>();
    }
    Tmp dout;
    HIP_TRY(ctx, dout.alloc(ctx, vec_bytes(ctx, total)));
    int rc = sparse_minus_mask_impl(ctx, iter, C, ptrs.data(), k, total, n_jobs, sorted, nullptr, dout.as<uint64_t>());
    if (rc) return rc;
    D2H(out, dout.p, vec_bytes(ctx, total));
    return FLASHE_OK;
}

int flashe_sparse_dense_mask(flashe_ctx *ctx, uint32_t iter, int n_lists, const uint8_t *const *sel, uint64_t total, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (total == 0) return FLASHE_OK;
    if (n_lists < 0 || (n_lists && !sel) || !out) return fail(ctx, FLASHE_EINVAL, "bad arguments");
    std::vector<Tmp> ds(n_lists);
    std::vector<const uint8_t *> ptrs(n_lists);
    for (int i = 0; i < n_lists; i++) {
        HIP_TRY(ctx, ds[i].alloc(ctx, static_cast<size_t>(total)));
        H2D(ds[i].p, sel[i], static_cast<size_t>(total));
        ptrs[i] = ds[i].as<uint8_t>();
    }
    Tmp dout;
    HIP_TRY(ctx, dout.alloc(ctx, vec_bytes(ctx, total)));
    int rc = flashe_sparse_dense_mask_dev(ctx, iter, n_lists, ptrs.data(), total, dout.as<uint64_t>());
    if (rc) return rc;
    D2H(out, dout.p, vec_bytes(ctx, total));
    return FLASHE_OK;
}

int flashe_quantize(flashe_ctx *ctx, uint64_t n, const void *x, int x_is_f64, double alpha, int element_bits, const double *u,
                    uint64_t *q)
{
    CHECK_CTX(ctx);
    if (n == 0) return FLASHE_OK;
    if (!x || !u || !q) return fail(ctx, FLASHE_EINVAL, "null vector");
    const size_t xb = static_cast<size_t>(n) * (x_is_f64 ? 8 : 4);
    Tmp dx, du, dq;
    HIP_TRY(ctx, dx.alloc(ctx, xb));
    HIP_TRY(ctx, du.alloc(ctx, static_cast<size_t>(n) * 8));
    HIP_TRY(ctx, dq.alloc(ctx, static_cast<size_t>(n) * 8));
    H2D(dx.p, x, xb);
    H2D(du.p, u, static_cast<size_t>(n) * 8);
    int rc = flashe_quantize_dev(ctx, n, dx.p, x_is_f64, alpha, element_bits, du.as<double>(), dq.as<uint64_t>());
    if (rc) return rc;
    D2H(q, dq.p, static_cast<size_t>(n) * 8);
    return FLASHE_OK;
}

int flashe_unquantize(flashe_ctx *ctx, uint64_t n, const uint64_t *v, int v_limbs, double alpha, int element_bits, int num_clients,
                      double *out)
{
    CHECK_CTX(ctx);
    if (n == 0) return FLASHE_OK;
    if (!v || !out) return fail(ctx, FLASHE_EINVAL, "null vector");
    if (v_limbs != 1 && v_limbs != 2) return fail(ctx, FLASHE_EINVAL, "v_limbs must be 1 or 2");
    Tmp dv, dout;
    HIP_TRY(ctx, dv.alloc(ctx, static_cast<size_t>(n) * v_limbs * 8));
    HIP_TRY(ctx, dout.alloc(ctx, static_cast<size_t>(n) * 8));
    H2D(dv.p, v, static_cast<size_t>(n) * v_limbs * 8);
    int rc = flashe_unquantize_dev(ctx, n, dv.as<uint64_t>(), v_limbs, alpha, element_bits, num_clients, dout.as<double>());
    if (rc) return rc;
    D2H(out, dout.p, static_cast<size_t>(n) * 8);
    return FLASHE_OK;
}

int flashe_batch(flashe_ctx *ctx, uint64_t n, const uint64_t *vals, int field_bits, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (n == 0) return FLASHE_OK;
    if (!vals || !out) return fail(ctx, FLASHE_EINVAL, "null vector");
    int rc = check_field_bits(ctx, field_bits);
    if (rc) return rc;
    const uint64_t bs = ctx->int_bits / field_bits, nb = (n + bs - 1) / bs;
    Tmp dv, dout;
    HIP_TRY(ctx, dv.alloc(ctx, static_cast<size_t>(n) * 8));
    HIP_TRY(ctx, dout.alloc(ctx, vec_bytes(ctx, nb)));
    H2D(dv.p, vals, static_cast<size_t>(n) * 8);
    rc = flashe_batch_dev(ctx, n, dv.as<uint64_t>(), field_bits, dout.as<uint64_t>());
    if (rc) return rc;
    D2H(out, dout.p, vec_bytes(ctx, nb));
    return FLASHE_OK;
}

int flashe_unbatch(flashe_ctx *ctx, uint64_t n_batches, const uint64_t *in, int field_bits, uint64_t *out)
{
    CHECK_CTX(ctx);
    if (n_batches == 0) return FLASHE_OK;
    if (!in || !out) return fail(ctx, FLASHE_EINVAL, "null vector");
    int rc = check_field_bits(ctx, field_bits);
    if (rc) return rc;
    const uint64_t bs = ctx->int_bits / field_bits;
    Tmp di, dout;
    HIP_TRY(ctx, di.alloc(ctx, vec_bytes(ctx, n_batches)));
    HIP_TRY(ctx, dout.alloc(ctx, static_cast<size_t>(n_batches * bs) * 8));
    H2D(di.p, in, vec_bytes(ctx, n_batches));
    rc = flashe_unbatch_dev(ctx, n_batches, di.as<uint64_t>(), field_bits, dout.as<uint64_t>());
    if (rc) return rc;
    D2H(out, dout.p, static_cast<size_t>(n_batches * bs) * 8);
    return FLASHE_OK;
}

int flashe_sparsify(flashe_ctx *ctx, uint64_t n, uint64_t k, const void *x, int x_is_f64, void *residual, uint32_t *loc, void *vals)
{
    CHECK_CTX(ctx);
    if (n == 0 || k == 0) return FLASHE_OK;
    if (!x || !loc || !vals) return fail(ctx, FLASHE_EINVAL, "null vector");
    const size_t es = x_is_f64 ? 8 : 4;
    Tmp dx, dr, dl, dv;
    HIP_TRY(ctx, dx.alloc(ctx, n * es));
    HIP_TRY(ctx, dl.alloc(ctx, k * 4));
    HIP_TRY(ctx, dv.alloc(ctx, k * es));
    H2D(dx.p, x, n * es);
    if (residual) { HIP_TRY(ctx, dr.alloc(ctx, n * es)); H2D(dr.p, residual, n * es); }
    int rc = flashe_sparsify_dev(ctx, n, k, dx.p, x_is_f64, residual ? dr.p : nullptr, dl.as<uint32_t>(), dv.p);
    if (rc) return rc;
    if (residual) HIP_TRY(ctx, hipMemcpyAsync(residual, dr.p, n * es, hipMemcpyDeviceToHost, ctx->env.stream));
    HIP_TRY(ctx, hipMemcpyAsync(loc, dl.p, k * 4, hipMemcpyDeviceToHost, ctx->env.stream));
    D2H(vals, dv.p, k * es);
    return FLASHE_OK;
}

int flashe_sparsify_batch(flashe_ctx *ctx, int n_layers, const uint64_t *n, const uint64_t *k, const void *x, int x_is_f64, void *residual,
                          uint32_t *loc, void *vals)
{
    CHECK_CTX(ctx);
    if (n_layers < 0 || (n_layers && (!n || !k))) return fail(ctx, FLASHE_EINVAL, "sparsify_batch: bad arguments");
    uint64_t total = 0, total_k = 0;
    for (int l = 0; l < n_layers; l++) { total += n[l]; total_k += k[l]; }
    if (total == 0 || total_k == 0) return FLASHE_OK;
    if (!x || !loc || !vals) return fail(ctx, FLASHE_EINVAL, "null vector");
    const size_t es = x_is_f64 ? 8 : 4;
    Tmp dx, dr, dl, dv;
    HIP_TRY(ctx, dx.alloc(ctx, total * es));
    HIP_TRY(ctx, dl.alloc(ctx, total_k * 4));
    HIP_TRY(ctx, dv.alloc(ctx, total_k * es));
    H2D(dx.p, x, total * es);
    if (residual) { HIP_TRY(ctx, dr.alloc(ctx, total * es)); H2D(dr.p, residual, total * es); }
    int rc = flashe_sparsify_batch_dev(ctx, n_layers, n, k, dx.p, x_is_f64, residual ? dr.p : nullptr, dl.as<uint32_t>(), dv.p);
    if (rc) return rc;
    if (residual) HIP_TRY(ctx, hipMemcpyAsync(residual, dr.p, total * es, hipMemcpyDeviceToHost, ctx->env.stream));
    HIP_TRY(ctx, hipMemcpyAsync(loc, dl.p, total_k * 4, hipMemcpyDeviceToHost, ctx->env.stream));
    D2H(vals, dv.p, total_k * es);
    return FLASHE_OK;
}

#ifdef FLASHE_TUNING
// tuning build only: phase cycle sums of span_prf_kernel's workgroup 0 (FLASHE_SPAN_PROBE=9), tests/perf/sparse_phases.py
int flashe_tune_span_prf_cycles(flashe_ctx *ctx, unsigned long long *out8, int reset)
{
    CHECK_CTX(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream));
    HIP_TRY(ctx, flashe::span_prf_cycles(out8, reset != 0));
    return FLASHE_OK;
}
// tuning build only (not in include/flashe.h): the two-workgroups-per-CU experiment, tests/perf/ab_reduce_2wg.py
int flashe_tune_reduce_decrypt_probe(flashe_ctx *ctx, int variant, uint32_t iter, uint32_t add_idx, uint32_t minus_idx, int C,
                                     const uint64_t *const *cts_dev, uint64_t n, uint64_t *out_dev)
{
    CHECK_CTX(ctx);
    if (!cts_dev || !out_dev) return fail(ctx, FLASHE_EINVAL, "null vector");
    HIP_TRY(ctx, launch_reduce_decrypt_probe(ctx->env, variant, iter, add_idx, minus_idx, C, cts_dev, n, out_dev));
    return FLASHE_OK;
}
#endif

}  // extern "C"
