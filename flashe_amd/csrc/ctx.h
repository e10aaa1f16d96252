// Private to the host side of libflashe_hip.so (abi.hip, comm.hip): the context object behind the opaque flashe_ctx of
// include/flashe.h, and the error-reporting helpers every entry point uses.
#pragma once
#include "flashe.h"
#include "kernels.h"
#include "blockpool.h"

#include <string>
#include <vector>

struct flashe_ctx {
    int device = 0;
    int device_cus = 0;       // compute units of the device (env.num_cus is what launches may use: flashe_ctx_set_cu_limit)
    int int_bits = 0;
    int limbs = 0;
    bool own_stream = false;
    flashe::LaunchEnv env{};
    uint32_t *te0_dev = nullptr;
    uint32_t *rkw_dev = nullptr;
    uint32_t *rkp_dev = nullptr;
    std::string err;
    // scratch buffers owned by the ctx (grown on demand, reused across calls)
    struct Buf { void *p = nullptr; size_t cap = 0; };
    Buf summaries;    // packed-aggregate block summaries
    Buf stream_tmp;   // whole-vector mask stream for the sparse paths
    Buf acc_tmp[2];   // ping-pong partial sums when a packed reduce has more than kMaxOps operands
    Buf sp_ws;        // sparsifier workspace (select state, histogram, per-block counts)
    Buf bounds;       // span reduce: first entry of every client in every span
    Buf mt_ws;        // flashe_mt19937_random_dev: state in / out and the substream windows
    Buf codec_tab;    // layer table of the fused quantise / unquantise over a flattened model
    // ctx-resident mask precompute (flashe_prepare_encrypt / flashe_prepare_decrypt): the masks of the reference's next_iter_*_prepared
    // caches (jzf_flashe.py:599-666) stay in HBM inside the ctx and are consumed by the next flashe_encrypt_prepared* /
    // flashe_decrypt_prepared* call; the blocks are kept for the next round's masks
    struct Prepared { Buf add, minus; uint64_t n = 0; bool valid = false, has_minus = false; uint32_t iter = 0, add_idx = 0, minus_idx = 0; };
    Prepared prep_enc, prep_dec;
    // staging blocks of the host-pointer twins: hipMalloc / hipFree cost more than the kernels on LeNet-sized vectors and more than
    // the PCIe transfer on 160 MB ones, so blocks are kept and reused within a byte budget (the twins are synchronous: a block is
    // free again when its call returns)
    flashe_pool::StagingPool *staging = nullptr;   // (policy in blockpool.h, so that it can be tested on the host under sanitizers)
    bool capturing = false;   // between flashe_graph_begin and flashe_graph_end
    uint32_t key_epoch = 0;   // bumped by flashe_ctx_set_key: a graph replays the key it was captured with
    hipEvent_t ev_copy[2] = {nullptr, nullptr};   // hand-off events of the pipelined host-pointer twins (created on first use)
    uint32_t *err_flag_host = nullptr;   // host-mapped word the sparse kernels set when they skip an out-of-range location
};

namespace flashe_host {

// Records the message on ctx (or, ctx == NULL, as this thread's context-creation error) and returns `code`.
int fail(flashe_ctx *ctx, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));

}  // namespace flashe_host

#define HIP_TRY(ctx, expr)                                                                           \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return flashe_host::fail(ctx, e_ == hipErrorOutOfMemory ? FLASHE_ENOMEM : FLASHE_EIO, "%s failed: %s", #expr, \
                                     hipGetErrorString(e_));                                         \
    } while (0)

#define CHECK_CTX(ctx)                                                        \
    do {                                                                      \
        if (!(ctx)) return FLASHE_EINVAL;                                     \
        HIP_TRY(ctx, hipSetDevice((ctx)->device));                            \
    } while (0)
