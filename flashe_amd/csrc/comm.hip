// Multi-GPU exchange of the sharded round (include/flashe.h, "multi-GPU"): RCCL over xGMI, one process per GPU.
//
// What the reference does at the application level -- the arbiter gathers every client model, reduces them in Python and
// broadcasts the aggregate (jzf_aggregator.py:292-308, :404-430, :502-508) -- becomes ONE exchange step between the GPUs of a
// node: a reduce-scatter mod 2^b of the per-GPU partial aggregates, then an all-gather of the decrypted slices.  RCCL has no
// 128-bit integer type and its ncclSum does not wrap at 2^b, so the reduce-scatter is built from point-to-point transfers
// (grouped ncclSend / ncclRecv: every GPU pair of an MI355X node has its own xGMI link, so the W - 1 transfers of a rank run
// concurrently instead of as a per-link-bound ring) plus the local mod-add kernel.
//
// librccl.so (570 MB) is loaded on first use with dlopen, so single-GPU users of libflashe_hip.so never pay for it.
#include "ctx.h"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using flashe_host::fail;

namespace {

struct RcclApi {
    void *handle = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;          // optional
    decltype(&ncclGetVersion) GetVersion = nullptr;        // optional
};

RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("FLASHE_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            api.error = dlerror();
        }
        if (!api.handle) return;
        bool ok = true;
        auto sym = [&](const char *name) { void *p = dlsym(api.handle, name); if (!p) { ok = false; api.error = std::string("missing symbol ") + name; } return p; };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) { dlclose(api.handle); api.handle = nullptr; return; }
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.handle, "ncclCommCount"));
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(api.handle, "ncclGetVersion"));
    });
    return api;
}

}  // namespace

struct flashe_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    double *scratch_dev = nullptr;      // two doubles for the host-value all-reduce
};

#define NCCL_TRY(ctx, expr)                                                                                   \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) return fail(ctx, FLASHE_EIO, "%s failed: %s", #expr, rccl().GetErrorString(r_)); \
    } while (0)

static_assert(sizeof(ncclUniqueId) == FLASHE_RCCL_ID_BYTES, "flashe.h must carry the size of ncclUniqueId");

extern "C" {

int flashe_rccl_unique_id(uint8_t id[FLASHE_RCCL_ID_BYTES])
{
    if (!id) return FLASHE_EINVAL;
    RcclApi &api = rccl();
    if (!api.handle) return fail(nullptr, FLASHE_ENODEV, "librccl.so could not be loaded: %s", api.error.c_str());
    ncclUniqueId u;
    const ncclResult_t r = api.GetUniqueId(&u);
    if (r != ncclSuccess) return fail(nullptr, FLASHE_EIO, "ncclGetUniqueId: %s", api.GetErrorString(r));
    memcpy(id, &u, sizeof u);
    return FLASHE_OK;
}

int flashe_rccl_init(flashe_ctx *ctx, const uint8_t id[FLASHE_RCCL_ID_BYTES], int rank, int world, flashe_comm **out)
{
    CHECK_CTX(ctx);
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(ctx, FLASHE_EINVAL, "flashe_rccl_init: bad arguments (rank %d of %d)", rank, world);
    *out = nullptr;
    RcclApi &api = rccl();
    if (!api.handle) return fail(ctx, FLASHE_ENODEV, "librccl.so could not be loaded: %s", api.error.c_str());
    flashe_comm *c = new (std::nothrow) flashe_comm();
    if (!c) return fail(ctx, FLASHE_ENOMEM, "out of host memory");
    c->rank = rank; c->world = world; c->device = ctx->device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t r = api.CommInitRank(&c->comm, world, u, rank);          // collective: returns once every rank has joined
    if (r != ncclSuccess) { delete c; return fail(ctx, FLASHE_EIO, "ncclCommInitRank(rank %d of %d): %s", rank, world, api.GetErrorString(r)); }
    if (hipMalloc(&c->scratch_dev, 64) != hipSuccess) { (void)api.CommDestroy(c->comm); delete c; return fail(ctx, FLASHE_ENOMEM, "hipMalloc(comm scratch)"); }
    *out = c;
    return FLASHE_OK;
}

int flashe_rccl_destroy(flashe_comm *comm)
{
    if (!comm) return FLASHE_OK;
    (void)hipSetDevice(comm->device);
    if (comm->scratch_dev) (void)hipFree(comm->scratch_dev);
    if (comm->comm) (void)rccl().CommDestroy(comm->comm);
    delete comm;
    return FLASHE_OK;
}

int flashe_rccl_version(int *version)
{
    if (!version) return FLASHE_EINVAL;
    *version = 0;
    RcclApi &api = rccl();
    if (!api.handle) return fail(nullptr, FLASHE_ENODEV, "librccl.so could not be loaded: %s", api.error.c_str());
    if (!api.GetVersion || api.GetVersion(version) != ncclSuccess) return fail(nullptr, FLASHE_EIO, "ncclGetVersion failed");
    return FLASHE_OK;
}

int flashe_rccl_rank(const flashe_comm *comm) { return comm ? comm->rank : FLASHE_EINVAL; }
// What RCCL itself says the communicator spans (ncclCommCount), so that a caller can prove the group really has `world` ranks;
// the value passed to flashe_rccl_init if this RCCL lacks the query.
int flashe_rccl_world(const flashe_comm *comm)
{
    if (!comm) return FLASHE_EINVAL;
    int count = 0;
    if (comm->comm && rccl().CommCount && rccl().CommCount(comm->comm, &count) == ncclSuccess) return count;
    return comm->world;
}

// Piece p of `send` (at send + p * send_stride) goes to rank p; the piece received from rank p lands at recv + p * recv_stride.
int flashe_rccl_all_to_all(flashe_ctx *ctx, flashe_comm *comm, const void *send_dev, size_t send_stride, void *recv_dev, size_t recv_stride,
                           size_t bytes)
{
    CHECK_CTX(ctx);
    if (!comm || (bytes && (!send_dev || !recv_dev))) return fail(ctx, FLASHE_EINVAL, "flashe_rccl_all_to_all: null argument");
    if (bytes == 0) return FLASHE_OK;
    if (send_stride < bytes || recv_stride < bytes) return fail(ctx, FLASHE_EINVAL, "flashe_rccl_all_to_all: stride smaller than the piece");
    RcclApi &api = rccl();
    const char *s = static_cast<const char *>(send_dev);
    char *r = static_cast<char *>(recv_dev);
    // the rank's own piece never touches the fabric (FLASHE_RCCL_SELF_SENDRECV=1 sends it through RCCL as well: lets a
    // one-GPU box exercise the grouped send / recv path)
    static const bool self_through_rccl = [] { const char *e = getenv("FLASHE_RCCL_SELF_SENDRECV"); return e && atoi(e) != 0; }();
    if (!self_through_rccl) {
        HIP_TRY(ctx, hipMemcpyAsync(r + comm->rank * recv_stride, s + comm->rank * send_stride, bytes, hipMemcpyDeviceToDevice, ctx->env.stream));
        if (comm->world == 1) return FLASHE_OK;
    }
    NCCL_TRY(ctx, api.GroupStart());
    // A Send / Recv that fails inside the group must not leave the group open (every later RCCL call of this thread would be queued
    // into it and never issued): close it first, then report the FIRST failure.
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    for (int step = self_through_rccl ? 0 : 1; step < comm->world && first == ncclSuccess; step++) {
        // rank r sends to r + step and receives from r - step: every pair (and xGMI link) is used once per step
        const int to = (comm->rank + step) % comm->world, from = (comm->rank - step + comm->world) % comm->world;
        first = api.Send(s + to * send_stride, bytes, ncclUint8, to, comm->comm, ctx->env.stream);
        if (first != ncclSuccess) { what = "ncclSend"; break; }
        first = api.Recv(r + from * recv_stride, bytes, ncclUint8, from, comm->comm, ctx->env.stream);
        if (first != ncclSuccess) what = "ncclRecv";
    }
    const ncclResult_t end = api.GroupEnd();
    if (first != ncclSuccess) return fail(ctx, FLASHE_EIO, "flashe_rccl_all_to_all: %s failed inside the group (closed): %s", what, api.GetErrorString(first));
    if (end != ncclSuccess) return fail(ctx, FLASHE_EIO, "flashe_rccl_all_to_all: ncclGroupEnd: %s", api.GetErrorString(end));
    return FLASHE_OK;
}

int flashe_rccl_all_gather(flashe_ctx *ctx, flashe_comm *comm, const void *send_dev, void *recv_dev, size_t bytes)
{
    CHECK_CTX(ctx);
    if (!comm || (bytes && (!send_dev || !recv_dev))) return fail(ctx, FLASHE_EINVAL, "flashe_rccl_all_gather: null argument");
    if (bytes == 0) return FLASHE_OK;
    NCCL_TRY(ctx, rccl().AllGather(send_dev, recv_dev, bytes, ncclUint8, comm->comm, ctx->env.stream));
    return FLASHE_OK;
}

// The mod-2^b reduce-scatter RCCL cannot express (no 128-bit type, no wrap at 2^b): all-to-all of slices into recv_dev
// (world x slice_elems elements of scratch), then out_slice = sum of the received slices (+ extra_dev, e.g. the decrypt
// mask difference, which turns the reduced slice into the decrypted slice) on the ctx stream.
int flashe_rccl_reduce_scatter_modadd(flashe_ctx *ctx, flashe_comm *comm, const uint64_t *partial_dev, uint64_t slice_elems,
                                      uint64_t *recv_dev, const uint64_t *extra_dev, uint64_t *out_slice_dev)
{
    CHECK_CTX(ctx);
    if (!comm || (slice_elems && (!partial_dev || !recv_dev || !out_slice_dev))) return fail(ctx, FLASHE_EINVAL, "flashe_rccl_reduce_scatter_modadd: null argument");
    if (slice_elems == 0) return FLASHE_OK;
    const size_t bytes = static_cast<size_t>(slice_elems) * ctx->limbs * 8;
    if (bytes % 16) return fail(ctx, FLASHE_EINVAL, "slices must be a multiple of 16 bytes (an even number of one-limb elements)");
    int rc = flashe_rccl_all_to_all(ctx, comm, partial_dev, bytes, recv_dev, bytes, bytes);
    if (rc) return rc;
    std::vector<const uint64_t *> ops;
    for (int g = 0; g < comm->world; g++) ops.push_back(recv_dev + static_cast<size_t>(g) * slice_elems * ctx->limbs);
    if (extra_dev) ops.push_back(extra_dev);
    return flashe_aggregate_elem_dev(ctx, static_cast<int>(ops.size()), ops.data(), slice_elems, out_slice_dev);
}

// int_bits <= 64: the cross-GPU mod-add as RCCL's own all-reduce (ncclAllReduce(ncclUint64, ncclSum) wraps mod
// 2^64, the mask to int_bits follows): buf[j] = (sum over ranks of buf[j]) mod 2^b on every rank, in place, on the ctx stream.
int flashe_rccl_allreduce_modadd_u64(flashe_ctx *ctx, flashe_comm *comm, uint64_t *buf_dev, uint64_t count)
{
    CHECK_CTX(ctx);
    if (!comm || (count && !buf_dev)) return fail(ctx, FLASHE_EINVAL, "flashe_rccl_allreduce_modadd_u64: bad arguments");
    if (ctx->limbs != 1) return fail(ctx, FLASHE_EINVAL, "flashe_rccl_allreduce_modadd_u64 needs int_bits <= 64 (wider moduli: flashe_rccl_reduce_scatter_modadd)");
    if (count == 0) return FLASHE_OK;
    // (one rank: nothing to add; FLASHE_RCCL_SELF_SENDRECV=1 makes the call anyway, so that a one-GPU box exercises the real collective)
    static const bool through_rccl = [] { const char *e = getenv("FLASHE_RCCL_SELF_SENDRECV"); return e && atoi(e) != 0; }();
    if (comm->world > 1 || through_rccl) NCCL_TRY(ctx, rccl().AllReduce(buf_dev, buf_dev, count, ncclUint64, ncclSum, comm->comm, ctx->env.stream));
    if (ctx->int_bits < 64) return flashe_combine_dev(ctx, count, buf_dev, 1, nullptr, nullptr, buf_dev);      // & (2^b - 1)
    return FLASHE_OK;
}

// Host-value all-reduce (timing and agreement between ranks): op 0 = max, 1 = min, 2 = sum.  Synchronous.
int flashe_rccl_allreduce_f64(flashe_ctx *ctx, flashe_comm *comm, double *value, int op)
{
    CHECK_CTX(ctx);
    if (!comm || !value || op < 0 || op > 2) return fail(ctx, FLASHE_EINVAL, "flashe_rccl_allreduce_f64: bad arguments");
    if (comm->world == 1) { HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream)); return FLASHE_OK; }
    HIP_TRY(ctx, hipMemcpyAsync(comm->scratch_dev, value, sizeof(double), hipMemcpyHostToDevice, ctx->env.stream));
    const ncclRedOp_t rop = op == 0 ? ncclMax : op == 1 ? ncclMin : ncclSum;
    NCCL_TRY(ctx, rccl().AllReduce(comm->scratch_dev, comm->scratch_dev + 1, 1, ncclDouble, rop, comm->comm, ctx->env.stream));
    HIP_TRY(ctx, hipMemcpyAsync(value, comm->scratch_dev + 1, sizeof(double), hipMemcpyDeviceToHost, ctx->env.stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->env.stream));
    return FLASHE_OK;
}

int flashe_rccl_barrier(flashe_ctx *ctx, flashe_comm *comm)
{
    double one = 1.0;
    return flashe_rccl_allreduce_f64(ctx, comm, &one, 2);
}

}  // extern "C"
