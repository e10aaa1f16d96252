// Host-side bookkeeping of device memory blocks in libflashe_hip.so, kept free of HIP types: the policy code (size classes, byte
// budgets, eviction, the deferred synchronisation that makes reuse safe) compiles with a mock backend under ASan / UBSan on a machine
// without a GPU (tests/host_blockpool_check.cpp, tools/asan_cpu.sh) -- the one place round 2 found a real bug was a pool's eviction loop.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <unordered_map>
#include <vector>

namespace flashe_pool {

// What the pools need from the device runtime.  Every call returns 0 or a (backend-defined) non-zero error code.
struct Backend {
    virtual ~Backend() {}
    virtual int alloc(void **p, size_t bytes) = 0;
    virtual int release(void *p) = 0;             // may synchronise the device (hipFree does)
    virtual int sync_all() = 0;                    // every stream of the device idle
    virtual void wipe(void *p, size_t bytes) = 0;  // zero a block that held plaintexts / ciphertexts before it leaves the process
};

// Small blocks in powers of two from 4 KiB, large ones in 2-MiB steps (a 160 MB vector must not hold 256 MB).
inline size_t size_class(size_t bytes)
{
    if (bytes > (static_cast<size_t>(8) << 20)) {
        const size_t step = static_cast<size_t>(2) << 20;
        return (bytes + step - 1) / step * step;
    }
    size_t cap = 4096;
    while (cap < bytes) cap <<= 1;
    return cap;
}

// Caching allocator behind flashe_dev_alloc / flashe_dev_free, one per device.
//
// Why: a hipMalloc + hipFree pair of a 160 MB block costs ~7 ms on this platform -- more than moving the block over PCIe and 25x the
// kernel that fills it -- and the drop-in API's device-resident results (DeviceVector) are allocated and dropped every round.
// Freed blocks are parked (within a byte budget, oldest evicted first) and handed out again for requests of the same size class.
//
// Safety of reuse: hipFree synchronises the device, so a caller may free a block while kernels of ANOTHER stream still read it.  A
// parked block keeps that guarantee lazily: it remembers the allocator's epoch at the time it was parked, every device-wide
// synchronisation the allocator performs starts a new epoch, and a block is handed out again only after such a synchronisation has
// happened since it was parked (one synchronisation covers every block parked before it).
class DeviceCache {
public:
    DeviceCache(Backend *backend, size_t budget_bytes) : be_(backend), budget_(budget_bytes) {}

    int alloc(size_t bytes, void **out)
    {
        *out = nullptr;
        const size_t cap = size_class(bytes ? bytes : 16);
        std::lock_guard<std::mutex> lock(mu_);
        int best = -1;
        for (size_t i = 0; i < parked_.size(); i++)
            if (parked_[i].cap == cap && (best < 0 || parked_[i].epoch < parked_[best].epoch)) best = static_cast<int>(i);
        if (best >= 0) {
            const Parked b = parked_[best];
            if (b.epoch == epoch_) {                       // nothing has synchronised the device since it was parked
                const int rc = be_->sync_all();
                if (rc) return rc;
                epoch_++;
            }
            parked_.erase(parked_.begin() + best);
            held_ -= cap;
            live_[b.p] = cap;
            *out = b.p;
            hits_++;
            return 0;
        }
        void *p = nullptr;
        int rc = be_->alloc(&p, cap);
        if (rc) {                                           // out of memory: give everything parked back and try once more
            trim_locked();
            rc = be_->alloc(&p, cap);
            if (rc) return rc;
        }
        live_[p] = cap;
        *out = p;
        misses_++;
        return 0;
    }

    int release(void *p)
    {
        if (!p) return 0;
        std::lock_guard<std::mutex> lock(mu_);
        auto it = live_.find(p);
        if (it == live_.end()) return be_->release(p);      // not ours (allocated before the cache existed, or by the caller)
        const size_t cap = it->second;
        live_.erase(it);
        if (cap > budget_) return be_->release(p);
        while (held_ + cap > budget_ && !parked_.empty()) {
            size_t oldest = 0;
            for (size_t i = 1; i < parked_.size(); i++)
                if (parked_[i].epoch < parked_[oldest].epoch) oldest = i;
            const Parked v = parked_[oldest];
            parked_.erase(parked_.begin() + oldest);
            held_ -= v.cap;
            settle_before_wipe(v);
            be_->wipe(v.p, v.cap);
            const int rc = be_->release(v.p);
            if (rc) { (void)be_->release(p); return rc; }
        }
        if (held_ + cap > budget_) return be_->release(p);
        parked_.push_back(Parked{p, cap, epoch_});
        held_ += cap;
        return 0;
    }

    void trim()
    {
        std::lock_guard<std::mutex> lock(mu_);
        trim_locked();
    }

    size_t held_bytes() const { std::lock_guard<std::mutex> lock(mu_); return held_; }
    size_t live_blocks() const { std::lock_guard<std::mutex> lock(mu_); return live_.size(); }
    size_t parked_blocks() const { std::lock_guard<std::mutex> lock(mu_); return parked_.size(); }
    uint64_t hits() const { std::lock_guard<std::mutex> lock(mu_); return hits_; }
    uint64_t misses() const { std::lock_guard<std::mutex> lock(mu_); return misses_; }
    void set_budget(size_t bytes) { std::lock_guard<std::mutex> lock(mu_); budget_ = bytes; }

private:
    struct Parked { void *p; size_t cap; uint64_t epoch; };
    // A block parked in the CURRENT epoch may still be written by a kernel in flight (callers drop a buffer right after enqueueing
    // the work that uses it, and the ctx streams are non-blocking: the null-stream memset of wipe() does not wait for them).  Wiping
    // it now would leave plaintext / ciphertext residue behind the wipe: synchronise the device first, which also opens a new epoch.
    // (The synchronisation is device-wide, like the one hipFree implies: a stream capture running on ANOTHER ctx of the device is
    // invalidated by it -- capture sequences are run once beforehand so that no allocation, hence no eviction, happens inside one.)
    void settle_before_wipe(const Parked &v)
    {
        if (v.epoch == epoch_) {
            (void)be_->sync_all();
            epoch_++;
        }
    }
    void trim_locked()
    {
        for (const Parked &b : parked_) { settle_before_wipe(b); be_->wipe(b.p, b.cap); (void)be_->release(b.p); }
        parked_.clear();
        held_ = 0;
    }
    Backend *be_;
    size_t budget_;
    mutable std::mutex mu_;
    std::unordered_map<void *, size_t> live_;
    std::vector<Parked> parked_;
    size_t held_ = 0;
    uint64_t epoch_ = 0, hits_ = 0, misses_ = 0;
};

// Staging blocks of the synchronous host-pointer twins, one pool per ctx (not thread-safe, like the ctx): a call leases the device
// blocks it stages its operands in and gives them back when it returns.  Blocks are kept between calls within a byte budget --
// a hipMalloc + hipFree pair per 160 MB block cost 7 ms against 2.9 ms for the transfer itself (tests/perf/e2e_calls.py) -- the
// smallest free block that fits is taken; when a new block is needed and the budget or the slot table is full, parked blocks are
// dropped largest first; a request that still does not fit (or exceeds the whole budget) is a plain allocation, slot -1, which the
// caller releases through the backend.
class StagingPool {
public:
    StagingPool(Backend *backend, size_t budget_bytes, size_t max_blocks = 64) : be_(backend), budget_(budget_bytes), max_blocks_(max_blocks) {}

    // *p = a device block of at least `bytes`; *slot = its pool slot, or -1 for a plain allocation.  Returns the backend's error code.
    int lease(size_t bytes, void **p, int *slot)
    {
        if (bytes == 0) bytes = 16;
        *p = nullptr; *slot = -1;
        int best = -1;
        for (size_t i = 0; i < blocks_.size(); i++)
            if (!blocks_[i].used && blocks_[i].cap >= bytes && (best < 0 || blocks_[i].cap < blocks_[best].cap)) best = static_cast<int>(i);
        if (best < 0) {
            const size_t cap = size_class(bytes);
            size_t held = 0;
            for (const Block &b : blocks_) held += b.cap;
            auto no_slot = [&] {
                if (blocks_.size() < max_blocks_) return false;
                for (const Block &b : blocks_)
                    if (!b.used && b.cap == 0) return false;
                return true;
            };
            // make room: drop parked blocks, largest first (they are all too small for this request); a request beyond the whole
            // budget is a plain allocation and evicts nothing
            while (cap <= budget_ && (held + cap > budget_ || no_slot())) {
                int victim = -1;
                for (size_t i = 0; i < blocks_.size(); i++)
                    if (!blocks_[i].used && blocks_[i].cap && (victim < 0 || blocks_[i].cap > blocks_[victim].cap)) victim = static_cast<int>(i);
                if (victim < 0) break;                   // nothing parked is left to give up
                // live leases hold slot numbers: only the LAST slot can be removed without renumbering; any other victim is freed in
                // place and its slot kept as an empty one
                be_->wipe(blocks_[victim].p, blocks_[victim].cap);
                (void)be_->release(blocks_[victim].p);
                held -= blocks_[victim].cap;
                if (static_cast<size_t>(victim) + 1 == blocks_.size()) blocks_.pop_back();
                else { blocks_[victim].p = nullptr; blocks_[victim].cap = 0; }
            }
            if (held + cap <= budget_) {
                int slot_new = -1;
                for (size_t i = 0; i < blocks_.size(); i++)
                    if (!blocks_[i].used && blocks_[i].cap == 0) { slot_new = static_cast<int>(i); break; }
                if (slot_new >= 0 || blocks_.size() < max_blocks_) {
                    void *q = nullptr;
                    const int rc = be_->alloc(&q, cap);
                    if (rc) return rc;
                    if (slot_new >= 0) blocks_[slot_new] = Block{q, cap, false};
                    else { blocks_.push_back(Block{q, cap, false}); slot_new = static_cast<int>(blocks_.size()) - 1; }
                    best = slot_new;
                }
            }
        }
        if (best >= 0) {
            blocks_[best].used = true;
            *p = blocks_[best].p; *slot = best;
            return 0;
        }
        return be_->alloc(p, bytes);
    }

    void give_back(int slot)
    {
        if (slot >= 0 && static_cast<size_t>(slot) < blocks_.size()) blocks_[slot].used = false;
    }

    // every block goes back to the device, wiped (they held plaintexts and ciphertexts)
    void destroy()
    {
        for (Block &b : blocks_)
            if (b.p) { be_->wipe(b.p, b.cap); (void)be_->release(b.p); }
        blocks_.clear();
    }

    size_t held_bytes() const { size_t h = 0; for (const Block &b : blocks_) h += b.cap; return h; }
    size_t slots() const { return blocks_.size(); }
    size_t leased() const { size_t k = 0; for (const Block &b : blocks_) k += b.used ? 1 : 0; return k; }

private:
    struct Block { void *p; size_t cap; bool used; };
    Backend *be_;
    size_t budget_, max_blocks_;
    std::vector<Block> blocks_;
};

}  // namespace flashe_pool
