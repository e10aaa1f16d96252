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
    void trim_locked()
    {
        for (const Parked &b : parked_) { be_->wipe(b.p, b.cap); (void)be_->release(b.p); }
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

}  // namespace flashe_pool
