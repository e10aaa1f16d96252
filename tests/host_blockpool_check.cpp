// Host check of flashe_amd/csrc/blockpool.h (the caching device allocator behind flashe_dev_alloc / flashe_dev_free) with a mock
// backend, meant to be built with -fsanitize=address,undefined: a seeded random workload of allocations and frees against a model,
// checking the invariants the design promises -- a block is never handed out twice, a parked block is reused only after a
// device-wide synchronisation that happened after it was parked, the parked bytes never exceed the budget, evicted / trimmed blocks
// are wiped before they go back, out-of-memory gives parked blocks up and retries, and nothing leaks.
#include "blockpool.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <random>
#include <set>

using namespace flashe_pool;

struct Mock final : Backend {
    std::map<void *, size_t> blocks;      // everything the "device" currently has allocated
    std::set<void *> wiped;
    size_t capacity, used = 0;
    uint64_t syncs = 0, allocs = 0, frees = 0;
    // sync count at the moment the model saw a block parked: a wipe of that block must come AFTER a later synchronisation (a kernel of a
    // non-blocking stream may still be writing a block parked in the current epoch: wiping first would leave residue behind the wipe)
    const std::map<void *, uint64_t> *parked_at = nullptr;
    uint64_t wipes_checked = 0;
    explicit Mock(size_t cap) : capacity(cap) {}
    int alloc(void **p, size_t bytes) override
    {
        if (used + bytes > capacity) return 2;                        // "hipErrorOutOfMemory"
        *p = malloc(real(bytes));               // (the "device" backs only the first page of a block with real memory: the
        memset(*p, 0xCD, real(bytes));          // policy sees the full sizes, the sanitizer still guards what the callers touch)
        blocks[*p] = bytes; used += bytes; allocs++;
        return 0;
    }
    int release(void *p) override
    {
        auto it = blocks.find(p);
        if (it == blocks.end()) { fprintf(stderr, "release of an unknown block\n"); abort(); }
        used -= it->second; blocks.erase(it); wiped.erase(p); free(p); frees++; syncs++;     // hipFree synchronises
        return 0;
    }
    int sync_all() override { syncs++; return 0; }
    void wipe(void *p, size_t bytes) override
    {
        if (parked_at) {
            auto it = parked_at->find(p);
            if (it != parked_at->end()) {
                if (syncs <= it->second) { fprintf(stderr, "a parked block was wiped before any synchronisation since it was parked\n"); abort(); }
                wipes_checked++;
            }
        }
        memset(p, 0, real(bytes)); wiped.insert(p);
    }
    static size_t real(size_t bytes) { return bytes < 4096 ? (bytes ? bytes : 1) : 4096; }
};

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main()
{
    CHECK(size_class(1) == 4096 && size_class(4097) == 8192 && size_class(8u << 20) == (8u << 20));
    CHECK(size_class((8u << 20) + 1) == (10u << 20) && size_class(160000000) % (2u << 20) == 0 && size_class(160000000) - 160000000 < (2u << 20));
    for (unsigned seed = 1; seed <= 20; seed++) {
        Mock mock(static_cast<size_t>(96) << 20);
        const size_t budget = static_cast<size_t>(seed % 4 == 0 ? 0 : 24) << 20;
        DeviceCache cache(&mock, budget);
        std::mt19937 rng(seed);
        struct Live { void *p; size_t want; };
        std::vector<Live> live;
        std::map<void *, uint64_t> parked_at;        // sync count of the mock when the model saw the block freed
        mock.parked_at = &parked_at;
        const size_t sizes[] = {1, 100, 4096, 5000, 70000, 1u << 20, (3u << 20) + 5, 9u << 20, 17u << 20, 30u << 20};
        for (int step = 0; step < 4000; step++) {
            const int op = rng() % 100;
            if (op < 55 || live.empty()) {
                const size_t want = sizes[rng() % (sizeof sizes / sizeof *sizes)];
                void *p = nullptr;
                const int rc = cache.alloc(want, &p);
                if (rc) { CHECK(rc == 2 && p == nullptr && cache.parked_blocks() == 0); continue; }       // a real OOM: everything parked was given up first
                CHECK(p != nullptr && mock.blocks.count(p) && mock.blocks[p] >= want);
                for (const Live &l : live) CHECK(l.p != p);                                          // never handed out twice
                auto it = parked_at.find(p);
                if (it != parked_at.end()) { CHECK(mock.syncs > it->second); parked_at.erase(it); }  // reused only after a later synchronisation
                memset(p, 0xAB, Mock::real(want));                                                               // the caller writes its block (ASan: in bounds)
                live.push_back(Live{p, want});
            } else if (op < 95) {
                const size_t i = rng() % live.size();
                const uint64_t before = mock.syncs;
                CHECK(cache.release(live[i].p) == 0);
                if (mock.blocks.count(live[i].p)) parked_at[live[i].p] = before;                     // parked (still allocated on the "device")
                live.erase(live.begin() + i);
                // whatever the eviction gave back is gone from the model
                for (auto it = parked_at.begin(); it != parked_at.end();) it = mock.blocks.count(it->first) ? std::next(it) : parked_at.erase(it);
            } else {
                cache.trim();
                CHECK(cache.parked_blocks() == 0 && cache.held_bytes() == 0);
                parked_at.clear();
            }
            CHECK(cache.held_bytes() <= budget);
            CHECK(cache.live_blocks() == live.size());
            CHECK(mock.blocks.size() == live.size() + cache.parked_blocks());
        }
        for (const Live &l : live) CHECK(cache.release(l.p) == 0);
        // a pointer the cache never handed out goes straight to the backend
        void *foreign = nullptr;
        CHECK(mock.alloc(&foreign, 123) == 0 && cache.release(foreign) == 0 && !mock.blocks.count(foreign));
        cache.trim();
        CHECK(mock.blocks.empty() && mock.used == 0 && mock.allocs == mock.frees);
        if (budget) CHECK(cache.hits() > 0);
    }
    // ---- the staging pool of the host-pointer twins: leases of a call are given back when it returns ----
    for (unsigned seed = 1; seed <= 12; seed++) {
        Mock mock(static_cast<size_t>(64) << 20);
        const size_t budget = static_cast<size_t>(seed % 5 == 0 ? 0 : seed % 3 == 0 ? 3 : 20) << 20;
        StagingPool pool(&mock, budget, seed % 4 == 0 ? 4 : 64);
        std::mt19937 rng(1000 + seed);
        const size_t sizes[] = {1, 4096, 70000, 1u << 20, (2u << 20) + 7, 5u << 20, 9u << 20, 13u << 20};
        for (int call = 0; call < 1200; call++) {
            // one "call": up to four leases alive at once, all returned at the end (what Tmp's destructors do)
            struct Lease { void *p; int slot; size_t want; };
            std::vector<Lease> mine;
            const int k = 1 + rng() % 4;
            for (int i = 0; i < k; i++) {
                const size_t want = sizes[rng() % (sizeof sizes / sizeof *sizes)];
                void *p = nullptr; int slot = -2;
                const int rc = pool.lease(want, &p, &slot);
                if (rc) { CHECK(rc == 2); continue; }                                   // the "device" is full: reported, nothing leaked
                CHECK(p && mock.blocks.count(p) && mock.blocks[p] >= want && slot >= -1);
                for (const Lease &l : mine) CHECK(l.p != p && (slot < 0 || l.slot != slot));   // two live leases never share a block
                memset(p, 0x5A, Mock::real(want));
                mine.push_back(Lease{p, slot, want});
            }
            CHECK(pool.leased() == static_cast<size_t>(std::count_if(mine.begin(), mine.end(), [](const Lease &l) { return l.slot >= 0; })));
            for (const Lease &l : mine) {
                if (l.slot >= 0) pool.give_back(l.slot);
                else CHECK(mock.release(l.p) == 0);                                     // plain allocation: the caller frees it
            }
            CHECK(pool.leased() == 0 && pool.held_bytes() <= budget && pool.slots() <= 64);
            CHECK(mock.used == pool.held_bytes());
        }
        pool.destroy();
        CHECK(mock.blocks.empty() && mock.allocs == mock.frees);
    }
    printf("BLOCKPOOL_OK\n");
    return 0;
}
