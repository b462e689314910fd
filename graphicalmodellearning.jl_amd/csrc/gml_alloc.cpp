// Device memory of the library goes through a small caching layer.
//
// Why: hipFree hands large blocks back to the driver lazily, and every so often a following hipMalloc of a multi-GB block
// then takes 0.4-1.3 s instead of 0.3 ms (measured on MI355X / ROCm 7: scripts/gpu_alloc_probe.py -- the third
// malloc / free of 10.5 GB in a process: 1 259 ms).  A handle's int8 workspace is 10.5 GB at the headline size, so a caller who
// creates a handle per learn() saw its 0.13 s solve become 0.6 s on some calls.  Freed blocks are therefore kept, per
// device and by exact size (the sizes are functions of the problem shape, so the next handle of the same shape finds
// them), up to a quarter of the device's memory; a failed allocation empties the cache and tries again;
// gml_trim_cache() (include/gml.h) returns everything to the driver.
#include "gml_dev.h"

#include <map>
#include <mutex>
#include <unordered_map>

namespace gml {
namespace {
struct Block {
    size_t bytes;
    int device;
};
struct Cached {
    void *p;
    unsigned long long seq; // order of release: the oldest block goes first when room is needed
};
struct Cache {
    std::mutex mu;
    std::unordered_map<void *, Block> live;                 // blocks handed out
    std::map<int, std::multimap<size_t, Cached>> free_;     // device -> size -> cached blocks
    std::map<int, size_t> cached, limit;                    // bytes cached / allowed per device
    unsigned long long seq = 0;
    long long user_limit = -1;                              // gml_set_cache_limit: bytes per device, 0 = no caching, -1 = a quarter of the device
};
Cache &cache() {
    static Cache *c = new Cache(); // leaked on purpose: the HIP runtime may be gone before static destructors run
    return *c;
}
constexpr size_t kMinCached = (size_t)1 << 20; // smaller blocks are not worth keeping

size_t release_device(Cache &c, int dev) { // caller holds the lock and has the device current
    size_t n = 0;
    for (auto &kv : c.free_[dev]) {
        (void)hipFree(kv.second.p);
        n += kv.first;
    }
    c.free_[dev].clear();
    c.cached[dev] = 0;
    return n;
}
} // namespace

hipError_t dev_malloc_bytes(void **out, size_t bytes) {
    *out = nullptr;
    if (bytes == 0) bytes = 1;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    Cache &c = cache();
    std::lock_guard<std::mutex> lock(c.mu);
    auto &fl = c.free_[dev];
    auto it = fl.find(bytes);
    if (it != fl.end()) {
        *out = it->second.p;
        fl.erase(it);
        c.cached[dev] -= bytes;
        c.live[*out] = {bytes, dev};
        return hipSuccess;
    }
    e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory && c.cached[dev] > 0) {
        (void)hipGetLastError();
        release_device(c, dev);
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) c.live[*out] = {bytes, dev};
    return e;
}

// A cached block is handed to the next caller of the same size on ANY stream or thread without further waiting, so nothing
// may still be using it when it enters the cache.  hipFree gave that guarantee implicitly (it synchronises the device);
// dev_free keeps it by synchronising the block's device before the block is cached.  Callers that release many blocks at
// once (the solver's arena, a handle's workspace) synchronise once themselves and use dev_free_synced.
static hipError_t free_impl(void *p, bool synced) {
    if (!p) return hipSuccess;
    Cache &c = cache();
    Block b;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.live.find(p);
        if (it == c.live.end()) return hipFree(p); // not ours (never happens for library memory)
        b = it->second;
        c.live.erase(it);
    }
    if (b.bytes < kMinCached) return hipFree(p);
    if (!synced) {
        // waited for OUTSIDE the lock: with one host thread per GPU (gml_multi) a drain of this block's device must not stall
        // the other devices' allocations.  Between the two critical sections the block is in neither table: nobody can hand it out.
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != b.device) (void)hipSetDevice(b.device);
        const hipError_t es = hipDeviceSynchronize();
        if (cur != b.device) (void)hipSetDevice(cur);
        if (es != hipSuccess) return hipFree(p); // a failed device: do not recycle anything it may still own
    }
    std::lock_guard<std::mutex> lock(c.mu);
    if (!c.limit.count(b.device)) {
        size_t freeb = 0, total = 0;
        int cur = 0;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(b.device);
        c.limit[b.device] = hipMemGetInfo(&freeb, &total) == hipSuccess ? total / 4 : 0;
        (void)hipSetDevice(cur);
    }
    const size_t lim = c.user_limit >= 0 ? (size_t)c.user_limit : c.limit[b.device];
    if (b.bytes > lim) return hipFree(p);
    // room for the newcomer: the blocks released longest ago go back to the driver first (one-off sizes -- a sort's scratch, a
    // raw upload -- do not squat in the cache)
    auto &fl = c.free_[b.device];
    while (c.cached[b.device] + b.bytes > lim && !fl.empty()) {
        auto oldest = fl.begin();
        for (auto jt = fl.begin(); jt != fl.end(); ++jt)
            if (jt->second.seq < oldest->second.seq) oldest = jt;
        (void)hipFree(oldest->second.p);
        c.cached[b.device] -= oldest->first;
        fl.erase(oldest);
    }
    fl.emplace(b.bytes, Cached{p, c.seq++});
    c.cached[b.device] += b.bytes;
    return hipSuccess;
}

hipError_t dev_free(void *p) { return free_impl(p, false); }
hipError_t dev_free_synced(void *p) { return free_impl(p, true); }

size_t dev_cached_bytes(int device) {
    Cache &c = cache();
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.cached.find(device);
    return it == c.cached.end() ? 0 : it->second;
}

hipError_t dev_mem_info(size_t *freeb, size_t *total) { // free memory as the library sees it: the driver's + its own cache
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipMemGetInfo(freeb, total);
    if (e == hipSuccess) *freeb += dev_cached_bytes(dev);
    return e;
}

size_t dev_trim_cache() {
    Cache &c = cache();
    std::lock_guard<std::mutex> lock(c.mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    size_t n = 0;
    for (auto &kv : c.free_) {
        if (kv.second.empty()) continue;
        (void)hipSetDevice(kv.first);
        n += release_device(c, kv.first);
    }
    (void)hipSetDevice(cur);
    return n;
}

void dev_set_cache_limit(long long bytes) {
    Cache &c = cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        c.user_limit = bytes < 0 ? -1 : bytes;
    }
    if (bytes == 0) (void)dev_trim_cache();
}

} // namespace gml

extern "C" int64_t gml_trim_cache(void) { return (int64_t)gml::dev_trim_cache(); }
extern "C" void gml_set_cache_limit(int64_t bytes_per_device) { gml::dev_set_cache_limit((long long)bytes_per_device); }
