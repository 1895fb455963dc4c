#include "uvc_alloc.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>

namespace {
struct Cache {
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks[16];           // per device: size -> block
    std::unordered_map<void *, std::pair<size_t, int>> live;  // block -> (size, device)
    size_t cached = 0, live_bytes = 0, limit = 0, n_real = 0, n_hit = 0;
};
Cache &cache() { static Cache c; return c; }
size_t round_up(size_t n) {
    if (n < 256) return 256;
    if (n < (1u << 20)) return (n + 4095) & ~(size_t)4095;
    const size_t step = (size_t)2 << 20;                       // 2 MiB granules for large blocks
    return (n + step - 1) / step * step;
}
void release_all_locked(Cache &c, int dev) {
    for (auto &kv : c.free_blocks[dev]) { (void)hipFree(kv.second); c.cached -= kv.first; }
    c.free_blocks[dev].clear();
}
}

extern "C" hipError_t uvc_dev_malloc(void **p, size_t bytes) {
    Cache &c = cache();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipMalloc(p, bytes);
    const size_t want = round_up(bytes);
    {
        std::lock_guard<std::mutex> g(c.mu);
        if (!c.limit) {
            const char *e = getenv("UVCGPU_CACHE_GB");
            size_t fr = 0, tot = 0;
            if (e) c.limit = (size_t)atoll(e) << 30; else if (hipMemGetInfo(&fr, &tot) == hipSuccess) c.limit = tot / 4; else c.limit = (size_t)16 << 30;
            if (!c.limit) c.limit = 1;
        }
        auto it = c.free_blocks[dev].lower_bound(want);
        if (it != c.free_blocks[dev].end() && it->first <= want + want / 2 + ((size_t)8 << 20)) {   // close enough a fit
            *p = it->second; c.live[*p] = { it->first, dev }; c.cached -= it->first; c.live_bytes += it->first; c.n_hit++;
            c.free_blocks[dev].erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {   // give the cached blocks back and try once more
        (void)hipGetLastError();
        { std::lock_guard<std::mutex> g(c.mu); release_all_locked(c, dev); }
        e = hipMalloc(p, want);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> g(c.mu);
    c.live[*p] = { want, dev }; c.live_bytes += want; c.n_real++;
    return hipSuccess;
}

extern "C" hipError_t uvc_dev_free(void *p) {
    if (!p) return hipSuccess;
    Cache &c = cache();
    size_t sz = 0; int dev = 0; bool keep = false;
    {
        std::lock_guard<std::mutex> g(c.mu);
        auto it = c.live.find(p);
        if (it == c.live.end()) { /* not ours (allocated before the cache existed) */ }
        else {
            sz = it->second.first; dev = it->second.second; c.live.erase(it); c.live_bytes -= sz;
            if (c.cached + sz <= c.limit) { c.free_blocks[dev].emplace(sz, p); c.cached += sz; keep = true; }
        }
    }
    return keep ? hipSuccess : hipFree(p);
}

extern "C" void uvc_dev_cache_stats(size_t *cached_bytes, size_t *live_bytes, size_t *real_mallocs, size_t *cache_hits) {
    Cache &c = cache();
    std::lock_guard<std::mutex> g(c.mu);
    if (cached_bytes) *cached_bytes = c.cached;
    if (live_bytes) *live_bytes = c.live_bytes;
    if (real_mallocs) *real_mallocs = c.n_real;
    if (cache_hits) *cache_hits = c.n_hit;
}
