// uvc_alloc.h -- device allocations of the host code go through a small caching allocator: hipMalloc / hipFree synchronise the whole
// device, which serialises worker threads that stream tiles through their own region handles (uvc1_main.cpp, uvc_amd/pipeline.py).
// A freed block is kept and handed to the next request of a similar size (same device); the cache is bounded (UVCGPU_CACHE_GB, default
// a quarter of the device memory) and is emptied when a real allocation fails.  Callers must not free a block that work in flight still
// uses: hipFree used to wait for the device, the cache does not -- the host code synchronises its own streams before it frees.
#ifndef UVC_ALLOC_H
#define UVC_ALLOC_H
#include <hip/hip_runtime.h>
#include <stddef.h>
extern "C" hipError_t uvc_dev_malloc(void **p, size_t bytes);
extern "C" hipError_t uvc_dev_free(void *p);
extern "C" void uvc_dev_cache_stats(size_t *cached_bytes, size_t *live_bytes, size_t *real_mallocs, size_t *cache_hits);
#endif
