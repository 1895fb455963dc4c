// uvc_inflate.hip -- BGZF blocks inflated on the device (the reader side of the path).
//
// The reference reads its BAM through htslib's bgzf_read -> zlib inflate, one stream per thread (grouping.cpp:617-731).  On the GPU box
// inflating the 650 MB of records behind a 1 Mb x 300x tile costs about one core-second of zlib, more than everything else the host does
// for the tile together, and bounds the files -> VCF chain near 8 M positions/s on 16 cores.  BGZF blocks are independent raw DEFLATE
// streams of <= 64 KiB, about 10 000 per tile:
//
//   one LANE per BGZF block, one wave per workgroup, one workgroup per CU.  The Huffman tables of the 64 decoders of a wave (2.3 KB each,
//   uvc_inflate_core.h) fill the CU's LDS; the decoder is a single loop of bounded steps (one symbol, <= 8 bytes of a match, one header
//   code length), so the lanes of a wave advance together.  Input is read with 4-byte loads, matches are copied 8 bytes at a time.
//
// The kernel keeps 160 of the 256 CUs busy for a full tile and leaves the rest (and every kernel that needs no LDS) to the calling
// pipeline's other streams.  The CRC-32 of each block is checked by the caller on the host copy (uvcio: 22 GB/s per core).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "uvcgpu.h"
#include "uvc_alloc.h"
#include "uvc_inflate_core.h"

extern "C" int uvcgpu_set_error(int code, const char *msg);   // uvc_host.cpp

#define DEV_INLINE __device__ __forceinline__
struct BgzfBlockDev { unsigned long long in_off, out_off; uint32_t in_len, out_len; };

__global__ void __launch_bounds__(64) k_bgzf_inflate(const uint8_t *comp, const BgzfBlockDev *blocks, int n, uint8_t *out, int32_t *status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef __attribute__((address_space(3))) InflState LdsState;   // ds_read / ds_write, not FLAT accesses
    LdsState *S = (LdsState *)lds_raw + threadIdx.x;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const BgzfBlockDev b = blocks[i];
    status[i] = uvc_inflate_block(comp + b.in_off, b.in_len, out + b.out_off, b.out_len, *S);
}

// One WAVE per BGZF block (UVCGPU_INFLATE_WAVE=1): the lane-per-block form above waits for its slowest lane and copies a match 8 bytes per step
// through global memory; here the 64 lanes decode one block together (uvc_inflate_block_t<true>), four blocks per workgroup, one table set
// (2.3 KB of LDS) per wave, a tile's ~8 500 blocks resident at once.
typedef InflStateT<10, 9> InflStateWave;   // one table set per wave: 3.9 KB, 32 waves of a CU = 125 KB of its 160 KB LDS
DEV_INLINE void bgzf_inflate_wave_body(const uint8_t *comp, const BgzfBlockDev *blocks, int n, uint8_t *out, int32_t *status, InflStateWave *lds_state) {
    typedef __attribute__((address_space(3))) InflStateWave LdsState;
    LdsState *S = (LdsState *)lds_state + (threadIdx.x >> 6);
    const int i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (i >= n) return;
    const BgzfBlockDev b = blocks[i];
    const int rc = uvc_inflate_block_t<true, 10, 9>(comp + b.in_off, b.in_len, out + b.out_off, b.out_len, *S, (uint32_t)(threadIdx.x & 63));
    if ((threadIdx.x & 63) == 0) status[i] = rc;
}
__global__ void __launch_bounds__(256) k_bgzf_inflate_wave(const uint8_t *comp, const BgzfBlockDev *blocks, int n, uint8_t *out, int32_t *status) {
    __shared__ __attribute__((aligned(16))) InflStateWave lds_state[4];
    bgzf_inflate_wave_body(comp, blocks, n, out, status, lds_state);
}
// the same with the register budget of eight waves per SIMD (every block of a 1 Mb tile resident at once; UVCGPU_INFLATE_WAVE=8)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_bgzf_inflate_wave8(const uint8_t *comp, const BgzfBlockDev *blocks, int n, uint8_t *out, int32_t *status) {
    __shared__ __attribute__((aligned(16))) InflStateWave lds_state[4];
    bgzf_inflate_wave_body(comp, blocks, n, out, status, lds_state);
}

namespace {
struct DevBuf { void *p = nullptr; size_t cap = 0; };
// per host thread: the staging buffers of the last call, reused (the fetch loop of a worker calls once per batch of blocks)
struct InflateCtx {
    DevBuf comp, out, blocks, status; hipStream_t stream = nullptr; int device = -1; bool attr_set = false;
    ~InflateCtx() { /* device memory goes back with the process: worker threads end with it */ }
};
thread_local InflateCtx g_ctx;
int ensure(DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    if (b.p) uvc_dev_free(b.p);
    b.p = nullptr; b.cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (uvc_dev_malloc(&b.p, want) != hipSuccess) { (void)hipGetLastError(); return -1; }
    b.cap = want;
    return 0;
}
}   // namespace

// Inflates n BGZF payloads (raw DEFLATE) that lie at comp + in_off[i] (in_len[i] bytes) into out + out_off[i] (out_len[i] bytes, the
// block's ISIZE); comp / out are host buffers.  Signature of uvcio_inflate_fn (include/uvcio.h): uvc1-mi355x hands this function to the
// reader.  Returns 0 or an error code (uvcgpu_last_error names the first bad block); the output of a failed call is undefined.
extern "C" int uvcgpu_bgzf_inflate(void *, const uint8_t *comp, int64_t comp_bytes, const int64_t *in_off, const int32_t *in_len, const int64_t *out_off, const int32_t *out_len,
                                   int64_t n, uint8_t *out, int64_t out_bytes) {
    if (n < 0 || comp_bytes < 0 || out_bytes < 0 || (n > 0 && (!comp || !in_off || !in_len || !out_off || !out_len || !out))) return uvcgpu_set_error(UVCGPU_EINVAL, "bgzf_inflate: bad argument");
    if (n == 0) return 0;
    if (n > (int64_t)1 << 24) return uvcgpu_set_error(UVCGPU_EINVAL, "bgzf_inflate: too many blocks in one call");
    try {
        std::vector<BgzfBlockDev> hb((size_t)n);
        for (int64_t i = 0; i < n; i++) {   // the kernel trusts these ranges: check them here
            if (in_off[i] < 0 || in_len[i] < 0 || in_off[i] + in_len[i] > comp_bytes || out_off[i] < 0 || out_len[i] < 0 || out_off[i] + out_len[i] > out_bytes)
                return uvcgpu_set_error(UVCGPU_EINVAL, "bgzf_inflate: a block lies outside its buffer");
            hb[(size_t)i] = BgzfBlockDev{ (unsigned long long)in_off[i], (unsigned long long)out_off[i], (uint32_t)in_len[i], (uint32_t)out_len[i] };
        }
        // The outputs must tile one span of `out` in block order (the reader's batches do): the span travels back as one copy, so a hole in
        // it would be overwritten with stale device bytes, and overlapping outputs would be two waves writing the same bytes.
        int64_t lo = out_off[0], hi = out_off[0];
        for (int64_t i = 0; i < n; i++) {
            if (out_off[i] != hi) return uvcgpu_set_error(UVCGPU_EINVAL, "bgzf_inflate: the outputs of the blocks must be contiguous and in order");
            hi += out_len[i];
        }
        if (hi <= lo) return 0;
        InflateCtx &C = g_ctx;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return uvcgpu_set_error(UVCGPU_EDEVICE, "bgzf_inflate: no device (uvcgpu_init on this thread first)");
        if (C.device != dev) {   // another device on this thread: the staging buffers and the stream of the old one go back first
            if (C.device >= 0) {
                (void)hipSetDevice(C.device);
                if (C.stream) { (void)hipStreamSynchronize(C.stream); (void)hipStreamDestroy(C.stream); }
                for (DevBuf *b : { &C.comp, &C.out, &C.blocks, &C.status }) if (b->p) uvc_dev_free(b->p);
                (void)hipSetDevice(dev);
            }
            C = InflateCtx(); C.device = dev;
        }
        if (!C.stream && hipStreamCreateWithFlags(&C.stream, hipStreamNonBlocking) != hipSuccess) return uvcgpu_set_error(UVCGPU_EDEVICE, "bgzf_inflate: hipStreamCreate");
        const size_t lds = sizeof(InflState) * 64;
        if (!C.attr_set) {
            if (hipFuncSetAttribute((const void *)k_bgzf_inflate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return uvcgpu_set_error(UVCGPU_EDEVICE, "bgzf_inflate: the device does not give one workgroup the LDS of 64 decoders");
            C.attr_set = true;
        }
        if (ensure(C.comp, (size_t)comp_bytes + 8) || ensure(C.out, (size_t)out_bytes + 8) || ensure(C.blocks, sizeof(BgzfBlockDev) * (size_t)n) || ensure(C.status, sizeof(int32_t) * (size_t)n))
            return uvcgpu_set_error(UVCGPU_ENOMEM, "bgzf_inflate: hipMalloc");
        if (hipMemcpyAsync(C.comp.p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, C.stream) != hipSuccess
            || hipMemcpyAsync(C.blocks.p, hb.data(), sizeof(BgzfBlockDev) * (size_t)n, hipMemcpyHostToDevice, C.stream) != hipSuccess
            || hipMemsetAsync(C.status.p, 0x7F, sizeof(int32_t) * (size_t)n, C.stream) != hipSuccess) return uvcgpu_set_error(UVCGPU_EDEVICE, "bgzf_inflate: H2D");
        const bool timing = (getenv("UVCGPU_TIMING") != nullptr);
        struct Events { hipEvent_t e0 = nullptr, e1 = nullptr; ~Events() { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); } } ev;   // destroyed on every return path
        hipEvent_t &e0 = ev.e0, &e1 = ev.e1;
        if (timing) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, C.stream); }
        const char *wenv = getenv("UVCGPU_INFLATE_WAVE");   // read per call: the tests run every form in one process
        const int wave_per_block = wenv ? atoi(wenv) : 8;   // default: a wave per block at 8 waves per SIMD; 1: the compiler's register budget; 0: a lane per block
        if (wave_per_block == 8) hipLaunchKernelGGL(k_bgzf_inflate_wave8, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, C.stream, (const uint8_t *)C.comp.p, (const BgzfBlockDev *)C.blocks.p, (int)n, (uint8_t *)C.out.p, (int32_t *)C.status.p);
        else if (wave_per_block) hipLaunchKernelGGL(k_bgzf_inflate_wave, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, C.stream, (const uint8_t *)C.comp.p, (const BgzfBlockDev *)C.blocks.p, (int)n, (uint8_t *)C.out.p, (int32_t *)C.status.p);
        else hipLaunchKernelGGL(k_bgzf_inflate, dim3((unsigned)((n + 63) / 64)), dim3(64), lds, C.stream, (const uint8_t *)C.comp.p, (const BgzfBlockDev *)C.blocks.p, (int)n, (uint8_t *)C.out.p, (int32_t *)C.status.p);
        if (hipGetLastError() != hipSuccess) return uvcgpu_set_error(UVCGPU_EDEVICE, "bgzf_inflate: kernel launch failed");
        if (timing) hipEventRecord(e1, C.stream);
        std::vector<int32_t> st((size_t)n);
        if (hipMemcpyAsync(st.data(), C.status.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, C.stream) != hipSuccess
            || hipMemcpyAsync(out + lo, (const uint8_t *)C.out.p + lo, (size_t)(hi - lo), hipMemcpyDeviceToHost, C.stream) != hipSuccess
            || hipStreamSynchronize(C.stream) != hipSuccess) return uvcgpu_set_error(UVCGPU_EDEVICE, "bgzf_inflate: kernel or D2H failed");
        if (timing) { float ms = 0; hipEventElapsedTime(&ms, e0, e1); fprintf(stderr, "[uvcgpu bgzf_inflate] %lld blocks, %.1f MB -> %.1f MB: kernel %.2f ms\n", (long long)n, comp_bytes / 1e6, (hi - lo) / 1e6, ms); }
        for (int64_t i = 0; i < n; i++) if (st[(size_t)i] != 0)
            return uvcgpu_set_error(UVCGPU_EINVAL, (std::string("bgzf_inflate: corrupt DEFLATE stream in block ") + std::to_string(i) + " (code " + std::to_string(st[(size_t)i]) + ")").c_str());
        return 0;
    } catch (const std::bad_alloc &) { return uvcgpu_set_error(UVCGPU_ENOMEM, "bgzf_inflate: out of host memory"); }
}
