// Micro-benchmark: what does the PMC counter FETCH_SIZE report for the access patterns of this library's kernels?
// MI355X_MICROARCH.md calibrates it for wide streaming reads only (16 B per lane: the counter shows exactly half the bytes) and says
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below reads a known
// set of bytes of a 2 GiB buffer exactly once (far beyond the 256 MiB Infinity Cache), so FETCH_SIZE x 1024 / bytes is the factor.
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define BUF_BYTES (2ull << 30)
// (1) 16 B per lane, streaming: the guide's case (expect 0.5)
__global__ void __launch_bounds__(256) k_stream16(const int4 *b, size_t n, int *out) {
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const int4 v = b[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678) out[0] = acc;
}
// (2) 4 B per lane, streaming: a wave reads 256 contiguous bytes (the plane reads of the position-centric kernels)
__global__ void __launch_bounds__(256) k_stream4(const int *b, size_t n, int *out) {
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= b[i];
    if (acc == 0x12345678) out[0] = acc;
}
// (3) 2 B per lane: a wave reads 128 contiguous bytes at a 2-byte-aligned pseudo-random place (base | quality of one read at 64 positions,
// k_frag16 / k_p2_fast); `seg` segments of 128 useful bytes each, no segment read twice (one per 512-byte slot, random phase inside it)
__global__ void __launch_bounds__(256) k_seg128(const unsigned short *b, size_t nseg, int *out) {
    const int lane = threadIdx.x & 63;
    int acc = 0;
    for (size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < nseg; w += ((size_t)gridDim.x * blockDim.x) >> 6) {
        const size_t slot = (w * 2654435761ull) % nseg;                 // a permutation-like walk over the slots (nseg is a power of two: odd multiplier)
        const size_t off = slot * 256 + ((slot * 40503ull) & 127);      // in ushorts: slot * 512 B + a phase of 0 .. 254 B
        acc ^= b[off + lane];
    }
    if (acc == 0x1234) out[0] = acc;   // (a value 16 bits can take: the compiler drops the loop otherwise)
}
// (4) 4 B per lane, every lane its own 64-byte sector (stride 64 B) or every second one (stride 128 B): the scoring gather's cell reads
template <int STRIDE_DW>
__global__ void __launch_bounds__(256) k_sparse4(const int *b, size_t ncell, int *out) {
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += (size_t)gridDim.x * blockDim.x) {
        const size_t c = (i * 2654435761ull) % ncell;                   // scattered: neighbouring lanes far apart
        acc ^= b[c * STRIDE_DW];
    }
    if (acc == 0x12345678) out[0] = acc;
}
// (5) 96-byte records, one per lane, as six 16-byte loads (FragFast in k_frag16): a wave's 64 records are 6 144 contiguous bytes
__global__ void __launch_bounds__(256) k_rec96(const int4 *b, size_t nrec, int *out) {
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
        const int4 *q = b + i * 6;
#pragma unroll
        for (int k = 0; k < 6; k++) { const int4 v = q[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678) out[0] = acc;
}
int main() {
    void *buf; int *out;
    if (hipMalloc(&buf, BUF_BYTES) != hipSuccess || hipMalloc((void **)&out, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, BUF_BYTES);
    hipDeviceSynchronize();
    const int blocks = 256 * 8;
    hipEvent_t a, e; hipEventCreate(&a); hipEventCreate(&e);
    auto timed = [&](const char *name, double bytes, auto launch) {
        hipEventRecord(a); launch(); hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, a, e);
        printf("%-12s useful bytes %.0f  %.3f ms  %.2f TB/s of useful bytes\n", name, bytes, ms, bytes / (ms * 1e-3) / 1e12);
    };
    timed("k_stream16", (double)BUF_BYTES, [&] { hipLaunchKernelGGL(k_stream16, dim3(blocks), dim3(256), 0, 0, (const int4 *)buf, BUF_BYTES / 16, out); });
    timed("k_stream4", (double)BUF_BYTES, [&] { hipLaunchKernelGGL(k_stream4, dim3(blocks), dim3(256), 0, 0, (const int *)buf, BUF_BYTES / 4, out); });
    const size_t nseg = (BUF_BYTES / 512) / 2;   // 2 Mi segments of 128 useful bytes in 512-byte slots (power of two); half the buffer
    timed("k_seg128", (double)nseg * 128, [&] { hipLaunchKernelGGL(k_seg128, dim3(blocks), dim3(256), 0, 0, (const unsigned short *)buf, nseg, out); });
    const size_t nc64 = BUF_BYTES / 64 / 2, nc128 = BUF_BYTES / 128 / 2;   // powers of two
    timed("k_sparse4<16>", (double)nc64 * 4, [&] { hipLaunchKernelGGL((k_sparse4<16>), dim3(blocks), dim3(256), 0, 0, (const int *)buf, nc64, out); });
    timed("k_sparse4<32>", (double)nc128 * 4, [&] { hipLaunchKernelGGL((k_sparse4<32>), dim3(blocks), dim3(256), 0, 0, (const int *)buf, nc128, out); });
    const size_t nrec = (BUF_BYTES / 96) & ~(size_t)63;
    timed("k_rec96", (double)nrec * 96, [&] { hipLaunchKernelGGL(k_rec96, dim3(blocks), dim3(256), 0, 0, (const int4 *)buf, nrec, out); });
    hipDeviceSynchronize();
    printf("useful bytes per kernel: stream16 %llu stream4 %llu seg128 %llu (sectors touched: 3 x 64 B per segment unless the phase is a multiple of 64 B) sparse4<16> %llu (64-B sectors %llu) sparse4<32> %llu (64-B sectors %llu, 128-B lines %llu) rec96 %llu\n",
           (unsigned long long)BUF_BYTES, (unsigned long long)BUF_BYTES, (unsigned long long)(nseg * 128), (unsigned long long)(nc64 * 4), (unsigned long long)(nc64 * 64),
           (unsigned long long)(nc128 * 4), (unsigned long long)(nc128 * 64), (unsigned long long)(nc128 * 128), (unsigned long long)(nrec * 96));
    return 0;
}
