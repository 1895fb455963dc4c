// Micro-benchmark: do SALU and VALU instructions of different waves of one SIMD overlap on gfx950?
// Each wave runs ITER iterations of (V VALU adds, S SALU adds), interleaved.  Build: hipcc --offload-arch=gfx950 -O3 issue_model.hip -o issue_model
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V, int S>
__global__ void __launch_bounds__(256) k(int iters, int *out) {
    int v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
    int s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (j * 4 < V) asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            if (j * 4 < S) asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        }
    }
    if (v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3 == 0x7fffffff) out[0] = 1;
}
template <int V, int S> void run(int waves_per_simd, int *d) {
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<V, S>), dim3(blocks), dim3(256), 0, 0, 10, d);
    hipEventRecord(a); hipLaunchKernelGGL((k<V, S>), dim3(blocks), dim3(256), 0, 0, iters, d); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // cycles per iteration per SIMD at 2.4 GHz, divided by instructions issued by ALL waves of the SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / iters;
    printf("V=%2d S=%2d waves/SIMD=%d  %.3f ms  %.1f cycles/iter/SIMD  per-wave-instr %.2f cycles  (VALU-only model %d, sum model %d, max model %d)\n", V, S, waves_per_simd, ms, cyc,
           cyc / ((V + S) * waves_per_simd), V * 4 * waves_per_simd, (V + S) * 4 * waves_per_simd, (V > S ? V : S) * 4 * waves_per_simd);
}
int main() {
    int *d; hipMalloc(&d, 4);
    for (int w : {1, 2, 4, 8}) { run<64, 0>(w, d); run<0, 64>(w, d); run<64, 64>(w, d); run<64, 32>(w, d); run<32, 64>(w, d); }
    return 0;
}
