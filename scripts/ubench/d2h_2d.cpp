// D2H of the score records: 2-D copy (136 rows of n*4 bytes out of a pitched array) against one contiguous copy of the same bytes.
// hipcc -O2 scripts/ubench/d2h_2d.cpp -o /tmp/d2h_2d && /tmp/d2h_2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t rows = 136, n = 54135, cap = 250001;
    int *d = nullptr; hipMalloc((void **)&d, rows * cap * 4);
    hipStream_t s; hipStreamCreate(&s);
    for (int pinned = 0; pinned < 2; pinned++) {
        int *h = nullptr;
        if (pinned) hipHostMalloc((void **)&h, rows * cap * 4, hipHostMallocDefault); else h = (int *)malloc(rows * cap * 4);
        for (size_t i = 0; i < rows * cap; i += 1024) h[i] = 1;
        for (int mode = 0; mode < 3; mode++) {
            double best = 1e9;
            for (int rep = 0; rep < 6; rep++) {
                hipStreamSynchronize(s);
                const double t0 = now();
                if (mode == 0) hipMemcpy2DAsync(h, cap * 4, d, cap * 4, n * 4, rows, hipMemcpyDeviceToHost, s);
                else if (mode == 1) hipMemcpyAsync(h, d, rows * n * 4, hipMemcpyDeviceToHost, s);
                else for (size_t r = 0; r < rows; r++) hipMemcpyAsync(h + r * cap, d + r * cap, n * 4, hipMemcpyDeviceToHost, s);
                hipStreamSynchronize(s);
                best = std::min(best, now() - t0);
            }
            printf("%s host, %s: %.3f ms = %.1f GB/s\n", pinned ? "pinned" : "pageable", mode == 0 ? "2-D copy (136 x 216 KB)" : mode == 1 ? "one 29 MB copy" : "136 row copies", best * 1e3, rows * n * 4 / best / 1e9);
        }
        if (pinned) hipHostFree(h); else free(h);
    }
    return 0;
}
