// does hipHostFree accept a small hipHostMalloc block that was the source of an async copy on a non-blocking stream?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (size_t bytes : { (size_t)5000, (size_t)70000, (size_t)3000000 }) {
        unsigned char *h = nullptr; void *d = nullptr;
        hipError_t e1 = hipHostMalloc((void **)&h, bytes, hipHostMallocDefault);
        hipMalloc(&d, bytes);
        memset(h, 1, bytes);
        hipMemcpyAsync(d, h + 64, bytes - 64, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        hipError_t e2 = hipHostFree(h);
        printf("%zu bytes: hipHostMalloc %s, hipHostFree %s\n", bytes, hipGetErrorString(e1), hipGetErrorString(e2));
        hipFree(d);
    }
    return 0;
}
