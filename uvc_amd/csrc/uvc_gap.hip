// uvc_gap.hip -- device radix sorts of the InDel allele pipeline (k_gap_keys / k_gap_alleles / k_gap_rows in uvc_kernels_acc.hip).
// rocPRIM lives in its own translation unit: its headers do not compile together with the kernel file's helpers.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>

extern "C" size_t uvc_gap_sort_tmp_bytes(size_t n) {
    size_t bytes = 0;
    unsigned long long *k = nullptr;
    rocprim::radix_sort_pairs(nullptr, bytes, k, k, k, k, n, 0, 64, (hipStream_t)0);
    return bytes;
}

// stable sort of (key, value) pairs by bits [0, end_bit) of the key
extern "C" int uvc_gap_sort(void *tmp, size_t tmp_bytes, const unsigned long long *kin, unsigned long long *kout, const unsigned long long *vin, unsigned long long *vout,
                            size_t n, int end_bit, hipStream_t s) {
    if (n == 0) return 0;
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, kin, kout, vin, vout, n, 0, (unsigned)end_bit, s) == hipSuccess ? 0 : -1;
}
