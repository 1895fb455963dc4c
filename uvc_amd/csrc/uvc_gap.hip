// uvc_gap.hip -- device radix sorts of the InDel allele pipeline (k_gap_keys / k_gap_alleles / k_gap_rows in uvc_kernels_acc.hip).
// rocPRIM lives in its own translation unit: its headers do not compile together with the kernel file's helpers.
#include <algorithm>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>
#include "uvcgpu.h"

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

// ---- device-side orders of set_reads (they replace host radix sorts of millions of ids) ----
// 32-bit keys and 32-bit ids: half the bytes of the 64-bit pair sort per radix pass
__global__ void __launch_bounds__(256) k_keys_pos_cls(const int32_t *pos, const int32_t *cls, int32_t beg, int shift, int64_t n, uint32_t *key, uint32_t *val) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = ((uint32_t)(pos[i] - beg) | ((uint32_t)cls[i] << shift)); val[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(256) k_gather4(const uint32_t *perm, int64_t n, const int32_t *a0, const int32_t *a1, const int32_t *a2, const int32_t *a3,
                                                 int32_t *o0, int32_t *o1, int32_t *o2, int32_t *o3) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t j = (int64_t)perm[i];
    o0[i] = a0[j]; o1[i] = a1[j]; o2[i] = a2[j]; o3[i] = a3[j];
}
// ids sorted by key: out_ids[k] = the k-th id, rank[id] = k for the first n_first ids (the others get -1)
__global__ void __launch_bounds__(256) k_rank_from_sorted(const uint32_t *perm, int64_t n, int64_t n_first, int32_t *out_ids, int32_t *rank) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int32_t id = (int32_t)perm[k];
    if (out_ids && k < n_first) out_ids[k] = id;
    if (rank) rank[id] = (k < n_first ? (int32_t)k : -1);
}
extern "C" size_t uvc_sort32_tmp_bytes(size_t n) {
    size_t bytes = 0;
    uint32_t *k = nullptr;
    rocprim::radix_sort_pairs(nullptr, bytes, k, k, k, k, std::max<size_t>(n, 1), 0, 32, (hipStream_t)0);
    return bytes;
}
// stable sort of ids 0..n-1 by (pos - beg) | cls << shift, all on the stream; work = 4 * n 32-bit words, the sorted ids are work + 3 * n
// key = (pos - beg) | cls << pos_bits with pos - beg < 2^pos_bits and cls < 2^cls_bits: the radix passes stop at the last used bit
// (three 8-bit passes for a 1 Mb tile instead of four)
extern "C" int uvc_sort_by_pos_cls(const int32_t *d_pos, const int32_t *d_cls, int32_t beg, int pos_bits, int cls_bits, int64_t n, uint32_t *work /* [4 n] */, void *tmp, size_t tmp_bytes, hipStream_t s) {
    if (n <= 0) return 0;
    if (pos_bits < 1 || cls_bits < 0 || pos_bits + cls_bits > 32) return -1;
    uint32_t *key = work, *key_s = work + n, *val = work + 2 * n, *val_s = work + 3 * n;
    hipLaunchKernelGGL(k_keys_pos_cls, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_pos, d_cls, beg, pos_bits, n, key, val);
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, key, key_s, val, val_s, (size_t)n, 0, (unsigned)(pos_bits + cls_bits), s) == hipSuccess ? 0 : -1;
}
extern "C" void uvc_launch_gather4(const uint32_t *perm, int64_t n, const int32_t *a0, const int32_t *a1, const int32_t *a2, const int32_t *a3, int32_t *o0, int32_t *o1, int32_t *o2, int32_t *o3, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_gather4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, perm, n, a0, a1, a2, a3, o0, o1, o2, o3);
}
extern "C" void uvc_launch_rank_from_sorted(const uint32_t *perm, int64_t n, int64_t n_first, int32_t *out_ids, int32_t *rank, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_rank_from_sorted, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, perm, n, n_first, out_ids, rank);
}

// ---- columns of chosen positions (uvcgpu_region_fetch_columns): every plane value of a position as one int64 row ----
// The planes are position-fastest, so the columns of one position are n_cols separate cache lines; a thread takes one (position, column)
// and the rows are written coalesced.  This is the read side of the record writer (O(emitted records)), not of the hot path.
struct ColGroups { const char *base[UVC_NUM_FIELD_GROUPS]; int32_t first_col[UVC_NUM_FIELD_GROUPS + 1]; int32_t elem[UVC_NUM_FIELD_GROUPS]; };
__global__ void __launch_bounds__(256) k_gather_columns(ColGroups G, int64_t npos, const int32_t *xs, int64_t n, int32_t n_cols, long long *out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n_cols) return;
    const int64_t i = t / n_cols; const int32_t c = (int32_t)(t - i * n_cols);
    const int64_t x = xs[i];
    long long v = 0;
    if (x >= 0 && x < npos) {
        int g = 0;
        while (g + 1 < UVC_NUM_FIELD_GROUPS && c >= G.first_col[g + 1]) g++;
        const int64_t at = (int64_t)(c - G.first_col[g]) * npos + x;
        v = (G.elem[g] == 8) ? ((const long long *)G.base[g])[at] : (long long)((const int32_t *)G.base[g])[at];
    }
    out[t] = v;
}
extern "C" void uvc_launch_gather_columns(const char *const *base, const int32_t *first_col, const int32_t *elem, int64_t npos, const int32_t *d_xs, int64_t n, long long *d_out, hipStream_t s) {
    ColGroups G;
    for (int g = 0; g < UVC_NUM_FIELD_GROUPS; g++) { G.base[g] = base[g]; G.first_col[g] = first_col[g]; G.elem[g] = elem[g]; }
    G.first_col[UVC_NUM_FIELD_GROUPS] = first_col[UVC_NUM_FIELD_GROUPS];
    const int32_t n_cols = first_col[UVC_NUM_FIELD_GROUPS];
    const int64_t total = n * n_cols;
    if (total > 0) hipLaunchKernelGGL(k_gather_columns, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, G, npos, d_xs, n, n_cols, d_out);
}
