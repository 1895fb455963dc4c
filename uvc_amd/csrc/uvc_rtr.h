// uvc_rtr.h -- interface between uvc_host.cpp and uvc_rtr.hip: the region side arrays built on the device.
#ifndef UVC_RTR_H
#define UVC_RTR_H
#include "uvc_device.h"

// Device scratch of one handle for the side-array kernels, sized for `cap` reference bases (uvc_rtr_work_bytes); one allocation, carved up
// by uvc_rtr_bind.  Everything in it is rewritten by each uvc_launch_region_tracks.
struct UvcRtrWork {
    int64_t cap;                // reference bases the buffers hold
    uint8_t *refchar;           // [cap + 1] the caller's ASCII reference (the repeat scan compares the characters themselves, main.hpp:828)
    int32_t *first;             // [vmax][ceil(cap / 1024)] first mismatch of ref[q] == ref[q + u] per 1024-chunk, then its suffix minimum
    int32_t *c_len, *c_alen, *c_info, *c_next, *exit1;   // [cap] per start: STR / any-TR track length, unit | any unit << 8 | indelphred << 16, next start of the walk, first start behind the 2048-chunk
    int32_t *entry;             // [ceil(cap / 2048)] where the walk enters each 2048-chunk (-1: jumped over)
    uint8_t *visited;           // [cap] the start is one the sequential walk of refstring2repeatvec stops at
    int32_t *long_n;            // [1] number of entries of long_list
    int32_t *long_list;         // [3 * cap] (start, track length, any-TR track length) of visited starts with a track longer than 1024
    int32_t *thr;               // [smax][bq_max] indel_phred as thresholds: thr[(u - 1) * bq_max + d] = smallest repeat count whose decphred is >= d
    int32_t *incs;              // [2][cap + 1] BAQ increments per position
    long long *btot;            // [2][ceil(cap / 1024)] their sums per k_rtr_tracks block
    void *scan_tmp; size_t scan_tmp_bytes;
};
size_t uvc_rtr_work_bytes(int64_t cap, int vmax, int smax, int bq_max, size_t *scan_tmp_bytes);
void uvc_rtr_bind(UvcRtrWork *W, char *base, int64_t cap, int vmax, int smax, int bq_max, size_t scan_tmp_bytes);
// Host side, once per handle (parameters only): the threshold form of indel_phred (main.hpp:794-801) for unit lengths 1..smax.
void uvc_rtr_thresholds(const UvcParams *P, int32_t *thr /* [smax][bq_max] */);
// refsym [npos], rtr0 [UVC_NRTR][npos], baq [2][npos] from the n = npos - 1 characters already at W->refchar.  Asynchronous on `s`.
// Returns 0 or a HIP error code (as int).
int uvc_launch_region_tracks(const UvcRtrWork *W, const UvcParams *P, int64_t npos, uint8_t *refsym, int32_t *rtr0, int64_t *baq, hipStream_t s);
#endif
