// uvc_group.hip -- family assignment on the device (include/uvcgroup.h; SURVEY rows a10 / a11, "next" N4).
//
//   k_g_pre     per alignment   fill_isrc_isr2_beg_end_with_aln + first-scan histograms + visited read names   grouping.cpp:347-415, 662-694
//   k_g_scan    per class       prefix sums of the begin + end counts                                           grouping.cpp:696-705
//   k_g_center  per bin         poscounter_to_pos2pcenter                                                       grouping.cpp:422-442
//   k_g_key     per alignment   second scan: snapped ends, amplicon tests, dedup_idflag, MolecularBarcode key    grouping.cpp:732-948, MolecularID.hpp:20-52
//   (rocPRIM radix sorts)       alns3 order: family key, strand, fragment (base-17 read-name hash), file order   grouping.cpp:545-566, 949-952
//   k_g_assign  per kept aln    run boundaries -> family / fragment ids
//
// Read names and UMIs are 2 x 64-bit hashes (uvcgroup.h).  A family key is reduced to two independent 64-bit mixes; two keys are
// the same family only if both agree.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <algorithm>
#include <string>
#include <vector>
#include "uvcgpu.h"
#include "uvcgroup.h"
#include "uvc_alloc.h"
#define hipMalloc(p, n) uvc_dev_malloc((void **)(p), (n))
#define hipFree(p) uvc_dev_free((void *)(p))

#define G_MAX_INSERT 2000        // MAX_INSERT_SIZE, common.hpp:64
#define G_MARGIN G_MAX_INSERT    // ARRPOS_MARGIN, grouping.cpp:22
#define G_OUTER 10               // ARRPOS_OUTER_RANGE
#define G_INNER 3                // ARRPOS_INNER_RANGE
typedef unsigned long long u64;

extern "C" int uvcgpu_set_error(int code, const char *msg);   // uvc_host.cpp

struct GCols { const int32_t *tid, *pos, *endpos, *mtid, *mpos, *isize; const uint16_t *flag; const uint8_t *mapq; const u64 *q31, *q17, *u31, *u17; const uint8_t *umi_kind; };
struct GWork {
    int64_t n; int fetch_size; u64 set_mask;
    int32_t *reason, *isize_norm, *cls, *tbeg, *tend;          // per alignment
    int32_t *begc, *endc, *b2c, *e2c;                          // [4][fetch_size]
    long long *border;                                          // [4][fetch_size + 1]
    u64 *set;                                                   // visited read names (open addressing, 0 = empty)
    u64 *key1, *key2, *qkey; int32_t *meta;                     // per alignment: family key mixes, base-17 name hash, strand | dflag << 8 | idflag << 16
    int32_t *ext;                                               // [0] min pos, [1] max endpos
    u64 *counters;                                              // [0] kept, [1] amplicon, [2] visited names, [3] hash collisions
};

__device__ __forceinline__ u64 mix64(u64 x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__device__ __forceinline__ u64 name_key(u64 a, u64 b) { const u64 k = mix64(a ^ mix64(b)); return k ? k : 1ull; }

__global__ void __launch_bounds__(256) k_g_pre(GWork W, GCols C, UvcGroupParams P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W.n) return;
    const int flag = C.flag[i], mapq = C.mapq[i], pos = C.pos[i], endpos = C.endpos[i], mpos = C.mpos[i];
    const int isize_raw = C.isize[i];
    const int isize = (abs(isize_raw) >= G_MAX_INSERT ? 0 : isize_raw);   // NORM_INSERT_SIZE, common.hpp:75
    W.isize_norm[i] = isize;
    // the call sites pass (kept_aln_min_aln_len, kept_aln_min_mapqual) into (min_mapqual, min_aln_len): swapped, reproduced as called
    const int min_mapqual = P.kept_aln_min_aln_len, min_aln_len = P.kept_aln_min_mapqual;
    const bool merge = (P.pair_end_merge == 0);
    int reason = UVC_FR_NOT_FILTERED, isrc = 0, isr2 = 0, tBeg = 0, tEnd = 0;
    if (flag & 0x4) reason = UVC_FR_NOT_MAPPED;
    else if ((flag & 0x900) != 0) reason = UVC_FR_NOT_PRIMARY_ALN;
    else if (mapq < min_mapqual) reason = UVC_FR_LOW_MAPQ;
    else if ((endpos - pos) < min_aln_len) reason = UVC_FR_LOW_ALN_LEN;
    else if (0 == isize && P.kept_aln_is_zero_isize_discarded) reason = UVC_FR_ZERO_ISIZE;
    else if (0 != isize && abs(isize) < P.kept_aln_min_isize) reason = UVC_FR_LOW_ISIZE;
    else if (0 != isize && abs(isize) > P.kept_aln_max_isize) reason = UVC_FR_HIGH_ISIZE;
    else {
        isrc = ((flag & 0x10) == 0x10);
        isr2 = (merge && (flag & 0x80) == 0x80 && (flag & 0x1) == 0x1);
        const int begpos = pos, endp = endpos - 1;
        if (!merge || ((flag & 0x1) == 0) || (flag & 0x8) || (0 == isize) || (abs(isize) >= G_MARGIN)) { tBeg = (isrc ? endp : begpos); tEnd = (isrc ? begpos : endp); }
        else {
            const int l = min(begpos, mpos), r = l + abs(isize) - 1;
            const bool strand = (((flag & 0x81) == 0x81) ? ((flag & 0x20) != 0) : ((flag & 0x10) != 0));
            tBeg = (strand ? r : l); tEnd = (strand ? l : r);
        }
        const int oB = min(tBeg, tEnd), oE = max(tBeg, tEnd);
        if (oB + (G_MARGIN - G_OUTER) <= P.fetch_tbeg || P.fetch_tend - 1 + (G_MARGIN - G_OUTER) <= oE) reason = UVC_FR_OUT_OF_RANGE;
        else if (P.end2end && !(oB <= P.fetch_tbeg && oE >= P.fetch_tend)) reason = UVC_FR_NOT_END_TO_END;
    }
    W.reason[i] = reason; W.cls[i] = isrc * 2 + isr2; W.tbeg[i] = tBeg; W.tend[i] = tEnd;
    if (reason != UVC_FR_NOT_FILTERED) return;
    const int c = isrc * 2 + isr2;
    const int bi = tBeg + G_MARGIN - P.fetch_tbeg, ei = tEnd + G_MARGIN - P.fetch_tbeg;
    if (bi >= 0 && bi < W.fetch_size) atomicAdd(&W.begc[(size_t)c * W.fetch_size + bi], 1);
    if (ei >= 0 && ei < W.fetch_size) atomicAdd(&W.endc[(size_t)c * W.fetch_size + ei], 1);
    const int mn = min(tBeg, tEnd), mx = max(tBeg, tEnd) + 2;
    if (!((mx <= P.fetch_tbeg) || (P.fetch_tend <= mn))) {   // visited_qnames.insert, grouping.cpp:690-692
        const u64 k = name_key(C.q31[i], C.q17[i]);
        u64 slot = k & W.set_mask;
        for (;;) {
            const u64 prev = atomicCAS(&W.set[slot], 0ull, k);
            if (prev == 0ull) { atomicAdd(&W.counters[2], 1ull); break; }
            if (prev == k) break;
            slot = (slot + 1) & W.set_mask;
        }
    }
}

// one block per class: border[c][i + 1] = sum_{j <= i} (begc + endc)
__global__ void __launch_bounds__(1024) k_g_scan(GWork W) {
    __shared__ long long buf[1024];
    const int c = blockIdx.x, t = threadIdx.x;
    const int32_t *b = W.begc + (size_t)c * W.fetch_size, *e = W.endc + (size_t)c * W.fetch_size;
    long long *out = W.border + (size_t)c * (W.fetch_size + 1);
    long long carry = 0;
    if (t == 0) out[0] = 0;
    for (int base = 0; base < W.fetch_size; base += 1024) {
        const int i = base + t;
        long long v = (i < W.fetch_size ? (long long)b[i] + e[i] : 0);
        buf[t] = v; __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) { long long a = (t >= off ? buf[t - off] : 0); __syncthreads(); buf[t] += a; __syncthreads(); }
        if (i < W.fetch_size) out[i + 1] = carry + buf[t];
        carry += buf[1023];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_g_center(GWork W, double mult) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = 4LL * W.fetch_size;
    if (t >= 2 * per) return;
    const bool is_end = (t >= per);
    const int64_t u = (is_end ? t - per : t);
    const int lo = (int)(u % W.fetch_size);
    const int32_t *cnt = (is_end ? W.endc : W.begc) + (u - lo);
    int32_t *cen = (is_end ? W.e2c : W.b2c) + (u - lo);
    if (lo < G_INNER || lo >= W.fetch_size - G_INNER) { cen[lo] = 0; return; }   // untouched bins keep the vector's initial 0
    const int locnt = cnt[lo];
    int center = lo, maxc = locnt;
    for (int hi = lo - G_INNER; hi < lo + G_INNER + 1; hi++) {
        const int hicnt = cnt[hi];
        const int d = (lo > hi ? lo - hi : hi - lo);
        if ((hicnt > maxc) && ((double)(hicnt + 1) > (double)(locnt + 1) * pow(mult, (double)d))) { center = hi; maxc = hicnt; }
    }
    cen[lo] = center;
}

__global__ void __launch_bounds__(256) k_g_key(GWork W, GCols C, UvcGroupParams P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W.n) return;
    W.key1[i] = ~0ull; W.key2[i] = 0; W.qkey[i] = C.q17[i]; W.meta[i] = 0;
    const int pos = C.pos[i], endpos = C.endpos[i];
    const long long win_lo = ((long long)P.fetch_tbeg > (G_MAX_INSERT + 1) ? (long long)P.fetch_tbeg - (G_MAX_INSERT + 1) : 0);
    if (pos < win_lo || endpos > (P.fetch_tend + G_MAX_INSERT + 1)) { W.reason[i] = UVC_FR_NOT_IN_WINDOW; return; }
    {
        const u64 k = name_key(C.q31[i], C.q17[i]);
        u64 slot = k & W.set_mask; bool found = false;
        for (;;) { const u64 v = W.set[slot]; if (v == k) { found = true; break; } if (v == 0ull) break; slot = (slot + 1) & W.set_mask; }
        if (!found) { W.reason[i] = UVC_FR_QNAME_NOT_VISITED; return; }
    }
    if (W.reason[i] != UVC_FR_NOT_FILTERED) return;
    atomicMin(&W.ext[0], pos); atomicMax(&W.ext[1], endpos);
    const int flag = C.flag[i], isize = W.isize_norm[i];
    const bool umi = (C.umi_kind[i] & 1), dup = (C.umi_kind[i] & 2);
    const int c = W.cls[i];
    const size_t co = (size_t)c * W.fetch_size;
    const int beg1 = W.tbeg[i] + G_MARGIN - P.fetch_tbeg, end1 = W.tend[i] + G_MARGIN - P.fetch_tbeg;
    const int beg2 = W.b2c[co + beg1], end2 = W.e2c[co + end1];
    const long long bc = W.begc[co + beg2], ec = W.endc[co + end2];
    const int iL = min(beg2 + 6, end2), iR = max(beg2, (end2 > 6 ? end2 - 6 : 0));
    const long long *bd = W.border + (size_t)c * (W.fetch_size + 1);
    const long long tot = bd[iR] - bd[iL];
    const double begratio = (double)(bc * (iR - iL) + 1) / (double)(tot + (iR - iL) + 1);
    const double endratio = (double)(ec * (iR - iL) + 1) / (double)(tot + (iR - iL) + 1);
    const bool b_amp = (begratio > P.dedup_amplicon_border_to_insert_cov_weak_avgDP_ratio && ((double)bc >= P.dedup_amplicon_border_weak_minDP) && ((double)bc >= (double)tot * P.dedup_amplicon_border_to_insert_cov_weak_totDP_ratio));
    const bool e_amp = (endratio > P.dedup_amplicon_border_to_insert_cov_weak_avgDP_ratio && ((double)ec >= P.dedup_amplicon_border_weak_minDP) && ((double)ec >= (double)tot * P.dedup_amplicon_border_to_insert_cov_weak_totDP_ratio));
    const bool b_str = (begratio > P.dedup_amplicon_border_to_insert_cov_strong_avgDP_ratio && ((double)bc >= P.dedup_amplicon_border_strong_minDP) && ((double)bc >= (double)tot * P.dedup_amplicon_border_to_insert_cov_strong_totDP_ratio));
    const bool e_str = (endratio > P.dedup_amplicon_border_to_insert_cov_strong_avgDP_ratio && ((double)ec >= P.dedup_amplicon_border_strong_minDP) && ((double)ec >= (double)tot * P.dedup_amplicon_border_to_insert_cov_strong_totDP_ratio));
    const bool amplicon = (b_str || e_str || (b_amp && e_amp));
    if (amplicon) atomicAdd(&W.counters[1], 1ull);
    int idflag;
    if (P.dedup_flag != 0) idflag = P.dedup_flag;
    else if (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) idflag = (umi ? 0x9 : (amplicon ? 0x7 : 0x3));
    else if (umi) idflag = ((b_str && e_amp && (double)bc > (double)ec * P.dedup_amplicon_end2end_ratio) ? 0x9 : ((e_str && b_amp && (double)ec > (double)bc * P.dedup_amplicon_end2end_ratio) ? 0xA : 0xB));
    else idflag = (amplicon ? 0x7 : 0x3);
    const bool preserved = ((flag & 0x1) && (!(flag & 0x4)) && (!(flag & 0x8)) && (abs(isize) >= (G_MAX_INSERT * 3 / 4) || isize == 0));
    const int begtid = ((!(flag & 0x4)) ? C.tid[i] : (INT32_MAX - 1));
    const int endtid = (((flag & 0x1) && !(flag & 0x8)) ? C.mtid[i] : (INT32_MAX - 1));
    const int beg3 = (preserved ? pos : (beg2 - G_MARGIN + P.fetch_tbeg));
    const int end3 = (preserved ? C.mpos[i] : (end2 - G_MARGIN + P.fetch_tbeg));
    const int strand = (((flag & 0x81) == 0x81) ? ((flag & 0x20) != 0) : ((flag & 0x10) != 0));
    const int dflag = (umi ? 0x1 : 0) + (dup ? 0x2 : 0) + (amplicon ? 0x4 : 0) + (preserved ? 0x8 : 0);
    // MolecularBarcode::createKey, MolecularID.hpp:20-52 (pairs compare lexicographically)
    int kb0 = -1, kb1 = -1, ke0 = -1, ke1 = -1;
    if (0x3 == (0x3 & idflag)) {
        const bool b_first = (begtid < endtid) || (begtid == endtid && beg3 <= end3);
        kb0 = (b_first ? begtid : endtid); kb1 = (b_first ? beg3 : end3); ke0 = (b_first ? endtid : begtid); ke1 = (b_first ? end3 : beg3);
    } else if (0x1 & idflag) { kb0 = begtid; kb1 = beg3; }
    else if (0x2 & idflag) { ke0 = endtid; ke1 = end3; }
    const u64 qa = ((0x4 & idflag) ? C.q31[i] : 0ull), qb = ((0x4 & idflag) ? C.q17[i] : 0ull);
    const u64 ua = (((0x8 & idflag) && umi) ? C.u31[i] : 0ull), ub = (((0x8 & idflag) && umi) ? C.u17[i] : 0ull);
    const u64 p0 = ((u64)(uint32_t)kb0 << 32) | (uint32_t)kb1, p1 = ((u64)(uint32_t)ke0 << 32) | (uint32_t)ke1, p2 = ((u64)dflag << 8) | (u64)idflag;
    u64 h1 = mix64(p0); h1 = mix64(h1 ^ p1); h1 = mix64(h1 ^ qa); h1 = mix64(h1 ^ qb); h1 = mix64(h1 ^ ua); h1 = mix64(h1 ^ ub); h1 = mix64(h1 ^ p2);
    u64 h2 = mix64(p2 * 0x2545F4914F6CDD1Dull + ub); h2 = mix64(h2 + ua); h2 = mix64(h2 + qb); h2 = mix64(h2 + qa); h2 = mix64(h2 + p1); h2 = mix64(h2 + p0);
    h1 = (h1 & ~1ull) | (u64)strand;       // the strand is the lowest sort bit inside a family
    if (h1 >= ~1ull) h1 -= 2;              // keep the sentinel of dropped alignments strictly largest
    W.key1[i] = h1; W.key2[i] = h2; W.meta[i] = strand | (dflag << 8) | (idflag << 16);
    atomicAdd(&W.counters[0], 1ull);
}

// sorted order: idx[k] = input index; boundary flags for the scans
__global__ void __launch_bounds__(256) k_g_flags(GWork W, const u64 *skey1, const int32_t *idx, int64_t n_kept, int32_t *fam_flag, int32_t *frag_flag) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_kept) return;
    int nf = 1, ng = 1;
    if (k > 0) {
        const int i = idx[k], j = idx[k - 1];
        const bool same_h1 = ((skey1[k] >> 1) == (skey1[k - 1] >> 1)), same_h2 = (W.key2[i] == W.key2[j]);
        if (same_h1 != same_h2) atomicAdd(&W.counters[3], 1ull);   // one mix collided: refuse rather than merge or split silently
        nf = !(same_h1 && same_h2);
        ng = (nf || skey1[k] != skey1[k - 1] || W.qkey[i] != W.qkey[j]);
    }
    fam_flag[k] = nf; frag_flag[k] = ng;
}
__global__ void __launch_bounds__(256) k_g_assign(GWork W, const int32_t *idx, int64_t n_kept, const int32_t *fam_scan, const int32_t *frag_scan, const int32_t *fam_flag,
                                                  int32_t *fam_id, int32_t *frag_id, uint8_t *fam_strand, uint8_t *fam_dflag, uint8_t *fam_idflag) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_kept) return;
    const int m = W.meta[idx[k]];
    const int f = fam_scan[k] - 1;
    fam_id[k] = f; frag_id[k] = frag_scan[k] - 1; fam_strand[k] = (uint8_t)(m & 1);
    if (fam_flag[k]) { fam_dflag[f] = (uint8_t)((m >> 8) & 0xFF); fam_idflag[f] = (uint8_t)((m >> 16) & 0xFF); }
}
__global__ void __launch_bounds__(256) k_g_iota(int32_t *v, int64_t n) { const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] = (int32_t)i; }
__global__ void __launch_bounds__(256) k_g_gather(const u64 *src, const int32_t *idx, u64 *dst, int64_t n) { const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[idx[i]]; }

namespace {
struct Pool {   // frees everything it handed out
    std::vector<void *> p;
    ~Pool() { (void)hipStreamSynchronize(0); for (void *q : p) hipFree(q); }   // the cache reuses a freed block at once: nothing may still run on it
    template <class T> T *get(size_t n, bool zero = false) {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        p.push_back(q);
        if (zero) hipMemset(q, 0, std::max<size_t>(n, 1) * sizeof(T));
        return (T *)q;
    }
    template <class T> T *up(const T *h, size_t n) { T *d = get<T>(n); if (d && n) hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice); return d; }
};
unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }
}

extern "C" {

void uvcgpu_group_params_default(UvcGroupParams *p) {
    memset(p, 0, sizeof(*p));
    p->struct_size = (int32_t)sizeof(UvcGroupParams);
#define UVC_GI(name, dflt) p->name = (int32_t)(dflt);
#define UVC_GD(name, dflt) p->name = (double)(dflt);
#include "uvc_group_params.def"
#undef UVC_GI
#undef UVC_GD
    p->inferred_sequencing_platform = UVC_PLATFORM_ILLUMINA;
}
// Hash.hpp:6-39
uint64_t uvcgpu_strnhash(const char *s, size_t n, uint64_t base) { uint64_t r = 0; for (size_t i = 0; i < n && s[i]; i++) r = r * base + (uint64_t)s[i]; return r; }
uint64_t uvcgpu_hash2hash(uint64_t a, uint64_t b) { return a * ((1UL << 31UL) - 1UL) + b; }
// grouping.cpp:763-786
int uvcgpu_qname_digest(const char *qname, int molecule_tag, int disable_duplex, uint64_t *q31, uint64_t *q17, uint64_t *u31, uint64_t *u17) {
    *q31 = uvcgpu_strnhash(qname, SIZE_MAX, 31UL); *q17 = uvcgpu_strnhash(qname, SIZE_MAX, 17UL);
    const size_t qname_len = strlen(qname);
    const char *h1 = strchr(qname, '#');
    const char *umi_beg = (h1 ? h1 + 1 : qname + qname_len);
    const char *h2 = strchr(umi_beg, '#');
    const char *umi_end = (h2 ? h2 : qname + qname_len);
    *u31 = *u17 = 0;
    if (!((umi_beg + 1 < umi_end) && (1 /* MOLECULE_TAG_NONE */ != molecule_tag))) return 0;
    const size_t umi_len = (size_t)(umi_end - umi_beg), umi_half = (umi_len - 1) / 2;
    *u31 = uvcgpu_strnhash(umi_beg, umi_len, 31UL); *u17 = uvcgpu_strnhash(umi_beg, umi_len, 17UL);
    return 1 | (((umi_len % 2 == 1) && ('+' == umi_beg[umi_half]) && !disable_duplex) ? 2 : 0);
}

// the same for a batch of NUL-terminated names (names + off[i]): what a BAM reader hands over
int uvcgpu_qname_digest_batch(const char *names, const int64_t *off, int64_t n, int molecule_tag, int disable_duplex,
                              uint64_t *q31, uint64_t *q17, uint64_t *u31, uint64_t *u17, uint8_t *umi_kind) {
    if (n < 0 || (n > 0 && (!names || !off || !q31 || !q17 || !u31 || !u17 || !umi_kind))) return uvcgpu_set_error(UVCGPU_EINVAL, "bad argument");
    for (int64_t i = 0; i < n; i++) umi_kind[i] = (uint8_t)uvcgpu_qname_digest(names + off[i], molecule_tag, disable_duplex, &q31[i], &q17[i], &u31[i], &u17[i]);
    return 0;
}


// bam2umihash (grouping.cpp:569-606, called :787-792): single-end reads without a UMI in their name are searched for an in-read UMI
// pattern (environment ONE_STEP_UMI_STRUCT of the reference, main.cpp:1224-1225; letters as seq_nt16_table codes, N = any base = a UMI
// letter), forward at the first five offsets, then reverse-complemented from the read's end.  A hit sets the "UMI found" bit of umi_kind;
// the hash of the UMI letters is returned too, although the reference's family key never reads it (its umistring stays empty there:
// umi_beg / umi_len come from the read name, grouping.cpp:929).  Bases arrive as the codes of UvcBamBatch (0..3 = ACGT, 4 = anything
// else, which the reference holds as its 4-bit code: an ambiguity letter in the READ therefore only matches an N of the pattern).
static int uvc_nt16_of_char(char c) {   // seq_nt16_table of htslib (SAM specification, section 4.2.3: "=ACMGRSVTWYHKDBN")
    switch (c) { case '=': return 0; case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'M': case 'm': return 3; case 'G': case 'g': return 4;
                 case 'R': case 'r': return 5; case 'S': case 's': return 6; case 'V': case 'v': return 7; case 'T': case 't': return 8; case 'W': case 'w': return 9;
                 case 'Y': case 'y': return 10; case 'H': case 'h': return 11; case 'K': case 'k': return 12; case 'D': case 'd': return 13; case 'B': case 'b': return 14; default: return 15; }
}
int uvcgpu_umi_in_read_batch(const char *umi_struct, const uint8_t *bases, const int64_t *seq_off, const int32_t *l_qseq, const uint16_t *flag, int64_t n,
                           uint8_t *umi_kind, uint64_t *umi_hash) {
    if (n < 0 || (n > 0 && (!bases || !seq_off || !l_qseq || !flag || !umi_kind))) return uvcgpu_set_error(UVCGPU_EINVAL, "bad argument");
    if (!umi_struct || !*umi_struct) return 0;
    int pat[256]; int np = 0;
    for (const char *c = umi_struct; *c && np < 256; c++) pat[np++] = uvc_nt16_of_char(*c);
    static const int code16[5] = { 1, 2, 4, 8, 15 };
    static const int rc16[16] = { 0, 8, 4, 3, 2, 5, 6, 7, 1, 9, 10, 11, 12, 13, 14, 15 };   // STATIC_REV_COMPLEMENT.table16, common.hpp:177-183
    for (int64_t r = 0; r < n; r++) {
        if (umi_hash) umi_hash[r] = 0;
        if ((umi_kind[r] & 1) || (flag[r] & 0x1)) continue;   // a UMI in the name wins; paired reads are not searched ("should be proton")
        const uint8_t *b = bases + seq_off[r]; const int lq = l_qseq[r];
        bool found = false; uint64_t h = 0;
        for (int is_rc = 0; is_rc < 2 && !found; is_rc++) for (int i = 0; i < 5 && !found; i++) {
            int patpos = 0; h = 0;
            for (int j = i; j < lq && patpos < np; j++) {
                const int raw = code16[b[is_rc ? (lq - 1 - j) : j] > 4 ? 4 : b[is_rc ? (lq - 1 - j) : j]];
                const int base = (is_rc ? rc16[raw] : raw);
                if (pat[patpos] == base || 15 == pat[patpos]) { if (15 == pat[patpos]) h = h * 16 + (uint64_t)base; patpos++; }
                else break;
            }
            if (patpos == np) found = true;
        }
        if (found) { umi_kind[r] |= 1; if (umi_hash) umi_hash[r] = h; }
    }
    return 0;
}

int uvcgpu_group_families(const UvcGroupParams *Pp, const UvcGroupInput *in, UvcGroupOut *out) {
    if (!Pp || !in || !out || Pp->struct_size != (int32_t)sizeof(UvcGroupParams)) return uvcgpu_set_error(UVCGPU_EINVAL, "bad argument / UvcGroupParams::struct_size");
    if (Pp->fetch_tend <= Pp->fetch_tbeg || in->n_alns < 0 || in->n_alns >= ((int64_t)1 << 31)) return uvcgpu_set_error(UVCGPU_EINVAL, "bad region or alignment count");
    const UvcGroupParams P = *Pp;
    const int64_t n = in->n_alns;
    Pool M;
    GCols C;
    C.tid = M.up(in->tid, n); C.pos = M.up(in->pos, n); C.endpos = M.up(in->endpos, n); C.mtid = M.up(in->mtid, n); C.mpos = M.up(in->mpos, n); C.isize = M.up(in->isize, n);
    C.flag = M.up(in->flag, n); C.mapq = M.up(in->mapq, n);
    C.q31 = (const u64 *)M.up(in->qname_hash31, n); C.q17 = (const u64 *)M.up(in->qname_hash17, n); C.u31 = (const u64 *)M.up(in->umi_hash31, n); C.u17 = (const u64 *)M.up(in->umi_hash17, n);
    C.umi_kind = M.up(in->umi_kind, n);
    GWork W;
    W.n = n; W.fetch_size = P.fetch_tend - P.fetch_tbeg + (G_MARGIN + G_OUTER) * 2;
    u64 cap = 1024; while (cap < (u64)(2 * n + 2)) cap <<= 1;
    W.set_mask = cap - 1;
    const size_t fs = (size_t)W.fetch_size;
    W.reason = M.get<int32_t>(n); W.isize_norm = M.get<int32_t>(n); W.cls = M.get<int32_t>(n); W.tbeg = M.get<int32_t>(n); W.tend = M.get<int32_t>(n);
    W.begc = M.get<int32_t>(4 * fs, true); W.endc = M.get<int32_t>(4 * fs, true); W.b2c = M.get<int32_t>(4 * fs); W.e2c = M.get<int32_t>(4 * fs);
    W.border = M.get<long long>(4 * (fs + 1)); W.set = M.get<u64>(cap, true);
    W.key1 = M.get<u64>(n); W.key2 = M.get<u64>(n); W.qkey = M.get<u64>(n); W.meta = M.get<int32_t>(n);
    W.ext = M.get<int32_t>(2); W.counters = M.get<u64>(4, true);
    u64 *skey_a = M.get<u64>(n), *skey_b = M.get<u64>(n);
    int32_t *idx_a = M.get<int32_t>(n), *idx_b = M.get<int32_t>(n), *fam_flag = M.get<int32_t>(n), *frag_flag = M.get<int32_t>(n), *fam_scan = M.get<int32_t>(n), *frag_scan = M.get<int32_t>(n);
    int32_t *d_fam = M.get<int32_t>(n), *d_frag = M.get<int32_t>(n);
    uint8_t *d_strand = M.get<uint8_t>(n), *d_dflag = M.get<uint8_t>(n), *d_idflag = M.get<uint8_t>(n);
    if (!C.umi_kind || !W.counters || !d_idflag || !frag_scan || !skey_b) return uvcgpu_set_error(UVCGPU_ENOMEM, "hipMalloc failed in uvcgpu_group_families");
    const int32_t ext0[2] = { INT32_MAX, 0 };
    hipMemcpy(W.ext, ext0, sizeof(ext0), hipMemcpyHostToDevice);
    hipStream_t s = nullptr;
    if (n) hipLaunchKernelGGL(k_g_pre, dim3(nblk(n)), dim3(256), 0, s, W, C, P);
    hipLaunchKernelGGL(k_g_scan, dim3(4), dim3(1024), 0, s, W);
    hipLaunchKernelGGL(k_g_center, dim3(nblk(8LL * W.fetch_size)), dim3(256), 0, s, W, P.dedup_center_mult);
    if (n) hipLaunchKernelGGL(k_g_key, dim3(nblk(n)), dim3(256), 0, s, W, C, P);
    u64 counters[4] = { 0, 0, 0, 0 };
    if (hipMemcpy(counters, W.counters, sizeof(counters), hipMemcpyDeviceToHost) != hipSuccess) return uvcgpu_set_error(UVCGPU_EDEVICE, hipGetErrorString(hipGetLastError()));
    const int64_t n_kept = (int64_t)counters[0];
    // alns3 order = stable sort by (family key | strand bit) of the stable sort by base-17 read-name hash of file order
    if (n) {
        hipLaunchKernelGGL(k_g_iota, dim3(nblk(n)), dim3(256), 0, s, idx_a, n);
        size_t tmp_bytes = 0;
        rocprim::radix_sort_pairs(nullptr, tmp_bytes, W.qkey, skey_a, idx_a, idx_b, (size_t)n, 0, 64, s);
        void *tmp = M.get<char>(tmp_bytes);
        if (!tmp) return uvcgpu_set_error(UVCGPU_ENOMEM, "hipMalloc(sort scratch)");
        if (rocprim::radix_sort_pairs(tmp, tmp_bytes, W.qkey, skey_a, idx_a, idx_b, (size_t)n, 0, 64, s) != hipSuccess) return uvcgpu_set_error(UVCGPU_EDEVICE, "radix sort 1");
        hipLaunchKernelGGL(k_g_gather, dim3(nblk(n)), dim3(256), 0, s, W.key1, idx_b, skey_a, n);
        size_t tmp2 = 0;
        rocprim::radix_sort_pairs(nullptr, tmp2, skey_a, skey_b, idx_b, idx_a, (size_t)n, 0, 64, s);
        void *tmpb = (tmp2 <= tmp_bytes ? tmp : M.get<char>(tmp2));
        if (!tmpb) return uvcgpu_set_error(UVCGPU_ENOMEM, "hipMalloc(sort scratch)");
        if (rocprim::radix_sort_pairs(tmpb, tmp2, skey_a, skey_b, idx_b, idx_a, (size_t)n, 0, 64, s) != hipSuccess) return uvcgpu_set_error(UVCGPU_EDEVICE, "radix sort 2");
    }
    int32_t n_fams = 0, n_frags = 0;
    if (n_kept) {   // skey_b / idx_a: sorted keys and input indices; the kept alignments are the first n_kept
        hipLaunchKernelGGL(k_g_flags, dim3(nblk(n_kept)), dim3(256), 0, s, W, skey_b, idx_a, n_kept, fam_flag, frag_flag);
        size_t sb = 0;
        rocprim::inclusive_scan(nullptr, sb, fam_flag, fam_scan, (size_t)n_kept, rocprim::plus<int32_t>(), s);
        void *st = M.get<char>(sb);
        if (!st) return uvcgpu_set_error(UVCGPU_ENOMEM, "hipMalloc(scan scratch)");
        rocprim::inclusive_scan(st, sb, fam_flag, fam_scan, (size_t)n_kept, rocprim::plus<int32_t>(), s);
        rocprim::inclusive_scan(st, sb, frag_flag, frag_scan, (size_t)n_kept, rocprim::plus<int32_t>(), s);
        hipLaunchKernelGGL(k_g_assign, dim3(nblk(n_kept)), dim3(256), 0, s, W, idx_a, n_kept, fam_scan, frag_scan, fam_flag, d_fam, d_frag, d_strand, d_dflag, d_idflag);
        hipMemcpy(&n_fams, fam_scan + (n_kept - 1), 4, hipMemcpyDeviceToHost);
        hipMemcpy(&n_frags, frag_scan + (n_kept - 1), 4, hipMemcpyDeviceToHost);
        hipMemcpy(counters, W.counters, sizeof(counters), hipMemcpyDeviceToHost);
        if (counters[3]) return uvcgpu_set_error(UVCGPU_EDEVICE, "family-key hash collision (two different keys agree in one of the two 64-bit key mixes, probability ~2^-64 per pair): refusing to group rather than merge or split families silently");
    }
    int32_t ext[2];
    hipMemcpy(ext, W.ext, sizeof(ext), hipMemcpyDeviceToHost);
    if (n) { hipMemcpy(out->filter_reason, W.reason, n * 4, hipMemcpyDeviceToHost); hipMemcpy(out->isize_norm, W.isize_norm, n * 4, hipMemcpyDeviceToHost); }
    if (n_kept) {
        hipMemcpy(out->order, idx_a, n_kept * 4, hipMemcpyDeviceToHost); hipMemcpy(out->fam_id, d_fam, n_kept * 4, hipMemcpyDeviceToHost); hipMemcpy(out->frag_id, d_frag, n_kept * 4, hipMemcpyDeviceToHost);
        hipMemcpy(out->fam_strand, d_strand, n_kept, hipMemcpyDeviceToHost); hipMemcpy(out->fam_dflag, d_dflag, n_fams, hipMemcpyDeviceToHost); hipMemcpy(out->fam_idflag, d_idflag, n_fams, hipMemcpyDeviceToHost);
    }
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return uvcgpu_set_error(UVCGPU_EDEVICE, "device error in uvcgpu_group_families");
    out->n_kept = n_kept; out->n_fams = n_fams; out->n_frags = n_frags; out->extended_inclu_beg_pos = ext[0]; out->extended_exclu_end_pos = ext[1];
    out->n_amplicon = (int64_t)counters[1]; out->n_visited_qnames = (int64_t)counters[2];
    return 0;
}

}  // extern "C"
