// uvc_rtr.hip -- the region side arrays as device kernels:
//   refstring2repeatvec                 main.hpp:803-874   (RegionalTandemRepeat per reference base: STR track and any-TR track)
//   is_indel_context_more_STR           main.hpp:699-721
//   indel_phred                         main.hpp:794-801   (through a threshold table built from the parameters, see uvc_rtr_thresholds)
//   region_repeatvec_to_baq_offsetarr   main.cpp:400-429   (two BAQ prefix-sum arrays)
//   CHAR_TO_SYMBOL                      main_conversion.hpp:473-488
//
// The reference walks the start positions one after the other (`refpos += max(unit * count, skip + 1) - skip`) and at every start it stops
// at runs `while (ref[q] == ref[q + unit]) q++` for 35 unit lengths.  Here:
//   1. k_rtr_first / k_rtr_suffix   where the comparison ref[q] == ref[q + u] first fails per 1024-chunk and unit, and the suffix minimum of
//                                   that over the chunks: a run that leaves a block's window ends at the next chunk's entry (no per-position loop
//                                   over a run: a 1 Mb run of N costs what random sequence costs).
//   2. k_rtr_cand                   per start: the run ends of all units as a suffix-minimum scan in LDS, the better-repeat rule over the units
//                                   in order, what the start would propose (track lengths, units, indelphred) and where the walk goes next;
//                                   the `next` pointers of a 2048-chunk are doubled in LDS until they leave the chunk.
//   3. k_rtr_chain                  the walk over the chunks: entry of chunk k + 1 = exit of the entry of chunk k (one thread, <= n / 2048 hops,
//                                   the common hop -- entry at the chunk's first base -- out of LDS).
//   4. k_rtr_mark                   the starts the walk stops at, per chunk: pointer jumping from the chunk's entry (J_k = next^(2^k) in LDS).
//   5. k_rtr_tracks                 track of a position = the longest proposal that covers it, the earliest start among equals (the reference
//                                   overwrites on strict > in start order): 64-bit LDS maxima of (length, -start); proposals longer than 1024
//                                   go through a list.  Also the per-position BAQ increments.
//   6. rocPRIM inclusive scan (int64) + k_div10.
#include <algorithm>
#include <cmath>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "uvc_rtr.h"

namespace {
#define RDEV __device__ __forceinline__
constexpr int TC = 1024;          // chunk of the first-mismatch table
constexpr int SC = 2048;          // starts per block of k_rtr_cand / k_rtr_mark
constexpr int WIN = SC + TC;      // reference window of a k_rtr_cand block: a run that is still open at its end continues at a TC boundary
constexpr int NT = 1024;          // threads of the chunk kernels
constexpr int TB = 1024;          // target positions per k_rtr_tracks block
constexpr int TL = 1024;          // proposals up to this length are pushed from the block's own window, longer ones come from the list
constexpr int LEVELS = 11;        // 2^11 = SC
constexpr int INF = 0x7fffffff;

RDEV int rmin(int a, int b) { return a < b ? a : b; }
RDEV int rmax(int a, int b) { return a > b ? a : b; }

// is_indel_context_more_STR, main.hpp:699-721 (note rank2 multiplies by rulen1 when rc2 <= 1, as the reference does)
RDEV bool str_better(int ulen1, int cnt1, int ulen2, int cnt2, int umax) {
    if (ulen2 * cnt2 == 0) return true;
    if (ulen1 > umax || ulen2 > umax) return (ulen1 < ulen2 || (ulen1 == ulen2 && cnt1 > cnt2));
    int r1 = (cnt1 <= 1 ? (-cnt1 * ulen1) : ((cnt1 - 1) * ulen1));
    int r2 = (cnt2 <= 1 ? (-cnt2 * ulen1) : ((cnt2 - 1) * ulen2));
    if (0 == cnt1 || 0 == ulen1) r1 = -100;
    if (0 == cnt2 || 0 == ulen2) r2 = -100;
    return r1 > r2;
}
RDEV uint8_t base_code(uint8_t c) {   // CHAR_TO_SYMBOL, main_conversion.hpp:473-488
    switch (c) { case 'A': case 'a': return UVC_BASE_A; case 'C': case 'c': return UVC_BASE_C; case 'G': case 'g': return UVC_BASE_G; case 'T': case 't': return UVC_BASE_T;
                 case 'I': case 'i': return UVC_LINK_M; case '-': case '_': return UVC_LINK_D1; default: return UVC_BASE_N; }
}
// reverse (suffix) minimum over the lanes of a wave: inclusive result per lane
RDEV int wave_suffix_min(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_down(v, d); if (lane + d < 64) v = rmin(v, o); }
    return v;
}

// first[u - 1][chunk] = smallest q of the chunk with !(q + u < n && ref[q] == ref[q + u]), INF when the chunk has none; also the symbol codes
__global__ void __launch_bounds__(256) k_rtr_first(const uint8_t *ref, int n, int64_t npos, int vmax, int nchunk, int32_t *first, uint8_t *refsym) {
    __shared__ uint8_t sref[TC + 256];
    __shared__ int wmin[256][4];
    const int t = threadIdx.x, base = blockIdx.x * TC;
    for (int w = t; w < TC + vmax; w += 256) sref[w] = (base + w < n ? ref[base + w] : 0);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) { const int64_t i = (int64_t)base + t + 256 * j; if (i < npos) refsym[i] = (i < n ? base_code(sref[t + 256 * j]) : 0); }
    if (blockIdx.x == gridDim.x - 1) for (int64_t i = (int64_t)gridDim.x * TC + t; i < npos; i += 256) refsym[i] = 0;   // npos = n + 1 may start a chunk of its own
    for (int u = 1; u <= vmax; u++) {
        int v = INF;
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            const int w = t + 256 * j, q = base + w;
            if (q < n && !(q + u < n && sref[w] == sref[w + u])) v = q;   // descending j: the smallest q survives
        }
        for (int d = 32; d > 0; d >>= 1) v = rmin(v, __shfl_xor(v, d));
        if ((t & 63) == 0) wmin[u][t >> 6] = v;
    }
    __syncthreads();
    if (t >= 1 && t <= vmax) first[(size_t)(t - 1) * nchunk + blockIdx.x] = rmin(rmin(wmin[t][0], wmin[t][1]), rmin(wmin[t][2], wmin[t][3]));
}
// in place: first[u - 1][k] = min over k' >= k
__global__ void __launch_bounds__(256) k_rtr_suffix(int nchunk, int32_t *first) {
    __shared__ int wtot[4];
    __shared__ int carry_s;
    int32_t *f = first + (size_t)blockIdx.x * nchunk;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int carry = INF;
    for (int hi = nchunk; hi > 0; hi -= 256) {
        const int idx = hi - 256 + t;
        int v = (idx >= 0 ? f[idx] : INF);
        v = wave_suffix_min(v, lane);
        if (lane == 0) wtot[wave] = v;
        __syncthreads();
        for (int w = wave + 1; w < 4; w++) v = rmin(v, wtot[w]);
        v = rmin(v, carry);
        if (idx >= 0) f[idx] = v;
        if (t == 0) carry_s = v;   // thread 0 holds the lowest index of the tile: its value is the minimum of everything from there on
        __syncthreads();
        carry = carry_s;
        __syncthreads();
    }
}

// what one start proposes, and where the walk goes from it
struct Best { int u, c, end; };
__global__ void __launch_bounds__(NT) k_rtr_cand(const uint8_t *ref, int n, int smax, int vmax, int bq_max, const int32_t *thr, const int32_t *suf, int nchunk,
                                                 int32_t *c_len, int32_t *c_alen, int32_t *c_info, int32_t *c_next, int32_t *exit1) {
    extern __shared__ uint8_t dyn[];            // WIN + vmax reference characters
    __shared__ int nm[2][WIN];                  // window-relative index of the first failing comparison at or behind each window element, as far as
                                                // the element's own wave (192 elements) sees; INF = look at the waves behind.  Two buffers: one barrier per unit
    __shared__ int wtot[2][NT / 64];
    __shared__ int nx[SC];
    uint8_t *sref = dyn;
    const int t = threadIdx.x, lane = t & 63;
    const int cs = blockIdx.x * SC, ce = rmin(cs + SC, n);
    for (int w = t; w < WIN + vmax; w += NT) sref[w] = (cs + w < n ? ref[cs + w] : 0);
    Best b[2], a[2];
#pragma unroll
    for (int k = 0; k < 2; k++) { b[k].u = 0; b[k].c = 0; b[k].end = cs + t + NT * k; a[k] = b[k]; }
    const int open_chunk = (cs + WIN) / TC;   // a run that is still open at the end of the window fails first at or behind this chunk
    __syncthreads();
    for (int u = 1; u <= vmax; u++) {
        const int buf = u & 1;
        // window elements 3t .. 3t + 2 of this thread: suffix minimum of the failing positions inside the wave
        const int w0 = 3 * t;
        int v[3];
        int run = INF;
#pragma unroll
        for (int j = 2; j >= 0; j--) {
            const int w = w0 + j, q = cs + w;
            if (q >= n || !(q + u < n && sref[w] == sref[w + u])) run = w;
            v[j] = run;
        }
        const int incl = wave_suffix_min(run, lane);
        int excl = __shfl_down(incl, 1); if (lane == 63) excl = INF;
        if (lane == 0) wtot[buf][t >> 6] = incl;
#pragma unroll
        for (int j = 0; j < 3; j++) nm[buf][w0 + j] = rmin(v[j], excl);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int s = t + NT * k, at = cs + s;
            if (at >= n) continue;
            int e = nm[buf][s];
            for (int w = s / 192 + 1; e == INF && w < NT / 64; w++) e = wtot[buf][w];   // the waves lie in window order: the first one that has a failure has the nearest
            const int q = (e == INF ? suf[(size_t)(u - 1) * nchunk + open_chunk] : cs + e);
            const int c = (q - at) / u + 1;
            if (u <= smax && str_better(u, c, b[k].u, b[k].c, smax)) { b[k].u = u; b[k].c = c; b[k].end = q + u; }
            if (str_better(u, c, a[k].u, a[k].c, vmax)) { a[k].u = u; a[k].c = c; a[k].end = q + u; }
        }
    }
    int nxt[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int s = t + NT * k, at = cs + s;
        nxt[k] = INF;
        if (at >= n) { nx[s] = INF; continue; }
        const int len = rmin(b[k].end, n) - at, alen = rmin(a[k].end, n) - at;
        const int cnt = len / b[k].u;
        // indel_phred(slip_rate * del_to_ins, unit, cnt) as a count of thresholds (uvc_rtr_thresholds), then indel_BQ_max - min(indel_BQ_max - 1, dec)
        const int32_t *th = thr + (size_t)(b[k].u - 1) * bq_max;
        int dec = 0;
        for (int d = 1; d < bq_max; d++) dec += (cnt >= th[d]);
        const int phred = bq_max - dec;
        const int skip = smax + b[k].u;
        nxt[k] = at + rmax(b[k].u * b[k].c, skip + 1) - skip;
        c_len[at] = len; c_alen[at] = alen; c_info[at] = b[k].u | (a[k].u << 8) | (phred << 16); c_next[at] = nxt[k];
        nx[s] = nxt[k];
    }
    // exit of every start: follow `next` until it leaves the chunk (pointer doubling, 2^LEVELS = SC hops at most)
    for (int r = 0; r < LEVELS; r++) {
        __syncthreads();
        int v2[2];
#pragma unroll
        for (int k = 0; k < 2; k++) { const int v1 = nx[t + NT * k]; v2[k] = (v1 < ce ? nx[v1 - cs] : v1); }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; k++) nx[t + NT * k] = v2[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; k++) { const int at = cs + t + NT * k; if (at < n) exit1[at] = nx[t + NT * k]; }
}

// entry[k] = the start at which the walk enters chunk k, -1 when it jumps over the chunk.  One wave: 64 chunks per step while the walk
// enters each chunk at its first base (what it does wherever no long repeat straddles a chunk boundary), one dependent load per chunk otherwise.
__global__ void __launch_bounds__(256) k_rtr_chain(int n, int nsc, const int32_t *exit1, int32_t *entry, int32_t *long_n) {
    const int t = threadIdx.x;
    for (int k = t; k < nsc; k += 256) entry[k] = -1;
    if (t == 0) *long_n = 0;
    __syncthreads();
    if (t >= 64) return;
    int e = 0;
    while (e < n) {
        const int k = e / SC;
        if (e == k * SC) {
            const int kk = k + t;
            const int xv = (kk < nsc ? exit1[(size_t)kk * SC] : INF);             // exit of chunk kk when it is entered at its first base
            const unsigned long long ok = __ballot(kk < nsc && xv == (kk + 1) * SC);
            const int run = (~ok == 0 ? 64 : __builtin_ctzll(~ok));              // chunks k .. k + run - 1 hand over to the next chunk's first base
            const int last = (run < 64 ? run : 63);                               // lane whose exit the walk continues from
            if (t <= last && kk < nsc) entry[kk] = kk * SC;
            e = __shfl(xv, last);
        } else {
            if (t == 0) entry[k] = e;
            e = exit1[e];
        }
    }
}

__global__ void __launch_bounds__(NT) k_rtr_mark(int n, const int32_t *entry, const int32_t *c_next, const int32_t *c_len, const int32_t *c_alen,
                                                 uint8_t *visited, int32_t *long_n, int32_t *long_list) {
    __shared__ uint16_t J[LEVELS][SC];
    __shared__ uint8_t m[SC];
    const int t = threadIdx.x, cs = blockIdx.x * SC, ce = rmin(cs + SC, n);
    const int e = entry[blockIdx.x];
    if (e < 0) {
#pragma unroll
        for (int k = 0; k < 2; k++) { const int at = cs + t + NT * k; if (at < n) visited[at] = 0; }
        return;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int s = t + NT * k, at = cs + s;
        const int nx = (at < n ? c_next[at] : INF);
        J[0][s] = (uint16_t)(nx < ce ? nx - cs : 0xFFFF);
        m[s] = (at == e);
    }
    for (int l = 1; l < LEVELS; l++) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; k++) { const int s = t + NT * k; const uint16_t j = J[l - 1][s]; J[l][s] = (j == 0xFFFF ? (uint16_t)0xFFFF : J[l - 1][j]); }
    }
    // from the entry: after level l every start at a multiple of 2^l hops is marked.  A mark set in the same round by another thread only
    // ever lies on the walk too (J of a start on the walk is on the walk), so the unordered reads are harmless.
    for (int l = LEVELS - 1; l >= 0; l--) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; k++) { const int s = t + NT * k; const uint16_t j = J[l][s]; if (m[s] && j != 0xFFFF) m[j] = 1; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int s = t + NT * k, at = cs + s;
        if (at >= n) continue;
        visited[at] = m[s];
        if (m[s]) {
            const int len = c_len[at], alen = c_alen[at];
            if (len > TL || alen > TL) { const int i = atomicAdd(long_n, 1); long_list[3 * (size_t)i] = at; long_list[3 * (size_t)i + 1] = len; long_list[3 * (size_t)i + 2] = alen; }
        }
    }
}

RDEV unsigned long long track_key(int len, int at) { return ((unsigned long long)(unsigned)len << 32) | (unsigned)(INF - at); }   // longest first, then the earliest start
RDEV void lds_max(unsigned long long *p, unsigned long long v) { __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__global__ void __launch_bounds__(NT) k_rtr_tracks(int n, int64_t npos, int bq_max, int polymerase_size, int str_phred_per_region, int nonstr_phred_per_base,
                                                   const uint8_t *visited, const int32_t *c_len, const int32_t *c_alen, const int32_t *c_info,
                                                   const int32_t *long_n, const int32_t *long_list, int32_t *rtr, int32_t *incs /* [2][npos] */, long long *btot /* [2][blocks] */) {
    __shared__ unsigned long long keyS[TB], keyA[TB];
    __shared__ int ov[NT * 3];
    __shared__ int n_ov;
    const int t = threadIdx.x, cs = blockIdx.x * TB, ce = rmin(cs + TB, n);
    keyS[t] = 0; keyA[t] = 0;
    if (t == 0) n_ov = 0;
    __syncthreads();
    // proposals of the starts in [cs - TL, ce) that are at most TL long
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int at = cs - TL + t + NT * k;
        if (at < 0 || at >= ce || !visited[at]) continue;
        const int len = c_len[at], alen = c_alen[at];
        if (len <= TL) { const unsigned long long key = track_key(len, at); for (int i = rmax(at, cs); i < rmin(at + len, ce); i++) lds_max(&keyS[i - cs], key); }
        if (alen <= TL) { const unsigned long long key = track_key(alen, at); for (int i = rmax(at, cs); i < rmin(at + alen, ce); i++) lds_max(&keyA[i - cs], key); }
    }
    // the long ones: filter the list for this block's range NT entries at a time, then every position looks at what is left
    const int nl = *long_n;
    for (int base = 0; base < nl; base += NT) {
        __syncthreads();
        if (t == 0) n_ov = 0;
        __syncthreads();
        const int i = base + t;
        if (i < nl) {
            const int at = long_list[3 * (size_t)i], len = long_list[3 * (size_t)i + 1], alen = long_list[3 * (size_t)i + 2];
            if (at < ce && at + rmax(len, alen) > cs) { const int o = atomicAdd(&n_ov, 1); ov[3 * o] = at; ov[3 * o + 1] = len; ov[3 * o + 2] = alen; }
        }
        __syncthreads();
        const int no = n_ov, p = cs + t;
        if (p < ce) {
            unsigned long long bs = 0, ba = 0;
            for (int o = 0; o < no; o++) {
                const int at = ov[3 * o], len = ov[3 * o + 1], alen = ov[3 * o + 2];
                if (len > TL && at <= p && p < at + len) { const unsigned long long key = track_key(len, at); if (key > bs) bs = key; }
                if (alen > TL && at <= p && p < at + alen) { const unsigned long long key = track_key(alen, at); if (key > ba) ba = key; }
            }
            if (bs) lds_max(&keyS[t], bs);
            if (ba) lds_max(&keyA[t], ba);
        }
    }
    __syncthreads();
    const int p = cs + t;
    int inc[2] = { 0, 0 };
    if (p < ce) {
        // every position lies in [a, next(a)) of the start a the walk stopped at in front of it, and next(a) <= a + track length: a key is never 0
        // (the guard only keeps a broken invariant from becoming a wild read)
        const unsigned long long ks = keyS[t] ? keyS[t] : track_key(0, p), ka = keyA[t] ? keyA[t] : track_key(0, p);
        const int tl = (int)(ks >> 32), at_s = INF - (int)(unsigned)ks, atl = (int)(ka >> 32), at_a = INF - (int)(unsigned)ka;
        const int info_s = c_info[at_s], info_a = c_info[at_a];
        const int ul = info_s & 0xFF, phred = info_s >> 16, aul = (info_a >> 8) & 0xFF;
        // region_repeatvec_to_baq_offsetarr, main.cpp:400-429: the increments (both arrays divide by the STR unit length, as the reference does)
#pragma unroll
        for (int any = 0; any < 2; any++) {
            const int l2 = any ? atl : tl, reps = l2 / ul;
            inc[any] = (reps >= 3 || (reps >= 2 && l2 >= polymerase_size)) ? ((str_phred_per_region * 10) / l2 + 1) : (nonstr_phred_per_base * 10);
        }
        const int vals[UVC_NRTR] = { at_s, tl, ul, phred, at_a, atl, aul };
#pragma unroll
        for (int f = 0; f < UVC_NRTR; f++) rtr[(size_t)f * npos + p] = vals[f];
        incs[p] = inc[0]; incs[npos + p] = inc[1];
        if (p == n - 1) {   // region_repeatvec.push_back(LAST(region_repeatvec)), main.hpp:872
#pragma unroll
            for (int f = 0; f < UVC_NRTR; f++) rtr[(size_t)f * npos + n] = vals[f];
            incs[n] = inc[0]; incs[npos + n] = inc[1];
        }
    }
    // what this block's positions add to the prefix sums (the repeated last entry is not part of any total: k_rtr_baq reads it itself)
    __shared__ long long wtot2[2][NT / 64];
#pragma unroll
    for (int a = 0; a < 2; a++) {
        long long v = inc[a];
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
        if ((t & 63) == 0) wtot2[a][t >> 6] = v;
    }
    __syncthreads();
    if (t < 2) { long long v = 0; for (int w = 0; w < NT / 64; w++) v += wtot2[t][w]; btot[(size_t)t * gridDim.x + blockIdx.x] = v; }
}
// region_repeatvec_to_baq_offsetarr, main.cpp:400-429: prefix sums of the increments from the region start, then / 10.  Every block sums the
// totals of the blocks in front of it itself (k_rtr_tracks left one per block), then scans its own 1024 positions.
__global__ void __launch_bounds__(NT) k_rtr_baq(int nb_tracks, int64_t npos, const int32_t *incs /* [2][npos] */, const long long *btot /* [2][nb_tracks] */, int64_t *baq) {
    __shared__ long long wsum[2][NT / 64];
    __shared__ long long base_s[2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, cs = blockIdx.x * TB;
    const int np = (int)npos;
    // the blocks in front of this one
    long long pre[2] = { 0, 0 };
    for (int i = t; i < (int)blockIdx.x && i < nb_tracks; i += NT) { pre[0] += btot[i]; pre[1] += btot[(size_t)nb_tracks + i]; }
    long long own[2] = { 0, 0 };
    const int p = cs + t;
    if (p < np) { own[0] = incs[p]; own[1] = incs[npos + p]; }
#pragma unroll
    for (int a = 0; a < 2; a++) {
        long long v = pre[a];
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
        if (lane == 0) wsum[a][wave] = v;
    }
    __syncthreads();
    if (t < 2) { long long v = 0; for (int w = 0; w < NT / 64; w++) v += wsum[t][w]; base_s[t] = v; }
    __syncthreads();
    long long incl[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
        long long v = own[a];
        for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(v, d); if (lane >= d) v += o; }
        incl[a] = v;
        if (lane == 63) wsum[a][wave] = v;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; a++) {
        long long v = incl[a] + base_s[a];
        for (int w = 0; w < wave; w++) v += wsum[a][w];
        if (p < np) baq[(size_t)a * npos + p] = v / 10;
    }
}
size_t a256(size_t b) { return (b + 255) & ~(size_t)255; }
}   // namespace

size_t uvc_rtr_work_bytes(int64_t cap, int vmax, int smax, int bq_max, size_t *scan_tmp_bytes) {
    const size_t n = (size_t)cap, ntc = (n + TC - 1) / TC + 1, nsc = (n + SC - 1) / SC + 1;
    *scan_tmp_bytes = 0;
    return a256(n + 1) + a256(4 * (size_t)vmax * ntc) + 5 * a256(4 * n) + a256(4 * nsc) + a256(n) + a256(4) + a256(12 * n) + a256(4 * (size_t)smax * bq_max) + a256(8 * (n + 2)) + a256(16 * ntc) + 256;
}
void uvc_rtr_bind(UvcRtrWork *W, char *b, int64_t cap, int vmax, int smax, int bq_max, size_t scan_tmp_bytes) {
    const size_t n = (size_t)cap, ntc = (n + TC - 1) / TC + 1, nsc = (n + SC - 1) / SC + 1;
    W->cap = cap;
    W->refchar = (uint8_t *)b; b += a256(n + 1);
    W->first = (int32_t *)b; b += a256(4 * (size_t)vmax * ntc);
    W->c_len = (int32_t *)b; b += a256(4 * n); W->c_alen = (int32_t *)b; b += a256(4 * n); W->c_info = (int32_t *)b; b += a256(4 * n);
    W->c_next = (int32_t *)b; b += a256(4 * n); W->exit1 = (int32_t *)b; b += a256(4 * n);
    W->entry = (int32_t *)b; b += a256(4 * nsc);
    W->visited = (uint8_t *)b; b += a256(n);
    W->long_n = (int32_t *)b; b += a256(4);
    W->long_list = (int32_t *)b; b += a256(12 * n);
    W->thr = (int32_t *)b; b += a256(4 * (size_t)smax * bq_max);
    W->incs = (int32_t *)b; b += a256(8 * (n + 2));
    W->btot = (long long *)b; b += a256(16 * ntc);
    W->scan_tmp = nullptr; W->scan_tmp_bytes = scan_tmp_bytes;
}

// indel_phred (main.hpp:794-801) is evaluated on the host, with the host's libm, into thresholds: for a unit length the value is a
// non-decreasing step function of the repeat count, and refstring2repeatvec only uses min(indel_BQ_max - 1, value).  The kernels count
// thresholds instead of evaluating log1p / exp / log, whose device versions are not correctly rounded (a floor() away from a flip).
void uvc_rtr_thresholds(const UvcParams *P, int32_t *thr) {
    const int smax = P->indel_str_repeatsize_max, bq = P->indel_BQ_max;
    const double ampfact = P->indel_polymerase_slip_rate * P->indel_del_to_ins_err_ratio;
    auto dec = [&](int u, int64_t cnt) -> int64_t {
        const int64_t span = (int64_t)u * cnt;
        const double slips = (span > 64 ? (double)(span - 8) : std::log1p(std::exp((double)span - 8.0))) * ampfact / ((double)(u * u));
        return (int64_t)std::floor(-10 * std::log((1.0 - 2.220446049250313e-16) / (slips + 1.0)) / std::log(10.0));
    };
    for (int u = 1; u <= smax; u++) for (int d = 0; d < bq; d++) {
        const int64_t top = ((int64_t)1 << 30) / u;
        int32_t v = 0x7fffffff;
        if (dec(u, top) >= d) { int64_t lo = 0, hi = top; while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (dec(u, mid) >= d) hi = mid; else lo = mid + 1; } v = (int32_t)lo; }
        thr[(size_t)(u - 1) * bq + d] = v;
    }
}

int uvc_launch_region_tracks(const UvcRtrWork *W, const UvcParams *P, int64_t npos, uint8_t *refsym, int32_t *rtr0, int64_t *baq, hipStream_t s) {
    const int n = (int)(npos - 1), smax = P->indel_str_repeatsize_max, vmax = P->indel_vntr_repeatsize_max, bq = P->indel_BQ_max;
    const int ntc = (n + TC - 1) / TC, nsc = (n + SC - 1) / SC;
    const int nbt = (n + TB - 1) / TB;
    hipLaunchKernelGGL(k_rtr_first, dim3(ntc), dim3(256), 0, s, W->refchar, n, npos, vmax, ntc, W->first, refsym);
    hipLaunchKernelGGL(k_rtr_suffix, dim3(vmax), dim3(256), 0, s, ntc, W->first);
    hipLaunchKernelGGL(k_rtr_cand, dim3(nsc), dim3(NT), (size_t)(WIN + vmax), s, W->refchar, n, smax, vmax, bq, W->thr, W->first, ntc, W->c_len, W->c_alen, W->c_info, W->c_next, W->exit1);
    hipLaunchKernelGGL(k_rtr_chain, dim3(1), dim3(256), 0, s, n, nsc, W->exit1, W->entry, W->long_n);
    hipLaunchKernelGGL(k_rtr_mark, dim3(nsc), dim3(NT), 0, s, n, W->entry, W->c_next, W->c_len, W->c_alen, W->visited, W->long_n, W->long_list);
    hipLaunchKernelGGL(k_rtr_tracks, dim3(nbt), dim3(NT), 0, s, n, npos, bq, (int)std::round(P->indel_polymerase_size), P->indel_str_phred_per_region, P->indel_nonSTR_phred_per_base,
                       W->visited, W->c_len, W->c_alen, W->c_info, W->long_n, W->long_list, rtr0, W->incs, W->btot);
    hipLaunchKernelGGL(k_rtr_baq, dim3((unsigned)((npos + TB - 1) / TB)), dim3(NT), 0, s, nbt, npos, W->incs, W->btot, baq);
    return (int)hipGetLastError();
}
