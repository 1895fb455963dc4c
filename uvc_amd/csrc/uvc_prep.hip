// uvc_prep.hip -- the read-dependent preparation of uvcgpu_region_set_reads as device kernels: what every later kernel needs to know
// about the alns3 nesting (family -> strand -> fragment -> alignment, main.hpp:3672) and about each CIGAR before the first pass runs.
// Input: the caller's UvcReadSoA columns, already in HBM.  Output: per-read facts (bam_endpos, simple / InDel kind, table / item /
// InDel-event offsets), FragRec / FsRec records with the reference's span rules (fillTidBegEndFromAlns1 / 2, main.hpp:658-697), the
// unit lists of the family kernels, the fragment lists of the statistics kernels and the P2 work-list entries.
// Everything is a per-read map, a prefix sum (own multi-column scans: tile sums in the producer, k_tile_tops, an apply pass) or a per-fragment / per-unit fold;
// two small read-backs size the allocations.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "uvc_prep.h"

namespace {
#define PDEV __device__ __forceinline__
PDEV int pmin(int a, int b) { return a < b ? a : b; }
PDEV int pmax(int a, int b) { return a > b ? a : b; }
enum { PC_MATCH = 0, PC_INS = 1, PC_DEL = 2, PC_REF_SKIP = 3, PC_SOFT_CLIP = 4, PC_HARD_CLIP = 5, PC_PAD = 6, PC_EQUAL = 7, PC_DIFF = 8 };
PDEV bool is_m(int op) { return op == PC_MATCH || op == PC_EQUAL || op == PC_DIFF; }
// one atomic per wave instead of one per lane: millions of same-address atomics serialise (k_read_facts took 8.6 ms with them, 0.2 without)
PDEV int wave_max(int v) { for (int d = 32; d > 0; d >>= 1) v = pmax(v, __shfl_xor(v, d)); return v; }
PDEV int wave_sum(int v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d); return v; }

// ---- several prefix sums over one index in three launches (tile sums inside the producing kernel, one block over the tile sums, one apply
// pass that writes every offset column) instead of a library scan -- two launches -- per column.  Tile = 2 048 consecutive elements.
#define SC_BLOCK 256
#define SC_ITEMS 8
#define SC_TILE (SC_BLOCK * SC_ITEMS)
PDEV long long wave_sum64(long long v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d); return v; }
// block-wide sums of NC per-thread values; the result is valid in thread 0
template <int NC> PDEV void block_sums(long long (&v)[NC], long long (*sh)[NC] /* [4][NC] */) {
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NC; c++) v[c] = wave_sum64(v[c]);
    if ((threadIdx.x & 63) == 0) { for (int c = 0; c < NC; c++) sh[wv][c] = v[c]; }
    __syncthreads();
    if (threadIdx.x == 0) { for (int c = 0; c < NC; c++) v[c] = sh[0][c] + sh[1][c] + sh[2][c] + sh[3][c]; }
}
// exclusive scan of every column of tile_sums [NC][ntiles] in place, the column totals to totals[NC]; one block of 1 024 threads
template <int NC> __global__ void __launch_bounds__(1024) k_tile_tops(long long *tile_sums, int ntiles, long long *totals) {
    __shared__ long long sh[1024];
    for (int c = 0; c < NC; c++) {
        long long carry = 0;
        for (int b0 = 0; b0 < ntiles; b0 += 1024) {
            const int i = b0 + (int)threadIdx.x;
            const long long v = (i < ntiles ? tile_sums[(size_t)c * ntiles + i] : 0);
            sh[threadIdx.x] = v;
            __syncthreads();
            for (int d = 1; d < 1024; d <<= 1) {
                const long long o = (threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0);
                __syncthreads();
                sh[threadIdx.x] += o;
                __syncthreads();
            }
            if (i < ntiles) tile_sums[(size_t)c * ntiles + i] = carry + sh[threadIdx.x] - v;
            const long long tot = sh[1023];
            __syncthreads();
            carry += tot;
        }
        if (threadIdx.x == 0) totals[c] = carry;
    }
}
// exclusive prefix of this thread's NC values over the block's threads (thread order), the tile's prefix added: v[c] becomes the exclusive prefix
template <int NC> PDEV void block_excl(long long (&v)[NC], const long long *tile_pref /* [NC][ntiles] */, int ntiles, int tile, long long (*shw)[NC] /* [4][NC] */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long inc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        inc[c] = v[c];
        for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(inc[c], d); if (lane >= d) inc[c] += o; }
    }
    if (lane == 63) { for (int c = 0; c < NC; c++) shw[wv][c] = inc[c]; }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; c++) {
        long long base = tile_pref[(size_t)c * ntiles + tile];
        for (int q = 0; q < 4; q++) if (q < wv) base += shw[q][c];
        v[c] = base + inc[c] - v[c];
    }
}

// up to three prefix sums over (possibly different) index ranges in three launches: k_cols_sums, k_tile_tops<3>, k_cols_apply
struct ScanCols { const void *in[3]; void *out[3]; int64_t n[3]; int in64[3], out64[3], inclusive[3]; };
PDEV long long col_load(const ScanCols &C, int c, int64_t i) { return i < C.n[c] ? (C.in64[c] ? ((const long long *)C.in[c])[i] : (long long)((const int32_t *)C.in[c])[i]) : 0LL; }
__global__ void __launch_bounds__(SC_BLOCK) k_cols_sums(ScanCols C, long long *tile_sums, int ntiles) {
    __shared__ long long sh[4][3];
    long long v[3] = { 0, 0, 0 };
    for (int k = 0; k < SC_ITEMS; k++) {
        const int64_t i = (int64_t)blockIdx.x * SC_TILE + (int64_t)k * SC_BLOCK + threadIdx.x;
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] += col_load(C, c, i);
    }
    block_sums<3>(v, sh);
    if (threadIdx.x == 0) { for (int c = 0; c < 3; c++) tile_sums[(size_t)c * ntiles + blockIdx.x] = v[c]; }
}
__global__ void __launch_bounds__(SC_BLOCK) k_cols_apply(ScanCols C, const long long *tile_pref, int ntiles) {
    __shared__ long long shw[4][3];
    const int64_t i0 = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    long long e[3][SC_ITEMS], v[3] = { 0, 0, 0 };
#pragma unroll
    for (int c = 0; c < 3; c++) { for (int k = 0; k < SC_ITEMS; k++) { e[c][k] = col_load(C, c, i0 + k); v[c] += e[c][k]; } }
    block_excl<3>(v, tile_pref, ntiles, (int)blockIdx.x, shw);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        for (int k = 0; k < SC_ITEMS; k++) {
            const int64_t i = i0 + k;
            if (i >= C.n[c]) break;
            const long long o = (C.inclusive[c] ? v[c] + e[c][k] : v[c]);
            if (C.out64[c]) ((long long *)C.out[c])[i] = o; else ((int32_t *)C.out[c])[i] = (int32_t)o;
            v[c] += e[c][k];
        }
    }
}

struct Stage1 {   // device counters of the first stage (read back once)
    int32_t err, max_aln_span, n_frags, n_fs, n_complex, any_amplicon;
    int32_t p2_cls[4]; int32_t max_p2_span, pad_;
    int64_t n_p2, table_rows, item_slots, gap_slots, ins_total;
};
struct Stage2 {
    int32_t err, max_frag_span, max_unit_span, max_unit_frags, n_generic, n_dup, n_sweep, n_frag_strand0, max_frag_depth, pad_;
    int64_t work, dup_work;
};

// ---- stage 1: what the CIGAR of each read says (nothing here depends on another read except the two "is a new ..." flags) ----
__global__ void __launch_bounds__(256) k_read_facts(UvcPrepIn in, int32_t rbeg, int32_t rend, int seg_eligible,
                                                    int32_t *endpos, int32_t *kind, int32_t *dflag_of, int32_t *new_frag, int32_t *new_fs, int32_t *is_complex,
                                                    int32_t *n_p2, int64_t *gaps, int64_t *trows, int64_t *items, int64_t *ins, Stage1 *T, long long *tile_sums, int ntiles) {
    int w_err = 0, w_span = 0, w_p2span = 0, w_cls[4] = { 0, 0, 0, 0 }, w_amp = 0;   // this lane's contributions to the region-wide counters
    long long col[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // this lane's share of the tile's column sums (new_frag, new_fs, is_complex, n_p2, trows, items, ins, gaps)
    // one block per tile of SC_TILE consecutive reads (the tile sums feed the offset scans of k_facts_apply)
    for (int k8 = 0; k8 < SC_ITEMS; k8++) {
    const int64_t i = (int64_t)blockIdx.x * SC_TILE + (int64_t)k8 * SC_BLOCK + threadIdx.x;
    if (i >= in.n_reads) break;
    const int32_t nc = in.n_cigar[i], lq = in.l_qseq[i], pos = in.pos[i];
    int32_t o_end = pos + 1, o_kind = 1, o_np2 = 0; int64_t o_gaps = 0, o_trows = 0, o_items = 0, o_ins = 0;
    int err = 0;
    const int fam = in.fam_id[i], strand = in.fam_strand[i];
    const bool bad_fam = (fam < 0 || fam >= in.n_fams || strand < 0 || strand > 1);   // kept apart from err: a later check must not hide it (fam indexes fam_dflag below)
    if (bad_fam) err = 5;
    if (in.seq_off[i] < 0 || in.seq_off[i] + lq > in.n_bases || in.cigar_off[i] < 0 || in.cigar_off[i] + nc > in.n_cigar_ops || nc < 1) err = 1;
    if (!err) {
        const uint32_t *cg = in.cigars + in.cigar_off[i];
        int32_t e = pos; int64_t q = 0, del_total = 0; int n_m = 0; bool simple = true;
        for (int k = 0; k < nc; k++) {
            const int op = (int)(cg[k] & 0xF); const int32_t len = (int32_t)(cg[k] >> 4);
            if (op > PC_DIFF) { err = 2; break; }
            if (is_m(op) || op == PC_DEL || op == PC_REF_SKIP) e += len;
            if (is_m(op) || op == PC_INS || op == PC_SOFT_CLIP) q += len;
            if (is_m(op)) n_m++;
            else if (!(op == PC_SOFT_CLIP || op == PC_HARD_CLIP)) simple = false;
            if (op == PC_INS || op == PC_DEL) o_gaps++;
            if (op == PC_INS) o_ins += len;
            if (op == PC_DEL) del_total += len;
        }
        if (!err) {
            if (e == pos) e = pos + 1;   // bam_endpos of a read without reference-consuming ops
            if (q != lq) err = 3;
        }
        if (!err) {
            if (n_m != 1 || nc > 3) simple = false;
            if (simple && nc == 3) { const int o0 = cg[0] & 0xF, o2 = cg[2] & 0xF; if (is_m(o0) || is_m(o2)) simple = false; }
            if (pos < rbeg || e > rend - 1) err = 4;
        }
        if (!err) {
            // P2 work list (k_p2_fast): a simple alignment is one entry; an InDel read contributes its M runs (used when its InDels are all high-quality)
            bool ok = simple || seg_eligible;
            int32_t rp = pos, np2 = 0, span = 1;
            for (int k = 0; k < nc && ok; k++) {
                const int op = (int)(cg[k] & 0xF); const int32_t len = (int32_t)(cg[k] >> 4);
                if (is_m(op)) { if (rp - pos > 65535 || e - (rp + len) > 65535) ok = false; np2++; span = pmax(span, len); rp += len; }
                else if (op == PC_DEL) rp += len;
                else if (op == PC_INS || op == PC_SOFT_CLIP || op == PC_HARD_CLIP) {}
                else ok = false;   // N / P: keep the sequential path
            }
            o_end = e; o_np2 = (ok ? np2 : 0);
            // kind 2 = candidate for the simple path: k_aln_prelude demotes it to 1 when the read has a low-quality InDel
            o_kind = simple ? 0 : (ok ? 2 : 1);
            if (!simple) { o_trows = (e - pos) + 1; o_items = 2 * (int64_t)lq + 2 * del_total + nc + 4; }
            else { o_gaps = 0; o_ins = 0; w_span = pmax(w_span, e - pos); }
            if (o_np2) {
                const int fl = in.flag[i];
                const int cls = ((fl & 0x10) ? 1 : 0) | ((((fl & 0x81) == 0x81) ? ((fl & 0x20) != 0) : ((fl & 0x10) != 0)) ? 2 : 0);   // is-reverse | bam_get_strand << 1 (common.hpp:89)
                w_cls[0] += (cls == 0) * o_np2; w_cls[1] += (cls == 1) * o_np2; w_cls[2] += (cls == 2) * o_np2; w_cls[3] += (cls == 3) * o_np2; w_p2span = pmax(w_p2span, span);
            }
        }
    }
    if (err) { w_err = pmax(w_err, err); o_kind = 1; o_gaps = 0; o_ins = 0; o_np2 = 0; }
    endpos[i] = o_end; kind[i] = o_kind; n_p2[i] = o_np2; gaps[i] = o_gaps; trows[i] = o_trows; items[i] = o_items; ins[i] = o_ins; is_complex[i] = (o_kind != 0);
    const bool ok_fam = !bad_fam;
    const int df = ok_fam ? (int)in.fam_dflag[fam] : 0;
    dflag_of[i] = df;
    if (df & 0x4) w_amp = 1;
    const bool nfs = (i == 0) || in.fam_id[i - 1] != fam || in.fam_strand[i - 1] != strand;
    const bool nfr = (nfs || in.frag_id[i - 1] != in.frag_id[i]);
    new_fs[i] = nfs; new_frag[i] = nfr;
    col[0] += nfr; col[1] += nfs; col[2] += (o_kind != 0); col[3] += o_np2; col[4] += o_trows; col[5] += o_items; col[6] += o_ins; col[7] += o_gaps;
    }
    {
        __shared__ long long shc[4][8];
        block_sums<8>(col, shc);
        if (threadIdx.x == 0) { for (int c = 0; c < 8; c++) tile_sums[(size_t)c * ntiles + blockIdx.x] = col[c]; }
    }
    // wave, then block, then one set of atomics per block (the grid is a few thousand blocks)
    __shared__ int sh[4][8];
    w_err = wave_max(w_err); w_span = wave_max(w_span); w_p2span = wave_max(w_p2span); w_amp = wave_max(w_amp);
    for (int c = 0; c < 4; c++) w_cls[c] = wave_sum(w_cls[c]);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[wv][0] = w_err; sh[wv][1] = w_span; sh[wv][2] = w_p2span; sh[wv][3] = w_amp; for (int c = 0; c < 4; c++) sh[wv][4 + c] = w_cls[c]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int v[8];
        for (int q = 0; q < 8; q++) { v[q] = sh[0][q]; for (int w = 1; w < 4; w++) v[q] = (q < 4 ? pmax(v[q], sh[w][q]) : v[q] + sh[w][q]); }
        if (v[0]) atomicMax(&T->err, v[0]);
        if (v[1]) atomicMax(&T->max_aln_span, v[1]);
        if (v[2]) atomicMax(&T->max_p2_span, v[2]);
        if (v[3]) T->any_amplicon = 1;
        for (int c = 0; c < 4; c++) if (v[4 + c]) atomicAdd(&T->p2_cls[c], v[4 + c]);
    }
}

// the prefix sums of stage 1 -> indices / offsets (-1 offsets for simple alignments), the list of InDel reads, the totals
__global__ void __launch_bounds__(SC_BLOCK) k_facts_apply(int64_t n, const long long *tile_pref, int ntiles, const long long *totals, const int32_t *kind, const int32_t *new_frag, const int32_t *new_fs,
                                                          const int32_t *n_p2, const int64_t *trows, const int64_t *items, const int64_t *gaps,
                                                          int32_t *frag_of, int32_t *fs_of, int32_t *p2_first, int64_t *table_off, int64_t *item_off, int64_t *gap_off, int32_t *complex_ids, Stage1 *T) {
    __shared__ long long shw[4][7];
    const int64_t i0 = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;   // this thread's SC_ITEMS consecutive reads
    long long v[7] = { 0, 0, 0, 0, 0, 0, 0 };   // new_frag, new_fs, is_complex, n_p2, trows, items, gaps (the sum of the inserted lengths is only needed as a total)
    int e_nf[SC_ITEMS], e_ns[SC_ITEMS], e_k[SC_ITEMS], e_p2[SC_ITEMS]; long long e_tr[SC_ITEMS], e_it[SC_ITEMS], e_gp[SC_ITEMS];
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        const int64_t i = i0 + k; const bool in = (i < n);
        e_nf[k] = in ? new_frag[i] : 0; e_ns[k] = in ? new_fs[i] : 0; e_k[k] = in ? kind[i] : 0; e_p2[k] = in ? n_p2[i] : 0;
        e_tr[k] = in ? trows[i] : 0; e_it[k] = in ? items[i] : 0; e_gp[k] = in ? gaps[i] : 0;
        v[0] += e_nf[k]; v[1] += e_ns[k]; v[2] += (e_k[k] != 0); v[3] += e_p2[k]; v[4] += e_tr[k]; v[5] += e_it[k]; v[6] += e_gp[k];
    }
    // (column 6 of the tile sums is the inserted length: skipped here)
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        long long inc[7];
#pragma unroll
        for (int c = 0; c < 7; c++) { inc[c] = v[c]; for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(inc[c], d); if (lane >= d) inc[c] += o; } }
        if (lane == 63) { for (int c = 0; c < 7; c++) shw[wv][c] = inc[c]; }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 7; c++) {
            long long base = tile_pref[(size_t)(c < 6 ? c : 7) * ntiles + blockIdx.x];
            for (int q = 0; q < 4; q++) if (q < wv) base += shw[q][c];
            v[c] = base + inc[c] - v[c];
        }
    }
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        const int64_t i = i0 + k;
        if (i >= n) break;
        v[0] += e_nf[k]; v[1] += e_ns[k];                 // inclusive: the index of the read's fragment / unit is the count of heads up to it, minus one
        frag_of[i] = (int32_t)v[0] - 1; fs_of[i] = (int32_t)v[1] - 1;
        p2_first[i] = (int32_t)v[3];
        if (e_k[k] == 0) { table_off[i] = -1; item_off[i] = -1; gap_off[i] = -1; }
        else { table_off[i] = v[4]; item_off[i] = v[5]; gap_off[i] = v[6]; complex_ids[v[2]] = (int32_t)i; }
        v[2] += (e_k[k] != 0); v[3] += e_p2[k]; v[4] += e_tr[k]; v[5] += e_it[k]; v[6] += e_gp[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        T->n_frags = (int32_t)totals[0]; T->n_fs = (int32_t)totals[1]; T->n_complex = (int32_t)totals[2];
        T->n_p2 = totals[3]; T->table_rows = totals[4]; T->item_slots = totals[5]; T->ins_total = totals[6]; T->gap_slots = totals[7];
    }
}

// ---- stage 2: the nesting ----
__global__ void __launch_bounds__(256) k_mark_first(UvcPrepIn in, const int32_t *new_frag, const int32_t *new_fs, const int32_t *frag_of, const int32_t *fs_of,
                                                    int32_t *frag_first, int32_t *fs_first_frag, int32_t *fam_fs, int32_t n_frags, int32_t n_fs, Stage2 *T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { frag_first[n_frags] = (int32_t)in.n_reads; fs_first_frag[n_fs] = n_frags; }
    if (i >= in.n_reads) return;
    if (new_frag[i]) frag_first[frag_of[i]] = (int32_t)i;
    if (new_fs[i]) {
        fs_first_frag[fs_of[i]] = frag_of[i];
        const int slot = in.fam_id[i] * 2 + in.fam_strand[i];
        if (atomicCAS(&fam_fs[slot], -1, fs_of[i]) != -1) atomicMax(&T->err, 6);   // reads of one (fam_id, fam_strand) are not contiguous
    }
}
// one thread per fragment: fillTidBegEndFromAlns1 (main.hpp:658-673) with its cumulative "+1 per alignment"
__global__ void __launch_bounds__(256) k_build_frags(UvcPrepIn in, UvcParams P, int32_t rend, const int32_t *endpos, const int32_t *kind, const int32_t *fs_of, const int32_t *dflag_of,
                                                     const int32_t *frag_first, int32_t n_frags, FragRec *frags, int32_t *sweep_flag, int32_t *frag_beg, int32_t *frag_strand, Stage2 *T) {
    int w_span = 0, w_s0 = 0;
    // a grid of at most 1024 blocks strides over the fragments and leaves ONE pair of atomics per block: with a wave's pair per 64 fragments the
    // 31 000 same-line atomics of a 1 M-fragment tile were most of the kernel's time
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < n_frags; f += gridDim.x * blockDim.x) {
    FragRec r; memset(&r, 0, sizeof(r));
    const int a0 = frag_first[f], a1 = frag_first[f + 1];
    r.aln_beg = a0; r.aln_end = a1; r.beg = INT32_MAX; r.end = 0; r.fs = fs_of[a0]; r.strand = in.fam_strand[a0]; r.dflag = dflag_of[a0];
    bool all_simple = true;
    for (int k = a0; k < a1; k++) {
        r.beg = pmin(r.beg, in.pos[k]); r.end = pmax(r.end, endpos[k]) + 1;
        r.normMQ = pmax(r.normMQ, (int)in.mapq[k]);
        all_simple = all_simple && (kind[k] == 0);
    }
    r.end = pmin(r.end, rend);
    const bool amplicon_gated = (((r.dflag & 0x4) || ((P.primerlen > 0) && !(0x2 & P.primer_flag))) && !(P.tn_is_paired && (0x1 & P.primer_flag)));
    r.stat_kind = (all_simple && (a1 - a0) <= 2 && !amplicon_gated) ? 0 : 1;
    sweep_flag[f] = r.stat_kind; frag_beg[f] = r.beg; frag_strand[f] = r.strand;
    frags[f] = r;
    w_span = pmax(w_span, r.end - r.beg); w_s0 += (r.strand == 0);
    }
    __shared__ int sh[4][2];
    w_span = wave_max(w_span); w_s0 = wave_sum(w_s0);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = w_span; sh[threadIdx.x >> 6][1] = w_s0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int sp = pmax(pmax(sh[0][0], sh[1][0]), pmax(sh[2][0], sh[3][0])), s0 = sh[0][1] + sh[1][1] + sh[2][1] + sh[3][1];
        if (sp) atomicMax(&T->max_frag_span, sp);
        if (s0) atomicAdd(&T->n_frag_strand0, s0);
    }
}
// one thread per family-strand unit: fillTidBegEndFromAlns2 (main.hpp:675-697), the duplex partner, which kernels take it
__global__ void __launch_bounds__(256) k_build_units(UvcPrepIn in, UvcParams P, int32_t rend, const int32_t *endpos, const int32_t *dflag_of, const int32_t *frag_first, const int32_t *fs_first_frag,
                                                     const int32_t *fam_fs, int32_t n_fs, FsRec *fss, int32_t *generic_flag, int64_t *gen_span, Stage2 *T) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    int w_span = 0, w_frags = 0;
    if (u < n_fs) {
    FsRec r; memset(&r, 0, sizeof(r));
    r.frag_beg = fs_first_frag[u]; r.frag_end = fs_first_frag[u + 1];
    const int a0 = frag_first[r.frag_beg], a1 = frag_first[r.frag_end];
    r.beg = INT32_MAX; r.end = 0; r.strand = in.fam_strand[a0]; r.fam = in.fam_id[a0]; r.dflag = dflag_of[a0];
    for (int k = a0; k < a1; k++) { r.beg = pmin(r.beg, in.pos[k]); r.end = pmax(r.end, endpos[k]) + 1; }
    r.end = pmin(r.end, rend);
    r.other_fs = fam_fs[r.fam * 2 + (1 - r.strand)];
    const bool singleton_ok = (P.fam_thres_dup1add >= 2 && P.fam_thres_dup2add >= 2 && P.fam_thres_emperr_all_flat_snv >= 2 && P.fam_thres_emperr_all_flat_indel >= 2);
    const bool duplex = ((r.dflag & 0x2) && r.other_fs >= 0);
    r.generic = ((r.frag_end - r.frag_beg) >= 2 || duplex || !singleton_ok) ? 1 : 0;
    r.work_off = 0;
    fss[u] = r;
    generic_flag[u] = r.generic; gen_span[u] = r.generic ? (int64_t)(r.end - r.beg) : 0;
    if (r.generic) { w_span = r.end - r.beg; w_frags = r.frag_end - r.frag_beg; }
    }
    w_span = wave_max(w_span); w_frags = wave_max(w_frags);
    if ((threadIdx.x & 63) == 0) { if (w_span) atomicMax(&T->max_unit_span, w_span); if (w_frags) atomicMax(&T->max_unit_frags, w_frags); }
}
__global__ void __launch_bounds__(256) k_units_post(int32_t rend, int32_t n_fs, FsRec *fss, const int32_t *generic_rank, const int64_t *work_off, int32_t *generic_fs,
                                                    int32_t *dup_flag, int64_t *dup_span) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_fs) return;
    FsRec &r = fss[u];
    int32_t df = 0; int64_t ds = 0;
    if (r.generic) { r.work_off = work_off[u]; generic_fs[generic_rank[u]] = u; }
    if ((r.dflag & 0x2) && r.other_fs >= 0 && r.strand == 0) {
        const FsRec &o = fss[r.other_fs];
        df = 1; ds = (int64_t)(pmax(r.end, pmin(o.end, rend)) - pmin(r.beg, o.beg));
    }
    dup_flag[u] = df; dup_span[u] = ds;
}
__global__ void __launch_bounds__(256) k_compact_dups(int32_t n_fs, const int32_t *dup_flag, const int32_t *dup_rank, const int64_t *dup_off_all, int32_t *dup_units, int64_t *dup_off) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_fs || !dup_flag[u]) return;
    dup_units[dup_rank[u]] = u; dup_off[dup_rank[u]] = dup_off_all[u];
}
__global__ void __launch_bounds__(256) k_frags_post(int32_t n_frags, FragRec *frags, const FsRec *fss, const int32_t *sweep_flag, const int32_t *sweep_rank, int32_t *sweep_frags,
                                                    int32_t rbeg, int32_t *depth_diff) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frags) return;
    frags[f].singleton = fss[frags[f].fs].generic ? 0 : 1;
    if (sweep_flag[f]) sweep_frags[sweep_rank[f]] = f;
    if (depth_diff) { atomicAdd(&depth_diff[frags[f].beg - rbeg], 1); atomicAdd(&depth_diff[frags[f].end - rbeg], -1); }
}
__global__ void k_stage2_totals(int32_t n_fs, int32_t n_frags, const int32_t *generic_rank, const int32_t *generic_flag, const int64_t *work_off, const int64_t *gen_span,
                                const int32_t *dup_rank, const int32_t *dup_flag, const int64_t *dup_off_all, const int64_t *dup_span,
                                const int32_t *sweep_rank, const int32_t *sweep_flag, Stage2 *T) {
    if (n_fs > 0) { const int l = n_fs - 1; T->n_generic = generic_rank[l] + generic_flag[l]; T->work = work_off[l] + gen_span[l]; T->n_dup = dup_rank[l] + dup_flag[l]; T->dup_work = dup_off_all[l] + dup_span[l]; }
    if (n_frags > 0) { const int l = n_frags - 1; T->n_sweep = sweep_rank[l] + sweep_flag[l]; }
}
__global__ void __launch_bounds__(256) k_max_i32(const int32_t *v, int64_t n, int32_t *out) {
    int32_t m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = pmax(m, v[i]);
    for (int d = 32; d > 0; d >>= 1) m = pmax(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
// the P2 work-list entries at their places (one thread per alignment; the order by (class, begin) is made afterwards)
__global__ void __launch_bounds__(256) k_p2_entries(UvcPrepIn in, const int32_t *n_p2, const int32_t *p2_first, int32_t *p_aln, int32_t *p_beg, int32_t *p_end, int32_t *p_qb, int32_t *p_cls) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= in.n_reads || n_p2[i] == 0) return;
    const int32_t nc = in.n_cigar[i];
    const uint32_t *cg = in.cigars + in.cigar_off[i];
    const int fl = in.flag[i];
    const int32_t cls = ((fl & 0x10) ? 1 : 0) | ((((fl & 0x81) == 0x81) ? ((fl & 0x20) != 0) : ((fl & 0x10) != 0)) ? 2 : 0);
    int32_t rp = in.pos[i]; int64_t qp = 0; int64_t w = p2_first[i];
    for (int k = 0; k < nc; k++) {
        const int op = (int)(cg[k] & 0xF); const int32_t len = (int32_t)(cg[k] >> 4);
        if (is_m(op)) { p_aln[w] = (int32_t)i; p_beg[w] = rp; p_end[w] = rp + len; p_qb[w] = (int32_t)((in.seq_off[i] + qp - rp) & 0xFFFFFFFFLL); p_cls[w] = cls; w++; rp += len; qp += len; }
        else if (op == PC_INS || op == PC_SOFT_CLIP) qp += len;
        else if (op == PC_DEL) rp += len;
    }
}
__global__ void __launch_bounds__(256) k_unit_keys(const FsRec *fss, const int32_t *generic_fs, int32_t n, int32_t *beg_of, int32_t *zero) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    beg_of[k] = fss[generic_fs[k]].beg; zero[k] = 0;
}
__global__ void __launch_bounds__(256) k_take_units(const uint32_t *perm, const int32_t *generic_fs, int32_t n, int32_t *sorted) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    sorted[k] = generic_fs[(int32_t)perm[k]];
}

unsigned nblk(int64_t n, int b) { return (unsigned)std::max<int64_t>((n + b - 1) / b, 1); }
}   // namespace

// ---- compact input forms (UvcReadSoA::seq_off / cigar_off == NULL, UvcReadSoA::bases4): offsets by prefix sums, BAM's 4-bit base codes
// through seq_nt16_int[] (htslib: A C G T -> 0..3, everything else 4) into the one-byte-per-base array the kernels read
namespace {
// seq_nt16_int[] as a nibble table: codes 1 2 4 8 (A C G T) -> 0 1 2 3, everything else 4
#define NT16_INT_LUT 0x4444444344424104ULL
PDEV unsigned nt16_int(unsigned code) { return (unsigned)(NT16_INT_LUT >> (4 * code)) & 0xF; }
// eight bases: four packed bytes + eight quality bytes -> the eight base symbols (one byte each) and base | qual << 8 (two bytes each)
PDEV void unpack8(uint32_t w, unsigned long long q, unsigned long long &bb, uint4 &o) {
    bb = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const unsigned byte = (w >> (8 * (j >> 1))) & 0xFF;
        bb |= (unsigned long long)nt16_int((j & 1) ? (byte & 15) : (byte >> 4)) << (8 * j);
    }
    o.x = (uint32_t)((bb & 0xFF) | ((q & 0xFF) << 8) | (((bb >> 8) & 0xFF) << 16) | (((q >> 8) & 0xFF) << 24));
    o.y = (uint32_t)(((bb >> 16) & 0xFF) | (((q >> 16) & 0xFF) << 8) | (((bb >> 24) & 0xFF) << 16) | (((q >> 24) & 0xFF) << 24));
    o.z = (uint32_t)(((bb >> 32) & 0xFF) | (((q >> 32) & 0xFF) << 8) | (((bb >> 40) & 0xFF) << 16) | (((q >> 40) & 0xFF) << 24));
    o.w = (uint32_t)(((bb >> 48) & 0xFF) | (((q >> 48) & 0xFF) << 8) | (((bb >> 56) & 0xFF) << 16) | (((q >> 56) & 0xFF) << 24));
}
// the offset columns of the compact input form: exclusive prefix sums of l_qseq, n_cigar and ceil(l_qseq / 2) over the reads, three at once
__global__ void __launch_bounds__(SC_BLOCK) k_len_sums(const int32_t *l_qseq, const int32_t *n_cigar, int64_t n, long long *tile_sums, int ntiles) {
    __shared__ long long sh[4][3];
    long long v[3] = { 0, 0, 0 };
    for (int k = 0; k < SC_ITEMS; k++) {
        const int64_t i = (int64_t)blockIdx.x * SC_TILE + (int64_t)k * SC_BLOCK + threadIdx.x;
        if (i < n) { const int32_t lq = l_qseq[i]; v[0] += lq; v[1] += n_cigar[i]; v[2] += (lq + 1) >> 1; }
    }
    block_sums<3>(v, sh);
    if (threadIdx.x == 0) { for (int c = 0; c < 3; c++) tile_sums[(size_t)c * ntiles + blockIdx.x] = v[c]; }
}
__global__ void __launch_bounds__(SC_BLOCK) k_len_apply(const int32_t *l_qseq, const int32_t *n_cigar, int64_t n, const long long *tile_pref, int ntiles, int64_t *seq_off, int64_t *cigar_off, int64_t *b4_off) {
    __shared__ long long shw[4][3];
    const int64_t i0 = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    int lq[SC_ITEMS], nc[SC_ITEMS];
    long long v[3] = { 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) { const int64_t i = i0 + k; lq[k] = (i < n ? l_qseq[i] : 0); nc[k] = (i < n ? n_cigar[i] : 0); v[0] += lq[k]; v[1] += nc[k]; v[2] += (lq[k] + 1) >> 1; }
    block_excl<3>(v, tile_pref, ntiles, (int)blockIdx.x, shw);
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        const int64_t i = i0 + k;
        if (i >= n) break;
        if (seq_off) seq_off[i] = v[0];
        if (cigar_off) cigar_off[i] = v[1];
        if (b4_off) b4_off[i] = v[2];
        v[0] += lq[k]; v[1] += nc[k]; v[2] += (lq[k] + 1) >> 1;
    }
}
// General form (reads of odd length: every read starts on a byte of the packed column): sixteen lanes per read, eight bases per lane and turn
// with one 4-byte, one 8-byte load and an 8- and a 16-byte store at the read's own (unaligned) offsets; the last one to seven bases of a read
// one by one.  (One wave per read with a lane per packed byte: 1.45 ms for the 2.7 M reads of a 200 kb x 2000x tile.)
__global__ void __launch_bounds__(256) k_pack_bq4_reads(const uint8_t *b4, int64_t n_b4, const int64_t *b4_off, const int64_t *seq_off, const int32_t *l_qseq, const uint8_t *quals, int64_t n, int64_t n_bases,
                                                        uint8_t *bases, uint16_t *bq, int32_t *bad) {
    const int sub = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
    for (int64_t i = grp; i < n; i += ngrp) {
        const int32_t lq = l_qseq[i];
        const int64_t so = seq_off[i], bo = b4_off[i];
        if (lq < 0 || so < 0 || so + lq > n_bases || bo + ((lq + 1) >> 1) > n_b4) { if (sub == 0) *bad = 1; continue; }
        for (int k = 8 * sub; k < lq; k += 128) {
            if (k + 8 <= lq) {
                uint32_t w; unsigned long long q, bb; uint4 o;
                __builtin_memcpy(&w, b4 + bo + (k >> 1), 4); __builtin_memcpy(&q, quals + so + k, 8);
                unpack8(w, q, bb, o);
                __builtin_memcpy(bases + so + k, &bb, 8); __builtin_memcpy(bq + so + k, &o, 16);
            } else {
                for (int k2 = k; k2 < lq; k2++) {
                    const unsigned byte = b4[bo + (k2 >> 1)];
                    const unsigned b0 = nt16_int((k2 & 1) ? (byte & 15) : (byte >> 4));
                    bases[so + k2] = (uint8_t)b0; bq[so + k2] = (uint16_t)(b0 | (quals[so + k2] << 8));
                }
            }
        }
    }
}
// Dense form (every read has an even length, so base g is nibble g of the packed column): eight bases per thread, wide loads and stores
__global__ void __launch_bounds__(256) k_pack_bq4_dense(const uint32_t *b4, const unsigned long long *quals, int64_t n8, unsigned long long *bases, uint4 *bq) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t w = b4[i]; const unsigned long long q = quals[i];
        unsigned long long bb; uint4 o;
        unpack8(w, q, bb, o);
        bases[i] = bb;
        bq[i] = o;
    }
}
// the dense form is only right when no read has an odd length: checked where the lengths are
__global__ void __launch_bounds__(256) k_any_odd(const int32_t *l_qseq, int64_t n, int32_t *bad) {
    int odd = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) odd |= (l_qseq[i] & 1) | (l_qseq[i] < 0);
    if (__any(odd) && (threadIdx.x & 63) == 0) *bad = 1;
}
}   // namespace
extern "C" size_t uvc_prep_compact_tmp_bytes(int64_t n) { return (size_t)(3 * ((n + SC_TILE - 1) / SC_TILE) + 3) * sizeof(long long) + 64; }
// seq_off / cigar_off / b4_off: outputs (device, n entries each) or NULL when the caller supplied that column; with bases4: bases_out (n_bases
// bytes) and bq_out (n_bases x base | qual << 8) are written here, so that no separate packing pass runs behind it
extern "C" int uvc_prep_compact(const int32_t *l_qseq, const int32_t *n_cigar, int64_t n, int64_t n_bases, const uint8_t *bases4, int64_t n_b4, const uint8_t *quals,
                                int64_t *seq_off_out, const int64_t *seq_off_in, int64_t *cigar_off_out, int64_t *b4_off, uint8_t *bases_out, uint16_t *bq_out, int32_t *bad,
                                void *tmp, size_t tmp_bytes, hipStream_t s) {
    hipError_t e = hipSuccess;
    const int ntiles = (int)((n + SC_TILE - 1) / SC_TILE);
    if ((size_t)(3 * ntiles + 3) * sizeof(long long) > tmp_bytes) return (int)hipErrorInvalidValue;
    long long *tile_sums = (long long *)tmp;
    const bool dense_form = bases4 && seq_off_out && 2 * n_b4 == n_bases && (n_bases & 7) == 0 && !(((uintptr_t)bases4) & 3) && !(((uintptr_t)quals | (uintptr_t)bases_out) & 7) && !(((uintptr_t)bq_out) & 15);
    if ((seq_off_out || cigar_off_out || (bases4 && !dense_form)) && n > 0) {
        hipLaunchKernelGGL(k_len_sums, dim3((unsigned)ntiles), dim3(SC_BLOCK), 0, s, l_qseq, n_cigar, n, tile_sums, ntiles);
        hipLaunchKernelGGL(k_tile_tops<3>, dim3(1), dim3(1024), 0, s, tile_sums, ntiles, tile_sums + (size_t)3 * ntiles);
        hipLaunchKernelGGL(k_len_apply, dim3((unsigned)ntiles), dim3(SC_BLOCK), 0, s, l_qseq, n_cigar, n, tile_sums, ntiles, seq_off_out, cigar_off_out, (bases4 && !dense_form) ? b4_off : (int64_t *)nullptr);
    }
    if (e == hipSuccess && bases4) {
        // reads back to back (offsets derived here) with 2 x bytes == bases can only be all-even lengths: nibble g is base g
        if (dense_form) {
            hipLaunchKernelGGL(k_any_odd, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, l_qseq, n, bad);
            hipLaunchKernelGGL(k_pack_bq4_dense, dim3((unsigned)std::min<int64_t>((n_bases / 8 + 255) / 256 + 1, 16384)), dim3(256), 0, s, (const uint32_t *)bases4, (const unsigned long long *)quals, n_bases / 8,
                               (unsigned long long *)bases_out, (uint4 *)bq_out);
        } else {
            hipLaunchKernelGGL(k_pack_bq4_reads, dim3((unsigned)std::min<int64_t>((n + 15) / 16, 65536)), dim3(256), 0, s, bases4, n_b4, b4_off, seq_off_out ? seq_off_out : seq_off_in, l_qseq, quals, n, n_bases, bases_out, bq_out, bad);
        }
    }
    return (int)e;
}

static inline int pos_bits_of(int64_t npos) { int b = 1; while (((int64_t)1 << b) < npos + 1) b++; return b; }
extern "C" int uvc_sort_by_pos_cls(const int32_t *d_pos, const int32_t *d_cls, int32_t beg, int pos_bits, int cls_bits, int64_t n, uint32_t *work, void *tmp, size_t tmp_bytes, hipStream_t s);
extern "C" size_t uvc_sort32_tmp_bytes(size_t n);

#define PREP_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { snprintf(errmsg, (size_t)errcap, "%s: %s", #call, hipGetErrorString(e_)); return UVCGPU_EDEVICE; } } while (0)

extern "C" int uvc_prep_reads(const UvcPrepIn *inp, const UvcParams *P, int32_t rbeg, int32_t rend, int64_t npos, UvcPrepAlloc alloc, void *ctx, hipStream_t s,
                              UvcPrepOut *out, char *errmsg, int errcap) {
    const UvcPrepIn &in = *inp;
    const int64_t n = in.n_reads;
    memset(out, 0, sizeof(*out));
    errmsg[0] = 0;
    auto A = [&](size_t bytes, int zero) -> void * { return alloc(ctx, std::max<size_t>(bytes, 8), zero); };
#define ALLOC(ptr, T, count, zero) do { ptr = (T *)A(sizeof(T) * (size_t)(count), zero); if (!ptr) { snprintf(errmsg, (size_t)errcap, "hipMalloc(%s)", #ptr); return UVCGPU_ENOMEM; } } while (0)
    const int seg_eligible = (UVC_PLATFORM_IONTORRENT != P->inferred_sequencing_platform) && rbeg >= 65536 && P->bias_thres_interfering_indel <= 10000;
    // ---- stage 1
    int32_t *new_frag, *new_fs, *is_complex, *n_p2; int64_t *gaps, *trows, *items, *ins;
    ALLOC(out->endpos, int32_t, n, 0); ALLOC(out->kind, int32_t, n, 0); ALLOC(out->dflag_of, int32_t, n, 0); ALLOC(out->frag_of, int32_t, n, 0); ALLOC(out->fs_of, int32_t, n, 0);
    ALLOC(out->table_off, int64_t, n, 0); ALLOC(out->item_off, int64_t, n, 0); ALLOC(out->gap_off, int64_t, n, 0); ALLOC(out->p2_first, int32_t, n, 0);
    ALLOC(new_frag, int32_t, n, 0); ALLOC(new_fs, int32_t, n, 0); ALLOC(is_complex, int32_t, n, 0); out->is_complex = is_complex; ALLOC(n_p2, int32_t, n, 0); ALLOC(gaps, int64_t, n, 0);
    ALLOC(trows, int64_t, n, 0); ALLOC(items, int64_t, n, 0); ALLOC(ins, int64_t, n, 0);
    Stage1 *dT1; ALLOC(dT1, Stage1, 1, 1);
    Stage2 *dT2; ALLOC(dT2, Stage2, 1, 1);
    // sized for the LARGEST scan issued below: the per-read ones (n) and the fragment-depth scan over npos + 1 positions (nf >= 65536)
    const size_t tmp_bytes = (size_t)(3 * ((std::max<int64_t>(n, npos + 2) + SC_TILE - 1) / SC_TILE) + 3) * sizeof(long long) + 64;   // tile sums of the largest three-column scan (units and fragments never outnumber the reads)
    void *tmp; ALLOC(tmp, char, tmp_bytes, 0);
    // one block per tile of SC_TILE reads; the eight offset columns come from the tile sums in two more launches (k_tile_tops, k_facts_apply)
    const int ntiles = (int)((n + SC_TILE - 1) / SC_TILE);
    long long *tile_sums; ALLOC(tile_sums, long long, (size_t)8 * ntiles + 8, 0);
    long long *col_totals = tile_sums + (size_t)8 * ntiles;
    ALLOC(out->complex_ids, int32_t, n, 0);   // (sized by the reads: the number of InDel reads is only known behind the scan that fills it)
    hipLaunchKernelGGL(k_read_facts, dim3((unsigned)ntiles), dim3(SC_BLOCK), 0, s, in, rbeg, rend, seg_eligible, out->endpos, out->kind, out->dflag_of, new_frag, new_fs, is_complex, n_p2, gaps, trows, items, ins, dT1, tile_sums, ntiles);
    hipLaunchKernelGGL(k_tile_tops<8>, dim3(1), dim3(1024), 0, s, tile_sums, ntiles, col_totals);
    hipLaunchKernelGGL(k_facts_apply, dim3((unsigned)ntiles), dim3(SC_BLOCK), 0, s, n, tile_sums, ntiles, col_totals, out->kind, new_frag, new_fs, n_p2, trows, items, gaps,
                       out->frag_of, out->fs_of, out->p2_first, out->table_off, out->item_off, out->gap_off, out->complex_ids, dT1);
    // up to three prefix sums at once (k_cols_sums, k_tile_tops<3>, k_cols_apply); `tmp` holds the tile sums
    auto scan3 = [&](ScanCols C) {
        int64_t nmax = 0; for (int c = 0; c < 3; c++) nmax = std::max(nmax, C.n[c]);
        if (nmax <= 0) return;
        const int nt = (int)((nmax + SC_TILE - 1) / SC_TILE);
        long long *ts = (long long *)tmp;
        hipLaunchKernelGGL(k_cols_sums, dim3((unsigned)nt), dim3(SC_BLOCK), 0, s, C, ts, nt);
        hipLaunchKernelGGL(k_tile_tops<3>, dim3(1), dim3(1024), 0, s, ts, nt, ts + (size_t)3 * nt);
        hipLaunchKernelGGL(k_cols_apply, dim3((unsigned)nt), dim3(SC_BLOCK), 0, s, C, ts, nt);
    };
    auto col = [](ScanCols &C, int c, const void *in, void *out, int64_t n, int in64, int out64, int inclusive) { C.in[c] = in; C.out[c] = out; C.n[c] = n; C.in64[c] = in64; C.out64[c] = out64; C.inclusive[c] = inclusive; };
    Stage1 T1;
    PREP_HIP(hipMemcpyAsync(&T1, dT1, sizeof(T1), hipMemcpyDeviceToHost, s));
    PREP_HIP(hipStreamSynchronize(s));
    if (T1.err) {
        const char *m = T1.err == 1 ? "read offsets out of range" : T1.err == 2 ? "unsupported CIGAR op (process_cigar throws, main_conversion.hpp:902-916)" : T1.err == 3 ? "CIGAR query length != l_qseq"
                      : T1.err == 4 ? "read outside region" : "fam_id / fam_strand out of range";
        snprintf(errmsg, (size_t)errcap, "%s", m);
        return T1.err == 2 ? UVCGPU_EUNSUPPORTED : UVCGPU_EINVAL;
    }
    out->n_frags = T1.n_frags; out->n_fs = T1.n_fs; out->n_complex = T1.n_complex; out->n_simple = (int32_t)(n - T1.n_complex); out->n_p2 = T1.n_p2; out->table_rows = T1.table_rows;
    out->item_slots = T1.item_slots; out->gap_slots = T1.gap_slots; out->ins_total = T1.ins_total; out->max_aln_span = std::max(T1.max_aln_span, 1); out->any_amplicon = T1.any_amplicon;
    out->max_p2_span = std::max(T1.max_p2_span, 1);
    out->p2_off[0] = 0; for (int c = 0; c < 4; c++) out->p2_off[c + 1] = out->p2_off[c] + T1.p2_cls[c];
    // ---- stage 2
    const int32_t nf = T1.n_frags, nu = T1.n_fs;
    int32_t *frag_first, *fs_first_frag, *fam_fs, *sweep_flag, *sweep_rank, *generic_flag, *generic_rank, *dup_flag, *dup_rank; int64_t *gen_span, *work_off, *dup_span, *dup_off_all;
    ALLOC(frag_first, int32_t, nf + 1, 0); ALLOC(fs_first_frag, int32_t, nu + 1, 0); ALLOC(fam_fs, int32_t, (size_t)in.n_fams * 2, 0);
    PREP_HIP(hipMemsetAsync(fam_fs, 0xFF, sizeof(int32_t) * std::max<size_t>((size_t)in.n_fams * 2, 1), s));
    ALLOC(sweep_flag, int32_t, nf, 0); ALLOC(sweep_rank, int32_t, nf, 0); ALLOC(generic_flag, int32_t, nu, 0); ALLOC(generic_rank, int32_t, nu, 0); ALLOC(dup_flag, int32_t, nu, 0); ALLOC(dup_rank, int32_t, nu, 0);
    ALLOC(gen_span, int64_t, nu, 0); ALLOC(work_off, int64_t, nu, 0); ALLOC(dup_span, int64_t, nu, 0); ALLOC(dup_off_all, int64_t, nu, 0);
    ALLOC(out->frags, FragRec, nf, 0); ALLOC(out->fss, FsRec, nu, 0); ALLOC(out->frag_beg, int32_t, nf, 0); ALLOC(out->frag_strand, int32_t, nf, 0);
    hipLaunchKernelGGL(k_mark_first, dim3(nblk(n, 256)), dim3(256), 0, s, in, new_frag, new_fs, out->frag_of, out->fs_of, frag_first, fs_first_frag, fam_fs, nf, nu, dT2);
    hipLaunchKernelGGL(k_build_frags, dim3(std::min(nblk(nf, 256), 1024u)), dim3(256), 0, s, in, *P, rend, out->endpos, out->kind, out->fs_of, out->dflag_of, frag_first, nf, out->frags, sweep_flag, out->frag_beg, out->frag_strand, dT2);
    hipLaunchKernelGGL(k_build_units, dim3(nblk(nu, 256)), dim3(256), 0, s, in, *P, rend, out->endpos, out->dflag_of, frag_first, fs_first_frag, fam_fs, nu, out->fss, generic_flag, gen_span, dT2);
    { ScanCols C; memset(&C, 0, sizeof(C)); col(C, 0, generic_flag, generic_rank, nu, 0, 0, 0); col(C, 1, gen_span, work_off, nu, 1, 1, 0); col(C, 2, sweep_flag, sweep_rank, nf, 0, 0, 0); scan3(C); }
    // generic_fs needs its size before k_units_post can fill it: the upper bound n_fs is small (4 B per unit)
    ALLOC(out->generic_fs, int32_t, nu, 0);
    hipLaunchKernelGGL(k_units_post, dim3(nblk(nu, 256)), dim3(256), 0, s, rend, nu, out->fss, generic_rank, work_off, out->generic_fs, dup_flag, dup_span);
    { ScanCols C; memset(&C, 0, sizeof(C)); col(C, 0, dup_flag, dup_rank, nu, 0, 0, 0); col(C, 1, dup_span, dup_off_all, nu, 1, 1, 0); scan3(C); }
    ALLOC(out->dup_units, int32_t, nu, 0); ALLOC(out->dup_off, int64_t, nu, 0);
    hipLaunchKernelGGL(k_compact_dups, dim3(nblk(nu, 256)), dim3(256), 0, s, nu, dup_flag, dup_rank, dup_off_all, out->dup_units, out->dup_off);
    ALLOC(out->sweep_frags, int32_t, nf, 0);
    int32_t *depth = nullptr;
    if (nf >= 65536) { ALLOC(depth, int32_t, npos + 2, 1); }   // fragment depth bound: below 65 536 fragments the count itself bounds it
    hipLaunchKernelGGL(k_frags_post, dim3(nblk(nf, 256)), dim3(256), 0, s, nf, out->frags, out->fss, sweep_flag, sweep_rank, out->sweep_frags, rbeg, depth);
    if (depth) {
        { ScanCols C; memset(&C, 0, sizeof(C)); col(C, 0, depth, depth, npos + 1, 0, 0, 1); scan3(C); }
        hipLaunchKernelGGL(k_max_i32, dim3(256), dim3(256), 0, s, depth, npos + 1, &dT2->max_frag_depth);
    }
    hipLaunchKernelGGL(k_stage2_totals, dim3(1), dim3(1), 0, s, nu, nf, generic_rank, generic_flag, work_off, gen_span, dup_rank, dup_flag, dup_off_all, dup_span, sweep_rank, sweep_flag, dT2);
    // the P2 work-list entries (unsorted) while the totals travel
    const int64_t np2 = T1.n_p2;
    ALLOC(out->p2_aln, int32_t, np2, 0); ALLOC(out->p2_beg, int32_t, np2, 0); ALLOC(out->p2_end, int32_t, np2, 0); ALLOC(out->p2_qb, int32_t, np2, 0); ALLOC(out->p2_cls, int32_t, np2, 0);
    hipLaunchKernelGGL(k_p2_entries, dim3(nblk(n, 256)), dim3(256), 0, s, in, n_p2, out->p2_first, out->p2_aln, out->p2_beg, out->p2_end, out->p2_qb, out->p2_cls);
    Stage2 T2;
    PREP_HIP(hipMemcpyAsync(&T2, dT2, sizeof(T2), hipMemcpyDeviceToHost, s));
    PREP_HIP(hipStreamSynchronize(s));
    if (T2.err) { snprintf(errmsg, (size_t)errcap, "reads of one (fam_id, fam_strand) are not contiguous"); return UVCGPU_EINVAL; }
    out->n_generic = T2.n_generic; out->n_dup = T2.n_dup; out->n_sweep = T2.n_sweep; out->work = T2.work; out->dup_work = T2.dup_work;
    out->max_frag_span = std::max(T2.max_frag_span, 1); out->max_unit_span = std::max(T2.max_unit_span, 1); out->max_unit_frags = T2.max_unit_frags; out->n_frag_strand0 = T2.n_frag_strand0;
    out->max_frag_depth = (nf < 65536 ? nf : T2.max_frag_depth);
    // the generic units ordered by begin, for the position-window family kernels (stable: equal begins keep the unit order)
    ALLOC(out->generic_sorted, int32_t, std::max(T2.n_generic, 1), 0);
    if (T2.n_generic > 0) {
        const int32_t ng = T2.n_generic;
        int32_t *beg_of, *zero; uint32_t *work; void *stmp;
        const size_t sb = uvc_sort32_tmp_bytes((size_t)ng);
        ALLOC(beg_of, int32_t, ng, 0); ALLOC(zero, int32_t, ng, 0); ALLOC(work, uint32_t, 4 * (size_t)ng, 0); ALLOC(stmp, char, sb + 16, 0);
        hipLaunchKernelGGL(k_unit_keys, dim3(nblk(ng, 256)), dim3(256), 0, s, out->fss, out->generic_fs, ng, beg_of, zero);
        if (uvc_sort_by_pos_cls(beg_of, zero, rbeg, pos_bits_of(npos), 0, ng, work, stmp, sb, s) != 0) { snprintf(errmsg, (size_t)errcap, "device sort of the units failed"); return UVCGPU_EDEVICE; }
        hipLaunchKernelGGL(k_take_units, dim3(nblk(ng, 256)), dim3(256), 0, s, work + 3 * (size_t)ng, out->generic_fs, ng, out->generic_sorted);
    }
    PREP_HIP(hipGetLastError());
    return 0;
#undef ALLOC
}
