// uvc_kernels_acc.hip -- hand-written gfx950 kernels for the accumulate half of the UVC hot path.
//
// Decomposition.  The reference walks every read 5 times and scatters `+=`
// into ~5.5 KB of per-position state (main.hpp:2543-3594).  Here the loops are turned inside out:
// a wavefront owns 64 consecutive reference positions (one lane = one position), loops over the
// alignments / fragments that overlap its window (wave-uniform control flow; the per-read scalars of
// 64 records are loaded one per lane and broadcast with v_readlane; base/qual bytes come through a
// bounds-checked buffer descriptor, coalesced across lanes) and keeps the accumulators of the two
// symbols that matter at a position -- the reference base and LINK_M -- in registers.  Everything
// rare takes a sparse path: mismatching bases through per-wave LDS queues and a lane-per-item
// kernel, InDel symbols through wave-per-read kernels, rare symbols of the fragment pass through
// fire-and-forget atomics.  All updates are integer adds or maxima, so any order gives
// bit-identical results.  The kernels are issue-bound:
// the code below is written to minimise VALU + SALU instructions per (read, 64 positions).
//
//   k_pack_bq        per read base  base | qual << 8
//   k_aln_prelude    per read       nge/ngo/clips, xm1500, bm1500s, penalties, eligibility   main.hpp:1795-1885
//   k_build_p2list   per entry      P2 work list (simple reads + M runs of InDel reads), 4 orientation classes
//   k_correct_bq     per read       apply_bq_err_correction3 (on request)                     grouping.cpp:459-543
//   k_prep_fast      per position   P1 for simple reads                                       main.hpp:924-1204
//   k_prep_slow      wave per read  P1 for reads with InDels (atomics)
//   k_thres          per position   P1b, also edits rtr.indelphred                            main.hpp:1206-1299
//   k_p2_fast<L,B>   per position   P2 + dealwith_segbias, LINK_M / read base instantiations   main.hpp:1360-1595, 1762-2296
//   k_p2_mism        per queued base  the same for bases that differ from the reference
//   k_p2_slow<BIAS>  wave per read  CIGAR walk of InDel reads: P2 items (BIAS) or the BASE_QUALITY_MAX contribution table
//   k_p2_items       wave per read  applies the items
//   k_fragstat_*     per fragment   covered / near-mutation position counts, M runs of InDel fragments   main.hpp:2738-2756
//   k_frag_generic   wave per fragment chunk  P3 at the positions next to InDels / of fragments with > 2 alignments
//   k_frag           per position   P3 + P3b, and P4/P5 of singleton families                 main.hpp:2620-2830, 2832-3594
//   k_fam_stat/p4/p5 per (family-strand unit, position): multi-fragment families              main.hpp:2883-3522
//   k_duplex         per (duplex family, position)                                            main.hpp:3523-3550
//   k_p5b            per (position, strand)  bucket -> quality for family consensus           main.hpp:3552-3591
//   k_gap_keys/alleles/rows  per (family, InDel position)  the allele-keyed counters of the InDel symbols   main.hpp:2710-2717, 3196-3546
// uvc_launch_accumulate at the end of the file orders them on two streams.
#include "uvc_device.h"

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
DEV int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Workgroups are dealt to the 8 XCDs round-robin (block b -> XCD b % 8) and each XCD has its own L2.  Position-window kernels
// therefore give XCD k one contiguous eighth of the region, so that neighbouring windows, which read the same alignments,
// share an L2.  Placement only affects speed, never results.
DEV int xcd_block() {
    const int nb = gridDim.x, b = blockIdx.x, q = nb >> 3, r = nb & 7, k = b & 7, i = b >> 3;
    return k * q + (k < r ? k : r) + i;
}

DEV int bcast(int v, int j) { return __builtin_amdgcn_readlane(v, j); }   // j must be wave-uniform
#ifdef UVC_FRAG_COARSE   // debug build (make XFLAGS=-DUVC_FRAG_COARSE): s_memtime at the three stages of k_frag, printed for a few waves
#define COARSE_T(v) unsigned long long v; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory");
#else
#define COARSE_T(v)
#endif
// base | qual << 8 of read byte `idx` through a raw buffer descriptor: out-of-range indices (lanes outside the read) return 0
DEV __amdgpu_buffer_rsrc_t bq_rsrc(const RegionDev &R) { return __builtin_amdgcn_make_buffer_rsrc((void *)R.bq, 0, (int)R.bq_bytes, 0x00020000); }
DEV int bq_load(__amdgpu_buffer_rsrc_t rs, int idx) { return (int)__builtin_amdgcn_raw_buffer_load_b16(rs, idx << 1, 0, 0); }
struct Chunk16 { int v[16]; };
DEV void load_chunk16(const FastRec *base, int kk, int hi, Chunk16 &c) {   // lane <- record kk (zeros past the end: rend = 0 never overlaps)
    if (kk < hi) {
        const int4 *q = (const int4 *)(base + kk);
        const int4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
        c.v[0] = a0.x; c.v[1] = a0.y; c.v[2] = a0.z; c.v[3] = a0.w; c.v[4] = a1.x; c.v[5] = a1.y; c.v[6] = a1.z; c.v[7] = a1.w;
        c.v[8] = a2.x; c.v[9] = a2.y; c.v[10] = a2.z; c.v[11] = a2.w; c.v[12] = a3.x; c.v[13] = a3.y; c.v[14] = a3.z; c.v[15] = a3.w;
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) c.v[i] = 0;
    }
}
DEV int lower_bound_frec(const FastRec *a, int n, int key) {   // first index with a[i].pos >= key
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid].pos < key) lo = mid + 1; else hi = mid; }
    return lo;
}
// Window index: for every 64-position window w and every begin-sorted work list, the records a wave of window w walks:
// [first record with begin >= window begin - longest span + 1, first record with begin >= window end).  A binary search per wave and list is
// ~20 dependent memory round trips (8 lists' bounds in k_p2_fast: an eighth of a wave's life); the table costs one load each.
// Lists: 0 = frec (k_prep_fast), 1..4 = the four sub-lists of frec2 (k_p2_fast), 5..6 = the two strands of ffast (k_frag),
// 7 = the generic family-strand units by begin (the window kernels of the family passes: two loads per search step there).
#define WIN_LISTS 8
DEV int win_lo(const RegionDev &R, int list, int w) { return R.win[((size_t)list * 2) * R.nwin + w]; }
DEV int win_hi(const RegionDev &R, int list, int w) { return R.win[((size_t)list * 2 + 1) * R.nwin + w]; }
__global__ void __launch_bounds__(256) k_win_index(RegionDev R, int list_beg, int list_end) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_list = 2 * (int64_t)R.nwin;
    if (t >= per_list * (list_end - list_beg)) return;
    const int list = list_beg + (int)(t / per_list), which = (int)((t % per_list) / R.nwin), w = (int)(t % R.nwin);
    const char *base; int stride, lo, hi, span;
    if (list == 0) { base = (const char *)&R.frec[0].pos; stride = sizeof(FastRec); lo = 0; hi = R.n_fast; span = R.max_aln_span; }
    else if (list <= 4) { base = (const char *)&R.frec2[0].pos; stride = sizeof(FastRec); lo = R.p2_off[list - 1]; hi = R.p2_off[list]; span = R.max_p2_span; }
    else if (list <= 6) { base = (const char *)&R.ffast[0].beg; stride = sizeof(FragFast); lo = R.frag_off[list - 5]; hi = R.frag_off[list - 4]; span = R.max_frag_span; }
    else { base = nullptr; stride = 0; lo = 0; hi = R.n_generic_fs; span = R.max_unit_span; }
    const int w0 = R.beg + 64 * w, key = (which == 0 ? w0 - span + 1 : w0 + 64);
    if (list <= 6) { while (lo < hi) { const int mid = (lo + hi) >> 1; if (*(const int32_t *)(base + (size_t)mid * stride) < key) lo = mid + 1; else hi = mid; } }
    else { while (lo < hi) { const int mid = (lo + hi) >> 1; if (R.fss[R.generic_sorted[mid]].beg < key) lo = mid + 1; else hi = mid; } }
    R.win[((size_t)list * 2 + which) * R.nwin + w] = lo;
}
DEV int lower_bound_pos(const AlnRec *a, int n, int key) {   // first index with a[i].pos >= key
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid].pos < key) lo = mid + 1; else hi = mid; }
    return lo;
}

struct SegAcc {   // SegFormatInfoSet (main_conversion.hpp:645-691) + the VQ a1/a2 sums + bqsum of ONE symbol at ONE position
    int s[UVC_NSEG32];
    long long l[UVC_NSEG64];
    int lp[UVC_NSEG64];   // k_p2_fast: 32-bit partial sums of the 64-bit counters over one chunk of 64 reads (each term < 2^25), folded into l[] per chunk
    int a1BQf, a1BQr, a2BQf, a2BQr, bq;
    DEV void zero() { for (int i = 0; i < UVC_NSEG32; i++) s[i] = 0; for (int i = 0; i < UVC_NSEG64; i++) { l[i] = 0; lp[i] = 0; } a1BQf = a1BQr = a2BQf = a2BQr = bq = 0; }
    DEV void fold() { for (int i = 0; i < UVC_NSEG64; i++) { l[i] += (long long)lp[i]; lp[i] = 0; } }
};

DEV void seg_flush(const RegionDev &R, const SegAcc &A, int sym, int64_t x) {
    mark_sym(R, sym, x);   // (with RegionDev::occ: scoring looks at the marked symbols of a position only)
#pragma unroll
    for (int f = 0; f < UVC_NSEG32; f++) if (A.s[f]) atomicAdd(&S32(R, f, sym, x), A.s[f]);
#pragma unroll
    for (int f = 0; f < UVC_NSEG64; f++) if (A.l[f]) add64(&S64(R, f, sym, x), A.l[f]);
    if (A.a1BQf) atomicAdd(&VQP(R, UVC_VQ_a1BQf, sym, x), A.a1BQf);
    if (A.a1BQr) atomicAdd(&VQP(R, UVC_VQ_a1BQr, sym, x), A.a1BQr);
    if (A.a2BQf) atomicAdd(&VQP(R, UVC_VQ_a2BQf, sym, x), A.a2BQf);
    if (A.a2BQr) atomicAdd(&VQP(R, UVC_VQ_a2BQr, sym, x), A.a2BQr);
    if (A.bq) atomicAdd(&BQS(R, sym, x), A.bq);
}

// first writer of (sym, x) after the planes were zeroed: plain stores, no read-modify-write
DEV void seg_store(const RegionDev &R, const SegAcc &A, int sym, int64_t x) {
#pragma unroll
    for (int f = 0; f < UVC_NSEG32; f++) if (A.s[f]) S32(R, f, sym, x) = A.s[f];
#pragma unroll
    for (int f = 0; f < UVC_NSEG64; f++) if (A.l[f]) S64(R, f, sym, x) = A.l[f];
    if (A.a1BQf) VQP(R, UVC_VQ_a1BQf, sym, x) = A.a1BQf;
    if (A.a1BQr) VQP(R, UVC_VQ_a1BQr, sym, x) = A.a1BQr;
    if (A.a2BQf) VQP(R, UVC_VQ_a2BQf, sym, x) = A.a2BQf;
    if (A.a2BQr) VQP(R, UVC_VQ_a2BQr, sym, x) = A.a2BQr;
    if (A.bq) BQS(R, sym, x) = A.bq;
}

struct PosThres { int t[UVC_NTHRES]; };

// per-read quantities that dealwith_segbias needs; all wave-uniform in the fast kernels
struct SegRead {
    int pos, rend, flag, mapq, isize, frag_pos_L, frag_pos_R, xm1500, clip_cnt, dflag;
    long long baq_pos, baq_last, baq2_last;
};

DEV SegRead make_segread(const RegionDev &R, const AlnRec &a) {
    SegRead r;
    r.pos = a.pos; r.rend = a.rend; r.flag = a.flag; r.mapq = a.mapq; r.isize = a.isize;
    r.frag_pos_L = imin(a.pos, a.mpos); r.frag_pos_R = r.frag_pos_L + abs(a.isize);
    r.xm1500 = a.xm1500; r.clip_cnt = a.clip_cnt; r.dflag = a.dflag;
    r.baq_pos = a.baq_pos; r.baq_last = a.baq_last; r.baq2_last = a.baq2_last;
    return r;
}

// update_bidirectional_bias, main.hpp:1318-1358
DEV void bidir(int &LP1, int &LP2, int &RP1, int &RP2, long long &LPL, long long &RPL, int L1, int L2, int R1, int R2, long long nl, long long nr, bool tier2, int n_indel) {
    if (nl + n_indel >= L1) LP1 += 1;
    if ((nl + n_indel >= L2) && tier2) LP2 += 1;
    if (nr >= R1) RP1 += 1;
    if ((nr >= R2) && tier2) RP2 += 1;
    LPL += nl; RPL += nr;
}

// dealwith_segbias<isGap>, main.hpp:1360-1595 (COMPILATION_ENABLE_XMGOT == 0)
template <bool isGap>
DEV void segbias(SegAcc &A, const UvcParams &P, const SegRead &r, const PosThres &T, int rpos, long long baq_p, long long baq2_p,
                 int bq, int bm1500, int cigar_op, int indel_len, int dist_to_interfering_indel) {
    const bool is_assay_amplicon = ((r.dflag & 0x4) || ((P.primerlen > 0) && !(0x2 & P.primer_flag)));
    const bool normal_filter_primers = (P.tn_is_paired && (0x1 & P.primer_flag));
    const bool is_assay_UMI = (r.dflag & 0x1);
    const int seg_l_baq1 = (int)(baq_p - r.baq_pos + 1);
    const int _seg_r_baq = (int)(r.baq_last - baq_p + 1);
    const int seg_r_baq1 = (isGap ? (int)lmin((long long)_seg_r_baq, r.baq2_last - baq2_p + 7) : _seg_r_baq);
    const int seg_l_nbases = rpos - r.pos + 1;
    const int seg_r_nbases = r.rend - rpos;
    const bool is_high_readlen = (P.central_readlen >= P.microadjust_median_readlen_thres);
    const int seg_l_baq = (is_high_readlen ? seg_l_baq1 : imax(seg_l_baq1, seg_l_nbases * P.microadjust_BAQ_per_base_x1024 / 1024));
    const int seg_r_baq = (is_high_readlen ? seg_r_baq1 : imax(seg_r_baq1, seg_r_nbases * P.microadjust_BAQ_per_base_x1024 / 1024));
    const int frag_l_nbases2 = ((r.isize != 0) ? imin(rpos - r.frag_pos_L + 1, MAX_INSERT_SIZE) : MAX_INSERT_SIZE);
    const int frag_r_nbases2 = ((r.isize != 0) ? imin(r.frag_pos_R - rpos, MAX_INSERT_SIZE) : MAX_INSERT_SIZE);
    const bool is_normal = ((r.isize != 0) || (0 == (r.flag & 0x1)));
    const bool isrc = (r.flag & 0x10) != 0;
    const bool strand = ((r.flag & 0x81) == 0x81) ? ((r.flag & 0x20) != 0) : ((r.flag & 0x10) != 0);   // bam_get_strand, common.hpp:89

    if (isrc) { A.a1BQr += bq; A.a2BQr += bq * bq / SQR_QUAL_DIV; } else { A.a1BQf += bq; A.a2BQf += bq * bq / SQR_QUAL_DIV; }
    A.s[UVC_S_aMQs] += r.mapq;
    A.s[UVC_S_aDPff] += (!strand && !isrc); A.s[UVC_S_aDPfr] += (!strand && isrc);
    A.s[UVC_S_aDPrf] += (strand && !isrc);  A.s[UVC_S_aDPrr] += (strand && isrc);
    if (imin(dist_to_interfering_indel, imin(seg_l_nbases, seg_r_nbases)) >= P.bias_thres_interfering_indel) A.s[UVC_S_aP3] += 1;
    if (0 == r.clip_cnt) A.s[UVC_S_aNC] += 1;
    if (isrc) A.l[UVC_S64_aLIT] += ((r.isize != 0) ? frag_l_nbases2 : 0);
    else      A.l[UVC_S64_aRIT] += ((r.isize != 0) ? frag_r_nbases2 : 0);

    const int _LPxT = T.t[UVC_T_aLPxT], RPxT = T.t[UVC_T_aRPxT];
    const int LPxT = (isGap ? _LPxT : imin(_LPxT, RPxT));
    const bool is_far_from_edge = (seg_l_nbases + ((C_INS == cigar_op) ? (int)nnminus(indel_len, P.microadjust_nobias_pos_indel_maxlen) : 0) >= LPxT) && (seg_r_nbases >= RPxT);
    const int thres_highBAQ = P.bias_thres_highBAQ + (isGap ? 0 : 3);
    const bool is_unaffected_by_edge = (seg_l_baq >= thres_highBAQ && seg_r_baq >= thres_highBAQ);
    const int min_dist2iend = ((r.flag & 0x1) ? imin(frag_l_nbases2, frag_r_nbases2) : (isrc ? seg_r_nbases : seg_l_nbases));
    if (is_far_from_edge && is_unaffected_by_edge && (min_dist2iend > P.primerlen2 || !is_assay_amplicon)) A.s[UVC_S_aP1] += 1;
    if (is_assay_UMI || !is_assay_amplicon) A.s[UVC_S_aP2] += 1;

    int ampfact2 = 100;
    if (bq < P.bias_thres_PFBQ1) ampfact2 = 100 * (bq * bq) / (P.bias_thres_PFBQ1 * P.bias_thres_PFBQ1);
    A.s[UVC_S_aPF1] += (isGap ? imin(100, ampfact2) : (100 * ampfact2 / 100));
    ampfact2 = 100;
    if (bq < P.bias_thres_PFBQ2) ampfact2 = 100 * (bq * bq) / (P.bias_thres_PFBQ2 * P.bias_thres_PFBQ2);
    A.s[UVC_S_aPF2] += (isGap ? imin(100, ampfact2) : (100 * ampfact2 / 100));
    if (!isGap) {
        A.s[UVC_S_a2XM2] += (r.xm1500 > 20 ? (100 * (20 * 20) / (r.xm1500 * r.xm1500)) : 100);
        A.s[UVC_S_a2BM2] += (bm1500 > 20 ? (100 * (20 * 20) / (bm1500 * bm1500)) : 100);
    }
    if (((!isGap) && bq >= P.bias_thres_highBQ) || (isGap && dist_to_interfering_indel >= P.bias_thres_interfering_indel)) {
        const bool tier2 = (isGap || bq >= P.bias_thres_highBQ);
        if (is_far_from_edge) {
            long long LPL = 0, RPL = 0;
            bidir(A.s[UVC_S_aLP1], A.s[UVC_S_aLP2], A.s[UVC_S_aRP1], A.s[UVC_S_aRP2], LPL, RPL,
                  T.t[UVC_T_aLP1t], T.t[UVC_T_aLP2t], T.t[UVC_T_aRP1t], T.t[UVC_T_aRP2t], seg_l_nbases, seg_r_nbases, tier2, indel_len);
            A.s[UVC_S_aLPL] += (int)LPL; A.s[UVC_S_aRPL] += (int)RPL;
        }
        if (is_unaffected_by_edge) {
            bidir(A.s[UVC_S_aLB1], A.s[UVC_S_aLB2], A.s[UVC_S_aRB1], A.s[UVC_S_aRB2], A.l[UVC_S64_aLBL], A.l[UVC_S64_aRBL],
                  P.bias_thres_BAQ1, P.bias_thres_BAQ2, P.bias_thres_BAQ1, P.bias_thres_BAQ2, seg_l_baq, seg_r_baq, tier2, 0);
        }
        A.s[UVC_S_aBQ2] += 1;
    }
    const bool mate_ok = ((0 == (r.flag & 0x8)) || (0 == (r.flag & 0x1)));
    const bool is_l_nonbiased = (mate_ok && seg_l_nbases > seg_r_nbases);
    const bool is_r_nonbiased = (mate_ok && seg_l_nbases < seg_r_nbases);
    const bool pos_good = ((!is_assay_amplicon) || (!normal_filter_primers) || (is_far_from_edge && is_unaffected_by_edge));
    if (isrc) {
        const int d = frag_l_nbases2;
        if ((d >= T.t[UVC_T_aLI1t]) && (d <= T.t[UVC_T_aLI1T] || isGap) && (is_normal || (isGap && is_l_nonbiased))) A.s[UVC_S_aLI1] += 1;
        if ((d >= T.t[UVC_T_aLI2t]) && (d <= T.t[UVC_T_aLI2T] || isGap) && (is_normal || (isGap && is_l_nonbiased))) { if (pos_good) A.s[UVC_S_aLI2] += 1; }
        if (pos_good) A.s[UVC_S_aLIr] += 1;
    } else {
        const int d = frag_r_nbases2;
        if ((d >= T.t[UVC_T_aRI1t]) && (d <= T.t[UVC_T_aRI1T] || isGap) && (is_normal || (isGap && is_r_nonbiased))) A.s[UVC_S_aRI1] += 1;
        if ((d >= T.t[UVC_T_aRI2t]) && (d <= T.t[UVC_T_aRI2T] || isGap) && (is_normal || (isGap && is_r_nonbiased))) { if (pos_good) A.s[UVC_S_aRI2] += 1; }
        if (pos_good) A.s[UVC_S_aRIf] += 1;
    }
}

// Lane masks.  A per-lane condition is kept as the 64-bit ballot of its compare (an SGPR pair), conditions are combined
// on the scalar unit, and a counter takes the mask as the carry-in of one v_addc: one vector instruction per counter.
typedef unsigned long long wmask;
#define BAL(c) __builtin_amdgcn_ballot_w64(c)
DEV wmask umask(bool b) { return b ? ~0ull : 0ull; }   // wave-uniform condition
DEV void addm(int &acc, wmask m) { asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(acc) : "s"(m) : "vcc"); }
DEV int selm(wmask m, int v) { int r; asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m)); return r; }

// dealwith_segbias<true> for LINK_M (lanes with hasL) and dealwith_segbias<false> for the read base (lanes with hasB) at one
// position of a simple alignment: cigar_op = M, indel_len = 0, dist_to_interfering_indel = 10000 (main.hpp:1360-1595).
// Everything that does not depend on the symbol is computed once for both.
// ISRC / STRAND (bam_get_strand, common.hpp:89) are compile-time: the P2 work list is split by them, so the direction-specific
// counters are fixed registers in each instantiation and the loop has no branch on the read's orientation.
// PLAIN: the region holds no amplicon-flagged family, no primer length is set and the reads are long (is_high_readlen), so the amplicon arms
// and the microadjust arm are compiled out.
template <bool ISRC, bool STRAND, bool PLAIN>
DEV void segbias_simple(SegAcc &AL, SegAcc &AB, const UvcParams &P, const SegRead &r, const PosThres &T, int rpos, long long baq_p, long long baq2_p,
                        bool hasL, bool hasB, int bqL, int bqB, int xm_inc, int bm_inc, const int *amp1, const int *amp2) {
    const bool amplicon = (PLAIN ? false : ((r.dflag & 0x4) || ((P.primerlen > 0) && !(0x2 & P.primer_flag))));
    const bool normal_filter_primers = (P.tn_is_paired && (0x1 & P.primer_flag));
    const bool is_assay_UMI = (r.dflag & 0x1);
    constexpr bool isrc = ISRC, strand = STRAND;
    const bool is_normal = ((r.isize != 0) || (0 == (r.flag & 0x1)));
    const bool mate_ok = ((0 == (r.flag & 0x8)) || (0 == (r.flag & 0x1)));
    // PLAIN also means is_high_readlen (main.hpp:1404): the microadjust arm (two full-width integer multiplies per read and wave, quarter rate) is compiled out
    const bool hrl = (PLAIN ? true : (P.central_readlen >= P.microadjust_median_readlen_thres));
    const int l_nb = rpos - r.pos + 1, r_nb = r.rend - rpos;
    // differences of BAQ prefix sums: truncation to 32 bits commutes with the subtraction
    const int l_baq1 = (int)((unsigned)baq_p - (unsigned)r.baq_pos + 1u);
    const int r_baq1B = (int)((unsigned)r.baq_last - (unsigned)baq_p + 1u);
    const int r_baq1L = imin(r_baq1B, (int)((unsigned)r.baq2_last - (unsigned)baq2_p + 7u));
    const int kb = P.microadjust_BAQ_per_base_x1024;
    const int l_baq = (hrl ? l_baq1 : imax(l_baq1, l_nb * kb / 1024));
    const int r_baqB = (hrl ? r_baq1B : imax(r_baq1B, r_nb * kb / 1024));
    const int r_baqL = (hrl ? r_baq1L : imax(r_baq1L, r_nb * kb / 1024));
    const bool has_isize = (r.isize != 0);
    const int fl2 = (has_isize ? imin(rpos - r.frag_pos_L + 1, MAX_INSERT_SIZE) : MAX_INSERT_SIZE);
    const int fr2 = (has_isize ? imin(r.frag_pos_R - rpos, MAX_INSERT_SIZE) : MAX_INSERT_SIZE);
    // masks
    const wmask m_p3 = BAL(imin(10000, imin(l_nb, r_nb)) >= P.bias_thres_interfering_indel);
    const int LPxT_L = T.t[UVC_T_aLPxT], RPxT = T.t[UVC_T_aRPxT], LPxT_B = imin(LPxT_L, RPxT);
    const wmask m_rfar = BAL(r_nb >= RPxT);
    const wmask m_farL = BAL(l_nb >= LPxT_L) & m_rfar, m_farB = BAL(l_nb >= LPxT_B) & m_rfar;
    const int hb = P.bias_thres_highBAQ;
    const wmask m_unaffL = BAL(l_baq >= hb) & BAL(r_baqL >= hb), m_unaffB = BAL(l_baq >= hb + 3) & BAL(r_baqB >= hb + 3);
    const int min_dist2iend = ((r.flag & 0x1) ? imin(fl2, fr2) : (isrc ? r_nb : l_nb));
    const wmask m_iend = umask(!amplicon) | BAL(min_dist2iend > P.primerlen2);
    const wmask m_lp1 = BAL(l_nb >= T.t[UVC_T_aLP1t]), m_lp2 = BAL(l_nb >= T.t[UVC_T_aLP2t]);
    const wmask m_rp1 = BAL(r_nb >= T.t[UVC_T_aRP1t]), m_rp2 = BAL(r_nb >= T.t[UVC_T_aRP2t]);
    const wmask m_lb1 = BAL(l_baq >= P.bias_thres_BAQ1), m_lb2 = BAL(l_baq >= P.bias_thres_BAQ2);
    const wmask m_goodU = umask((!amplicon) || (!normal_filter_primers));
    const wmask m_goodL = m_goodU | (m_farL & m_unaffL), m_goodB = m_goodU | (m_farB & m_unaffB);
    // insert-end thresholds of this read's direction
    const int d = (isrc ? fl2 : fr2);
    wmask m_i1lo, m_i1hi, m_i2lo, m_i2hi, m_nonb;
    if (isrc) {
        m_i1lo = BAL(d >= T.t[UVC_T_aLI1t]); m_i1hi = BAL(d <= T.t[UVC_T_aLI1T]); m_i2lo = BAL(d >= T.t[UVC_T_aLI2t]); m_i2hi = BAL(d <= T.t[UVC_T_aLI2T]);
        m_nonb = BAL(l_nb > r_nb);
    } else {
        m_i1lo = BAL(d >= T.t[UVC_T_aRI1t]); m_i1hi = BAL(d <= T.t[UVC_T_aRI1T]); m_i2lo = BAL(d >= T.t[UVC_T_aRI2t]); m_i2hi = BAL(d <= T.t[UVC_T_aRI2T]);
        m_nonb = BAL(l_nb < r_nb);
    }
    const wmask m_okB = umask(is_normal), m_okL = m_okB | (umask(mate_ok) & m_nonb);
    const int p2 = (is_assay_UMI || !amplicon), nc = (0 == r.clip_cnt);
    auto common = [&](SegAcc &A, int bq) {
        const int bq2 = (int)(__umul24((unsigned)bq, (unsigned)bq) / SQR_QUAL_DIV);   // bq < 2^12: the 24-bit multiply is exact and full rate
        if (isrc) { A.a1BQr += bq; A.a2BQr += bq2; } else { A.a1BQf += bq; A.a2BQf += bq2; }
        A.bq += bq;
        A.s[UVC_S_aMQs] += r.mapq;
        if (strand) { if (isrc) A.s[UVC_S_aDPrr] += 1; else A.s[UVC_S_aDPrf] += 1; }
        else        { if (isrc) A.s[UVC_S_aDPfr] += 1; else A.s[UVC_S_aDPff] += 1; }
        addm(A.s[UVC_S_aP3], m_p3);
        A.s[UVC_S_aNC] += nc; A.s[UVC_S_aP2] += p2;
        if (has_isize) { if (isrc) A.lp[UVC_S64_aLIT] += fl2; else A.lp[UVC_S64_aRIT] += fr2; }
    };
    // (in_all: `in` is one wave-uniform mask, all lanes or none -- a mask, not a branch around the block: the loop has no branch per read for it)
    auto bias = [&](SegAcc &A, bool in_all, wmask in, wmask far, wmask unaff, wmask m_rb1, wmask m_rb2, int r_baq) {
        const wmask a = (in & far), b = (in & unaff);
        addm(A.s[UVC_S_aLP1], a & m_lp1); addm(A.s[UVC_S_aLP2], a & m_lp2); addm(A.s[UVC_S_aRP1], a & m_rp1); addm(A.s[UVC_S_aRP2], a & m_rp2);
        A.s[UVC_S_aLPL] += selm(a, l_nb); A.s[UVC_S_aRPL] += selm(a, r_nb);
        addm(A.s[UVC_S_aLB1], b & m_lb1); addm(A.s[UVC_S_aLB2], b & m_lb2); addm(A.s[UVC_S_aRB1], b & m_rb1); addm(A.s[UVC_S_aRB2], b & m_rb2);
        A.lp[UVC_S64_aLBL] += selm(b, l_baq); A.lp[UVC_S64_aRBL] += selm(b, r_baq);
        if (in_all) A.s[UVC_S_aBQ2] += (in ? 1 : 0); else addm(A.s[UVC_S_aBQ2], in);
        addm(A.s[UVC_S_aP1], far & unaff & m_iend);
    };
    if (hasB) {
        common(AB, bqB);
        AB.s[UVC_S_aPF1] += amp1[imin(bqB, 255)]; AB.s[UVC_S_aPF2] += amp2[imin(bqB, 255)];   // "100 * amp / 100" == amp (main.hpp:1472-1519)
        AB.s[UVC_S_a2XM2] += xm_inc; AB.s[UVC_S_a2BM2] += bm_inc;
        // the base side enters the bias blocks when bq >= highBQ (tier2 then holds)
        bias(AB, false, BAL(bqB >= P.bias_thres_highBQ), m_farB, m_unaffB, BAL(r_baqB >= P.bias_thres_BAQ1), BAL(r_baqB >= P.bias_thres_BAQ2), r_baqB);
        if (isrc) { addm(AB.s[UVC_S_aLI1], m_i1lo & m_i1hi & m_okB); addm(AB.s[UVC_S_aLI2], m_i2lo & m_i2hi & m_okB & m_goodB); addm(AB.s[UVC_S_aLIr], m_goodB); }
        else      { addm(AB.s[UVC_S_aRI1], m_i1lo & m_i1hi & m_okB); addm(AB.s[UVC_S_aRI2], m_i2lo & m_i2hi & m_okB & m_goodB); addm(AB.s[UVC_S_aRIf], m_goodB); }
    }
    if (hasL) {
        common(AL, bqL);
        AL.s[UVC_S_aPF1] += imin(100, amp1[imin(bqL, 255)]); AL.s[UVC_S_aPF2] += imin(100, amp2[imin(bqL, 255)]);
        // the gap side enters when dist_to_interfering_indel (10000) >= bias_thres_interfering_indel
        bias(AL, true, umask(10000 >= P.bias_thres_interfering_indel), m_farL, m_unaffL, BAL(r_baqL >= P.bias_thres_BAQ1), BAL(r_baqL >= P.bias_thres_BAQ2), r_baqL);   // (aP1 inside does not look at `in`)
        if (isrc) { addm(AL.s[UVC_S_aLI1], m_i1lo & m_okL); addm(AL.s[UVC_S_aLI2], m_i2lo & m_okL & m_goodL); addm(AL.s[UVC_S_aLIr], m_goodL); }
        else      { addm(AL.s[UVC_S_aRI1], m_i1lo & m_okL); addm(AL.s[UVC_S_aRI2], m_i2lo & m_okL & m_goodL); addm(AL.s[UVC_S_aRIf], m_goodL); }
    }
}

DEV void load_thres(const RegionDev &R, PosThres &T, int64_t x) {
#pragma unroll
    for (int f = 0; f < UVC_NTHRES; f++) T.t[f] = TH(R, f, x);
}

// primer gating of updateByAln, main.hpp:1872-1875, 1895
DEV void primer_window(const UvcParams &P, const AlnRec &a, int &ibeg, int &iend) {
    const bool isrc = (a.flag & 0x10) != 0;
    ibeg = ((a.isize != 0) ? (imin(a.pos, a.mpos) + P.primerlen) : ((isrc && (0x0 == (0x1 & a.flag))) ? 0 : (a.pos + P.primerlen)));
    iend = ((a.isize != 0) ? (int)nnminus(imin(a.pos, a.mpos) + abs(a.isize), P.primerlen) : ((isrc && (0x0 == (0x1 & a.flag))) ? (int)nnminus(a.rend, P.primerlen) : INT32_MAX));
}

// ------------------------------------------------------------------------------------------------
// k_aln_prelude: one thread per alignment (main.hpp:1795-1885 prelude of updateByAln)
// ------------------------------------------------------------------------------------------------
struct RawReads {   // the caller's columns as they are (uvcgpu_region_set_reads) + what uvc_prep.hip derives per read
    const int32_t *pos, *endpos, *mpos, *isize, *nm, *l_qseq, *n_cigar, *frag, *fs, *dflag, *kind, *fast_rank;
    const uint16_t *flag; const uint8_t *mapq;
    const int64_t *seq_off, *cigar_off, *table_off, *item_off, *gap_off;
};

// does the read have an InDel next to low base qualities (main.hpp:1817-1859)?  Then dist_to_interfering_indel varies along the
// read and its M runs cannot take the simple path of k_p2_fast.
DEV bool has_lowbq_indel(const UvcParams &P, const AlnRec &a, const uint32_t *cigar, const uint8_t *quals) {
    auto Q = [&](int q) -> int { return (int)quals[imin(imax(q, 0), a.l_qseq - 1)]; };
    int qpos = 0;
    for (int i = 0; i < a.n_cigar; i++) {
        const int op = cig_op(cigar[i]), len = cig_len(cigar[i]);
        if (op == C_MATCH || op == C_EQUAL || op == C_DIFF || op == C_SOFT_CLIP) qpos += len;
        else if (op == C_INS) {
            for (int q2 = qpos - imin(qpos, 1); q2 < imin(qpos + len + 1, a.rend); q2++) if (Q(q2) < P.bias_thres_interfering_indel_BQ) return true;
            qpos += len;
        } else if (op == C_DEL) {
            if (imin(Q(imax(1, qpos) - 1), Q(qpos)) <= P.bias_thres_interfering_indel_BQ) return true;
        }
    }
    return false;
}

// digest of alignment `a` (index `id`) for the position-centric kernels, covering [cbeg, cend) with query offset qb_lo
DEV void fill_fastrec(FastRec &f, const AlnRec &a, int id, int cbeg, int cend, int32_t qb_lo) {
    f.pos = cbeg; f.rend = cend; f.qb_lo = qb_lo; f.aln = id;
    f.fmd = (a.flag & 0xFFFF) | ((a.mapq & 0xFF) << 16) | ((a.dflag & 0xFF) << 24); f.isize = a.isize; f.mpos = a.mpos; f.xm1500 = a.xm1500;
    // per-read constants of dealwith_segbias<false>: the a2XM2 / a2BM2 increments (main.hpp:1521-1522), each <= 100
    int bv[5];
    for (int s2 = 0; s2 < 5; s2++) bv[s2] = (a.bm1500[s2] > 20 ? (100 * (20 * 20) / (a.bm1500[s2] * a.bm1500[s2])) : 100);
    const int xv = (a.xm1500 > 20 ? (100 * (20 * 20) / (a.xm1500 * a.xm1500)) : 100);
    f.bmv = bv[0] | (bv[1] << 8) | (bv[2] << 16) | (bv[3] << 24); f.xbv = bv[4] | (xv << 8);
    f.bm4c = ((a.clip_cnt & 0xF) << 16) | (a.nogap_penal & 0xFFFF);   // nogap_penal is negative when NM < the InDel lengths
    f.clips = (a.lclip_oplen & 0xFFFF) | (a.rclip_oplen << 16);
    f.baq_pos = (int32_t)a.baq_pos; f.baq_last = (int32_t)a.baq_last; f.baq2_last = (int32_t)a.baq2_last;
    f.ext = ((cbeg - a.pos) & 0xFFFF) | ((a.rend - cend) << 16);
}

// P2 work list: entry j covers [cbeg[j], cend[j]) of alignment aln[j] (sorted by (class, cbeg)).  The entry of a simple alignment is the
// record k_aln_prelude already made for the P1 list (frec, the same alignment under its rank there): 64 bytes copied instead of the
// alignment's 200-byte record read again; only the M runs of InDel reads are built from their alignment records.
__global__ void __launch_bounds__(256) k_build_p2list(RegionDev R, const int32_t *fast_rank, const int32_t *aln, const int32_t *cbeg, const int32_t *cend, const int32_t *qb) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= R.n_fast2) return;
    const int id = aln[j], rk = fast_rank[id];
    if (rk >= 0) {
        const uint4 *src = (const uint4 *)&R.frec[rk]; uint4 *dst = (uint4 *)&R.frec2[j];
        const uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
        dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3;
        return;
    }
    FastRec f;
    const AlnRec &a = R.alns[id];
    // an InDel read that k_aln_prelude found ineligible keeps its slot (the list stays sorted) but covers nothing
    fill_fastrec(f, a, id, cbeg[j], (a.kind == 1 ? cbeg[j] : cend[j]), qb[j]);
    R.frec2[j] = f;
}

// Mismatching bases per base symbol (bm1500s, main.hpp:1860-1863) of the simple alignments.  Eight lanes per alignment, eight bases per lane
// and step: read bases and reference codes are bytes below 0x80, so "differs" and "is symbol s" are byte-parallel tests on two 8-byte words
// (unaligned loads: the hardware takes them) and a count is a popcount.  The eight lanes of an alignment read 64 consecutive bytes, so a
// load of the wave touches eight cache lines; with a lane per alignment it touched 64 (8 bytes of each, 150 bytes apart) and the kernel was
// bound by the L2 -> L1 traffic of lines read sixteen times over (0.39 ms per 2 M reads; a wave per alignment was 0.59 ms, a thread per
// alignment walking byte by byte 2.0 ms).  The five counts wait in the alignment's own FastRec slot.
DEV unsigned long long load8u(const uint8_t *p) { unsigned long long v; __builtin_memcpy(&v, p, 8); return v; }
__global__ void __launch_bounds__(256) k_aln_bm(RegionDev R, RawReads W) {
    const unsigned long long K7F = 0x7F7F7F7F7F7F7F7FULL, K80 = 0x8080808080808080ULL, K01 = 0x0101010101010101ULL;
    const int sub = threadIdx.x & 7;
    for (int id = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; id < R.n_alns; id += (gridDim.x * blockDim.x) >> 3) {
        const int rk = W.fast_rank[id];
        if (rk < 0) continue;   // (the eight lanes of the alignment leave together)
        const int pos = W.pos[id], len = W.endpos[id] - pos;
        const uint32_t c0 = R.cigars[W.cigar_off[id]];
        const uint8_t *b = R.bases + W.seq_off[id] + (cig_op(c0) == C_SOFT_CLIP ? cig_len(c0) : 0);
        const uint8_t *rf = R.refsym + (pos - R.beg);
        int cnt[5] = { 0, 0, 0, 0, 0 };
        for (int k = 8 * sub; k + 8 <= len; k += 64) {
            const unsigned long long bb = load8u(b + k), rr = load8u(rf + k);
            const unsigned long long mism = ((bb ^ rr) + K7F) & K80;          // 0x80 in every byte that differs (all bytes are < 0x80)
#pragma unroll
            for (int s2 = 0; s2 < 5; s2++) cnt[s2] += __popcll(mism & ~((bb ^ (K01 * (unsigned)s2)) + K7F) & K80);
        }
        { const int k = (len & ~7) + sub;   // the last len % 8 bases, one per lane
          if (k < len) { const int bs = b[k]; const bool mm = (bs != rf[k]);
#pragma unroll
                         for (int s2 = 0; s2 < 5; s2++) cnt[s2] += (mm && bs == s2); } }
#pragma unroll
        for (int s2 = 0; s2 < 5; s2++) { cnt[s2] += __shfl_xor(cnt[s2], 1); cnt[s2] += __shfl_xor(cnt[s2], 2); cnt[s2] += __shfl_xor(cnt[s2], 4); }
        if (sub == 0) {
            int32_t *dst = (int32_t *)&R.frec[rk];
#pragma unroll
            for (int s2 = 0; s2 < 5; s2++) dst[s2] = cnt[s2];
        }
    }
}

// the prelude of one alignment; returns the number of its bases that go through the mismatch queue of k_p2_fast
DEV int aln_prelude_one(const RegionDev &R, const RawReads &W, const UvcParams &P, const int id, AlnRec &out) {
    AlnRec a;
    a.pos = W.pos[id]; a.rend = W.endpos[id]; a.mpos = W.mpos[id]; a.isize = W.isize[id]; a.flag = W.flag[id]; a.mapq = W.mapq[id];
    a.dflag = W.dflag[id]; a.l_qseq = W.l_qseq[id]; a.seq_off = W.seq_off[id]; a.cigar_off = W.cigar_off[id]; a.table_off = W.table_off[id]; a.item_off = W.item_off[id]; a.gap_off = W.gap_off[id];
    a.n_cigar = W.n_cigar[id]; a.kind = W.kind[id]; a.frag = W.frag[id]; a.fs = W.fs[id]; a.id = id;
    const uint32_t *cigar = R.cigars + a.cigar_off;
    const uint8_t *bases = R.bases + a.seq_off;
    int nge = 0, ngo = 0, clip_cnt = 0;
    for (int i = 0; i < a.n_cigar; i++) {
        const int op = cig_op(cigar[i]);
        if (C_INS == op || C_DEL == op) { nge += cig_len(cigar[i]); ngo++; }
        if (C_SOFT_CLIP == op || C_HARD_CLIP == op) clip_cnt++;
        if (op > C_DIFF) atomicExch(R.err, UVCGPU_EUNSUPPORTED);   // BAM_CBACK etc.: process_cigar throws, main_conversion.hpp:911-915
    }
    const int nm_cnt = (W.nm[id] >= 0 ? W.nm[id] : nge);
    const int qlen = a.rend - a.pos;
    a.xm1500 = (nm_cnt - nge) * 1500 / qlen;
    a.go1500 = ngo * 1500 / qlen;
    a.clip_cnt = clip_cnt;
    int bm[5] = { 0, 0, 0, 0, 0 };
    int qpos = 0, rpos = a.pos, lclip_q = 0, m_index = -1;
    const int rk0 = W.fast_rank[id];
    if (rk0 >= 0) {   // simple alignment: k_aln_bm counted its mismatching bases with a whole wave and left them in this read's own FastRec slot
        const int32_t *pre = (const int32_t *)&R.frec[rk0];
        for (int s2 = 0; s2 < 5; s2++) bm[s2] = pre[s2];
        for (int i = 0; i < a.n_cigar && m_index < 0; i++) {
            const int op = cig_op(cigar[i]);
            if (op == C_MATCH || op == C_EQUAL || op == C_DIFF) { m_index = i; lclip_q = qpos; }
            else if (op == C_SOFT_CLIP) qpos += cig_len(cigar[i]);
        }
    } else
    for (int i = 0; i < a.n_cigar; i++) {
        const int op = cig_op(cigar[i]), len = cig_len(cigar[i]);
        if (op == C_MATCH || op == C_EQUAL || op == C_DIFF) {
            if (m_index < 0) { m_index = i; lclip_q = qpos; }
            for (int k = 0; k < len; k++) { const int b = bases[qpos]; if (R.refsym[rpos - R.beg] != b) bm[b] += 1; qpos++; rpos++; }
        } else if (op == C_INS || op == C_SOFT_CLIP) qpos += len;
        else if (op == C_DEL || op == C_REF_SKIP) rpos += len;
    }
    for (int s = 0; s < 5; s++) a.bm1500[s] = bm[s] * 1500 / qlen;
    a.lclip_len = ((a.n_cigar > 0 && cig_op(cigar[0]) == C_SOFT_CLIP) ? cig_len(cigar[0]) : 0);
    a.rclip_len = ((a.n_cigar > 0 && cig_op(cigar[a.n_cigar - 1]) == C_SOFT_CLIP) ? cig_len(cigar[a.n_cigar - 1]) : 0);
    const int by_clip = imax(a.lclip_len, a.rclip_len) / 6;
    const int by_nm = (a.xm1500 + a.go1500) / 30;
    a.indel_penal = imin(1, by_nm + by_clip);
    a.nogap_penal = imin(4, by_nm + by_clip) + 1;
    // NM below the InDel lengths (malformed, but the reference computes with it) makes the penalties negative; the 16-bit value
    // fields of Item / Contrib hold the results down to here
    if (by_nm + by_clip < -30000) atomicExch(R.err, UVCGPU_EUNSUPPORTED);   // (an NM tag tens of thousands below the InDel lengths of the read)
    a.lclip_q = lclip_q; a.m_index = m_index;
    a.lclip_oplen = 0; a.rclip_oplen = 0;
    if (a.kind == 0) {
        if (m_index > 0) a.lclip_oplen = cig_len(cigar[m_index - 1]);
        if (m_index + 1 < a.n_cigar) a.rclip_oplen = cig_len(cigar[m_index + 1]);
    }
    a.qbase = a.seq_off + lclip_q - a.pos;
    a.baq_pos = BAQ1(R, a.pos); a.baq_last = BAQ1(R, a.rend - 1); a.baq2_last = BAQ2(R, a.rend - 1);
    if (a.kind == 2 && has_lowbq_indel(P, a, cigar, R.quals + a.seq_off)) a.kind = 1;
    const int rk = W.fast_rank[id];
    int n_mis = 0;
    a.n_mutc = bm[0] + bm[1] + bm[2] + bm[3] + bm[4] + ngo;
    out = a;
    if (rk >= 0 || W.kind[id] == 2) n_mis = bm[0] + bm[1] + bm[2] + bm[3] + bm[4];   // every alignment that can be on the P2 work list: its mismatching bases go through the mismatch queue
    if (rk >= 0) {
        FastRec f;
        fill_fastrec(f, a, id, a.pos, a.rend, (int32_t)(a.qbase & 0xFFFFFFFFLL));
        R.frec[rk] = f;
    }
    return n_mis;
}
// One thread per alignment.  The 200-byte records leave through LDS: written from registers, a lane's record is 25 stores that each touch 64
// cache lines of the wave; the block's 256 records are one contiguous 51 200-byte span of alns[] and are copied out with coalesced 16-byte stores.
__global__ void __launch_bounds__(256) k_aln_prelude(RegionDev R, RawReads W, UvcParams P) {
    static_assert(sizeof(AlnRec) % 8 == 0 && (256 * sizeof(AlnRec)) % 16 == 0, "the staged copy moves the block's records as 16-byte words");
    __shared__ __attribute__((aligned(16))) AlnRec stage[256];
    const int id0 = blockIdx.x * blockDim.x, id = id0 + threadIdx.x;
    __shared__ int mis_s[4];
    int n_mis = (id < R.n_alns) ? aln_prelude_one(R, W, P, id, stage[threadIdx.x]) : 0;
    for (int d = 32; d > 0; d >>= 1) n_mis += __shfl_xor(n_mis, d);   // one atomic per block: same-address atomics serialise
    if ((threadIdx.x & 63) == 0) mis_s[threadIdx.x >> 6] = n_mis;
    __syncthreads();
    if (threadIdx.x == 0) { const int t = mis_s[0] + mis_s[1] + mis_s[2] + mis_s[3]; if (t) atomicAdd(R.mis_total, (unsigned long long)t); }
    const int nrec = imin(256, R.n_alns - id0);
    if (nrec == 256) {
        const uint4 *src = (const uint4 *)&stage[0]; uint4 *dst = (uint4 *)(R.alns + id0);
        for (int i = threadIdx.x; i < (int)(256 * sizeof(AlnRec) / 16); i += 256) dst[i] = src[i];
    } else {
        const unsigned long long *src = (const unsigned long long *)&stage[0]; unsigned long long *dst = (unsigned long long *)(R.alns + id0);
        for (int i = threadIdx.x; i < nrec * (int)(sizeof(AlnRec) / 8); i += 256) dst[i] = src[i];
    }
}

// ------------------------------------------------------------------------------------------------
// P1 (update_seg_format_prep_sets_by_aln, main.hpp:924-1204)
// ------------------------------------------------------------------------------------------------
// SNV / DNV run detection started at ref position s of alignment a (main.hpp:1025-1046).
// `q_of_r` maps the walk to the query: the reference advances query and reference together
// regardless of the CIGAR, so q = q_s + (r - s).
DEV void snv_dnv_scatter(const RegionDev &R, const uint8_t *qbases, int q_s, int l_qseq, int apos, int rend, int s) {
    int nq = q_s, nr = s;
    int refsymbol = UVC_BASE_NN, readsymbol = UVC_NUM_SYMBOLS;
    while (refsymbol != readsymbol && nq < l_qseq && nr < rend) {
        refsymbol = R.refsym[nr - R.beg];
        readsymbol = qbases[nq];
        nq++; nr++;
    }
    if (nr == s + 2) for (int r = imax(apos, s - 1); r < imin(nr, rend); r++) atomicAdd(&P32(R, UVC_P_a_snv_dp, r - R.beg), 1);
    if (nr > s + 2)  for (int r = imax(apos, s - 1); r < imin(nr, rend); r++) atomicAdd(&P32(R, UVC_P_a_dnv_dp, r - R.beg), 1);
}

DEV void clip_event(const RegionDev &R, const UvcParams &P, int rpos, int i, int len, int pcr_dp_inc) {   // main.hpp:1183-1199
    const int delta = ((0 == i) ? 0 : -1);
    if (pcr_dp_inc) {
        for (int r2 = rpos + delta - P.microadjust_near_clip_dist; r2 <= rpos + delta + P.microadjust_near_clip_dist; r2++)
            if (R.beg <= r2 && r2 < R.end) atomicAdd(&P32(R, UVC_P_a_near_pcr_clip_dp, r2 - R.beg), pcr_dp_inc);
    }
    if ((0 == pcr_dp_inc) && (len >= P.microadjust_alignment_clip_min_len)) {
        const int r2 = rpos + delta;
        if (R.beg <= r2 && r2 < R.end) atomicAdd(&P32(R, UVC_P_a_near_long_clip_dp, r2 - R.beg), 1);
    }
}

DEV int simple_base_value(const UvcParams &P, const AlnRec &a, int p, const uint8_t *quals_qbase, bool proton);

// A mismatching base of a simple alignment: decide with the reference's fragment consensus (main.hpp:2658-2726) whether
// position p is a high-quality mutation of the fragment and, if so, record it once (by the first mismatching alignment).
DEV void mut_event(const RegionDev &R, const UvcParams &P, const AlnRec &a, int p, int my_ref) {
    const FragRec &f = R.frags[a.frag];
    if (f.stat_kind != 0) return;
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    int m[6] = { 0, 0, 0, 0, 0, 0 };
    int first_mis = -1;
    for (int k = f.aln_beg; k < f.aln_end; k++) {
        const AlnRec &al = R.alns[k];
        if (p < al.pos || p >= al.rend) continue;
        const int sym = R.bases[al.qbase + p];
        if (sym != my_ref && first_mis < 0) first_mis = k;
        const int v = simple_base_value(P, al, p, R.quals + al.qbase, proton);
        m[sym] = imax(m[sym], v);
    }
    if (first_mis != a.id) return;
    int cs = UVC_BASE_NN, cc = 0, ct = 0;
    for (int s2 = 0; s2 < 6; s2++) { if (cc < m[s2]) { cs = s2; cc = m[s2]; } ct += m[s2]; }
    const int con_qual = cc * 2 - ct;
    const bool highBQ = (proton ? true : (con_qual >= P.bias_thres_highBQ));
    if (symbols_mutated(my_ref, cs) && highBQ) {
        const int idx = atomicAdd(&R.frag_nmut[a.frag], 1);
        if (idx < UVC_MAXEV) R.frag_mut[(size_t)a.frag * UVC_MAXEV + idx] = p;
    }
}

#define PREPQ_CAP 192   // per-wave LDS queue of (alignment, position) pairs whose base differs from the reference
// ------------------------------------------------------------------------------------------------
// k_prep_sums: the P1 counters of the simple alignments that do not depend on the bases, as interval sums (the scheme of k_frag_sums).
// a_dp, a_pcr_dp, a_umi_dp, a_qlen, a_XM1500, a_LIDP, a_RIDP are sums of per-read constants over the reads that cover a position;
// a_LI / a_RI add min(distance to the insert end, MAX_INSERT_SIZE) (main.hpp:1006-1017), which is p * A(p) + B(p) with A the number of
// covering reads on the linear piece and B the sum of their offsets plus MAX_INSERT_SIZE per read on the clamped piece.  One block per
// tile: a lane per read adds +v / -v at the ends of its pieces in LDS, prefix sums, plain stores (the planes are zero; k_prep_fast and
// k_prep_slow add theirs with atomics afterwards).  k_prep_fast keeps the queue of mismatching bases and the counters behind base quality.
// ------------------------------------------------------------------------------------------------
#define PSUM_TILE 1024
enum { PS_DP = 0, PS_PCR, PS_UMI, PS_QLEN, PS_XM, PS_LIDP, PS_RIDP, PS_LIA, PS_RIA, PS_N32 };
__global__ void __launch_bounds__(256) k_prep_sums(RegionDev R, UvcParams P) {
    __shared__ int d[PS_N32][PSUM_TILE + 1];
    __shared__ unsigned long long d64[2][PSUM_TILE + 1];   // B of a_LI, a_RI
    __shared__ int wtot[4];
    __shared__ unsigned long long wtot64[4];
    const int t0 = R.beg + (int)blockIdx.x * PSUM_TILE, t1 = imin(t0 + PSUM_TILE, R.beg + (int)R.npos);
    for (int i = threadIdx.x; i < PS_N32 * (PSUM_TILE + 1); i += 256) (&d[0][0])[i] = 0;
    for (int i = threadIdx.x; i < 2 * (PSUM_TILE + 1); i += 256) (&d64[0][0])[i] = 0ull;
    __syncthreads();
    const int w_first = (int)blockIdx.x * (PSUM_TILE / 64), w_last = imin(w_first + PSUM_TILE / 64, R.nwin) - 1;
    const int lo = win_lo(R, 0, w_first), hi = win_hi(R, 0, w_last);
    auto put = [&](int f, int a, int b, int v) {   // += v on [a, b) of plane f, clipped to the tile
        a = imax(a, t0); b = imin(b, t1);
        if (a < b && v != 0) { atomicAdd(&d[f][a - t0], v); atomicAdd(&d[f][b - t0], -v); }
    };
    auto put64 = [&](int f, int a, int b, long long v) {
        a = imax(a, t0); b = imin(b, t1);
        if (a < b && v != 0) { atomicAdd(&d64[f][a - t0], (unsigned long long)v); atomicAdd(&d64[f][b - t0], (unsigned long long)(-v)); }
    };
    for (int k = lo + (int)threadIdx.x; k < hi; k += 256) {
        const int4 *q4 = (const int4 *)(R.frec + k);
        const int4 h0 = q4[0], h1 = q4[1];   // pos rend qb_lo aln | fmd isize mpos xm1500
        const int apos = h0.x, rend = h0.y;
        if (rend <= t0 || apos >= t1) continue;
        const int dflag = (h1.x >> 24) & 0xFF;
        put(PS_DP, apos, rend, 1); put(PS_PCR, apos, rend, (dflag & 0x4) ? 1 : 0); put(PS_UMI, apos, rend, (dflag & 0x1) ? 1 : 0);
        put(PS_QLEN, apos, rend, rend - apos); put(PS_XM, apos, rend, h1.w);
        if (h1.y != 0) {
            const int fl = imin(apos, h1.z);
            if (h1.x & 0x10) {   // a_LI: min(p - fl + 1, MAX) = p + (1 - fl) while p < fl + MAX
                const int brk = imax(apos, imin(rend, (int)lmin((long long)fl + MAX_INSERT_SIZE, (long long)INT32_MAX)));
                put(PS_LIDP, apos, rend, 1);
                put(PS_LIA, apos, brk, 1); put64(0, apos, brk, 1LL - fl); put64(0, brk, rend, MAX_INSERT_SIZE);
            } else {             // a_RI: min(fr - p, MAX) = fr - p once p > fr - MAX
                const long long fr = (long long)fl + (long long)abs(h1.y);
                const int brk = imax(apos, imin(rend, (int)lmax(lmin(fr - MAX_INSERT_SIZE + 1, (long long)INT32_MAX), (long long)INT32_MIN)));
                put(PS_RIDP, apos, rend, 1);
                put64(1, apos, brk, MAX_INSERT_SIZE); put(PS_RIA, brk, rend, 1); put64(1, brk, rend, fr);
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n_here = t1 - t0;
    const int64_t xb = (int64_t)blockIdx.x * PSUM_TILE;
    int liA[4] = {0, 0, 0, 0}, riA[4] = {0, 0, 0, 0};
    const int plane_of[PS_N32] = { UVC_P_a_dp, UVC_P_a_pcr_dp, UVC_P_a_umi_dp, UVC_P_a_qlen, UVC_P_a_XM1500, UVC_P_a_LIDP, UVC_P_a_RIDP, -1, -1 };
    for (int f = 0; f < PS_N32; f++) {
        int v[4], run = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { run += d[f][threadIdx.x * 4 + i]; v[i] = run; }
        int inc = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        int before = inc - run;
        for (int i = 0; i < w; i++) before += wtot[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = threadIdx.x * 4 + i, val = before + v[i];
            if (f == PS_LIA) liA[i] = val; else if (f == PS_RIA) riA[i] = val;
            else if (x < n_here && val != 0) P32(R, plane_of[f], xb + x) = val;
        }
    }
    for (int f = 0; f < 2; f++) {
        unsigned long long v[4], run = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { run += d64[f][threadIdx.x * 4 + i]; v[i] = run; }
        unsigned long long inc = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wtot64[w] = inc;
        __syncthreads();
        unsigned long long before = inc - run;
        for (int i = 0; i < w; i++) before += wtot64[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = threadIdx.x * 4 + i;
            const long long pp = (long long)t0 + x;
            const long long val = (long long)(before + v[i]) + (f == 0 ? pp * liA[i] : -pp * riA[i]);
            if (x < n_here && val != 0) P64(R, f == 0 ? UVC_P_a_LI : UVC_P_a_RI, xb + x) = val;
        }
    }
}

// SPLIT: a region with fewer windows than the chip has wave slots (a deep, short panel region: 200 kb are three waves per SIMD, each walking
// thousands of reads) gives every window to a BLOCK: its four waves take every fourth chunk of the window's reads.  The kernel ends in
// atomics anyway, so the partial sums need no reduction.
template <bool SPLIT>
__global__ void __launch_bounds__(256) k_prep_fast(RegionDev R, UvcParams P) {
    __shared__ int2 prepq[4][PREPQ_CAP];
    const int lane = threadIdx.x & 63;
    const int wave = SPLIT ? wave_uniform((int)xcd_block()) : wave_uniform((int)((xcd_block() * blockDim.x + threadIdx.x) >> 6));
    const int64_t x0 = (int64_t)wave * 64;
    if (x0 >= R.npos) return;
    const int w0 = R.beg + (int)x0;
    const int p = w0 + lane;
    const int64_t x = x0 + lane;
    const bool valid = x < R.npos;
    const int my_ref = valid ? R.refsym[x] : 0;
    const int my_baq = valid ? (int)R.baq[x] : 0;
    int ldist = 0, rdist = 0, hbq = 0;
    long long lbaq = 0, rbaq = 0;
    const int lo = wave_uniform(win_lo(R, 0, (int)(x0 >> 6))), hi = wave_uniform(win_hi(R, 0, (int)(x0 >> 6)));
    COARSE_T(ct1)
    const __amdgpu_buffer_rsrc_t rs = bq_rsrc(R);
    // Mismatching bases are rare per lane but present in a large share of the iterations of a wave, and their handling (SNV / DNV
    // run detection, fragment mutation events) is a chain of dependent loads.  They are queued per wave and handled 64 at a time,
    // one per lane, in wave-uniform control flow, so the main loop never waits on them.
    int2 *myq = prepq[threadIdx.x >> 6];
    int nq = 0;   // wave-uniform: only updated in uniform control flow
    auto drain = [&]() {
        for (int i = lane; i < nq; i += 64) {
            const int2 e = myq[i];
            const AlnRec &a = R.alns[e.x];
            snv_dnv_scatter(R, R.bases + a.seq_off, (int)(a.qbase + e.y - a.seq_off), a.l_qseq, a.pos, a.rend, e.y);
            mut_event(R, P, a, e.y, R.refsym[e.y - R.beg]);
        }
        nq = 0;
    };
    for (int k0 = lo + (SPLIT ? 64 * (int)(threadIdx.x >> 6) : 0); k0 < hi; k0 += (SPLIT ? 256 : 64)) {
        Chunk16 c;
        load_chunk16(R.frec, k0 + lane, hi, c);
        const int n = imin(64, hi - k0);
        // the base | quality bytes of the chunk's reads are requested four reads ahead (one load per read and lane, HBM / L2 latency each:
        // with one read of look-ahead the wave waited for most of them)
        auto fetch = [&](int j) { return bq_load(rs, bcast(c.v[2], j) + p); };   // low word of qbase + p: exact for lanes inside the read, harmless elsewhere
        auto one = [&](int j, int bqn) {
            const int b = bqn & 0xFF, q = (bqn >> 8) & 0xFF;
            const int apos = bcast(c.v[0], j), rend = bcast(c.v[1], j);
            if (rend <= w0) return;
            const bool cover = (valid && p >= apos && p < rend);
            const wmask mm = BAL(cover && b != my_ref);
            if (mm) {
                if (nq > PREPQ_CAP - 64) drain();
                if (cover && b != my_ref) myq[nq + (int)__builtin_popcountll(mm & ((1ull << lane) - 1ull))] = make_int2(bcast(c.v[3], j), p);   // (FastRec::aln, position)
                nq += (int)__builtin_popcountll(mm);
            }
            if (cover) {   // (the counters that do not look at the base: k_prep_sums)
                if (q >= P.bias_thres_highBQ) {
                    ldist += p - apos + 1; rdist += rend - p;
                    lbaq += (my_baq - bcast(c.v[12], j) + 1);
                    rbaq += (bcast(c.v[13], j) - my_baq + 1);
                    hbq += 1;
                }
                const int clips = bcast(c.v[11], j);
                if (clips != 0) {
                    const int lclip = clips & 0xFFFF, rclip = (clips >> 16) & 0xFFFF;
                    const int pcr_inc = (((bcast(c.v[4], j) >> 24) & 0x4) ? 1 : 0);
                    if (p == apos && lclip > 0) clip_event(R, P, apos, 0, lclip, pcr_inc);
                    if (p == rend - 1 && rclip > 0) clip_event(R, P, rend, 1, rclip, pcr_inc);   // any index > 0 gives rpos_delta = -1
                }
            }
        };
        int r0 = fetch(0), r1 = (1 < n ? fetch(1) : 0), r2 = (2 < n ? fetch(2) : 0), r3 = (3 < n ? fetch(3) : 0);
        for (int j = 0; j < n; j += 4) {
            { const int v = r0; if (j + 4 < n) r0 = fetch(j + 4); one(j, v); }
            if (j + 1 < n) { const int v = r1; if (j + 5 < n) r1 = fetch(j + 5); one(j + 1, v); }
            if (j + 2 < n) { const int v = r2; if (j + 6 < n) r2 = fetch(j + 6); one(j + 2, v); }
            if (j + 3 < n) { const int v = r3; if (j + 7 < n) r3 = fetch(j + 7); one(j + 3, v); }
        }
    }
    if (nq > 0) drain();
    COARSE_T(ct2)
    if (!valid) return;
    // k_prep_slow runs concurrently on the side stream and adds to the same fields: atomics
    if (hbq) {
        atomicAdd(&P32(R, UVC_P_a_highBQ_dp, x), hbq);
        atomicAdd(&P32(R, UVC_P_a_l_dist_sum, x), ldist); atomicAdd(&P32(R, UVC_P_a_r_dist_sum, x), rdist);
        add64(&P64(R, UVC_P_a_l_BAQ_sum, x), lbaq); add64(&P64(R, UVC_P_a_r_BAQ_sum, x), rbaq);
    }
#ifdef UVC_FRAG_COARSE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    COARSE_T(ct3)
    if (lane == 0 && ((int)(x0 >> 6) % 2999) == 7) printf("coarse prep wave %d lists %llu epilogue %llu\n", (int)(x0 >> 6), ct2 - ct1, ct3 - ct2);
#endif
}

// P1 for one alignment whose CIGAR has InDels / unusual clip layouts: one wave per read, lanes stride over the bases of
// each CIGAR op (every update of P1 is an independent integer add once the per-read totals of the first loop are known)
__global__ void __launch_bounds__(64) k_prep_slow(RegionDev R, UvcParams P) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= R.n_complex) return;
    const AlnRec &a = R.alns[R.complex_ids[t]];
    const uint32_t *cigar = R.cigars + a.cigar_off;
    const uint8_t *bases = R.bases + a.seq_off, *quals = R.quals + a.seq_off;
    const int off = R.beg, rend = a.rend, apos = a.pos, n_cigar = a.n_cigar;
    const long long baq_last = R.end - 1;
    int nge = 0, ngo = 0, insbaq_sum = 0, delbaq_sum = 0, inslen_sum = 0, dellen_sum = 0;
    int qpos = 0, rpos = apos;
    for (int i = 0; i < n_cigar; i++) {
        const int op = cig_op(cigar[i]), len = cig_len(cigar[i]);
        if (C_INS == op) { nge += len; ngo++; insbaq_sum += (int)(BAQ1(R, lmin((long long)rpos + len, baq_last)) - BAQ1(R, rpos)); inslen_sum += len; qpos += len; }
        else if (C_DEL == op) { nge += len; ngo++; delbaq_sum += (int)(BAQ1(R, lmin((long long)rpos + len, baq_last)) - BAQ1(R, rpos)); dellen_sum += len; rpos += len; }
        else if (op == C_MATCH || op == C_EQUAL || op == C_DIFF) { qpos += len; rpos += len; }
        else if (op == C_REF_SKIP) rpos += len;
        else if (op == C_SOFT_CLIP) qpos += len;
    }
    const int qlen = rend - apos;
    const int xm1500 = a.xm1500, go1500 = a.go1500;
    const int avg_gaplen = nge / imax(1, ngo);
    const int frag_pos_L = imin(apos, a.mpos), frag_pos_R = frag_pos_L + abs(a.isize);
    const bool isrc = (a.flag & 0x10) != 0;
    const int pcr_dp_inc = ((a.dflag & 0x4) ? 1 : 0), umi_dp_inc = ((a.dflag & 0x1) ? 1 : 0);
    const int atd = P.indel_adj_tracklen_dist;
    const int nrtr = (int)R.npos;
    const int isize = a.isize, l_qseq = a.l_qseq;
    const long long baq_apos = BAQ1(R, apos), baq_rlast = BAQ1(R, rend - 1);
    qpos = 0; rpos = apos;
    for (int i = 0; i < n_cigar; i++) {
        const int op = cig_op(cigar[i]), len = cig_len(cigar[i]);
        if (op == C_MATCH || op == C_EQUAL || op == C_DIFF) {
            for (int j = lane; j < len; j += 64) {
                const int rp = rpos + j, qp = qpos + j;
                const int64_t x = rp - off;
                if (pcr_dp_inc) atomicAdd(&P32(R, UVC_P_a_pcr_dp, x), pcr_dp_inc);
                if (umi_dp_inc) atomicAdd(&P32(R, UVC_P_a_umi_dp, x), umi_dp_inc);
                atomicAdd(&P32(R, UVC_P_a_dp, x), 1);
                atomicAdd(&P32(R, UVC_P_a_qlen, x), qlen);
                atomicAdd(&P32(R, UVC_P_a_XM1500, x), xm1500);
                atomicAdd(&P32(R, UVC_P_a_GO1500, x), go1500);
                atomicAdd(&P32(R, UVC_P_a_GAPLEN, x), avg_gaplen);
                if (isize != 0) {
                    if (isrc) { add64(&P64(R, UVC_P_a_LI, x), imin(rp - frag_pos_L + 1, MAX_INSERT_SIZE)); atomicAdd(&P32(R, UVC_P_a_LIDP, x), 1); }
                    else      { add64(&P64(R, UVC_P_a_RI, x), imin(frag_pos_R - rp, MAX_INSERT_SIZE));     atomicAdd(&P32(R, UVC_P_a_RIDP, x), 1); }
                }
                snv_dnv_scatter(R, bases, qp, l_qseq, apos, rend, rp);
                if (quals[qp] >= P.bias_thres_highBQ) {
                    atomicAdd(&P32(R, UVC_P_a_l_dist_sum, x), rp - apos + 1);
                    atomicAdd(&P32(R, UVC_P_a_r_dist_sum, x), rend - rp);
                    atomicAdd(&P32(R, UVC_P_a_inslen_sum, x), inslen_sum);
                    atomicAdd(&P32(R, UVC_P_a_dellen_sum, x), dellen_sum);
                    add64(&P64(R, UVC_P_a_l_BAQ_sum, x), (int)(BAQ1(R, rp) - baq_apos + 1));
                    add64(&P64(R, UVC_P_a_r_BAQ_sum, x), (int)(baq_rlast - BAQ1(R, rp) + 1));
                    add64(&P64(R, UVC_P_a_insBAQ_sum, x), insbaq_sum);
                    add64(&P64(R, UVC_P_a_delBAQ_sum, x), delbaq_sum);
                    atomicAdd(&P32(R, UVC_P_a_highBQ_dp, x), 1);
                }
            }
            qpos += len; rpos += len;
        } else if (op == C_INS || op == C_DEL) {
            const int i1 = imax(atd, rpos - off) - atd, i2 = imin(rpos - off + atd, nrtr - 1);
            const int t1 = RTRP(R, UVC_RTR_tracklen, i1), t2 = RTRP(R, UVC_RTR_tracklen, i2);
            const int unitlen2 = imax(1, (t1 > t2) ? RTRP(R, UVC_RTR_unitlen, i1) : RTRP(R, UVC_RTR_unitlen, i2));
            const int rtr_lo = imax((off + RTRP(R, UVC_RTR_begpos, i1)) - atd, apos);
            const int rtr_hi = imin((off + RTRP(R, UVC_RTR_begpos, i2) + t2) + atd, rend);
            const int inv100 = (int)(100u / ((0 == (unsigned)len % (unsigned)unitlen2) ? ((unsigned)len / (unsigned)unitlen2) : 4u));
            if (op == C_INS) {
                const int nbases = (int)((unsigned)len * (unsigned)P.indel_adj_indellen_perc / 100u);
                for (int r2 = imax(rpos - nbases, apos) + lane; r2 < imin(rpos + nbases, rend); r2 += 64) {
                    const int64_t x = r2 - off;
                    atomicAdd(&P32(R, UVC_P_a_near_ins_dp, x), 1);
                    add64(&P64(R, UVC_P_a_near_ins_pow2len, x), (long long)((unsigned)len * (unsigned)len));
                    add64(&P64(R, UVC_P_a_near_ins_l_pow2len, x), (long long)(r2 + 1 - (rpos - nbases)) * (r2 + 1 - (rpos - nbases)));
                    add64(&P64(R, UVC_P_a_near_ins_r_pow2len, x), (long long)((rpos + nbases) - r2) * ((rpos + nbases) - r2));
                    atomicAdd(&P32(R, UVC_P_a_near_ins_inv100len, x), inv100);
                }
                for (int r2 = rtr_lo + lane; r2 < rtr_hi; r2 += 64) atomicAdd(&P32(R, UVC_P_a_near_RTR_ins_dp, r2 - off), 1);
                if (lane == 0) atomicAdd(&P32(R, UVC_P_a_at_ins_dp, rpos - off), 1);
                qpos += len;
            } else {
                for (int r2 = rpos + lane; r2 < rpos + len; r2 += 64) {
                    const int64_t x = r2 - off;
                    if (pcr_dp_inc) atomicAdd(&P32(R, UVC_P_a_pcr_dp, x), pcr_dp_inc);
                    if (umi_dp_inc) atomicAdd(&P32(R, UVC_P_a_umi_dp, x), umi_dp_inc);
                    atomicAdd(&P32(R, UVC_P_a_dp, x), 1);
                    atomicAdd(&P32(R, UVC_P_a_qlen, x), qlen);
                    atomicAdd(&P32(R, UVC_P_a_highBQ_dp, x), 1);
                    atomicAdd(&P32(R, UVC_P_a_XM1500, x), xm1500);
                    atomicAdd(&P32(R, UVC_P_a_GO1500, x), go1500);
                    atomicAdd(&P32(R, UVC_P_a_GAPLEN, x), avg_gaplen);
                    if (isize != 0) {   // sic: the deletion start rpos, not r2 (main.hpp:1137-1145)
                        if (isrc) { add64(&P64(R, UVC_P_a_LI, x), imin(rpos - frag_pos_L + 1, MAX_INSERT_SIZE)); atomicAdd(&P32(R, UVC_P_a_LIDP, x), 1); }
                        else      { add64(&P64(R, UVC_P_a_RI, x), imin(frag_pos_R - rpos, MAX_INSERT_SIZE));     atomicAdd(&P32(R, UVC_P_a_RIDP, x), 1); }
                    }
                    atomicAdd(&P32(R, UVC_P_a_l_dist_sum, x), rpos - apos + 1);
                    atomicAdd(&P32(R, UVC_P_a_r_dist_sum, x), rend - rpos);
                    atomicAdd(&P32(R, UVC_P_a_inslen_sum, x), inslen_sum);
                    atomicAdd(&P32(R, UVC_P_a_dellen_sum, x), dellen_sum);
                    add64(&P64(R, UVC_P_a_l_BAQ_sum, rpos - off), (int)(BAQ1(R, rpos) - baq_apos + 1));   // sic: at rpos (main.hpp:1156-1157)
                    add64(&P64(R, UVC_P_a_r_BAQ_sum, rpos - off), (int)(baq_rlast - BAQ1(R, rpos) + 1));
                    add64(&P64(R, UVC_P_a_insBAQ_sum, x), insbaq_sum);
                    add64(&P64(R, UVC_P_a_delBAQ_sum, x), delbaq_sum);
                }
                const int nbases_l = (int)((unsigned)len * (unsigned)(P.indel_adj_indellen_perc - 100) / 100u);
                const int nbases_r = (int)((unsigned)len * (unsigned)P.indel_adj_indellen_perc / 100u);
                const int lpos = imax(rpos - nbases_l, apos), rpos_r = imin(rpos + nbases_r, rend) - 1;
                for (int r2 = lpos + lane; r2 <= rpos_r; r2 += 64) {
                    const int64_t x = r2 - off;
                    atomicAdd(&P32(R, UVC_P_a_near_del_dp, x), 1);
                    add64(&P64(R, UVC_P_a_near_del_pow2len, x), (long long)((unsigned)len * (unsigned)len));
                    add64(&P64(R, UVC_P_a_near_del_l_pow2len, x), (long long)(r2 - lpos + 1) * (r2 - lpos + 1));
                    add64(&P64(R, UVC_P_a_near_del_r_pow2len, x), (long long)(rpos_r - r2 + 1) * (rpos_r - r2 + 1));
                    atomicAdd(&P32(R, UVC_P_a_near_del_inv100len, x), inv100);
                }
                for (int r2 = rtr_lo + lane; r2 < rtr_hi; r2 += 64) atomicAdd(&P32(R, UVC_P_a_near_RTR_del_dp, r2 - off), 1);
                if (lane == 0) atomicAdd(&P32(R, UVC_P_a_at_del_dp, rpos - off), 1);
                rpos += len;
            }
        } else {
            if ((C_SOFT_CLIP == op || C_HARD_CLIP == op) && lane == 0) clip_event(R, P, rpos, i, len, pcr_dp_inc);
            if (op == C_REF_SKIP) rpos += len; else if (op == C_SOFT_CLIP) qpos += len;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// P1b (update_seg_format_thres_from_prep_sets, main.hpp:1206-1299)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_thres(RegionDev R, UvcParams P, int half_ratio_phred) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= R.npos) return;
    const bool is_normal = P.tumor_vcf_is_provided;
    const int segLIDP = imax(P32(R, UVC_P_a_LIDP, x), 1), segRIDP = imax(P32(R, UVC_P_a_RIDP, x), 1);
    const int ins_dp = P32(R, UVC_P_a_near_ins_dp, x), del_dp = P32(R, UVC_P_a_near_del_dp, x);
    const double ins_l = ceil(sqrt((double)(P64(R, UVC_P_a_near_ins_l_pow2len, x) / imax(ins_dp, 1))));
    const double del_l = ceil(sqrt((double)(P64(R, UVC_P_a_near_del_l_pow2len, x) / imax(del_dp, 1))));
    const double ins_r = ceil(sqrt((double)(P64(R, UVC_P_a_near_ins_r_pow2len, x) / imax(ins_dp, 1))));
    const double del_r = ceil(sqrt((double)(P64(R, UVC_P_a_near_del_r_pow2len, x) / imax(del_dp, 1))));
    const int dnv_border = ((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform && (P32(R, UVC_P_a_dnv_dp, x) * 2 > P32(R, UVC_P_a_snv_dp, x))) ? 10 : 0);
    TH(R, UVC_T_aLPxT, x) = (int)(fmax(ins_l, fmax(del_l, (double)dnv_border)) + P.bias_thres_aLPxT_add);
    TH(R, UVC_T_aRPxT, x) = (int)(fmax(ins_r, fmax(del_r, (double)dnv_border)) + P.bias_thres_aLPxT_add);
    int ip = RTRP(R, UVC_RTR_indelphred, x);
    if (ins_dp * P.indel_del_to_ins_err_ratio < del_dp) ip += half_ratio_phred;
    if (del_dp * P.indel_del_to_ins_err_ratio < ins_dp) ip -= half_ratio_phred;
    const int pc_inc1 = (int)(3 * 100 * imax(1, ins_dp + del_dp) / (imax(1, P32(R, UVC_P_a_near_ins_inv100len, x) + P32(R, UVC_P_a_near_del_inv100len, x)))) - 3;
    ip += ibetween(pc_inc1, 0, 6);
    RTRP(R, UVC_RTR_indelphred, x) = imax(ip, 0);
    const int p1T = (is_normal ? P.bias_thres_aLRI1NT_perc : P.bias_thres_aLRI1T_perc), p1t = (is_normal ? P.bias_thres_aLRI1Nt_perc : P.bias_thres_aLRI1t_perc);
    const long long LI = P64(R, UVC_P_a_LI, x), RI = P64(R, UVC_P_a_RI, x);
    TH(R, UVC_T_aLI1T, x) = (int)(LI * p1T / (segLIDP * 100) + P.bias_thres_aLRI1T_add);
    TH(R, UVC_T_aLI2T, x) = (int)(LI * P.bias_thres_aLRI2T_perc / (segLIDP * 100) + P.bias_thres_aLRI2T_add);
    TH(R, UVC_T_aLI1t, x) = (int)(LI * p1t / (segLIDP * 100));
    TH(R, UVC_T_aLI2t, x) = (int)(LI * P.bias_thres_aLRI2t_perc / (segLIDP * 100));
    TH(R, UVC_T_aRI1T, x) = (int)(RI * p1T / (segRIDP * 100) + P.bias_thres_aLRI1T_add);
    TH(R, UVC_T_aRI2T, x) = (int)(RI * P.bias_thres_aLRI2T_perc / (segRIDP * 100) + P.bias_thres_aLRI2T_add);
    TH(R, UVC_T_aRI1t, x) = (int)(RI * p1t / (segRIDP * 100));
    TH(R, UVC_T_aRI2t, x) = (int)(RI * P.bias_thres_aLRI2t_perc / (segRIDP * 100));
    const int pP1 = (is_normal ? P.bias_thres_aLRP1Nt_avgmul_perc : P.bias_thres_aLRP1t_avgmul_perc), pP2 = P.bias_thres_aLRP2t_avgmul_perc;
    const int pB1 = (is_normal ? P.bias_thres_aLRB1Nt_avgmul_perc : P.bias_thres_aLRB1t_avgmul_perc), pB2 = P.bias_thres_aLRB2t_avgmul_perc;
    const int hb = P32(R, UVC_P_a_highBQ_dp, x);
    const long long den = imax(1, hb * 100);
    const long long lds = P32(R, UVC_P_a_l_dist_sum, x), rds = P32(R, UVC_P_a_r_dist_sum, x);
    TH(R, UVC_T_aLP1t, x) = (int)nnminus(lds * pP1 / den, P.bias_thres_aLRP1t_minus);
    TH(R, UVC_T_aLP2t, x) = (int)nnminus(lds * pP2 / den, P.bias_thres_aLRP2t_minus);
    TH(R, UVC_T_aRP1t, x) = (int)nnminus(rds * pP1 / den, P.bias_thres_aLRP1t_minus);
    TH(R, UVC_T_aRP2t, x) = (int)nnminus(rds * pP2 / den, P.bias_thres_aLRP2t_minus);
    const long long pdel = P64(R, UVC_P_a_delBAQ_sum, x) / imax(1, hb);
    const long long lb = P64(R, UVC_P_a_l_BAQ_sum, x), rb = P64(R, UVC_P_a_r_BAQ_sum, x);
    TH(R, UVC_T_aLB1t, x) = (int)nnminus(lb * pB1 / den, P.bias_thres_aLRB1t_minus + pdel);
    TH(R, UVC_T_aLB2t, x) = (int)nnminus(lb * pB2 / den, P.bias_thres_aLRB2t_minus);
    TH(R, UVC_T_aRB1t, x) = (int)nnminus(rb * pB1 / den, P.bias_thres_aLRB1t_minus + pdel);
    TH(R, UVC_T_aRB2t, x) = (int)nnminus(rb * pB2 / den, P.bias_thres_aLRB2t_minus);
}

// ------------------------------------------------------------------------------------------------
// contribution of a SIMPLE alignment at reference position p (BASE_QUALITY_MAX values, no bias)
// main.hpp:1918-1924 (LINK_M), 1949-1980 (base)
// ------------------------------------------------------------------------------------------------
DEV int simple_link_value(const RegionDev &R, const UvcParams &P, const AlnRec &a, int p, const uint8_t *quals_qbase, bool proton) {
    const int64_t x = p - R.beg;
    const int noindel = imin(RTRP(R, UVC_RTR_indelphred, x - 1), RTRP(R, UVC_RTR_indelphred, x));
    const int qfromBQ2 = (proton ? imin((int)quals_qbase[p - 1], (int)quals_qbase[p]) : 80);
    return (int)nnminus(imin(qfromBQ2, noindel), a.nogap_penal) + 1;
}
DEV int simple_base_value(const UvcParams &P, const AlnRec &a, int p, const uint8_t *quals_qbase, bool proton) {
    const int q = quals_qbase[p];
    if (proton) {
        const int i2 = p - a.pos, len = a.rend - a.pos;
        if ((0 == i2) || (len - 1 == i2)) {
            // packed neighbouring cigar words never equal the bare op codes (main.hpp:1953-1956), so both flags hold at the op ends
            const bool next_gap = (len - 1 == i2), prev_gap = (0 == i2);
            const bool isrc2 = (0 != i2);
            const int qpos = (int)(p - a.pos) + a.lclip_q;
            int prev_base_phred = 1;
            if (isrc2 && (qpos + 1 < a.l_qseq)) prev_base_phred = quals_qbase[p + 1];
            if ((!isrc2) && (qpos > 0)) prev_base_phred = quals_qbase[p - 1];
            int adj = 100;
            if (next_gap) adj = imin(adj, (a.rclip_oplen > 0 ? a.rclip_oplen : 100));
            if (prev_gap) adj = imin(adj, (a.lclip_oplen > 0 ? a.lclip_oplen : 100));
            if (adj < 3) return imin(q, prev_base_phred) + imin(P.bq_phred_added_misma, P.bq_phred_added_indel);
            return imin(q, prev_base_phred) + P.bq_phred_added_misma;
        }
    }
    return q + P.bq_phred_added_misma;
}

// one queued mismatching base of a simple alignment: dealwith_segbias<false> into a scratch record, flushed with atomics
DEV void mis_apply(const RegionDev &R, const UvcParams &P, const MisItem &it) {
    const int64_t x = it.epos - R.beg;
    const int sym = it.symval & 0xFF, inc = it.symval >> 8;
    if ((unsigned)it.rank >= (unsigned)R.n_alns || x < 0 || x >= R.npos || sym > UVC_BASE_NN) { atomicExch(R.err, UVCGPU_EDEVICE); return; }   // corrupt queue entry
    const AlnRec &a = R.alns[it.rank];
    const SegRead sr = make_segread(R, a);
    PosThres T;
    load_thres(R, T, x);
    SegAcc A; A.zero();
    A.bq = inc;
    segbias<false>(A, P, sr, T, it.epos, R.baq[x], R.baq[R.npos + x], inc, a.bm1500[sym], C_MATCH, 0, 10000);
    seg_flush(R, A, sym, x);
}

#ifndef P2_AHEAD
#define P2_AHEAD 2
#endif
#define MISQ_CAP 192   // per-wave LDS queue of mismatching bases (flushed to the global queue when fewer than 64 slots are left)
// ------------------------------------------------------------------------------------------------
// P2 fast: updateByAln<SYMBOL_COUNT_SUM, bias> for simple alignments, one lane per position
// ------------------------------------------------------------------------------------------------
// Two instantiations, <true,false> for LINK_M and <false,true> for the read bases: each keeps one SegAcc in registers,
// which halves the accumulator footprint and doubles the waves per SIMD.
// SPLIT (see k_prep_fast): a block per window, its four waves take every fourth chunk of the window's entries; their counters are added up
// in LDS (red: one slot per counter and lane) and stored by the four waves together.
#define P2_RED_SLOTS (UVC_NSEG32 + 2 * UVC_NSEG64 + 5)
// the 64-bit counter f of the LDS reduction area: two int rows of 64 = one row of 64 unsigned long long
DEV unsigned long long *red64(int (*red)[64], int f) { return (unsigned long long *)&red[UVC_NSEG32 + 2 * f][0]; }
template <bool DO_L, bool DO_B, bool PLAIN, bool SPLIT = false>
DEV void p2_fast_body(const RegionDev &R, const UvcParams &P, int *amp1, int *amp2, MisItem (*misq)[DO_B ? MISQ_CAP : 1], int (*red)[64] = nullptr) {
    COARSE_T(ct0)
    {
        const int v = threadIdx.x;
        amp1[v] = (v < P.bias_thres_PFBQ1 ? 100 * (v * v) / (P.bias_thres_PFBQ1 * P.bias_thres_PFBQ1) : 100);
        amp2[v] = (v < P.bias_thres_PFBQ2 ? 100 * (v * v) / (P.bias_thres_PFBQ2 * P.bias_thres_PFBQ2) : 100);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = SPLIT ? wave_uniform((int)xcd_block()) : wave_uniform((int)((xcd_block() * blockDim.x + threadIdx.x) >> 6));
    const int64_t x0 = (int64_t)wave * 64;
    if (x0 >= R.npos) return;   // (block-uniform in the split form)
    if (SPLIT) { for (int i2 = threadIdx.x; i2 < P2_RED_SLOTS * 64; i2 += 256) (&red[0][0])[i2] = 0; __syncthreads(); }
    const int w0 = R.beg + (int)x0;
    const int p = w0 + lane;
    const int64_t x = x0 + lane;
    const bool valid = x < R.npos;
    const bool proton = (PLAIN ? false : (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform));
    const bool normal_filter_primers = (P.tn_is_paired && (0x1 & P.primer_flag));
    const int my_ref = valid ? R.refsym[x] : 0;
    PosThres T;
    long long baq_p = 0, baq2_p = 0;
    if (valid) { load_thres(R, T, x); baq_p = R.baq[x]; baq2_p = R.baq[R.npos + x]; }
    else { for (int f = 0; f < UVC_NTHRES; f++) T.t[f] = 0; }
    SegAcc Aref, Alink;   // only the one selected by the template arguments stays live
    Aref.zero(); Alink.zero();
    // LINK_M value of a simple read at this position is a per-position constant up to the read's penalty (main.hpp:1919-1923)
    int noindel80 = 80;
    if (DO_L && valid && x > 0) noindel80 = imin(80, imin(RTRP(R, UVC_RTR_indelphred, x - 1), RTRP(R, UVC_RTR_indelphred, x)));
    MisItem *myq = misq[DO_B ? (threadIdx.x >> 6) : 0];
    int nq = 0;   // wave-uniform: only updated in uniform control flow
    auto flush_queue = [&]() {
        int base = 0;
        if (lane == 0) base = atomicAdd(R.mis_cnt, nq);
        base = wave_uniform(base);
        for (int i = lane; i < nq; i += 64) {
            if (base + i < R.mis_cap) R.mis[base + i] = myq[i];
            else atomicExch(R.err, UVCGPU_EDEVICE);   // cannot happen: the queue holds every mismatching base of the simple alignments
        }
        nq = 0;
    };
    const __amdgpu_buffer_rsrc_t rs = bq_rsrc(R);
    // the work list is stored as four pos-sorted sub-lists, one per (is-reverse, bam_get_strand) class
    auto run_list = [&](auto IS, auto ST, int cls) {
    constexpr bool ISRC = decltype(IS)::value, STRAND = decltype(ST)::value;
    const int lo = wave_uniform(win_lo(R, 1 + cls, (int)(x0 >> 6))), hi = wave_uniform(win_hi(R, 1 + cls, (int)(x0 >> 6)));
    for (int k0 = lo + (SPLIT ? 64 * (int)(threadIdx.x >> 6) : 0); k0 < hi; k0 += (SPLIT ? 256 : 64)) {
        Chunk16 c;
        if (DO_B) load_chunk16(R.frec2, k0 + lane, hi, c);   // (the LINK-only pass reads its records through the scalar cache, below)
        const int n = imin(64, hi - k0);
        // (the base pass requests the base | quality bytes P2_AHEAD reads ahead: one load per read and lane at HBM / L2 latency)
        auto fetch = [&](int j) { return bq_load(rs, bcast(c.v[2], j) + p); };   // low word of qbase + p: exact for lanes inside the read, harmless elsewhere
        auto one = [&](auto K, int bqn) {   // K(i): dword i of the work-list record
            const int sym = bqn & 0xFF, q = (bqn >> 8) & 0xFF;
            const int apos = K(0), rend = K(1);
            if (rend <= w0) return;
            const int fmd = K(4);
            const int ext = K(15);
            SegRead sr;   // [apos, rend) is what this entry covers; the bias arithmetic uses the ends of the whole alignment
            sr.pos = apos - (ext & 0xFFFF); sr.rend = rend + (int)((unsigned)ext >> 16); sr.flag = fmd & 0xFFFF; sr.mapq = (fmd >> 16) & 0xFF; sr.dflag = (fmd >> 24) & 0xFF;
            sr.isize = K(5);
            const int mpos = K(6);
            sr.frag_pos_L = imin(sr.pos, mpos); sr.frag_pos_R = sr.frag_pos_L + abs(sr.isize);
            sr.xm1500 = K(7);
            const int bm4c = K(10);
            sr.clip_cnt = (bm4c >> 16) & 0xF;
            const int nogap = (int)(short)(bm4c & 0xFFFF);
            sr.baq_pos = K(12); sr.baq_last = K(13); sr.baq2_last = K(14);
            const bool is_assay_amplicon = (PLAIN ? false : ((sr.dflag & 0x4) || ((P.primerlen > 0) && !(0x2 & P.primer_flag))));
            bool gate = true;
            if (is_assay_amplicon && !normal_filter_primers) {   // primer gating, main.hpp:1872-1875, 1895
                constexpr bool isrc = ISRC;
                const int ibeg = ((sr.isize != 0) ? (sr.frag_pos_L + P.primerlen) : ((isrc && (0x0 == (0x1 & sr.flag))) ? 0 : (sr.pos + P.primerlen)));
                const int iend = ((sr.isize != 0) ? (int)nnminus(sr.frag_pos_L + abs(sr.isize), P.primerlen) : ((isrc && (0x0 == (0x1 & sr.flag))) ? (int)nnminus(sr.rend, P.primerlen) : INT32_MAX));
                gate = (ibeg <= p && p < iend);
            }
            // the queue bookkeeping below must run in wave-uniform control flow (nq is a scalar): no divergent `continue` before it
            // (one mask, not a chain of short-circuit tests: every `&&` of per-lane conditions is an exec-mask level of its own -- save, branch, restore)
            const bool cover = (valid & (p >= apos) & (p < rend) & gate);
            const bool hasL = DO_L && (p > apos), hasB = DO_B && (sym == my_ref);
            int inc = 0, incL = 0;
            if (proton) {   // IonTorrent values need neighbouring qualities and clip lengths: take them from the full record
                if (cover) {
                    const AlnRec &a = R.alns[K(3)];
                    if (DO_B) inc = simple_base_value(P, a, p, R.quals + a.qbase, true);
                    if (DO_L) incL = (p > apos ? simple_link_value(R, P, a, p, R.quals + a.qbase, true) : 0);
                }
            } else { inc = q + P.bq_phred_added_misma; incL = (int)nnminus(noindel80, nogap) + 1; }
            const wmask mm = DO_B ? BAL(cover && !hasB) : 0ull;
            if (DO_B && mm) {   // bases that differ from the reference go to the wave's queue; k_p2_mism applies them
                if (nq > MISQ_CAP - 64) flush_queue();
                if (cover && !hasB) { MisItem it; it.rank = K(3); it.epos = p; it.symval = sym | (inc << 8); myq[nq + (int)__builtin_popcountll(mm & ((1ull << lane) - 1ull))] = it; }
                nq += (int)__builtin_popcountll(mm);
            }
            // a pass with one side acts on a lane iff that side has something there (one exec-mask level instead of cover, then hasL / hasB inside)
            constexpr bool ONE_SIDE = (DO_B != DO_L);
            if (!ONE_SIDE ? cover : (DO_B ? (cover & hasB) : (cover & hasL))) {
                const int bmv = K(8), xbv = K(9);
                const int bm_inc = (sym < 4 ? ((bmv >> (8 * sym)) & 0xFF) : (xbv & 0xFF)), xm_inc = (xbv >> 8) & 0xFF;
                segbias_simple<ISRC, STRAND, PLAIN>(Alink, Aref, P, sr, T, p, baq_p, baq2_p, ONE_SIDE ? DO_L : hasL, ONE_SIDE ? DO_B : hasB, incL, inc, xm_inc, bm_inc, amp1, amp2);
            }
        };
        if (DO_B) {
            int r0 = fetch(0), r1 = (1 < n ? fetch(1) : 0);
#if P2_AHEAD == 4
            int r2 = (2 < n ? fetch(2) : 0), r3 = (3 < n ? fetch(3) : 0);
            for (int j = 0; j < n; j += 4) {
                { const int v = r0; if (j + 4 < n) r0 = fetch(j + 4); one([&](int i) { return bcast(c.v[i], j); }, v); }
                if (j + 1 < n) { const int v = r1; if (j + 5 < n) r1 = fetch(j + 5); one([&](int i) { return bcast(c.v[i], j + 1); }, v); }
                if (j + 2 < n) { const int v = r2; if (j + 6 < n) r2 = fetch(j + 6); one([&](int i) { return bcast(c.v[i], j + 2); }, v); }
                if (j + 3 < n) { const int v = r3; if (j + 7 < n) r3 = fetch(j + 7); one([&](int i) { return bcast(c.v[i], j + 3); }, v); }
            }
#else
#ifdef UVC_P2_BASE_VECREC
            for (int j = 0; j < n; j += 2) {
                { const int v = r0; if (j + 2 < n) r0 = fetch(j + 2); one([&](int i) { return bcast(c.v[i], j); }, v); }
                if (j + 1 < n) { const int v = r1; if (j + 3 < n) r1 = fetch(j + 3); one([&](int i) { return bcast(c.v[i], j + 1); }, v); }
            }
#else
            // the record's fields through the scalar cache as in the LINK pass; the per-lane copy of the chunk stays for the one field the byte
            // requests need two reads ahead (fetch)
            typedef int v4i_ __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(4))) const v4i_ *crec4;
            v4i_ a0, a1, a2, a3;
            const int kb = wave_uniform(k0), nb = wave_uniform(n);
            { crec4 q = (crec4)(R.frec2 + kb); a0 = q[0]; a1 = q[1]; a2 = q[2]; a3 = q[3]; }
            for (int j = 0; j < nb; j += 2) {
                { const int v = r0; if (j + 2 < nb) r0 = fetch(j + 2);
                  const v4i_ c0 = a0, c1 = a1, c2 = a2, c3 = a3;
                  if (j + 1 < nb) { crec4 q = (crec4)(R.frec2 + (kb + j + 1)); a0 = q[0]; a1 = q[1]; a2 = q[2]; a3 = q[3]; }
                  one([&](int i) { const v4i_ w = (i < 4 ? c0 : i < 8 ? c1 : i < 12 ? c2 : c3); return (int)w[i & 3]; }, v); }
                if (j + 1 < nb) { const int v = r1; if (j + 3 < nb) r1 = fetch(j + 3);
                  const v4i_ c0 = a0, c1 = a1, c2 = a2, c3 = a3;
                  if (j + 2 < nb) { crec4 q = (crec4)(R.frec2 + (kb + j + 2)); a0 = q[0]; a1 = q[1]; a2 = q[2]; a3 = q[3]; }
                  one([&](int i) { const v4i_ w = (i < 4 ? c0 : i < 8 ? c1 : i < 12 ? c2 : c3); return (int)w[i & 3]; }, v); }
            }
#endif
#endif
        } else {
            // The LINK pass reads no read bytes: its only per-read data is the 64-byte work-list record.  Straight into scalar registers (s_load through
            // the constant address space, the next record requested while this one is worked on) instead of a per-lane vector load of 64 records and
            // eleven v_readlane per read.
            typedef int v4i_ __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(4))) const v4i_ *crec4;
            v4i_ a0, a1, a2, a3;
            const int kb = wave_uniform(k0), nb = wave_uniform(n);   // (the split form's k0 depends on the wave's number in the block: uniform, but not to the compiler)
            { crec4 q = (crec4)(R.frec2 + kb); a0 = q[0]; a1 = q[1]; a2 = q[2]; a3 = q[3]; }
            for (int j = 0; j < nb; j++) {
                const v4i_ c0 = a0, c1 = a1, c2 = a2, c3 = a3;   // (sixteen s_mov; two register sets and a loop unrolled by two measured slower: 1.27 against 1.24 ms)
                if (j + 1 < nb) { crec4 q = (crec4)(R.frec2 + (kb + j + 1)); a0 = q[0]; a1 = q[1]; a2 = q[2]; a3 = q[3]; }
                one([&](int i) { const v4i_ w = (i < 4 ? c0 : i < 8 ? c1 : i < 12 ? c2 : c3); return (int)w[i & 3]; }, 0);
            }
        }
        if (DO_B) Aref.fold();
        if (DO_L) Alink.fold();
    }
    };
    COARSE_T(ct1)
    run_list(std::false_type{}, std::false_type{}, 0);
    run_list(std::true_type{}, std::false_type{}, 1);
    run_list(std::false_type{}, std::true_type{}, 2);
    run_list(std::true_type{}, std::true_type{}, 3);
    if (DO_B && nq > 0) flush_queue();
    COARSE_T(ct2)
    if (SPLIT) {   // the four partial sets -> LDS -> one set (the two passes own different symbols: one symbol per kernel)
        const SegAcc &A = (DO_B ? Aref : Alink);
#pragma unroll
        for (int f = 0; f < UVC_NSEG32; f++) if (A.s[f]) atomicAdd(&red[f][lane], A.s[f]);
#pragma unroll
        for (int f = 0; f < UVC_NSEG64; f++) if (A.l[f]) atomicAdd(&red64(red, f)[lane], (unsigned long long)A.l[f]);
        if (A.a1BQf) atomicAdd(&red[UVC_NSEG32 + 2 * UVC_NSEG64 + 0][lane], A.a1BQf);
        if (A.a1BQr) atomicAdd(&red[UVC_NSEG32 + 2 * UVC_NSEG64 + 1][lane], A.a1BQr);
        if (A.a2BQf) atomicAdd(&red[UVC_NSEG32 + 2 * UVC_NSEG64 + 2][lane], A.a2BQf);
        if (A.a2BQr) atomicAdd(&red[UVC_NSEG32 + 2 * UVC_NSEG64 + 3][lane], A.a2BQr);
        if (A.bq) atomicAdd(&red[UVC_NSEG32 + 2 * UVC_NSEG64 + 4][lane], A.bq);
        __syncthreads();
        if (!valid) return;
        const int sym = (DO_B ? my_ref : UVC_LINK_M), wv = (int)(threadIdx.x >> 6);
        for (int f = wv; f < UVC_NSEG32; f += 4) { const int v = red[f][lane]; if (v) S32(R, f, sym, x) = v; }
        for (int f = wv; f < UVC_NSEG64; f += 4) { const long long v = (long long)red64(red, f)[lane]; if (v) S64(R, f, sym, x) = v; }
        if (wv == 0) {
            const int b0 = UVC_NSEG32 + 2 * UVC_NSEG64;
            if (red[b0][lane]) VQP(R, UVC_VQ_a1BQf, sym, x) = red[b0][lane];
            if (red[b0 + 1][lane]) VQP(R, UVC_VQ_a1BQr, sym, x) = red[b0 + 1][lane];
            if (red[b0 + 2][lane]) VQP(R, UVC_VQ_a2BQf, sym, x) = red[b0 + 2][lane];
            if (red[b0 + 3][lane]) VQP(R, UVC_VQ_a2BQr, sym, x) = red[b0 + 3][lane];
            if (red[b0 + 4][lane]) BQS(R, sym, x) = red[b0 + 4][lane];
        }
        return;
    }
    if (!valid) return;
    // k_p2_fast runs before every other writer of these planes (k_p2_mism, k_p2_items), and the two instantiations own
    // different symbols: plain stores
    if (DO_B) seg_store(R, Aref, my_ref, x);
    if (DO_L) seg_store(R, Alink, UVC_LINK_M, x);
#ifdef UVC_FRAG_COARSE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    COARSE_T(ct3)
    if (lane == 0 && (wave % 2999) == 7) printf("coarse p2<%d,%d> wave %d prologue %llu lists %llu epilogue %llu\n", (int)DO_L, (int)DO_B, wave, ct1 - ct0, ct2 - ct1, ct3 - ct2);
#endif
}
template <bool DO_L, bool DO_B, bool PLAIN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) k_p2_fast(RegionDev R, UvcParams P) {
    __shared__ int amp1[256], amp2[256];
    __shared__ MisItem misq[DO_B ? 4 : 1][DO_B ? MISQ_CAP : 1];
    p2_fast_body<DO_L, DO_B, PLAIN>(R, P, amp1, amp2, misq);
}
template <bool DO_L, bool DO_B, bool PLAIN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) k_p2_fast_split(RegionDev R, UvcParams P) {
    __shared__ int amp1[256], amp2[256];
    __shared__ MisItem misq[DO_B ? 4 : 1][DO_B ? MISQ_CAP : 1];
    __shared__ __attribute__((aligned(8))) int red[P2_RED_SLOTS][64];
    p2_fast_body<DO_L, DO_B, PLAIN, true>(R, P, amp1, amp2, misq, red);
}

// ------------------------------------------------------------------------------------------------
// k_p2_mism: the queued mismatching bases of simple alignments, one lane per item
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_p2_mism(RegionDev R, UvcParams P) {
    const int n = imin(*R.mis_cnt, R.mis_cap);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) mis_apply(R, P, R.mis[t]);
}

// ------------------------------------------------------------------------------------------------
// sequential updateByAln for one complex alignment (main.hpp:1762-2296).
//   BIAS = true : SYMBOL_COUNT_SUM + dealwith_segbias, flushed with atomics (P2)
//   BIAS = false: BASE_QUALITY_MAX values written to the contribution table (used by P3/P4/P5)
// ------------------------------------------------------------------------------------------------
DEV int indel_len_rusize_phred_dev(int indel_len, int repeatunit_size) {   // main.hpp:757-790
    const int t[19] = { 0, 0, 3, 5, 6, 7, 8, 8, 9, 10, 10, 10, 11, 11, 11, 12, 12, 12, 13 };
    if (0 == (indel_len % repeatunit_size)) return t[imin(indel_len / repeatunit_size, 18)];
    return t[imin(indel_len, 18)];
}
DEV int indel_phred_dev(double ampfact, int rs, int rn) {   // main.hpp:794-801
    const int region_size = rs * rn;
    const double num_slips = (region_size > 64 ? (double)(region_size - 8) : log1p(exp((double)region_size - (double)8))) * ampfact / ((double)(rs * rs));
    const double pr = (1.0 - 2.220446049250313e-16) / (num_slips + 1.0);
    return (int)floor(-10 * log(pr) / log(10.0));
}
DEV bool more_STR(int rulen1, int rc1, int rulen2, int rc2, int strmax) {   // is_indel_context_more_STR, main.hpp:699-721
    if (rulen2 * rc2 == 0) return true;
    if (rulen1 > strmax || rulen2 > strmax) return (rulen1 < rulen2 || (rulen1 == rulen2 && rc1 > rc2));
    int rank1 = (rc1 <= 1 ? (-rc1 * rulen1) : ((rc1 - 1) * rulen1));
    int rank2 = (rc2 <= 1 ? (-rc2 * rulen1) : ((rc2 - 1) * rulen2));
    if (0 == rc1 || 0 == rulen1) rank1 = -100;
    if (0 == rc2 || 0 == rulen2) rank2 = -100;
    return rank1 > rank2;
}
DEV int ref_to_phredvalue_dev(int &n_units, int &max_rn, int &rs_at_max, const RegionDev &R, const UvcParams &P, int refidx, int max_phred, double ampfact, int oplen, int op) {   // main.hpp:876-922
    const int n = (int)R.npos - 1;   // refstring.size()
    max_rn = 0; rs_at_max = 0;
    for (int rs = 1; rs <= P.indel_str_repeatsize_max; rs++) {
        int q = refidx;
        while (q + rs < n && R.refsym[q] == R.refsym[q + rs]) q++;
        const int rn = (q - refidx) / rs + 1;
        if (more_STR(rs, rn, rs_at_max, max_rn, P.indel_str_repeatsize_max)) { max_rn = rn; rs_at_max = rs; }
    }
    if (oplen == rs_at_max && op == C_DEL) ampfact *= P.indel_del_to_ins_err_ratio;
    const int decphred = indel_phred_dev(ampfact, rs_at_max, max_rn);
    if (rs_at_max * (max_rn - 1) >= 6 - 1) n_units = ((0 == oplen % rs_at_max) ? (oplen / rs_at_max) : ((1 == oplen) ? 1 : 0));
    else n_units = 1 + (oplen / 6);
    return max_phred - imin(max_phred, decphred) + indel_len_rusize_phred_dev(oplen, rs_at_max);
}
DEV int proton_cigarlen2phred(int cigarlen) { const int t[13] = { 0, 0, 9, 14, 18, 21, 23, 25, 27, 29, 30, 31, 32 }; return t[imin(cigarlen, 12)]; }

DEV void table_put(const RegionDev &R, Contrib *row, int sym, int v) {
    const uint16_t vv = (uint16_t)imin(imax(v, 0), 65535);
    if (sym <= UVC_BASE_NN) {
        if (row->bsym_p1 == 0 || row->bsym_p1 == sym + 1) { row->bval = (row->bsym_p1 == sym + 1) ? (uint16_t)imax((int)row->bval, (int)vv) : vv; row->bsym_p1 = (uint8_t)(sym + 1); }
        else atomicExch(R.err, UVCGPU_EDEVICE);   // two different bases of one read at one reference position: no CIGAR does that
        return;
    }
    const int k = sym - UVC_LINK_M;   // one slot per LINK symbol: any pile fits
    if (row->lmask & (1 << k)) row->lval[k] = (uint16_t)imax((int)row->lval[k], (int)vv);
    else { row->lmask = (uint8_t)(row->lmask | (1 << k)); row->lval[k] = vv; }
}

template <bool BIAS>
__global__ void __launch_bounds__(64) k_p2_slow(RegionDev R, UvcParams P) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R.n_complex) return;
    const AlnRec a = R.alns[R.complex_ids[t]];
    const uint32_t *cigar = R.cigars + a.cigar_off;
    const uint8_t *bases = R.bases + a.seq_off, *quals = R.quals + a.seq_off;
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    const int off = R.beg, rend = a.rend, n_cigar = a.n_cigar;
    const bool is_assay_amplicon = ((a.dflag & 0x4) || ((P.primerlen > 0) && !(0x2 & P.primer_flag)));
    const bool normal_filter_primers = (P.tn_is_paired && (0x1 & P.primer_flag));
    const int addMis = P.bq_phred_added_misma, addInd = P.bq_phred_added_indel;
    Contrib *table = BIAS ? nullptr : (R.table + a.table_off);
    auto Q = [&](int q) -> int { return (int)quals[imin(imax(q, 0), a.l_qseq - 1)]; };
    Item *items = BIAS ? (R.items + a.item_off) : nullptr;
    int n_items = 0;
    auto emit = [&](bool gap, int epos, int sym, int v, int bm, int op, int indel_len, int dist) {
        if (BIAS) {   // the bias update itself is applied by k_p2_items, in parallel over the items
            Item it; it.epos = epos; it.sym = (uint8_t)sym; it.flags = (uint8_t)((gap ? 1 : 0) | (op << 1)); it.val = (uint16_t)imin(imax(v, 0), 65535);
            it.dist = (uint16_t)imin(imax(dist, 0), 65535); it.indel_len = (uint16_t)imin(indel_len, 65535); it.pad2 = 0;
            items[n_items++] = it;
        } else if (epos >= a.pos && epos <= a.rend) table_put(R, table + (epos - a.pos), sym, v);
        else atomicExch(R.err, UVCGPU_EDEVICE);   // would leave the read's rows
    };
    // low-BQ InDel positions (main.hpp:1817-1859): a list of the read's own in global memory (one slot per I / D op + two sentinels), so
    // that a read with any number of them is walked like every other read
    int32_t *indel_rposs = R.ir_list + a.gap_off + 2 * (int64_t)t; int n_ir = 0;
    if (BIAS) indel_rposs[n_ir++] = 0;
    int nge = 0;
    {
        int qpos = 0, rpos = a.pos;
        for (int i = 0; i < n_cigar; i++) {
            const int op = cig_op(cigar[i]), len = cig_len(cigar[i]);
            if (op == C_MATCH || op == C_EQUAL || op == C_DIFF) { qpos += len; rpos += len; }
            else if (op == C_INS) {
                nge += len;
                bool low = false;
                for (int q2 = qpos - imin(qpos, 1); q2 < imin(qpos + len + 1, rend); q2++) if (Q(q2) < P.bias_thres_interfering_indel_BQ) low = true;
                if (low && BIAS) indel_rposs[n_ir++] = rpos;
                qpos += len;
            } else if (op == C_DEL) {
                nge += len;
                const bool low = (imin(Q(imax(1, qpos) - 1), Q(qpos)) <= P.bias_thres_interfering_indel_BQ);
                if (low && BIAS) indel_rposs[n_ir++] = rpos;
                rpos += len;
            } else if (op == C_REF_SKIP) rpos += len;
            else if (op == C_SOFT_CLIP) qpos += len;
        }
        if (BIAS) indel_rposs[n_ir++] = INT32_MAX;
    }
    int ir_idx = 0;
    int ibeg, iend;
    primer_window(P, a, ibeg, iend);
    const int atd = P.indel_adj_tracklen_dist, nrtr = (int)R.npos;
    const int xm1500 = a.xm1500;
    int qpos = 0, rpos = a.pos, incvalue = 1;
    int gi = 0;   // ordinal of the I / D op: slot of its AlnGap event
    auto gap_event = [&](int sym, int len, int weight) {
        AlnGap g; g.epos = rpos; g.sym = sym; g.len = len; g.qpos = qpos; g.weight = weight; g.aln = a.id; g.mark = 0; g.pad_ = 0;
        R.gap.ev[a.gap_off + gi] = g;
    };
    for (int i = 0; i < n_cigar; i++) {
        const int op = cig_op(cigar[i]), len = cig_len(cigar[i]);
        if (op == C_MATCH || op == C_EQUAL || op == C_DIFF) {
            if (BIAS && a.kind == 2) { rpos += len; qpos += len; continue; }   // the M runs of this read are on the P2 work list of k_p2_fast
            for (int i2 = 0; i2 < len; i2++) {
                if ((normal_filter_primers || !is_assay_amplicon) || (ibeg <= rpos && rpos < iend)) {
                    int dist = 10000;
                    if (BIAS && nge > 0) {
                        if (indel_rposs[ir_idx] <= rpos) ir_idx++;
                        const int prev_ir = indel_rposs[ir_idx - 1], next_ir = indel_rposs[ir_idx];
                        const int i1 = imax(rpos - off, atd) - atd, i2r = imin(rpos - off + atd, nrtr - 1);
                        const long long prevlen = nnminus(rpos - prev_ir, imax(rpos - (off + RTRP(R, UVC_RTR_begpos, i1)), TH(R, UVC_T_aLP1t, rpos - off)));
                        const long long nextlen = nnminus((long long)next_ir - rpos, imax((off + RTRP(R, UVC_RTR_begpos, i2r) + RTRP(R, UVC_RTR_tracklen, i2r)) - rpos, TH(R, UVC_T_aRP1t, rpos - off)));
                        dist = (int)lmin(prevlen, nextlen);
                    }
                    if (i2 > 0) {
                        const int noindel = imin(RTRP(R, UVC_RTR_indelphred, rpos - off - 1), RTRP(R, UVC_RTR_indelphred, rpos - off));
                        const int qfromBQ2 = (proton ? imin(Q(qpos - 1), Q(qpos)) : 80);
                        incvalue = (int)nnminus(imin(qfromBQ2, noindel), a.nogap_penal) + 1;
                        emit(true, rpos, UVC_LINK_M, incvalue, 0, op, 0, dist);
                    }
                    const int symbol = bases[qpos];
                    if (proton && ((0 == i2) || (len - 1 == i2))) {
                        const bool next_gap = (len - 1 == i2), prev_gap = (0 == i2);   // packed words != op codes, main.hpp:1953-1956
                        const bool isrc2 = (0 != i2);
                        int prev_base_phred = 1;
                        if (isrc2 && (qpos + 1 < a.l_qseq)) prev_base_phred = Q(qpos + 1);
                        if ((!isrc2) && (qpos > 0)) prev_base_phred = Q(qpos - 1);
                        int adj = 100;
                        if (next_gap) adj = imin(adj, ((i + 1 < n_cigar) ? cig_len(cigar[i + 1]) : 100));
                        if (prev_gap) adj = imin(adj, ((0 < i) ? cig_len(cigar[i - 1]) : 100));
                        if (adj < 3) incvalue = imin(Q(qpos), prev_base_phred) + imin(addMis, addInd);
                        else incvalue = imin(Q(qpos), prev_base_phred) + addMis;
                    } else incvalue = Q(qpos) + addMis;
                    emit(false, rpos, symbol, incvalue, a.bm1500[symbol], op, 0, dist);
                }
                rpos += 1; qpos += 1;
            }
        } else if (op == C_INS) {
            bool gapped = false;
            if ((normal_filter_primers || !is_assay_amplicon) || (ibeg <= rpos && rpos < iend)) {
                const int nbases2end = imin(qpos, a.l_qseq - (qpos + len));
                int inslen = len;
                if (nbases2end <= 0) {
                    incvalue = (0 != qpos ? Q(qpos - 1) : ((qpos + len < a.l_qseq) ? Q(qpos + len) : 1)) + addInd;
                } else {
                    int max_rn, rs_at_max;
                    int phredvalue = ref_to_phredvalue_dev(inslen, max_rn, rs_at_max, R, P, rpos - off, P.indel_BQ_max, P.indel_polymerase_slip_rate, len, op);
                    const int64_t x = rpos - off;
                    const int adp = P32(R, UVC_P_a_dp, x);
                    // adp == 0 is undefined behaviour in the reference (round(-inf) -> int); defined here and in the oracle as "no bonus"
                    const int phredinc = (adp > 0 ? (int)round(2 * ((10.0 / log(10.0)) * log((double)adp / (double)(1.0 + nnminus(adp, P32(R, UVC_P_a_at_ins_dp, x) + P32(R, UVC_P_a_at_del_dp, x)))))) : -1000000);
                    const int ratiothres = (!P.tumor_vcf_is_provided ? 2 : 4);
                    const bool multiallelic = (P64(R, UVC_P_a_near_ins_pow2len, x) * ratiothres > (long long)imax(1, P32(R, UVC_P_a_near_ins_dp, x)) * (long long)((unsigned)len * 3u));
                    if (1 == inslen && !multiallelic) phredvalue += ibetween(phredinc - 3, 0, 4);
                    const int thisdp = P32(R, UVC_P_a_at_ins_dp, x);
                    const int neardp = imax(P32(R, UVC_P_a_near_ins_dp, x), P32(R, UVC_P_a_near_RTR_ins_dp, x));
                    int insbase_minphred = 80;
                    for (int q2 = qpos; q2 < qpos + len; q2++) insbase_minphred = imin(insbase_minphred, Q(q2));
                    int ancbase_minphred = 80;
                    if (qpos > 0) ancbase_minphred = imin(ancbase_minphred, Q(qpos - 1));
                    if (qpos + len + 1 < a.l_qseq) ancbase_minphred = imin(ancbase_minphred, Q(qpos + len + 1));
                    int minq = 80;
                    if (proton && (1 == len) && (1 == rs_at_max) && (1 < max_rn))
                        for (int qinc = 0; (qinc < max_rn + 2) && (qpos + qinc) < a.l_qseq; qinc++) if (bases[qpos + qinc] == bases[qpos]) minq = imin(minq, Q(qpos + qinc));
                    const bool isrc = (a.flag & 0x10) != 0;
                    const int qfromBQ1 = (proton ? imin(ancbase_minphred, minq) : imin(ancbase_minphred, insbase_minphred));
                    const int qfromBQ2 = ((thisdp * ratiothres <= neardp || (1 == len && (xm1500 >= P.microadjust_xm
                                 || ((a.lclip_len + P.microadjust_cliplen >= rpos - a.pos) && isrc) || ((a.rclip_len + P.microadjust_cliplen >= rend - a.pos) && !isrc))))
                            ? qfromBQ1 : (proton ? imin(qfromBQ1 + proton_cigarlen2phred(len), imax(3, qfromBQ1) * len) : 80));
                    incvalue = (int)nnminus(imin(qfromBQ2, phredvalue + addInd), a.indel_penal) + 1;
                }
                if (nbases2end >= P.indel_filter_edge_dist) {
                    const int symbol = (1 == inslen ? UVC_LINK_I1 : ((2 == inslen) ? UVC_LINK_I2 : UVC_LINK_I3P));
                    emit(true, rpos, symbol, imax(1, incvalue), 0, op, len, 10000);
                    if (!BIAS) {   // incIns, main.hpp:2101-2113
                        int incvalue2 = incvalue;
                        for (int i2 = 0; i2 < len; i2++) incvalue2 = imin(incvalue2, Q(qpos + i2) + addInd);
                        gap_event(symbol, len, imax(1, incvalue2)); gapped = true;
                    }
                }
            }
            if (!BIAS) { if (!gapped) gap_event(-1, len, 0); gi++; }
            qpos += len;
        } else if (op == C_DEL) {
            bool gapped = false;
            if ((normal_filter_primers || !is_assay_amplicon) || (ibeg <= rpos && rpos < iend)) {
                const int nbases2end = imin(qpos, a.l_qseq - qpos);
                int dellen = len;
                if (nbases2end <= 0) {
                    incvalue = (0 != qpos ? Q(qpos - 1) : ((qpos < a.l_qseq) ? Q(qpos) : 1)) + addInd;
                } else {
                    int max_rn, rs_at_max;
                    int phredvalue = ref_to_phredvalue_dev(dellen, max_rn, rs_at_max, R, P, rpos - off, P.indel_BQ_max, P.indel_polymerase_slip_rate, len, op);
                    const int64_t x = rpos - off;
                    const int adp = P32(R, UVC_P_a_dp, x);
                    // adp == 0 is undefined behaviour in the reference (round(-inf) -> int); defined here and in the oracle as "no bonus"
                    const int phredinc = (adp > 0 ? (int)round(2 * ((10.0 / log(10.0)) * log((double)adp / (double)(1.0 + nnminus(adp, P32(R, UVC_P_a_at_ins_dp, x) + P32(R, UVC_P_a_at_del_dp, x)))))) : -1000000);
                    if (1 == dellen) phredvalue += ibetween(phredinc - 3, 0, 4);
                    const int thisdp = P32(R, UVC_P_a_at_del_dp, x);
                    const int neardp = imax(P32(R, UVC_P_a_near_del_dp, x), P32(R, UVC_P_a_near_RTR_del_dp, x));
                    int minq = 80;
                    if (proton && (1 == len) && (1 == rs_at_max) && (1 < max_rn))
                        for (int qinc = 0; qinc < (max_rn + 2) && (qpos + qinc) < a.l_qseq; qinc++) if (bases[qpos + qinc] == bases[qpos]) minq = imin(minq, Q(qpos + qinc));
                    const int qfromBQ1 = imin(Q(qpos), imin(Q(qpos - 1), minq));
                    const int ratiothres = (!P.tumor_vcf_is_provided ? 2 : 4);
                    const int qfromBQ2 = ((thisdp * ratiothres <= neardp) ? (int)nnminus(qfromBQ1, 1) : (proton ? imin(qfromBQ1 + proton_cigarlen2phred(len), imax(3, qfromBQ1) * len) : 80));
                    const double delFA = ((double)(thisdp + 0.5) / (double)(adp + 1));
                    const int delFAQ = imax(0, P.microadjust_delFAQmax + (int)round(P.powlaw_exponent * ((10.0 / log(10.0)) * log(delFA))));
                    int prev_cidx = i, prev_rpos = rpos;
                    while ((0 != prev_cidx) && (C_INS != cig_op(cigar[prev_cidx]) || len != cig_len(cigar[prev_cidx]))) {
                        prev_cidx--;
                        const int o = cig_op(cigar[prev_cidx]);
                        if (C_MATCH == o || C_EQUAL == o || C_DIFF == o || C_DEL == o || C_REF_SKIP == o) prev_rpos -= cig_len(cigar[prev_cidx]);
                    }
                    int next_cidx = i, next_rpos = rpos + len;
                    while ((n_cigar - 1 != next_cidx) && (C_INS != cig_op(cigar[next_cidx]) || len != cig_len(cigar[next_cidx]))) {
                        next_cidx++;
                        const int o = cig_op(cigar[next_cidx]);
                        if (C_MATCH == o || C_EQUAL == o || C_DIFF == o || C_DEL == o || C_REF_SKIP == o) next_rpos += cig_len(cigar[next_cidx]);
                    }
                    const int qfromBAQl = (int)(BAQ1(R, rpos) - BAQ1(R, prev_rpos));
                    const int qfromBAQr = (int)(BAQ1(R, next_rpos) - BAQ1(R, rpos + len));
                    const int qfromBAQ = imax(delFAQ, imax(qfromBQ1, imin(qfromBAQl, qfromBAQr)));
                    incvalue = (int)nnminus(imin(qfromBQ2, imin(qfromBAQ, phredvalue + addInd)), a.indel_penal) + 1;
                }
                if (nbases2end >= P.indel_filter_edge_dist) {
                    const int symbol = (1 == dellen ? UVC_LINK_D1 : ((2 == dellen) ? UVC_LINK_D2 : UVC_LINK_D3P));
                    emit(true, rpos, symbol, imax(1, incvalue), 0, op, len, 10000);
                    if (!BIAS) { gap_event(symbol, len, imax(1, incvalue)); gapped = true; }   // incDel, main.hpp:2216
                    for (int r2 = rpos; r2 < imin(rpos + len, rend); r2++) {   // padded deletion, main.hpp:2219-2253
                        for (int k = 0; k < 2; k++) {
                            const int s = (k == 0 ? UVC_BASE_NN : UVC_LINK_NN);
                            const int pp = ((UVC_BASE_NN == s) ? r2 : (r2 + 1));
                            if (pp >= rend) continue;
                            int dist = 0;
                            if (BIAS) {
                                if (indel_rposs[ir_idx] <= rpos) ir_idx++;
                                const unsigned prev_ir = (unsigned)indel_rposs[ir_idx - 1], next_ir = (unsigned)indel_rposs[ir_idx];
                                const unsigned d1 = (unsigned)rpos - prev_ir, d2 = next_ir - (unsigned)rpos;
                                dist = (int)(d1 < d2 ? d1 : d2);
                            }
                            emit(true, pp, s, imax(1, incvalue), 0, op, len, dist);
                        }
                    }
                }
            }
            if (!BIAS) { if (!gapped) gap_event(-1, len, 0); gi++; }
            rpos += len;
        } else {
            if (op == C_REF_SKIP) rpos += len; else if (op == C_SOFT_CLIP) qpos += len;
        }
    }
    if (BIAS) R.item_cnt[t] = n_items;
}

template __global__ void k_p2_slow<true>(RegionDev, UvcParams);
template __global__ void k_p2_slow<false>(RegionDev, UvcParams);

// applies the items of one InDel read: one wave per read, lanes stride over the items
__global__ void __launch_bounds__(64) k_p2_items(RegionDev R, UvcParams P) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= R.n_complex) return;
    const AlnRec &a = R.alns[R.complex_ids[t]];
    const SegRead sr = make_segread(R, a);
    const Item *items = R.items + a.item_off;
    const int n = R.item_cnt[t];
    for (int k = lane; k < n; k += 64) {
        const Item it = items[k];
        const int64_t x = it.epos - R.beg;
        SegAcc A; A.zero(); A.bq = it.val;
        PosThres T; load_thres(R, T, x);
        const int op = (it.flags >> 1) & 0xF;
        const int dist = (it.dist == 65535 ? 10000 : (int)it.dist);   // 10000 / large distances are clamped on emit; every use compares with small thresholds
        if (it.flags & 1) segbias<true>(A, P, sr, T, it.epos, R.baq[x], R.baq[R.npos + x], it.val, 0, op, it.indel_len, dist);
        else              segbias<false>(A, P, sr, T, it.epos, R.baq[x], R.baq[R.npos + x], it.val, a.bm1500[it.sym], op, it.indel_len, dist);
        seg_flush(R, A, it.sym, x);
    }
}

// ------------------------------------------------------------------------------------------------
// contribution of ANY alignment at position p: simple ones are computed on the fly, complex ones
// come from the table.  Values are merged into the per-fragment symbol counts with MAX.
// ------------------------------------------------------------------------------------------------
DEV void aln_contrib_max(const RegionDev &R, const UvcParams &P, const AlnRec &a, int p, bool proton, int *cnt /*[NSYM]*/) {
    // an InDel read can carry an insertion symbol at rpos == rend (CIGAR ...M I [S]): its table has one row more than the read spans
    if (p < a.pos || p > a.rend || (p == a.rend && a.kind == 0)) return;
    if (a.kind == 0) {
        const bool is_assay_amplicon = ((a.dflag & 0x4) || ((P.primerlen > 0) && !(0x2 & P.primer_flag)));
        if (is_assay_amplicon && !(P.tn_is_paired && (0x1 & P.primer_flag))) {
            int ibeg, iend; primer_window(P, a, ibeg, iend);
            if (!(ibeg <= p && p < iend)) return;
        }
        const uint8_t *qq = R.quals + a.qbase;
        if (p > a.pos) { const int v = simple_link_value(R, P, a, p, qq, proton); cnt[UVC_LINK_M] = imax(cnt[UVC_LINK_M], v); }
        const int sym = R.bases[a.qbase + p];
        const int v = simple_base_value(P, a, p, qq, proton);
        cnt[sym] = imax(cnt[sym], v);
    } else {
        const Contrib c = R.table[a.table_off + (p - a.pos)];
        if (c.bsym_p1) cnt[c.bsym_p1 - 1] = imax(cnt[c.bsym_p1 - 1], (int)c.bval);
        if (c.lmask) {
#pragma unroll
            for (int k = 0; k < 8; k++) if (c.lmask & (1 << k)) cnt[UVC_LINK_M + k] = imax(cnt[UVC_LINK_M + k], (int)c.lval[k]);
        }
    }
}

DEV void frag_counts(const RegionDev &R, const UvcParams &P, const FragRec &f, int p, bool proton, int *cnt) {
#pragma unroll
    for (int s = 0; s < NSYM; s++) cnt[s] = 0;
    for (int k = f.aln_beg; k < f.aln_end; k++) aln_contrib_max(R, P, R.alns[k], p, proton, cnt);
}

// ------------------------------------------------------------------------------------------------
// per fragment: counts of covered and near-mutation positions (b10xSeqTlen / b10xSeqTNevents, main.hpp:2738-2756)
// ------------------------------------------------------------------------------------------------
// sequential sweep over the fragment span (fragments with InDel reads, > 2 reads, primer gating, or > UVC_MAXEV events)
DEV void fragstat_sweep(const RegionDev &R, const UvcParams &P, int fi) {
    const FragRec f = R.frags[fi];
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    const int nb = P.syserr_mut_region_n_bases;
    int n_cov = 0, n_near = 0;
    int last_mut = INT32_MIN / 2;
    int tail[32]; int th = 0, tn = 0;   // covered positions within the last nb that are not (yet) near a mutation
    int cnt[NSYM];
    for (int p = f.beg; p < f.end; p++) {
        frag_counts(R, P, f, p, proton, cnt);
        bool covered = false, mut = false;
        const int refsymbol = R.refsym[p - R.beg];
        for (int vi = 0; vi < 2; vi++) {
            const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
            int cs, cc, ct;
            fill_consensus(cnt, cs, cc, ct, st, st == UVC_LINK_SYMBOL, false);
            if (0 == ct) continue;
            covered = true;
            const int con_qual = cc * 2 - ct;
            const bool highBQ = (proton ? (UVC_BASE_SYMBOL == st || con_qual + 3 >= P.bias_thres_highBQ) : (UVC_LINK_SYMBOL == st || con_qual >= P.bias_thres_highBQ));
            if (symbols_mutated(refsymbol, cs) && highBQ) mut = true;
        }
        while (tn > 0 && tail[th] < p - nb) { th = (th + 1) & 31; tn--; }   // can no longer be reached by a later mutation
        if (mut) { n_near += tn; tn = 0; last_mut = p; }
        if (covered) {
            n_cov++;
            if (p - last_mut <= nb) n_near++;
            else { tail[(th + tn) & 31] = p; tn++; }
        }
    }
    R.frags[fi].n_cov = n_cov; R.frags[fi].n_near = n_near;
    FragFast &ff = R.ffast[R.frag_rank[fi]]; ff.n_cov = n_cov; ff.n_near = n_near;   // k_frag reads the digest
}

// one wave per fragment: lanes classify positions (covered / mutated) into LDS, then count covered positions and those within
// +-syserr_mut_region_n_bases of a mutation.  Spans beyond the LDS window fall back to the sequential sweep.
#define SWEEP_MAXSPAN 4096
__global__ void __launch_bounds__(64) k_fragstat_sweep(RegionDev R, UvcParams P, const int32_t *list, const int32_t *n_dev, int n_host) {
    __shared__ uint8_t flag[SWEEP_MAXSPAN];
    __shared__ int tot[2];
    const int lane = threadIdx.x;
    const int n = (n_dev ? *n_dev : n_host);
    const int nb = P.syserr_mut_region_n_bases;
    // spans beyond the LDS window are taken in chunks with a halo of nb positions either side (a mutation reaches nb positions far)
    const int step = SWEEP_MAXSPAN - 2 * nb;
    if (step < 64) { if (lane == 0 && blockIdx.x < n) atomicExch(R.err, UVCGPU_EUNSUPPORTED); return; }   // syserr_mut_region_n_bases beyond 2 000
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    // the list may be longer than the grid (the overflow form is launched before its length is known on the host): blocks stride over it
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
    const int fi = list[t];
    const FragRec &f = R.frags[fi];
    const int span = f.end - f.beg;
    __syncthreads();   // the previous fragment's totals have been read
    if (lane < 2) tot[lane] = 0;
    int cnt[NSYM];
    int n_cov = 0, n_near = 0;
    for (int c0 = 0; c0 < span; c0 += step) {
        const int w0 = imax(0, c0 - nb), w1 = imin(span, c0 + step + nb);
        __syncthreads();   // the previous chunk's flags have been read
        for (int i = w0 + lane; i < w1; i += 64) {
            const int p = f.beg + i;
            frag_counts(R, P, f, p, proton, cnt);
            int fl = 0;
            const int refsymbol = R.refsym[p - R.beg];
            for (int vi = 0; vi < 2; vi++) {
                const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
                int cs, cc, ct;
                fill_consensus(cnt, cs, cc, ct, st, st == UVC_LINK_SYMBOL, false);
                if (0 == ct) continue;
                fl |= 1;
                const int con_qual = cc * 2 - ct;
                const bool highBQ = (proton ? (UVC_BASE_SYMBOL == st || con_qual + 3 >= P.bias_thres_highBQ) : (UVC_LINK_SYMBOL == st || con_qual >= P.bias_thres_highBQ));
                if (symbols_mutated(refsymbol, cs) && highBQ) fl |= 2;
            }
            flag[i - w0] = (uint8_t)fl;
        }
        __syncthreads();
        for (int i = c0 + lane; i < imin(span, c0 + step); i += 64) {
            if (!(flag[i - w0] & 1)) continue;
            n_cov++;
            bool near = false;
            for (int j = imax(0, i - nb); j <= imin(span - 1, i + nb) && !near; j++) near = (flag[j - w0] & 2) != 0;
            if (near) n_near++;
        }
    }
    atomicAdd(&tot[0], n_cov); atomicAdd(&tot[1], n_near);
    __syncthreads();
    if (lane == 0) { R.frags[fi].n_cov = tot[0]; R.frags[fi].n_near = tot[1]; FragFast &ff = R.ffast[R.frag_rank[fi]]; ff.n_cov = tot[0]; ff.n_near = tot[1]; }
    }
}

// M runs and special range of one alignment for k_frag; false when it has more than one InDel or a reference skip
DEV bool aln_runs(const RegionDev &R, const AlnRec &a, int &pA, int &eA, int &qA, int &pB, int &eB, int &qB, int &sp_beg, int &sp_len) {
    const uint32_t *cigar = R.cigars + a.cigar_off;
    int rp = a.pos; long long qp = 0; int n_runs = 0, n_indel = 0;
    pA = eA = qA = pB = eB = qB = 0; sp_beg = a.pos; sp_len = 0;
    for (int i = 0; i < a.n_cigar; i++) {
        const int op = cig_op(cigar[i]), len = cig_len(cigar[i]);
        if (op == C_MATCH || op == C_EQUAL || op == C_DIFF) {
            const int qb = (int)((a.seq_off + qp - rp) & 0xFFFFFFFFLL);
            if (n_runs == 0) { pA = rp; eA = rp + len; qA = qb; } else if (n_runs == 1) { pB = rp; eB = rp + len; qB = qb; } else return false;
            n_runs++; rp += len; qp += len;
        } else if (op == C_INS) { if (++n_indel > 1) return false; sp_beg = rp; sp_len = 1; qp += len; }
        else if (op == C_DEL) { if (++n_indel > 1) return false; sp_beg = rp; sp_len = len + 1; rp += len; }
        else if (op == C_SOFT_CLIP) qp += len;
        else if (op == C_HARD_CLIP) {}
        else return false;
    }
    if (sp_beg + sp_len > a.rend + 1) sp_len = imax(0, a.rend + 1 - sp_beg);   // the last position that can carry a symbol is rend itself (insertion behind the last base)
    return n_runs >= 1 && sp_len < 32768;
}

// closed form for fragments of <= 2 simple alignments: coverage = union of the alignment spans, mutations = the event list
__global__ void __launch_bounds__(256) k_fragstat_fast(RegionDev R, UvcParams P) {
    const int fi = blockIdx.x * blockDim.x + threadIdx.x;
    if (fi >= R.n_frags) return;
    const FragRec f = R.frags[fi];
    FragFast ff;
    memset(&ff, 0, sizeof(ff));
    ff.beg = f.beg; ff.end = f.end; ff.fi = fi;
    ff.flags = (f.stat_kind ? 1 : 0) | (f.strand << 1) | (f.singleton << 2) | ((f.aln_end - f.aln_beg) << 3);
    ff.sq = (f.normMQ * f.normMQ) / SQR_QUAL_DIV;
    if (f.stat_kind != 0) {
        // fragments with InDel reads: k_frag can still take every position outside the InDels' neighbourhoods when the alignments
        // decompose into <= 2 M runs each (n_cov / n_near arrive from k_fragstat_sweep)
        const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
        const bool amplicon_gated = (((f.dflag & 0x4) || ((P.primerlen > 0) && !(0x2 & P.primer_flag))) && !(P.tn_is_paired && (0x1 & P.primer_flag)));
        const int n_aln = f.aln_end - f.aln_beg;
        bool ok = (!proton && !amplicon_gated && n_aln <= 2 && f.end - f.beg < 65536);
        int sb, sl;
        if (ok) {
            const AlnRec &x0 = R.alns[f.aln_beg];
            ok = aln_runs(R, x0, ff.pos0, ff.rend0, ff.qb0, ff.bpos0, ff.brend0, ff.bqb0, sb, sl);
            ff.nogap0 = x0.nogap_penal; ff.sp0 = (sb - f.beg) | (sl << 16);
            if (sb < f.beg) ok = false;
        }
        if (ok && n_aln == 2) {
            const AlnRec &x1 = R.alns[f.aln_beg + 1];
            ok = aln_runs(R, x1, ff.pos1, ff.rend1, ff.qb1, ff.bpos1, ff.brend1, ff.bqb1, sb, sl);
            ff.nogap1 = x1.nogap_penal; ff.sp1 = (sb - f.beg) | (sl << 16);
            if (sb < f.beg) ok = false;
        }
        if (ok) ff.flags = (ff.flags & ~1) | 0x100;
        R.ffast[R.frag_rank[fi]] = ff; R.ffast_u[fi] = *(const FragUnit *)&ff; return;
    }
    const int nm = R.frag_nmut[fi];
    // coverage intervals [a1,b1) u [a2,b2), disjoint and ordered
    const AlnRec &x0 = R.alns[f.aln_beg];
    ff.pos0 = x0.pos; ff.rend0 = x0.rend; ff.qb0 = (int32_t)(x0.qbase & 0xFFFFFFFFLL); ff.nogap0 = x0.nogap_penal;
    int a1 = x0.pos, b1 = x0.rend, a2 = 0, b2 = 0;
    if (f.aln_end - f.aln_beg == 2) {
        const AlnRec &x1 = R.alns[f.aln_beg + 1];
        ff.pos1 = x1.pos; ff.rend1 = x1.rend; ff.qb1 = (int32_t)(x1.qbase & 0xFFFFFFFFLL); ff.nogap1 = x1.nogap_penal;
        int c = x1.pos, d = x1.rend;
        if (c < a1) { int t = a1; a1 = c; c = t; t = b1; b1 = d; d = t; }
        if (c <= b1) { b1 = imax(b1, d); } else { a2 = c; b2 = d; }
    }
    // k_frag_sums counts LINK_M by micro_nogap_penal class 1..5 (main.hpp:1882-1885 gives min(4, ..) + 1; below 1 only when NM is smaller than
    // the InDel lengths it should contain): anything else keeps the per-position form
    if (ff.nogap0 < 1 || ff.nogap0 > 5 || (f.aln_end - f.aln_beg == 2 && (ff.nogap1 < 1 || ff.nogap1 > 5))) ff.flags |= 0x200;
    const int n_cov = (b1 - a1) + (b2 - a2);
    int n_near = 0;
    if (nm > UVC_MAXEV) {   // too many events for the closed form: the sweep kernel (overflow pass) fills n_cov / n_near of this record
        const int k = atomicAdd(R.n_overflow, 1); R.overflow_frags[k] = fi;
        R.ffast[R.frag_rank[fi]] = ff; R.ffast_u[fi] = *(const FragUnit *)&ff; return;
    }
    if (nm > 0) {
        const int nb = P.syserr_mut_region_n_bases;
        int ev[UVC_MAXEV];
        for (int i = 0; i < nm; i++) ev[i] = R.frag_mut[(size_t)fi * UVC_MAXEV + i];
        for (int i = 1; i < nm; i++) { const int v = ev[i]; int j = i - 1; while (j >= 0 && ev[j] > v) { ev[j + 1] = ev[j]; j--; } ev[j + 1] = v; }
        int lo = ev[0] - nb, hi = ev[0] + nb + 1;   // merged dilation interval [lo, hi)
        for (int i = 1; i <= nm; i++) {
            if (i < nm && ev[i] - nb <= hi) { hi = ev[i] + nb + 1; continue; }
            n_near += imax(0, imin(hi, b1) - imax(lo, a1)) + imax(0, imin(hi, b2) - imax(lo, a2));
            if (i < nm) { lo = ev[i] - nb; hi = ev[i] + nb + 1; }
        }
    }
    R.frags[fi].n_cov = n_cov; R.frags[fi].n_near = n_near;
    ff.n_cov = n_cov; ff.n_near = n_near;
    R.ffast[R.frag_rank[fi]] = ff; R.ffast_u[fi] = *(const FragUnit *)&ff;
}

// ------------------------------------------------------------------------------------------------
// k_frag_sums: what the plain fragments add to LINK_M and to "a base is here" at every position, as interval sums.
// For a fragment of <= 2 simple alignments the LINK_M consensus at p is max(noindel80(p) - micro_nogap_penal, 0) + 1 of the better mate that
// has a link at p (main.hpp:1919-1923, 339-349): given the position it depends on the fragment through the penalty class only, and bDP / bTA /
// bTB / bMQ / the singleton votes are per-fragment constants.  One block per (tile of FSUM_TILE positions, strand): a lane per fragment adds
// +v at the begin and -v at the end of each of its <= 3 class pieces in LDS, a block-wide prefix sum turns the differences into sums.
// k_frag then spends nothing per (fragment, position) on LINK_M, and on the base side only what depends on the base and its quality.
// ------------------------------------------------------------------------------------------------
#define FSUM_TILE 1024
__global__ void __launch_bounds__(256) k_frag_sums(RegionDev R, UvcParams P) {
    __shared__ int d[UVC_FSUM_N][FSUM_TILE + 1];
    __shared__ int wtot[4];
    const int strand = blockIdx.y;
    const int t0 = R.beg + (int)blockIdx.x * FSUM_TILE, t1 = imin(t0 + FSUM_TILE, R.beg + (int)R.npos);
    for (int i = threadIdx.x; i < UVC_FSUM_N * (FSUM_TILE + 1); i += 256) (&d[0][0])[i] = 0;
    __syncthreads();
    const int seg_beg = R.frag_off[strand], seg_end = R.frag_off[strand + 1];
    // the fragments that can reach the tile, from the window index (k_win_index, lists 5 / 6): first with beg >= t0 - max_frag_span + 1 ..
    // first with beg >= the end of the tile's last 64-position window; two binary searches over the list by every thread (forty dependent
    // loads in front of the block's work) otherwise
    (void)seg_beg; (void)seg_end;
    const int lo = win_lo(R, 5 + strand, (int)blockIdx.x * (FSUM_TILE / 64)), hi = win_hi(R, 5 + strand, (t1 - 1 - R.beg) >> 6);
    auto put = [&](int f, int a, int b, int v) {   // += v on [a, b) of plane f, clipped to the tile
        a = imax(a, t0); b = imin(b, t1);
        if (a < b) { atomicAdd(&d[f][a - t0], v); atomicAdd(&d[f][b - t0], -v); }
    };
    for (int k = lo + (int)threadIdx.x; k < hi; k += 256) {
        const int4 *q4 = (const int4 *)(R.ffast + k);
        const int4 h0 = q4[0];   // beg, end, fi, flags
        if ((h0.w & 0x301) || h0.y <= t0) continue;
        const int4 h1 = q4[1], h2 = q4[2], h3 = q4[3];   // pos0 rend0 pos1 rend1 | qb0 qb1 nogap0 nogap1 | sq n_cov n_near
        const bool has2 = (((h0.w >> 3) & 0xF) == 2);
        const int sing = (h0.w >> 2) & 1, sq = h3.x, n_cov = h3.y, n_near = h3.z;
        // k_frag only looks at positions of [beg, end)
        const int a0 = imax(h1.x, h0.x), a1 = imin(h1.y, h0.y), b0 = (has2 ? imax(h1.z, h0.x) : 0), b1 = (has2 ? imin(h1.w, h0.y) : 0);
        {   // base side: the union of [a0, a1) and [b0, b1)
            auto base = [&](int a, int b) { put(UVC_FSUM_BDP, a, b, 1); put(UVC_FSUM_BTA, a, b, n_cov); put(UVC_FSUM_BTB, a, b, n_near); put(UVC_FSUM_BMQ, a, b, sq); };
            if (a0 < a1 && b0 < b1 && imax(a0, b0) <= imin(a1, b1)) base(imin(a0, b0), imax(a1, b1));
            else { if (a0 < a1) base(a0, a1); if (b0 < b1) base(b0, b1); }
        }
        {   // LINK_M exists from the second base of a run on: (pos, rend); the better (smaller) penalty wins where the mates overlap
            const int la0 = imax(h1.x + 1, h0.x), lb0 = (has2 ? imax(h1.z + 1, h0.x) : 0);
            const int ea = h2.z, eb = h2.w;
            auto link = [&](int a, int b, int e) {
                if (a >= b) return;
                put(UVC_FSUM_LCNT + e - 1, a, b, 1); put(UVC_FSUM_LTA, a, b, n_cov); put(UVC_FSUM_LTB, a, b, n_near); put(UVC_FSUM_LMQ, a, b, sq);
                if (sing) put(UVC_FSUM_LSING, a, b, 1);
            };
            const int o0 = imax(la0, lb0), o1 = imin(a1, b1);
            if (la0 < a1 && lb0 < b1 && o0 < o1) {
                link(imin(la0, lb0), o0, la0 < lb0 ? ea : eb);
                link(o0, o1, imin(ea, eb));
                link(o1, imax(a1, b1), a1 > b1 ? ea : eb);
            } else { link(la0, a1, ea); link(lb0, b1, eb); }
        }
    }
    __syncthreads();
    // prefix sums: four consecutive positions per thread, the block's 256 partial sums by wave scan + the waves' totals
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int32_t *out = R.fsum + (size_t)strand * UVC_FSUM_N * R.npos + (size_t)blockIdx.x * FSUM_TILE;
    const int n_here = t1 - t0;
    for (int f = 0; f < UVC_FSUM_N; f++) {
        int v[4], run = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { run += d[f][threadIdx.x * 4 + i]; v[i] = run; }
        int inc = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wtot[w] = inc;
        __syncthreads();
        int before = inc - run;
        for (int i = 0; i < w; i++) before += wtot[i];
        __syncthreads();
        int32_t *o4 = out + (size_t)f * R.npos + threadIdx.x * 4;
        if (threadIdx.x * 4 + 4 <= n_here && (((uintptr_t)o4) & 15) == 0) *(int4 *)o4 = make_int4(before + v[0], before + v[1], before + v[2], before + v[3]);
        else {
#pragma unroll
            for (int i = 0; i < 4; i++) { const int x = threadIdx.x * 4 + i; if (x < n_here) o4[i] = before + v[i]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_frag_generic: the P3 / singleton P4-P5 contribution of the fragments k_frag cannot take in registers (InDel reads,
// more than two alignments, primer-gated reads; every fragment on IonTorrent).  One wave per fragment, lanes over its
// positions, everything added to the planes with atomics; the bucket histogram goes to the global bucket plane, which
// k_frag merges in P3b.  Runs before k_frag.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_frag_generic(RegionDev R, UvcParams P, const int32_t *list, int n_list) {
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    const bool padded_ignored = (P.microadjust_padded_deletion_flag & (proton ? 0x2 : 0x1)) != 0;
    const bool vcfgen = P.inferred_is_vcf_generated;
    // grid = (fragments, 64-position chunks of the longest fragment): the work of a fragment is a chain of dependent loads per
    // position, so more, shorter waves hide it better than one wave striding over the whole span
    for (int t = blockIdx.x; t < n_list; t += gridDim.x) {
        const int fi = (list ? list[t] : t);
        const FragRec &f = R.frags[fi];
        const int strand = f.strand;
        const bool singleton = f.singleton;
        const int fsq = (f.normMQ * f.normMQ) / SQR_QUAL_DIV;
        const FragFast &ff = R.ffast[R.frag_rank[fi]];
        const bool special_only = (ff.flags & 0x100) != 0;   // k_frag takes every other position of this fragment
        const int sp0 = ff.sp0, sp1 = ff.sp1;
        const int n0 = (special_only ? (sp0 >> 16) : 0), n1 = (special_only ? (sp1 >> 16) : 0);
        const int p_lo = imax(f.beg, R.beg), p_hi = imin(f.end, R.end);
        const int n_work = (special_only ? n0 + n1 : p_hi - p_lo);
        for (int w = (int)(blockIdx.y * 64 + threadIdx.x); w < n_work; w += 64 * gridDim.y) {
            int p;
            if (!special_only) p = p_lo + w;
            else {
                p = (w < n0 ? f.beg + (sp0 & 0xFFFF) + w : f.beg + (sp1 & 0xFFFF) + (w - n0));
                // a position in both ranges is done once, by the first
                if (w >= n0 && (unsigned)(p - (f.beg + (sp0 & 0xFFFF))) < (unsigned)n0) continue;
                if (p < p_lo || p >= p_hi) continue;
            }
            const int64_t x = p - R.beg;
            const int my_ref = R.refsym[x];
            int cnt[NSYM];
            frag_counts(R, P, f, p, proton, cnt);
            for (int vi = 0; vi < 2; vi++) {
                const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
                int cs, cc, ct;
                fill_consensus(cnt, cs, cc, ct, st, st == UVC_LINK_SYMBOL, false);
                if (0 == ct) continue;
                int cs4 = cs, cc4 = cc, ct4 = ct;
                if (st == UVC_BASE_SYMBOL && padded_ignored) fill_consensus(cnt, cs4, cc4, ct4, st, false, true);
                if (vcfgen) {
                    const int ad = S32(R, UVC_S_aDPff, cs, x) + S32(R, UVC_S_aDPfr, cs, x) + S32(R, UVC_S_aDPrf, cs, x) + S32(R, UVC_S_aDPrr, cs, x);
                    const int max_qual = 8 + BQS(R, cs, x) / imax(1, ad);   // get_avgBQ, main_conversion.hpp:791-796
                    int phredlike = imin(cc * 2 - ct, max_qual);
                    if (0x1 & P.fam_flag) phredlike = imin(phredlike, sscs_phred(P, my_ref, cs));
                    const int pbucket = imax(0, max_qual - phredlike);
                    if (pbucket < NBUCKETS) atomicAdd(&BKP(R, 0, cs, pbucket, x), 1);
                    mark_sym(R, cs, x);
                    atomicAdd(&FRP(R, strand, UVC_FRAG_bDP, cs, x), 1); atomicAdd(&FRP(R, strand, UVC_FRAG_bTA, cs, x), f.n_cov); atomicAdd(&FRP(R, strand, UVC_FRAG_bTB, cs, x), f.n_near);
                    atomicAdd(&VQP(R, UVC_VQ_bMQ, cs, x), fsq);
                }
                if (singleton) {   // con / mmm identities of a one-fragment unit (main.hpp:466-520)
                    const int adj = imax(cc4 * 2, ct4) - ct4;
                    const int thr = (st == UVC_BASE_SYMBOL ? P.fam_thres_highBQ_snv : 0);
                    if (adj >= thr && adj > 0) { mark_sym(R, cs4, x); atomicAdd(&FAP(R, strand, UVC_FAM_cDP12, cs4, x), 1); atomicAdd(&FAP(R, strand, UVC_FAM_cDP21, cs4, x), 1); }
                    const int adj5 = imax(cc * 2, ct) - ct;
                    if (adj5 > 0 && vcfgen) { mark_sym(R, cs, x); atomicAdd(&FAP(R, strand, UVC_FAM_cDP1, cs, x), 1); }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_frag: P3 + P3b for every fragment, and P4/P5 of singleton family-strand units.  One lane per
// position; per-lane bucket histograms of the two dense symbols live in LDS, rare symbols use the
// global bucket plane (each position has exactly one writer in this kernel).
// ------------------------------------------------------------------------------------------------
// PLAIN: VCF run, Illumina-like values, no SSCS table cap (fam_flag & 1), padded deletions counted: the common case without those arms
// H16: no position is covered by 65 536 fragments or more (host bound, RegionDev::max_frag_depth), so two buckets share one LDS word:
// 16 KiB per block instead of 32, six waves per SIMD instead of five.
#define RQ_CAP 128   // events per wave between two flushes (one record adds at most 64)
// SPLIT (see k_prep_fast): a block per window.  The four waves take every fourth chunk of the window's fragments and add into ONE set of LDS
// columns (bucket histogram, lacc, racc: a column per position, LDS atomics); behind the lists wave 0 alone holds the sums and runs the rest.
template <bool PLAIN, bool H16, bool SPLIT = false>
DEV void frag_body(const RegionDev &R, const UvcParams &P, unsigned (*hist)[NBUCKETS / (H16 ? 2 : 1)][SPLIT ? 64 : 256], unsigned long long (*rq)[RQ_CAP], int (*lacc)[SPLIT ? 64 : 256], int (*racc)[64] = nullptr) {
    const int col = (SPLIT ? (int)(threadIdx.x & 63) : (int)threadIdx.x);   // this thread's column of the LDS arrays
    auto hist_add = [&](int dense, int b) {   // ds_add_u32 without return
        if (H16) atomicAdd(&hist[dense][b >> 1][col], 1u << (16 * (b & 1))); else atomicAdd(&hist[dense][b][col], 1u);
    };
    auto hist_get = [&](int dense, int b) -> int {
        return H16 ? (int)((hist[dense][b >> 1][col] >> (16 * (b & 1))) & 0xFFFFu) : (int)hist[dense][b][col];
    };
    COARSE_T(ct0)
    const int lane = threadIdx.x & 63;
    const int wave = SPLIT ? wave_uniform((int)xcd_block()) : wave_uniform((int)((xcd_block() * blockDim.x + threadIdx.x) >> 6));
    const int wv = (int)(threadIdx.x >> 6);
    const int64_t x0 = (int64_t)wave * 64;
    if (x0 >= R.npos) return;
    const int w0 = R.beg + (int)x0;
    const int p = w0 + lane;
    const int64_t x = x0 + lane;
    const bool valid = x < R.npos;
    const bool proton = (PLAIN ? false : (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform));
    const bool padded_ignored = (PLAIN ? false : ((P.microadjust_padded_deletion_flag & (proton ? 0x2 : 0x1)) != 0));
    const bool vcfgen = (PLAIN ? true : (P.inferred_is_vcf_generated != 0));   // P3 belongs to updateByAlns3UsingBQ, skipped on FASTQ-only runs (main.hpp:3691)
    const int my_ref = valid ? R.refsym[x] : 0;
    if (!SPLIT || wv == 0) {
        for (int b = 0; b < NBUCKETS / (H16 ? 2 : 1); b++) { hist[0][b][col] = 0; hist[1][b][col] = 0; }
        for (int i = 0; i < (SPLIT ? 10 : 5); i++) lacc[i][col] = 0;
        if (SPLIT) for (int i = 0; i < 11; i++) racc[i][col] = 0;
    }
    if (SPLIT) __syncthreads();   // (block-uniform: the four waves share the window)
    // avgBQ + 8 (get_avgBQ, main_conversion.hpp:791-796) of the five read symbols and of LINK_M; LINK_M value of a simple read here
    auto maxq_at = [&](int sym) {
        const int ad = S32(R, UVC_S_aDPff, sym, x) + S32(R, UVC_S_aDPfr, sym, x) + S32(R, UVC_S_aDPrf, sym, x) + S32(R, UVC_S_aDPrr, sym, x);
        return 8 + BQS(R, sym, x) / imax(1, ad);
    };
    // the five per-symbol values are packed 16 bits each (a value is 8 + an average base quality) so that the per-lane lookup by
    // consensus symbol is a shift, not an indexed local array (which would live in scratch memory)
    unsigned long long mq_acgt = 0x0008000800080008ull;
    int mqN = 8, maxq_link = 8, noindel80 = 80;
    if (valid) {
        mq_acgt = (unsigned long long)imin(maxq_at(UVC_BASE_A), 65535) | ((unsigned long long)imin(maxq_at(UVC_BASE_C), 65535) << 16)
                | ((unsigned long long)imin(maxq_at(UVC_BASE_G), 65535) << 32) | ((unsigned long long)imin(maxq_at(UVC_BASE_T), 65535) << 48);
        mqN = maxq_at(UVC_BASE_N);
        maxq_link = maxq_at(UVC_LINK_M);
        if (x > 0) noindel80 = imin(80, imin(RTRP(R, UVC_RTR_indelphred, x - 1), RTRP(R, UVC_RTR_indelphred, x)));
    }
    auto maxq_base = [&](int cs) { return cs >= UVC_BASE_N ? mqN : (int)((mq_acgt >> (16 * cs)) & 0xFFFFull); };   // cs in A..N
    const int maxq_ref = maxq_base(my_ref);
    // dense accumulators of the reference symbol: [strand] x {bDP, bTA, bTB, cDP12 (== cDP21), cDP1}, bMQ.  Named structs, selected by
    // a wave-uniform branch on the strand: runtime-indexed local arrays would live in scratch memory (one VMEM round trip per update).
    struct DAcc { int bDP, bTA, bTB, c12, c1; };
    DAcc a_fr = {0,0,0,0,0}, a_rr = {0,0,0,0,0};   // {fwd,rev} of the reference symbol (LINK_M: interval sums + lacc, see the end)
    int bMQ_r = 0;
    // One (fragment, position, symbol type) consensus (cs = symbol, cc = its value, ct = total) -> P3 outputs and, for singleton units,
    // the P4/P5 identities:
    //   con = 1 vote for the fragment consensus when 2*max - tot passes the threshold (main.hpp:466-495) => cDP12, cDP21 (tot_count == 1)
    //   mmm = 2*max - tot when positive (main.hpp:497-520)                                              => cDP1
    // The reference symbol accumulates in registers / LDS.  Every other symbol is rare and becomes an EVENT in the wave's LDS queue:
    // the loop itself issues no store or atomic to memory, so the s_waitcnt vmcnt(0) in front of the next record's prefetched bytes
    // waits for those bytes only (with the atomics in the loop it waited for every one of them to reach the L2: two thirds of
    // the kernel's wave-cycles, SQ_WAIT_ANY).  flush_events turns the queue into fire-and-forget atomics, a lane per event.
    //   event: bit 63 valid | fragment index k << 24 | p3 << 23 | c12 << 22 | c1 << 21 | strand << 20 | bucket (31 = none) << 14 | cs4 << 10 | cs << 6 | lane
    unsigned long long *myq = rq[threadIdx.x >> 6];
    int nq = 0;   // wave-uniform: only updated in uniform control flow
    auto flush_events = [&]() {
#ifdef UVC_ABLATE_RARE
        nq = 0; return;
#endif
        for (int i = lane; i < nq; i += 64) {
            const unsigned long long e = myq[i];
            const int le = (int)(e & 63), cs = (int)((e >> 6) & 15), cs4 = (int)((e >> 10) & 15), pb = (int)((e >> 14) & 31), st = (int)((e >> 20) & 1);
            const int64_t xe = x0 + le;
            const FragFast *ff = R.ffast + (int)((e >> 24) & 0x7FFFFFFFull);
            // The cells of a position belong to its wave in this kernel: the adds need to be atomic among the lanes of this flush only, so
            // they are L2 atomics of workgroup scope.  (Agent scope sends each one to the memory side and makes the fence below a write-back +
            // invalidate of the XCD's whole L2 -- per wave, 15 000 times per launch: it was what kept every load of the kernel slow.)
            auto add_own = [](int32_t *cell, int v) { __hip_atomic_fetch_add(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
            const int sq = ff->sq, n_cov = ff->n_cov, n_near = ff->n_near;
            mark_sym(R, cs, xe); mark_sym(R, cs4, xe);
            if ((e >> 23) & 1) {
                if (pb < NBUCKETS) add_own(&BKP(R, 0, cs, pb, xe), 1);
                add_own(&FRP(R, st, UVC_FRAG_bDP, cs, xe), 1); add_own(&FRP(R, st, UVC_FRAG_bTA, cs, xe), n_cov); add_own(&FRP(R, st, UVC_FRAG_bTB, cs, xe), n_near);
                add_own(&VQP(R, UVC_VQ_bMQ, cs, xe), sq);
            }
            if ((e >> 22) & 1) { add_own(&FAP(R, st, UVC_FAM_cDP12, cs4, xe), 1); add_own(&FAP(R, st, UVC_FAM_cDP21, cs4, xe), 1); }
            if ((e >> 21) & 1) add_own(&FAP(R, st, UVC_FAM_cDP1, cs, xe), 1);
        }
        nq = 0;
    };
    // FULL: the fragment is not in k_frag_sums' interval sums, its per-fragment constants are added here.  Otherwise the sums hold them for
    // every position with a base, as if the base were the reference: a rare symbol takes its share back out of the reference's accumulators.
    // K(i): dword i of the fragment's record.  Returns the event of a rare symbol, or 0.
    auto apply = [&](bool full, DAcc &ar, int cs, int cc, int ct, int cs4, int cc4, int ct4, int max_qual, int strand, auto K, bool singleton, int k) -> unsigned long long {
        const bool is_ref = (cs == my_ref);
        int pb = 31;
        if (vcfgen) {
            const int con_qual = cc * 2 - ct;
            int phredlike = imin(con_qual, max_qual);
            if (!PLAIN && (0x1 & P.fam_flag)) phredlike = imin(phredlike, sscs_phred(P, my_ref, cs));
            const int pbucket = imax(0, max_qual - phredlike);
            if (is_ref) {
                if (pbucket < NBUCKETS) hist_add(0, pbucket);
                if (full) { ar.bDP += 1; ar.bTA += K(13); ar.bTB += K(14); bMQ_r += K(12); }
            } else {
                pb = imin(pbucket, 31);
                if (!full) { ar.bDP -= 1; ar.bTA -= K(13); ar.bTB -= K(14); bMQ_r -= K(12); }
            }
        }
        // (conditions as values, increments as adds of 0 / 1: a nest of per-lane ifs is an exec-mask level each -- save, branch, restore -- per record)
        const int adj = imax(cc4 * 2, ct4) - ct4;
        const bool v12 = singleton & (adj >= P.fam_thres_highBQ_snv) & (adj > 0);   // cDP12 == cDP21 for a singleton unit
        const bool r4 = (cs4 == my_ref);
        ar.c12 += ((v12 & r4) ? 1 : 0);
        const bool e12 = v12 & !r4;
        const int adj5 = imax(cc * 2, ct) - ct;
        const bool v1 = singleton & (adj5 > 0) & vcfgen;
        ar.c1 += ((v1 & is_ref) ? 1 : 0);
        const bool e1 = v1 & !is_ref;
        const bool e3 = (vcfgen && !is_ref);
        if (!(e3 || e12 || e1)) return 0ull;
        return (1ull << 63) | ((unsigned long long)(unsigned)k << 24) | ((unsigned long long)e3 << 23) | ((unsigned long long)e12 << 22) | ((unsigned long long)e1 << 21)
             | ((unsigned long long)strand << 20) | ((unsigned long long)pb << 14) | ((unsigned long long)(cs4 & 15) << 10) | ((unsigned long long)(cs & 15) << 6) | (unsigned long long)lane;
    };
    // LINK_M of a fragment outside the interval sums (few): value lv > 0 at this position, its consensus is a vote (threshold 0) and an mmm > 0.
    // Bucket and sums in LDS (lacc: bDP, bTA, bTB, bMQ, singleton votes of the strand being walked; moved to the planes after each list).
    auto link_full = [&](int lv, int strand, auto K, bool singleton) {
        if (vcfgen) {
            int phredlike = imin(lv, maxq_link);
            if (!PLAIN && (0x1 & P.fam_flag)) phredlike = imin(phredlike, sscs_phred(P, my_ref, UVC_LINK_M));
            const int pbucket = imax(0, maxq_link - phredlike);
            if (pbucket < NBUCKETS) hist_add(1, pbucket);
            const int lr = (SPLIT ? 5 * strand : 0);
            atomicAdd(&lacc[lr + 0][col], 1); atomicAdd(&lacc[lr + 1][col], K(13)); atomicAdd(&lacc[lr + 2][col], K(14)); atomicAdd(&lacc[lr + 3][col], K(12));
        }
        if (singleton) atomicAdd(&lacc[(SPLIT ? 5 * strand : 0) + 4][col], 1);
    };
    auto maxq_generic = [&](int cs) { return cs == UVC_LINK_M ? maxq_link : (cs <= UVC_BASE_N ? maxq_base(cs) : maxq_at(cs)); };
    const __amdgpu_buffer_rsrc_t rs = bq_rsrc(R);
    auto run_list = [&](auto ST, DAcc &ar) {
    constexpr int strand = decltype(ST)::value ? 1 : 0;
    const int lo = wave_uniform(win_lo(R, 5 + strand, (int)(x0 >> 6))), hi = wave_uniform(win_hi(R, 5 + strand, (int)(x0 >> 6)));
    for (int k0 = lo + (SPLIT ? 64 * wv : 0); k0 < hi; k0 += (SPLIT ? 256 : 64)) {
        // one FragFast (24 dwords) per lane, fields of record j broadcast with v_readlane; base/qual bytes of record j+1 are
        // requested before record j is processed
        int c[24];
        if (k0 + lane < hi) {
            const int4 *q4 = (const int4 *)(R.ffast + (k0 + lane));
#pragma unroll
            for (int i = 0; i < 6; i++) { const int4 t = q4[i]; c[4 * i] = t.x; c[4 * i + 1] = t.y; c[4 * i + 2] = t.z; c[4 * i + 3] = t.w; }
        } else {
#pragma unroll
            for (int i = 0; i < 24; i++) c[i] = 0;
        }
        const int n = imin(64, hi - k0);
        int bq0n = 0, bq1n = 0;
        int n_fl = 0, n_fend = 0;   // of the record whose bytes are on their way: read once, used by its iteration too
        auto issue = [&](int j) {
            const int fl = n_fl = bcast(c[3], j);
            n_fend = bcast(c[1], j);
            // the list holds every fragment that could reach the window (begin within the longest span): four in ten end in front of it,
            // and a fragment of the generic kind is not read here at all -- no bytes for those
            if (n_fend <= w0 || (fl & 1)) return;
            int i0 = bcast(c[8], j), i1 = bcast(c[9], j);
            if (fl & 0x100) {   // the lane's position may lie in run B of either alignment
                if (p >= bcast(c[16], j) && p < bcast(c[17], j)) i0 = bcast(c[18], j);
                if (p >= bcast(c[20], j) && p < bcast(c[21], j)) i1 = bcast(c[22], j);
            }
            bq0n = bq_load(rs, i0 + p);
            if (((fl >> 3) & 0xF) == 2) bq1n = bq_load(rs, i1 + p);
        };
        if (proton) break;   // IonTorrent values need neighbouring qualities: every fragment takes k_frag_generic
        issue(0);
        for (int j = 0; j < n; j++) {
            const int b0 = bq0n & 0xFF, q0 = (bq0n >> 8) & 0xFF, b1 = bq1n & 0xFF, q1 = (bq1n >> 8) & 0xFF;
            const int flags = n_fl, fend = n_fend;
            if (j + 1 < n) issue(j + 1);
            if (fend <= w0) continue;
            const int fbeg = bcast(c[0], j);
            if (flags & 1) continue;   // done by k_frag_generic
            // from here on the control flow stays wave-uniform down to the queue bookkeeping (nq is a scalar): per-lane conditions are predicates
            bool cover = (valid && p >= fbeg && p < fend);
            const bool singleton = (flags >> 2) & 1;
            auto K = [&](int i) { return bcast(c[i], j); };
            // consensus of <= 2 alignments in registers, written with selects (BASE_QUALITY_MAX merge, main.hpp:339-349)
            const int pos0 = bcast(c[4], j), rend0 = bcast(c[5], j), pos1 = bcast(c[6], j), rend1 = bcast(c[7], j);
            const bool has2 = (((flags >> 3) & 0xF) == 2);
            bool in0 = (p >= pos0 && p < rend0), in1 = (has2 && p >= pos1 && p < rend1);
            const bool full = (flags & 0x300) != 0;   // wave-uniform: not in k_frag_sums
            if (full) {
                const int nogap0 = bcast(c[10], j), nogap1 = bcast(c[11], j);
                bool lk0 = (in0 && p > pos0), lk1 = (in1 && p > pos1);   // LINK_M exists from the second base of a run on
                if (flags & 0x100) {
                    const int bpos0 = bcast(c[16], j), brend0 = bcast(c[17], j), bpos1 = bcast(c[20], j), brend1 = bcast(c[21], j);
                    const int sp0 = bcast(c[19], j), sp1 = bcast(c[23], j);
                    const bool inb0 = (p >= bpos0 && p < brend0), inb1 = (p >= bpos1 && p < brend1);
                    in0 = in0 || inb0; in1 = in1 || inb1;
                    lk0 = lk0 || (inb0 && p > bpos0); lk1 = lk1 || (inb1 && p > bpos1);
                    // positions next to an InDel carry InDel symbols / padded-deletion symbols: k_frag_generic does them for this fragment
                    const unsigned off = (unsigned)(p - fbeg);
                    if (off - (unsigned)(sp0 & 0xFFFF) < (unsigned)(sp0 >> 16) || off - (unsigned)(sp1 & 0xFFFF) < (unsigned)(sp1 >> 16)) cover = false;
                }
                // LINK_M: value of the better mate
                const int lv0 = (lk0 ? imax(noindel80 - nogap0, 0) + 1 : 0), lv1 = (lk1 ? imax(noindel80 - nogap1, 0) + 1 : 0);
                const int lv = imax(lv0, lv1);
                if (cover && lv > 0) link_full(lv, strand, K, singleton);
            }
            unsigned long long ev = 0ull;
            if (cover && (in0 || in1)) {
                const int v0 = q0 + P.bq_phred_added_misma, v1 = q1 + P.bq_phred_added_misma;
                const int A = (in0 ? v0 : 0), B = (in1 ? v1 : 0);
                const bool diff = (in0 && in1 && b0 != b1);
                const bool first = (v0 > v1) || (v0 == v1 && b0 < b1);
                const int cc = imax(A, B), ct = (diff ? A + B : cc);
                const int cs = ((in0 && (!diff || first)) ? b0 : b1);
                int cs4 = cs, cc4 = cc, ct4 = ct;
                if (padded_ignored) {   // fillConsensusCounts<false, true>: only A..T take part (main.hpp:410)
                    const int A4 = ((in0 && b0 <= UVC_BASE_T) ? v0 : 0), B4 = ((in1 && b1 <= UVC_BASE_T) ? v1 : 0);
                    const bool first4 = (A4 > B4) || (A4 == B4 && b0 < b1);
                    cc4 = imax(A4, B4); ct4 = (diff ? A4 + B4 : cc4);
                    cs4 = ((A4 == 0 && B4 == 0) ? UVC_BASE_T : (diff ? (first4 ? b0 : b1) : cs));
                }
                const int mq = maxq_base(cs);
                ev = apply(full, ar, cs, cc, ct, cs4, cc4, ct4, mq, strand, K, singleton, k0 + j);
            }
            const wmask mm = BAL(ev != 0ull);
            if (mm) {
                if (nq > RQ_CAP - 64) flush_events();
                if (ev != 0ull) myq[nq + (int)__builtin_popcountll(mm & ((1ull << lane) - 1ull))] = ev;
                nq += (int)__builtin_popcountll(mm);
            }
        }
    }
    };
    COARSE_T(ct1)
    // LINK_M sums of the fragments outside the interval sums, strand by strand: plain read-modify-write (these cells are this lane's)
    auto lacc_out = [&](int strand) {
        const int lr = (SPLIT ? 5 * strand : 0);
        const int n = lacc[lr + 0][col], ta = lacc[lr + 1][col], tb = lacc[lr + 2][col], mq = lacc[lr + 3][col], sg = lacc[lr + 4][col];
        if (valid && (n | sg)) {
            const int v0 = FRP(R, strand, UVC_FRAG_bDP, UVC_LINK_M, x), v1 = FRP(R, strand, UVC_FRAG_bTA, UVC_LINK_M, x), v2 = FRP(R, strand, UVC_FRAG_bTB, UVC_LINK_M, x);
            const int v3 = VQP(R, UVC_VQ_bMQ, UVC_LINK_M, x), v4 = FAP(R, strand, UVC_FAM_cDP12, UVC_LINK_M, x), v5 = FAP(R, strand, UVC_FAM_cDP21, UVC_LINK_M, x), v6 = FAP(R, strand, UVC_FAM_cDP1, UVC_LINK_M, x);
            if (n) { FRP(R, strand, UVC_FRAG_bDP, UVC_LINK_M, x) = v0 + n; FRP(R, strand, UVC_FRAG_bTA, UVC_LINK_M, x) = v1 + ta; FRP(R, strand, UVC_FRAG_bTB, UVC_LINK_M, x) = v2 + tb; VQP(R, UVC_VQ_bMQ, UVC_LINK_M, x) = v3 + mq; }
            if (sg) { FAP(R, strand, UVC_FAM_cDP12, UVC_LINK_M, x) = v4 + sg; FAP(R, strand, UVC_FAM_cDP21, UVC_LINK_M, x) = v5 + sg; if (vcfgen) FAP(R, strand, UVC_FAM_cDP1, UVC_LINK_M, x) = v6 + sg; }
        }
        if (!SPLIT) { for (int i = 0; i < 5; i++) lacc[i][col] = 0; }
    };
    run_list(std::false_type{}, a_fr);
    if (!SPLIT) lacc_out(0);
    run_list(std::true_type{}, a_rr);
    if (!SPLIT) lacc_out(1);
    if (nq > 0) flush_events();
    if (SPLIT) {
        // the partial sums of the four waves -> racc; every wave's adds to the planes must have reached the L2 before wave 0 reads them back
        const int part[11] = { a_fr.bDP, a_fr.bTA, a_fr.bTB, a_fr.c12, a_fr.c1, a_rr.bDP, a_rr.bTA, a_rr.bTB, a_rr.c12, a_rr.c1, bMQ_r };
#pragma unroll
        for (int i = 0; i < 11; i++) if (part[i]) atomicAdd(&racc[i][col], part[i]);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
        if (wv != 0) return;
        a_fr.bDP = racc[0][col]; a_fr.bTA = racc[1][col]; a_fr.bTB = racc[2][col]; a_fr.c12 = racc[3][col]; a_fr.c1 = racc[4][col];
        a_rr.bDP = racc[5][col]; a_rr.bTA = racc[6][col]; a_rr.bTB = racc[7][col]; a_rr.c12 = racc[8][col]; a_rr.c1 = racc[9][col]; bMQ_r = racc[10][col];
        lacc_out(0); lacc_out(1);
    }
    COARSE_T(ct2)
    if (!valid) return;
    // the adds above must have reached the L2 before the planes are read back (they pass through this CU's L1, which drops its copy of the line)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    COARSE_T(ce0)
    // the interval sums of the plain fragments (k_frag_sums): LINK_M by penalty class into the bucket histogram, the rest into the accumulators
    DAcc a_fl = {0,0,0,0,0}, a_rl = {0,0,0,0,0};
    int bMQ_l = 0;
    asm volatile("" ::: "memory");
    if (!proton) {
        int lcnt[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int32_t *S = R.fsum + (size_t)s * UVC_FSUM_N * R.npos + x;
            DAcc &ar = (s ? a_rr : a_fr), &al = (s ? a_rl : a_fl);
            const int sing = S[(size_t)UVC_FSUM_LSING * R.npos];
            al.c12 += sing;   // a LINK_M consensus is a vote of its own value > 0: cDP12 / cDP21, and cDP1 with P3
            if (vcfgen) {
                int n = 0;
#pragma unroll
                for (int e = 0; e < 5; e++) { const int c = S[(size_t)(UVC_FSUM_LCNT + e) * R.npos]; lcnt[e] += c; n += c; }
                al.bDP += n; al.bTA += S[(size_t)UVC_FSUM_LTA * R.npos]; al.bTB += S[(size_t)UVC_FSUM_LTB * R.npos]; bMQ_l += S[(size_t)UVC_FSUM_LMQ * R.npos];
                al.c1 += sing;
                ar.bDP += S[(size_t)UVC_FSUM_BDP * R.npos]; ar.bTA += S[(size_t)UVC_FSUM_BTA * R.npos]; ar.bTB += S[(size_t)UVC_FSUM_BTB * R.npos]; bMQ_r += S[(size_t)UVC_FSUM_BMQ * R.npos];
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int e = 0; e < 5; e++) {
            if (lcnt[e] == 0) continue;
            const int lv = imax(noindel80 - (e + 1), 0) + 1;
            int phredlike = imin(lv, maxq_link);
            if (!PLAIN && (0x1 & P.fam_flag)) phredlike = imin(phredlike, sscs_phred(P, my_ref, UVC_LINK_M));
            const int b = imax(0, maxq_link - phredlike);
            if (b < NBUCKETS) { if (H16) hist[1][b >> 1][col] += (unsigned)lcnt[e] << (16 * (b & 1)); else hist[1][b][col] += (unsigned)lcnt[e]; }
        }
    }
    // ---- the rest reads and writes this position's planes.  A read-modify-write per counter, one after the other, costs a memory round
    // trip each (measured with s_memtime: a quarter of a wave's life went here): every stage below issues all of its loads, then computes,
    // then stores.  Nobody else touches these cells in this kernel and the atomics above have landed.
    const bool has_generic = (R.n_sweep > 0 || proton);
    // (1) fragment depth of every symbol with the dense accumulators folded in; which symbols are present
    int totDP_b = 0, totDP_l = 0; unsigned present = 0;
#define STAGE_FENCE asm volatile("" ::: "memory");   // keeps the compiler from hoisting the next stage's loads into this one (registers: no scratch here)
    auto depth_stage = [&](auto LINKS) {
        constexpr int sb = decltype(LINKS)::value ? UVC_LINK_M : UVC_BASE_A, se = decltype(LINKS)::value ? UVC_LINK_NN : UVC_BASE_NN;
        int f0[se - sb + 1], f1[se - sb + 1];
#pragma unroll
        for (int s = sb; s <= se; s++) { f0[s - sb] = FRP(R, 0, UVC_FRAG_bDP, s, x); f1[s - sb] = FRP(R, 1, UVC_FRAG_bDP, s, x); }
#pragma unroll
        for (int s = sb; s <= se; s++) {
            int d0 = 0, d1 = 0;
            if (s <= UVC_BASE_NN && s == my_ref) { d0 = a_fr.bDP; d1 = a_rr.bDP; }
            if (s == UVC_LINK_M) { d0 = a_fl.bDP; d1 = a_rl.bDP; }
            if (d0) FRP(R, 0, UVC_FRAG_bDP, s, x) = f0[s - sb] + d0;
            if (d1) FRP(R, 1, UVC_FRAG_bDP, s, x) = f1[s - sb] + d1;
            const int nfr = f0[s - sb] + f1[s - sb] + d0 + d1;
            if (s <= UVC_BASE_NN) totDP_b += nfr; else totDP_l += nfr;
            if (nfr) present |= 1u << s;
        }
    };
    STAGE_FENCE
    depth_stage(std::false_type{});
    STAGE_FENCE
    depth_stage(std::true_type{});
    STAGE_FENCE
    COARSE_T(ce1)
    // (2) the other dense accumulators, one (symbol, strand) at a time: five loads, five stores
    auto flush5 = [&](const DAcc &a, int t, int sym) {
        if (!(a.bTA | a.bTB | a.c12 | a.c1)) return;
        const int v0 = FRP(R, t, UVC_FRAG_bTA, sym, x), v1 = FRP(R, t, UVC_FRAG_bTB, sym, x);
        const int v2 = FAP(R, t, UVC_FAM_cDP12, sym, x), v3 = FAP(R, t, UVC_FAM_cDP21, sym, x), v4 = FAP(R, t, UVC_FAM_cDP1, sym, x);
        if (a.bTA) FRP(R, t, UVC_FRAG_bTA, sym, x) = v0 + a.bTA;
        if (a.bTB) FRP(R, t, UVC_FRAG_bTB, sym, x) = v1 + a.bTB;
        if (a.c12) { FAP(R, t, UVC_FAM_cDP12, sym, x) = v2 + a.c12; FAP(R, t, UVC_FAM_cDP21, sym, x) = v3 + a.c12; }
        if (a.c1) FAP(R, t, UVC_FAM_cDP1, sym, x) = v4 + a.c1;
    };
    flush5(a_fr, 0, my_ref); STAGE_FENCE
    flush5(a_rr, 1, my_ref); STAGE_FENCE
    flush5(a_fl, 0, UVC_LINK_M); STAGE_FENCE
    flush5(a_rl, 1, UVC_LINK_M); STAGE_FENCE
    if (bMQ_r | bMQ_l) {
        const int q0 = VQP(R, UVC_VQ_bMQ, my_ref, x), q1 = VQP(R, UVC_VQ_bMQ, UVC_LINK_M, x);
        if (bMQ_r) VQP(R, UVC_VQ_bMQ, my_ref, x) = q0 + bMQ_r;
        if (bMQ_l) VQP(R, UVC_VQ_bMQ, UVC_LINK_M, x) = q1 + bMQ_l;
    }
    STAGE_FENCE
    COARSE_T(ce2)
    // (3) P3b (main.hpp:2801-2828): one pass per symbol that some lane of the wave has; its 16 buckets and three outputs in one batch of loads
    if (vcfgen) {
        unsigned long long todo = 0;   // wave-uniform set of symbols
#pragma unroll
        for (int s = 0; s < NSYM; s++) if (BAL((present >> s) & 1u)) todo |= 1ull << s;
        while (todo) {
            const int s = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            if (!((present >> s) & 1u)) continue;   // empty histogram -> all three outputs are 0
            const int dense = (s == my_ref ? 0 : (s == UVC_LINK_M ? 1 : -1));
            const bool global_b = (dense < 0 || has_generic);   // the global buckets hold the rare symbols and everything k_frag_generic added
            int h[NBUCKETS];
#pragma unroll
            for (int b = 0; b < NBUCKETS; b++) h[b] = (global_b ? BKP(R, 0, s, b, x) : 0);
            const int o0 = VQP(R, UVC_VQ_bIAQb, s, x), o1 = VQP(R, UVC_VQ_bIADb, s, x), o2 = VQP(R, UVC_VQ_bIDQb, s, x);
            const int max_qual = maxq_generic(s);
            if (global_b) {
#pragma unroll
                for (int b = 0; b < NBUCKETS; b++) if (h[b]) BKP(R, 0, s, b, x) = 0;   // clearSymbolBucketCount, main.hpp:2827
            }
            if (dense >= 0) {
#pragma unroll
                for (int b = 0; b < NBUCKETS; b++) h[b] += hist_get(dense, b);
            }
            int mv, ad2, bq2;
            infer_max_qual_regs(mv, ad2, bq2, max_qual, (s <= UVC_BASE_NN ? totDP_b : totDP_l), h);
            if (mv | ad2 | bq2) { VQP(R, UVC_VQ_bIAQb, s, x) = o0 + mv; VQP(R, UVC_VQ_bIADb, s, x) = o1 + ad2; VQP(R, UVC_VQ_bIDQb, s, x) = o2 + bq2; }
            STAGE_FENCE
        }
    }
#ifdef UVC_FRAG_COARSE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    COARSE_T(ct3)
    if (lane == 0 && (wave % 1499) == 7) printf("coarse wave %d prologue %llu lists %llu epilogue %llu = fence %llu sums+depth %llu flush %llu p3b %llu\n", wave, ct1 - ct0, ct2 - ct1, ct3 - ct2, ce0 - ct2, ce1 - ce0, ce2 - ce1, ct3 - ce2);
#endif
}

template <bool PLAIN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5,5))) k_frag(RegionDev R, UvcParams P) {
    __shared__ unsigned hist[2][NBUCKETS][256];   // [dense symbol][bucket][thread]: conflict-free
    __shared__ unsigned long long rq[4][RQ_CAP];
    __shared__ int lacc[5][256];
    frag_body<PLAIN, false>(R, P, hist, rq, lacc);
}
template <bool PLAIN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5,6))) k_frag16_split(RegionDev R, UvcParams P) {
    __shared__ unsigned hist[2][NBUCKETS / 2][64];
    __shared__ unsigned long long rq[4][RQ_CAP];
    __shared__ int lacc[10][64];
    __shared__ int racc[11][64];
    frag_body<PLAIN, true, true>(R, P, hist, rq, lacc, racc);
}
template <bool PLAIN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6,6))) k_frag16(RegionDev R, UvcParams P) {
    __shared__ unsigned hist[2][NBUCKETS / 2][256];   // 16 + 4 + 5 KiB: six blocks share a CU
    __shared__ unsigned long long rq[4][RQ_CAP];
    __shared__ int lacc[5][256];
    frag_body<PLAIN, true>(R, P, hist, rq, lacc);
}

// ------------------------------------------------------------------------------------------------
// generic family-strand units (more than one fragment, or UMI / duplex): one thread per (unit, position)
// ------------------------------------------------------------------------------------------------
// indel_len of the FAM2 position-bias test (main.hpp:3236-3246) for an insertion consensus: the number of fragments of the unit that carry
// the majority inserted sequence.  While the unit has no more insertion votes than microadjust_nobias_pos_indel_maxlen the value cannot
// matter (non_neg_minus gives 0 either way); above it the exact count comes from the allele pipeline (k_gap_alleles), found by the
// (family, position) key of the sorted candidates.
DEV int fam2_ins_len(const RegionDev &R, const UvcParams &P, const FsRec &u, int p, int votes) {
    if (votes <= P.microadjust_nobias_pos_indel_maxlen || R.gap.n_ev <= 0) return votes;
    const unsigned long long key = ((unsigned long long)u.fam << 26) | (unsigned long long)(p - R.beg);
    int lo = 0, hi = R.gap.n_ev;
    while (lo < hi) { const int m = (lo + hi) >> 1; if (R.gap.ckey_s[m] < key) lo = m + 1; else hi = m; }
    if (lo >= R.gap.n_ev || R.gap.ckey_s[lo] != key) return votes;   // cannot happen: an insertion vote comes from an insertion event
    return R.gap.maj[2 * lo + u.strand];
}
DEV int find_unit(const RegionDev &R, int64_t w) {   // last generic unit with work_off <= w
    int lo = 0, hi = R.n_generic_fs;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (R.fss[R.generic_fs[mid]].work_off <= w) lo = mid; else hi = mid; }
    return R.generic_fs[lo];
}

// builds con (votes) and optionally mmm (major-minus-minor BQ sums) of one unit at position p
// HAS_MMM selects the second output.  Fragments of <= 2 simple alignments (the FragFast record k_frag uses) are evaluated with the same
// select-based consensus as in k_frag; the others go through the per-alignment contribution path.
template <bool HAS_MMM, class Arr>
DEV void unit_counts(const RegionDev &R, const UvcParams &P, const FsRec &u, int p, bool proton, Arr con, Arr mmm) {
    const bool padded_ignored = (P.microadjust_padded_deletion_flag & (proton ? 0x2 : 0x1)) != 0;
    for (int s = 0; s < NSYM; s++) { con[s] = 0; if (HAS_MMM) mmm[s] = 0; }
    const int64_t x = p - R.beg;
    const int noindel80 = (x > 0 ? imin(80, imin(RTRP(R, UVC_RTR_indelphred, x - 1), RTRP(R, UVC_RTR_indelphred, x))) : 80);
    const __amdgpu_buffer_rsrc_t rs = bq_rsrc(R);
    int cnt[NSYM];
    for (int fi = u.frag_beg; fi < u.frag_end; fi++) {
        const FragFast &ff = R.ffast[R.frag_rank[fi]];
        const int fbeg = ff.beg, fend = ff.end, flags = ff.flags;
        if (p < fbeg || p >= fend) continue;
        if ((flags & 0x101) == 0 && !proton) {
            // register path (see k_frag): LINK_M value of the better mate; BASE_QUALITY_MAX merge of the two mates' bases
            const int pos0 = ff.pos0, rend0 = ff.rend0, pos1 = ff.pos1, rend1 = ff.rend1;
            const bool has2 = (((flags >> 3) & 0xF) == 2);
            const bool in0 = (p >= pos0 && p < rend0), in1 = (has2 && p >= pos1 && p < rend1);
            const int lv0 = ((in0 && p > pos0) ? imax(noindel80 - ff.nogap0, 0) + 1 : 0), lv1 = ((in1 && p > pos1) ? imax(noindel80 - ff.nogap1, 0) + 1 : 0);
            const int lv = imax(lv0, lv1);
            if (lv > 0) { con[UVC_LINK_M] += 1; if (HAS_MMM) mmm[UVC_LINK_M] += lv; }   // one symbol: 2*max - tot = lv, threshold 0 (main.hpp:466-520)
            if (in0 || in1) {
                const int bq0 = (in0 ? bq_load(rs, ff.qb0 + p) : 0), bq1 = (in1 ? bq_load(rs, ff.qb1 + p) : 0);
                const int b0 = bq0 & 0xFF, b1 = bq1 & 0xFF;
                const int v0 = ((bq0 >> 8) & 0xFF) + P.bq_phred_added_misma, v1 = ((bq1 >> 8) & 0xFF) + P.bq_phred_added_misma;
                const int A = (in0 ? v0 : 0), B = (in1 ? v1 : 0);
                const bool diff = (in0 && in1 && b0 != b1);
                const bool first = (v0 > v1) || (v0 == v1 && b0 < b1);
                const int cc = imax(A, B), ct = (diff ? A + B : cc);
                const int cs = ((in0 && (!diff || first)) ? b0 : b1);
                int cs4 = cs, cc4 = cc, ct4 = ct;
                if (padded_ignored) {   // fillConsensusCounts<false, true>: only A..T take part (main.hpp:410)
                    const int A4 = ((in0 && b0 <= UVC_BASE_T) ? v0 : 0), B4 = ((in1 && b1 <= UVC_BASE_T) ? v1 : 0);
                    const bool first4 = (A4 > B4) || (A4 == B4 && b0 < b1);
                    cc4 = imax(A4, B4); ct4 = (diff ? A4 + B4 : cc4);
                    cs4 = ((A4 == 0 && B4 == 0) ? UVC_BASE_T : (diff ? (first4 ? b0 : b1) : cs));
                }
                const int adj = imax(cc4 * 2, ct4) - ct4;
                if (adj >= P.fam_thres_highBQ_snv && adj > 0) con[cs4] += 1;
                if (HAS_MMM) { const int adj5 = imax(cc * 2, ct) - ct; if (adj5 > 0) mmm[cs] += adj5; }
            }
            continue;
        }
        const FragRec &f = R.frags[fi];
        frag_counts(R, P, f, p, proton, cnt);
        for (int st = 0; st < 2; st++) {
            int cs, cc, ct;
            if (st == UVC_LINK_SYMBOL) fill_consensus(cnt, cs, cc, ct, st, true, false);
            else fill_consensus(cnt, cs, cc, ct, st, false, padded_ignored);
            const int adj = imax(cc * 2, ct) - ct;
            if (adj >= (st == UVC_BASE_SYMBOL ? P.fam_thres_highBQ_snv : 0) && adj > 0) con[cs] += 1;
            if (HAS_MMM) {
                if (st == UVC_BASE_SYMBOL && padded_ignored) fill_consensus(cnt, cs, cc, ct, st, false, false);
                const int adj5 = imax(cc * 2, ct) - ct;
                if (adj5 > 0) mmm[cs] += adj5;
            }
        }
    }
}

// per-unit scalars: medians of read ends "as filled" (main.hpp:2916-2940) and the no-strict-bias window (:2959-2998)
__global__ void __launch_bounds__(64) k_fam_stat(RegionDev R, UvcParams P) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R.n_generic_fs) return;
    const int ui = R.generic_fs[t];
    FsRec u = R.fss[ui];
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    int n_l2r = 0, n_r2l = 0, qsum = 0, nq = 0;
    const int a_beg = R.frags[u.frag_beg].aln_beg, a_end = R.frags[u.frag_end - 1].aln_end;
    for (int k = a_beg; k < a_end; k++) { const AlnRec &a = R.alns[k]; if (a.flag & 0x10) n_r2l++; else n_l2r++; qsum += a.l_qseq; nq++; }
    auto nth = [&](bool rev, int idx) -> int {
        int c = 0;
        for (int k = a_beg; k < a_end; k++) { const AlnRec &a = R.alns[k]; if (((a.flag & 0x10) != 0) == rev) { if (c == idx) return rev ? a.pos : a.rend; c++; } }
        return 0;
    };
    u.l2r_end_median = (n_l2r > 0 ? (nth(false, (n_l2r - 1) / 2) + nth(false, n_l2r / 2)) / 2 : u.end);
    u.r2l_end_median = (n_r2l > 0 ? (nth(true, (n_r2l - 1) / 2) + nth(true, n_r2l / 2)) / 2 : u.beg);
    int nsb_min = u.end, nsb_max = u.beg;
    const int nfrags = u.frag_end - u.frag_beg;
    // the scan can only succeed for UMI families (or fam_flag & 0x2): is_fam_good needs it (main.hpp:2988-2989)
    if ((nfrags >= P.fam_thres_dup1add) && (qsum >= nq * P.fam_thres_qseqlen) && ((u.dflag & 0x1) || (P.fam_flag & 0x2))) {
        int con[NSYM];
        for (int dir = 0; dir < 2; dir++) {
            int b = (dir ? (u.end - 1) : u.beg), e = (dir ? (u.beg - 1) : u.end), step = (dir ? -1 : 1);
            for (int p = b; p != e; p += step) {
                unit_counts<false>(R, P, u, p, proton, con, con);
                int cs, cc, ct;
                fill_consensus(con, cs, cc, ct, UVC_BASE_SYMBOL, false, false);
                if (0 == ct) continue;
                const bool good = ((P.fam_thres_dup1add <= ct) && (cc * 100 >= ct * P.fam_thres_dup1perc) && ((u.dflag & 0x1) || (P.fam_flag & 0x2)));
                if (good && (UVC_BASE_N != cs) && (UVC_BASE_NN != cs)) { if (dir) nsb_max = p; else nsb_min = p; break; }
            }
        }
    }
    u.nsb_min = nsb_min; u.nsb_max = nsb_max;
    // k_fam_p4d leaves the (unit, position) cells under a fragment of the general kind (InDel reads, > 2 alignments) to k_fam_p4d_rest
    int has_general = (proton ? 1 : 0);
    for (int f = u.frag_beg; f < u.frag_end && !has_general; f++) if (R.ffast_u[f].v[3] & 0x101) has_general = 1;
    u.pad_ = has_general;
    R.fss[ui] = u;
}

// P4 of generic units (main.hpp:2999-3355; consensus FASTQ left out)
__global__ void __launch_bounds__(256) k_fam_p4(RegionDev R, UvcParams P) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= R.n_generic_work) return;
    const FsRec u = R.fss[find_unit(R, w)];
    const int p = u.beg + (int)(w - u.work_off);
    const int64_t x = p - R.beg;
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    const int strand = u.strand;
    __shared__ int con_s[NSYM][256];
    const LdsCounts<256> con = { &con_s[0][threadIdx.x] };
    unit_counts<false>(R, P, u, p, proton, con, con);
    for (int vi = 0; vi < 2; vi++) {
        const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
        int cs, cc, ct;
        fill_consensus(con, cs, cc, ct, st, false, false);
        if (0 == ct) continue;
        const bool is_fam_good = ((P.fam_thres_dup1add <= ct) && (cc * 100 >= ct * P.fam_thres_dup1perc) && ((u.dflag & 0x1) || (P.fam_flag & 0x2)));
        mark_sym(R, cs, x);
        atomicAdd(&FAP(R, strand, UVC_FAM_cDP12, cs, x), 1);
        if (1 == ct) atomicAdd(&FAP(R, strand, UVC_FAM_cDP21, cs, x), 1);
        if (!P.inferred_is_vcf_generated) continue;
        if (is_fam_good) {
            mark_fi(R, cs, x);
            atomicAdd(&FAP(R, strand, UVC_FAM_cDP2, cs, x), 1);
            int rbeg = imin(u.nsb_min, p), rend = imax(u.nsb_max, p);
            const bool nonconf_middle = (u.l2r_end_median <= (u.r2l_end_median + P.indel_adj_tracklen_dist));
            if (nonconf_middle && p < u.r2l_end_median) rend = imax(imin(u.l2r_end_median, imin(u.r2l_end_median, rend)), p);
            if (nonconf_middle && u.l2r_end_median < p) rbeg = imin(imax(u.l2r_end_median, imax(u.r2l_end_median, rbeg)), p);
            const bool isGap = (UVC_LINK_SYMBOL == st);
            const int bq = 90, dist = 1024 * 1024;
            if (((!isGap) && bq >= P.bias_thres_highBQ) || (isGap && dist >= P.bias_thres_highBQ)) {
                const bool tier2 = (isGap || bq >= P.bias_thres_highBQ);
                const int l_nb = (int)nnminus(p + 1, rbeg), r_nb = (int)nnminus(rend, p);
                const int _LPxT = TH(R, UVC_T_aLPxT, x), RPxT = TH(R, UVC_T_aRPxT, x);
                const int LPxT = (isGap ? _LPxT : imin(_LPxT, RPxT));
                // indel_len = majority COUNT of one inserted sequence among the unit's fragments (main.hpp:3239-3243): see fam2_ins_len
                const int indel_len = (is_ins(cs) ? fam2_ins_len(R, P, u, p, con[cs]) : 0);   // (a deletion's count is never read: main.hpp:3245)
                const bool far = (l_nb + (is_ins(cs) ? (int)nnminus(indel_len, P.microadjust_nobias_pos_indel_maxlen) : 0) >= LPxT) && (r_nb >= RPxT);
                if (far) {
                    int LP1 = 0, LP2 = 0, RP1 = 0, RP2 = 0; long long LPL = 0, RPL = 0;
                    bidir(LP1, LP2, RP1, RP2, LPL, RPL, TH(R, UVC_T_aLP1t, x), TH(R, UVC_T_aLP2t, x), TH(R, UVC_T_aRP1t, x), TH(R, UVC_T_aRP2t, x), l_nb, r_nb, true, 0);
                    if (LP1) atomicAdd(&FIP(R, UVC_FI_c2LP1, cs, x), LP1);
                    if (LP2) atomicAdd(&FIP(R, UVC_FI_c2LP2, cs, x), LP2);
                    if (RP1) atomicAdd(&FIP(R, UVC_FI_c2RP1, cs, x), RP1);
                    if (RP2) atomicAdd(&FIP(R, UVC_FI_c2RP2, cs, x), RP2);
                    atomicAdd(&FIP(R, UVC_FI_c2LPL, cs, x), (int)LPL); atomicAdd(&FIP(R, UVC_FI_c2RPL, cs, x), (int)RPL);
                }
                if ((int)nnminus(p + 1, u.nsb_min) >= P.bias_thres_strict_c2LRP0) atomicAdd(&FIP(R, UVC_FI_c2LP0, cs, x), 1);
                if ((int)nnminus(u.nsb_max, p) >= P.bias_thres_strict_c2LRP0) atomicAdd(&FIP(R, UVC_FI_c2RP0, cs, x), 1);
                const long long baq_last = R.end - 1;
                const int seg_l_baq = (int)(BAQ1(R, p) - BAQ1(R, lmax((long long)rbeg, nnminus(p, MAX_STR_N_BASES))) + 1);
                const long long rr = lmin((long long)rend - 1, lmin((long long)p + MAX_STR_N_BASES, baq_last));
                const int _seg_r_baq = (int)(BAQ1(R, rr) - BAQ1(R, p) + 1);
                const int seg_r_baq = (isGap ? (int)lmin((long long)_seg_r_baq, BAQ2(R, rr) - BAQ2(R, p) + 7) : _seg_r_baq);
                const int thres_highBAQ = P.bias_thres_highBAQ + (isGap ? 0 : 3);
                if (seg_l_baq >= thres_highBAQ && seg_r_baq >= thres_highBAQ) {
                    int LB1 = 0, LB2 = 0, RB1 = 0, RB2 = 0; long long LBL = 0, RBL = 0;
                    bidir(LB1, LB2, RB1, RB2, LBL, RBL, P.bias_thres_BAQ1, P.bias_thres_BAQ2, P.bias_thres_BAQ1, P.bias_thres_BAQ2, seg_l_baq, seg_r_baq, tier2, 0);
                    if (LB1) atomicAdd(&FIP(R, UVC_FI_c2LB1, cs, x), LB1);
                    if (LB2) atomicAdd(&FIP(R, UVC_FI_c2LB2, cs, x), LB2);
                    if (RB1) atomicAdd(&FIP(R, UVC_FI_c2RB1, cs, x), RB1);
                    if (RB2) atomicAdd(&FIP(R, UVC_FI_c2RB2, cs, x), RB2);
                    add64(&FI64P(R, UVC_FI64_c2LBL, cs, x), LBL); add64(&FI64P(R, UVC_FI64_c2RBL, cs, x), RBL);
                }
                atomicAdd(&FIP(R, UVC_FI_c2BQ2, cs, x), 1);
            }
        }
        if (P.fam_thres_dup2add <= ct && (cc * 100 >= ct * P.fam_thres_dup2perc)) atomicAdd(&FAP(R, strand, UVC_FAM_cDP3, cs, x), 1);
        const int flat = (is_subst(cs) ? P.fam_thres_emperr_all_flat_snv : P.fam_thres_emperr_all_flat_indel);
        const int perc = (is_subst(cs) ? P.fam_thres_emperr_con_perc_snv : P.fam_thres_emperr_con_perc_indel);
        if (ct < flat) continue;
        if (cc * 100 < ct * perc) continue;
        const int sb = (st == 0 ? UVC_BASE_A : UVC_LINK_M), se = (st == 0 ? UVC_BASE_NN : UVC_LINK_NN);
        int m = 0, M = 0;
        for (int s = sb; s <= se; s++) if (s != cs) { m += con[s]; M += ct; }
        mark_sym(R, cs, x);
        if (m) atomicAdd(&FAP(R, strand, UVC_FAM_cDPm, cs, x), m);
        atomicAdd(&FAP(R, strand, UVC_FAM_cDPM, cs, x), M);
    }
}

// P5 of generic units (main.hpp:3392-3513)
__global__ void __launch_bounds__(256) k_fam_p5(RegionDev R, UvcParams P) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= R.n_generic_work) return;
    const FsRec u = R.fss[find_unit(R, w)];
    const int p = u.beg + (int)(w - u.work_off);
    const int64_t x = p - R.beg;
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    const int strand = u.strand;
    const bool is_duplex_fam = (0x2 == (u.dflag & 0x2));
    const bool other_present = (u.other_fs >= 0);
    const bool will_inc_dscs = is_duplex_fam && other_present;
    const bool will_inc_sscs = is_duplex_fam && !other_present;
    __shared__ int con_s[NSYM][256], mmm_s[NSYM][256];
    const LdsCounts<256> con = { &con_s[0][threadIdx.x] }, mmm = { &mmm_s[0][threadIdx.x] };
    unit_counts<true>(R, P, u, p, proton, con, mmm);
    for (int vi = 0; vi < 2; vi++) {
        const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
        int cs, con_sumBQs, tot_sumBQs;
        fill_consensus(mmm, cs, con_sumBQs, tot_sumBQs, st, false, false);
        if (0 == tot_sumBQs) continue;
        const int con_nfrags = con[cs];
        int tot_nfrags = 0;
        const int sb = (st == 0 ? UVC_BASE_A : UVC_LINK_M), se = (st == 0 ? UVC_BASE_NN : UVC_LINK_NN);
        for (int s = sb; s <= se; s++) tot_nfrags += con[s];
        mark_sym(R, cs, x);
        atomicAdd(&FAP(R, strand, UVC_FAM_cDP1, cs, x), 1);
        if (will_inc_sscs && (!will_inc_dscs) && (tot_nfrags >= P.fam_thres_dup1add) && (con_nfrags * 100 >= tot_nfrags * P.fam_thres_dup1perc))
            atomicAdd(&FAP(R, strand, UVC_FAM_cDPD, cs, x), 1);
        const int avgBQ = ((0 == tot_nfrags) ? 1 : (con_sumBQs / tot_nfrags));
        const int majorcount = FAP(R, strand, UVC_FAM_cDPM, cs, x), minorcount = FAP(R, strand, UVC_FAM_cDPm, cs, x);
        const double prior_weight = 1.0 / (minorcount + 1.0);
        const double p2p = pow(10.0, (double)(-((float)avgBQ) / 10));   // phred2prob's float cast, main_conversion.hpp:885-888
        const double prob = (minorcount + prior_weight) / (majorcount + minorcount + prior_weight / p2p);
        const double realphred = -10 * log(prob) / log(10.0);
        const int indep_frag_phred = (int)round(((con_nfrags * 2) - tot_nfrags) * realphred);
        int confam_qual;
        if (UVC_LINK_SYMBOL == st) confam_qual = imax(1, imin(indep_frag_phred, P.fam_phred_indel_inc_before_barcode_labeling + (int)round(realphred)));
        else confam_qual = imax(1, imin(indep_frag_phred, (con_sumBQs * 2) - tot_sumBQs));
        const int max_qual = sscs_phred(P, R.refsym[x], cs) + (!P.tumor_vcf_is_provided ? 0 : 4);
        const int confam_qual2 = imin(confam_qual, max_qual);
        if (tot_nfrags >= P.fam_thres_dup1add) {
            const int pbucket = (max_qual - confam_qual2 + 2) / 4;
            if (pbucket >= 0 && pbucket < NBUCKETS) { atomicAdd(&BKP(R, strand, cs, pbucket, x), 1); R.p5flag[(size_t)strand * R.npos + x] = 1; }
        }
    }
}

// P4 (PASS 4, main.hpp:2883-3390) and P5 (PASS 5, main.hpp:3392-3513) of the generic units, one block per 64-position window.
// One thread per (unit, position) with an atomic per increment is bound by the L2 atomic units at depth (deep UMI panels put
// hundreds of units on a position).  Here the four waves of a block share the units that overlap the window, every lane keeps its
// position, and the increments of the two dense symbols (reference base, LINK_M) are collected in LDS and written once per position.
#define FAMW_SLOTS (2 * UVC_NFAM + UVC_NFAMINFO32)   // [strand][FAM field], then the FamFormatInfoSet i32 fields
struct FamAcc {
    int (*a32)[FAMW_SLOTS][64]; unsigned long long (*a64)[UVC_NFAMINFO64][64]; int (*bk)[2][NBUCKETS][64];
    const RegionDev *R; int64_t x; int lane, my_ref;
    DEV int dense(int cs) const { return !a32 ? -1 : (cs == my_ref ? 0 : (cs == UVC_LINK_M ? 1 : -1)); }   // without LDS accumulators (k_fam_p4d_rest) every increment is a global atomic
    // a window kernel (a32 set) owns its positions: its adds to the planes are L2 atomics of workgroup scope (add_own)
    DEV void gadd(int32_t *p, int v) const { if (a32) add_own(p, v); else atomicAdd(p, v); }
    DEV void fap(int strand, int f, int cs, int v) const { const int d = dense(cs); if (d >= 0) atomicAdd(&a32[d][strand * UVC_NFAM + f][lane], v); else { mark_sym(*R, cs, x); gadd(&FAP(*R, strand, f, cs, x), v); } }
    DEV void fi(int f, int cs, int v) const { const int d = dense(cs); if (d >= 0) atomicAdd(&a32[d][2 * UVC_NFAM + f][lane], v); else { occ_mark(*R, cs, x); mark_fi(*R, cs, x); gadd(&FIP(*R, f, cs, x), v); } }
    DEV void fi64(int f, int cs, long long v) const { const int d = dense(cs); if (d >= 0) atomicAdd(&a64[d][f][lane], (unsigned long long)v); else { occ_mark(*R, cs, x); mark_fi(*R, cs, x); if (a32) add64_own(&FI64P(*R, f, cs, x), v); else add64(&FI64P(*R, f, cs, x), v); } }
    DEV void bucket(int strand, int cs, int b) const { const int d = dense(cs); if (d >= 0) atomicAdd(&bk[d][strand][b][lane], 1); else gadd(&BKP(*R, strand, cs, b, x), 1); R->p5flag[(size_t)strand * R->npos + x] = 1; }
};

// The P4 increments of one (unit, position, symbol type) once the vote consensus (cs = symbol, cc = its votes, ct = all votes) is known
// (main.hpp:2999-3355); the body of k_fam_win<4> with con[] reduced to what it uses of it.
// what p4_apply reads of the position alone (thresholds, BAQ prefix sums): loaded once per lane, not once per unit
struct P4Pos { int LPxT, RPxT, LP1t, LP2t, RP1t, RP2t; long long baq1, baq2;
               const long long *lb1, *lb2; int lb_lo; };   // LB: the two BAQ prefix-sum arrays of [lb_lo, lb_lo + P4_BAQ_WIN) staged in LDS
#define P4_BAQ_WIN (64 + 2 * MAX_STR_N_BASES + 2)   // every BAQ index of a window's cells lies within MAX_STR_N_BASES of the window
template <bool LB = false>
DEV void p4_apply(const FamAcc &A, const RegionDev &R, const UvcParams &P, const FsRec &u, int p, int64_t x, int st, int cs, int cc, int ct, const P4Pos &Q) {
    auto baq1_at = [&](long long a) -> long long { if (LB) return Q.lb1[(int)a - Q.lb_lo]; return BAQ1(R, a); };
    auto baq2_at = [&](long long a) -> long long { if (LB) return Q.lb2[(int)a - Q.lb_lo]; return BAQ2(R, a); };
    const int strand = u.strand;
    {
                const bool is_fam_good = ((P.fam_thres_dup1add <= ct) && (cc * 100 >= ct * P.fam_thres_dup1perc) && ((u.dflag & 0x1) || (P.fam_flag & 0x2)));
                A.fap(strand, UVC_FAM_cDP12, cs, 1);
                if (1 == ct) A.fap(strand, UVC_FAM_cDP21, cs, 1);
                if (!P.inferred_is_vcf_generated) return;
                if (is_fam_good) {
                    A.fap(strand, UVC_FAM_cDP2, cs, 1);
                    int rbeg = imin(u.nsb_min, p), rend = imax(u.nsb_max, p);
                    const bool nonconf_middle = (u.l2r_end_median <= (u.r2l_end_median + P.indel_adj_tracklen_dist));
                    if (nonconf_middle && p < u.r2l_end_median) rend = imax(imin(u.l2r_end_median, imin(u.r2l_end_median, rend)), p);
                    if (nonconf_middle && u.l2r_end_median < p) rbeg = imin(imax(u.l2r_end_median, imax(u.r2l_end_median, rbeg)), p);
                    const bool isGap = (UVC_LINK_SYMBOL == st);
                    const int bq = 90, dist = 1024 * 1024;
                    if (((!isGap) && bq >= P.bias_thres_highBQ) || (isGap && dist >= P.bias_thres_highBQ)) {
                        const bool tier2 = (isGap || bq >= P.bias_thres_highBQ);
                        const int l_nb = (int)nnminus(p + 1, rbeg), r_nb = (int)nnminus(rend, p);
                        const int _LPxT = Q.LPxT, RPxT = Q.RPxT;
                        const int LPxT = (isGap ? _LPxT : imin(_LPxT, RPxT));
                        // indel_len = majority COUNT of one inserted sequence among the unit's fragments (main.hpp:3239-3243): see fam2_ins_len
                        const int indel_len = (is_ins(cs) ? fam2_ins_len(R, P, u, p, cc) : 0);   // cc = con[cs] of the vote consensus
                        const bool far = (l_nb + (is_ins(cs) ? (int)nnminus(indel_len, P.microadjust_nobias_pos_indel_maxlen) : 0) >= LPxT) && (r_nb >= RPxT);
                        if (far) {
                            int LP1 = 0, LP2 = 0, RP1 = 0, RP2 = 0; long long LPL = 0, RPL = 0;
                            bidir(LP1, LP2, RP1, RP2, LPL, RPL, Q.LP1t, Q.LP2t, Q.RP1t, Q.RP2t, l_nb, r_nb, true, 0);
                            if (LP1) A.fi(UVC_FI_c2LP1, cs, LP1);
                            if (LP2) A.fi(UVC_FI_c2LP2, cs, LP2);
                            if (RP1) A.fi(UVC_FI_c2RP1, cs, RP1);
                            if (RP2) A.fi(UVC_FI_c2RP2, cs, RP2);
                            A.fi(UVC_FI_c2LPL, cs, (int)LPL); A.fi(UVC_FI_c2RPL, cs, (int)RPL);
                        }
                        if ((int)nnminus(p + 1, u.nsb_min) >= P.bias_thres_strict_c2LRP0) A.fi(UVC_FI_c2LP0, cs, 1);
                        if ((int)nnminus(u.nsb_max, p) >= P.bias_thres_strict_c2LRP0) A.fi(UVC_FI_c2RP0, cs, 1);
                        const long long baq_last = R.end - 1;
                        const int seg_l_baq = (int)(Q.baq1 - baq1_at(lmax((long long)rbeg, nnminus(p, MAX_STR_N_BASES))) + 1);
                        const long long rr = lmin((long long)rend - 1, lmin((long long)p + MAX_STR_N_BASES, baq_last));
                        const int _seg_r_baq = (int)(baq1_at(rr) - Q.baq1 + 1);
                        const int seg_r_baq = (isGap ? (int)lmin((long long)_seg_r_baq, baq2_at(rr) - Q.baq2 + 7) : _seg_r_baq);
                        const int thres_highBAQ = P.bias_thres_highBAQ + (isGap ? 0 : 3);
                        if (seg_l_baq >= thres_highBAQ && seg_r_baq >= thres_highBAQ) {
                            int LB1 = 0, LB2 = 0, RB1 = 0, RB2 = 0; long long LBL = 0, RBL = 0;
                            bidir(LB1, LB2, RB1, RB2, LBL, RBL, P.bias_thres_BAQ1, P.bias_thres_BAQ2, P.bias_thres_BAQ1, P.bias_thres_BAQ2, seg_l_baq, seg_r_baq, tier2, 0);
                            if (LB1) A.fi(UVC_FI_c2LB1, cs, LB1);
                            if (LB2) A.fi(UVC_FI_c2LB2, cs, LB2);
                            if (RB1) A.fi(UVC_FI_c2RB1, cs, RB1);
                            if (RB2) A.fi(UVC_FI_c2RB2, cs, RB2);
                            A.fi64(UVC_FI64_c2LBL, cs, LBL); A.fi64(UVC_FI64_c2RBL, cs, RBL);
                        }
                        A.fi(UVC_FI_c2BQ2, cs, 1);
                    }
                }
                if (P.fam_thres_dup2add <= ct && (cc * 100 >= ct * P.fam_thres_dup2perc)) A.fap(strand, UVC_FAM_cDP3, cs, 1);
                const int flat = (is_subst(cs) ? P.fam_thres_emperr_all_flat_snv : P.fam_thres_emperr_all_flat_indel);
                const int perc = (is_subst(cs) ? P.fam_thres_emperr_con_perc_snv : P.fam_thres_emperr_con_perc_indel);
                if (ct < flat) return;
                if (cc * 100 < ct * perc) return;
                const int sb = (st == 0 ? UVC_BASE_A : UVC_LINK_M), se = (st == 0 ? UVC_BASE_NN : UVC_LINK_NN);
                const int m = ct - cc, M = ct * (se - sb);   // sum over the other symbols of the type of (their votes, all votes)
                if (m) A.fap(strand, UVC_FAM_cDPm, cs, m);
                A.fap(strand, UVC_FAM_cDPM, cs, M);
    }
}

// ---- p4_apply for the cells whose consensus symbol is one of the window kernel's two dense symbols (nearly all of them) ----
// p4_apply spends its time in the control flow around its ~20 conditional increments (exec-mask bookkeeping, the dense / rare select of every
// FamAcc call): 2.35 of k_fam_p4d's 5.4 ms on a 200 kb x 2000x tile.  Here every condition is a 0 / 1 value and every increment an
// unconditional LDS add of a value that may be 0, and what depends on (unit, position) only -- the two distances, the BAQ differences, the
// position-bias flags -- is computed once for the cell's LINK and BASE consensus (P4Cell) instead of once per symbol type.  Same sums.
struct P4Cell { int good_unit, l_nb, r_nb, lp1, lp2, rp1, rp2, lp0, rp0, seg_l_baq, seg_r_base, seg_r_gap; };
DEV void p4_cell(P4Cell &W, const RegionDev &R, const UvcParams &P, const FsRec &u, int p, const P4Pos &Q) {
    W.good_unit = (((u.dflag & 0x1) || (P.fam_flag & 0x2)) ? 1 : 0);
    int rbeg = imin(u.nsb_min, p), rend = imax(u.nsb_max, p);
    const bool nonconf_middle = (u.l2r_end_median <= (u.r2l_end_median + P.indel_adj_tracklen_dist));
    if (nonconf_middle && p < u.r2l_end_median) rend = imax(imin(u.l2r_end_median, imin(u.r2l_end_median, rend)), p);
    if (nonconf_middle && u.l2r_end_median < p) rbeg = imin(imax(u.l2r_end_median, imax(u.r2l_end_median, rbeg)), p);
    W.l_nb = (int)nnminus(p + 1, rbeg); W.r_nb = (int)nnminus(rend, p);
    // update_bidirectional_bias with tier2 = true, n_indel = 0 (bidir above)
    W.lp1 = (W.l_nb >= Q.LP1t); W.lp2 = (W.l_nb >= Q.LP2t); W.rp1 = (W.r_nb >= Q.RP1t); W.rp2 = (W.r_nb >= Q.RP2t);
    W.lp0 = ((int)nnminus(p + 1, u.nsb_min) >= P.bias_thres_strict_c2LRP0); W.rp0 = ((int)nnminus(u.nsb_max, p) >= P.bias_thres_strict_c2LRP0);
    const long long baq_last = R.end - 1;
    W.seg_l_baq = (int)(Q.baq1 - Q.lb1[(int)lmax((long long)rbeg, nnminus(p, MAX_STR_N_BASES)) - Q.lb_lo] + 1);
    const int rr = (int)lmin((long long)rend - 1, lmin((long long)p + MAX_STR_N_BASES, baq_last)) - Q.lb_lo;
    W.seg_r_base = (int)(Q.lb1[rr] - Q.baq1 + 1);
    W.seg_r_gap = (int)lmin((long long)W.seg_r_base, Q.lb2[rr] - Q.baq2 + 7);
}
// d = FamAcc::dense(cs) >= 0; GAP = the LINK symbol type (cs == LINK_M, never an insertion: indel_len = 0)
template <bool GAP>
DEV void p4_apply_dense(const FamAcc &A, int d, const UvcParams &P, const FsRec &u, int cc, int ct, const P4Pos &Q, const P4Cell &W) {
    int *b32 = &A.a32[d][0][A.lane];
    unsigned long long *b64 = &A.a64[d][0][A.lane];
#define P4ADD_FAP(f, v) atomicAdd(b32 + (u.strand * UVC_NFAM + (f)) * 64, (v))
#define P4ADD_FI(f, v) atomicAdd(b32 + (2 * UVC_NFAM + (f)) * 64, (v))
    P4ADD_FAP(UVC_FAM_cDP12, 1);
    P4ADD_FAP(UVC_FAM_cDP21, (1 == ct) ? 1 : 0);
    if (!P.inferred_is_vcf_generated) return;   // (uniform)
    const int good = ((P.fam_thres_dup1add <= ct) && (cc * 100 >= ct * P.fam_thres_dup1perc) && W.good_unit) ? 1 : 0;
    P4ADD_FAP(UVC_FAM_cDP2, good);
    const int bq = 90, dist = 1024 * 1024;
    if (GAP ? (dist >= P.bias_thres_highBQ) : (bq >= P.bias_thres_highBQ)) {   // (uniform; then tier2 is true for both types)
        const int LPxT = (GAP ? Q.LPxT : imin(Q.LPxT, Q.RPxT));
        const int gf = (good && (W.l_nb >= LPxT) && (W.r_nb >= Q.RPxT)) ? 1 : 0;
        P4ADD_FI(UVC_FI_c2LP1, gf & W.lp1); P4ADD_FI(UVC_FI_c2LP2, gf & W.lp2); P4ADD_FI(UVC_FI_c2RP1, gf & W.rp1); P4ADD_FI(UVC_FI_c2RP2, gf & W.rp2);
        P4ADD_FI(UVC_FI_c2LPL, gf ? W.l_nb : 0); P4ADD_FI(UVC_FI_c2RPL, gf ? W.r_nb : 0);
        P4ADD_FI(UVC_FI_c2LP0, good & W.lp0); P4ADD_FI(UVC_FI_c2RP0, good & W.rp0);
        const int seg_r_baq = (GAP ? W.seg_r_gap : W.seg_r_base);
        const int thres_highBAQ = P.bias_thres_highBAQ + (GAP ? 0 : 3);
        const int gb = (good && W.seg_l_baq >= thres_highBAQ && seg_r_baq >= thres_highBAQ) ? 1 : 0;
        P4ADD_FI(UVC_FI_c2LB1, gb & (W.seg_l_baq >= P.bias_thres_BAQ1 ? 1 : 0)); P4ADD_FI(UVC_FI_c2LB2, gb & (W.seg_l_baq >= P.bias_thres_BAQ2 ? 1 : 0));
        P4ADD_FI(UVC_FI_c2RB1, gb & (seg_r_baq >= P.bias_thres_BAQ1 ? 1 : 0)); P4ADD_FI(UVC_FI_c2RB2, gb & (seg_r_baq >= P.bias_thres_BAQ2 ? 1 : 0));
        atomicAdd(b64 + UVC_FI64_c2LBL * 64, (unsigned long long)(long long)(gb ? W.seg_l_baq : 0));
        atomicAdd(b64 + UVC_FI64_c2RBL * 64, (unsigned long long)(long long)(gb ? seg_r_baq : 0));
        P4ADD_FI(UVC_FI_c2BQ2, good);
    }
    P4ADD_FAP(UVC_FAM_cDP3, ((P.fam_thres_dup2add <= ct) && (cc * 100 >= ct * P.fam_thres_dup2perc)) ? 1 : 0);
    const int flat = (GAP ? P.fam_thres_emperr_all_flat_indel : P.fam_thres_emperr_all_flat_snv);
    const int perc = (GAP ? P.fam_thres_emperr_con_perc_indel : P.fam_thres_emperr_con_perc_snv);
    const bool e = (ct >= flat) && (cc * 100 >= ct * perc);
    P4ADD_FAP(UVC_FAM_cDPm, e ? ct - cc : 0);
    P4ADD_FAP(UVC_FAM_cDPM, e ? ct * (GAP ? (UVC_LINK_NN - UVC_LINK_M) : (UVC_BASE_NN - UVC_BASE_A)) : 0);
#undef P4ADD_FAP
#undef P4ADD_FI
}

// DG: P4 leaves a digest per (unit, position) -- the BQ-sum consensus P5 needs and the {1, 1}-threshold vote consensus of the duplex pass --
// so that the unit's fragments are walked once instead of three times (R.fam_digest, 32 B per cell); P5 then only reads it.
template <int PASS, bool DG>
__global__ void __launch_bounds__(256) k_fam_win(RegionDev R, UvcParams P) {
    __shared__ int con_s[(PASS == 5 && DG) ? 1 : 4][NSYM][(PASS == 5 && DG) ? 1 : 64];
    __shared__ int mmm_s[((PASS == 5) != DG) ? 4 : 1][NSYM][((PASS == 5) != DG) ? 64 : 1];   // needed by P4 with a digest and by P5 without
    __shared__ int a32[2][FAMW_SLOTS][64];
    __shared__ unsigned long long a64[2][UVC_NFAMINFO64][64];
    __shared__ int bk[PASS == 5 ? 2 : 1][2][NBUCKETS][PASS == 5 ? 64 : 1];
    __shared__ double p2p_s[PASS == 5 ? 128 : 1];   // phred2prob of every average quality a cell can have: one pow() per value and block, not one per cell
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t x0 = (int64_t)xcd_block() * 64;
    if (x0 >= R.npos) return;
    const int w0 = R.beg + (int)x0;
    // the units whose span can reach this window (sorted by begin)
    const int lo = win_lo(R, 7, (int)(x0 >> 6)), hi = win_hi(R, 7, (int)(x0 >> 6));
    if (lo >= hi) return;   // block-uniform
    for (int i = threadIdx.x; i < 2 * FAMW_SLOTS * 64; i += 256) (&a32[0][0][0])[i] = 0;
    for (int i = threadIdx.x; i < 2 * UVC_NFAMINFO64 * 64; i += 256) (&a64[0][0][0])[i] = 0ull;
    if (PASS == 5) for (int i = threadIdx.x; i < 2 * 2 * NBUCKETS * 64; i += 256) (&bk[0][0][0][0])[i] = 0;
    if (PASS == 5) for (int i = threadIdx.x; i < 128; i += 256) p2p_s[i] = pow(10.0, (double)(-((float)i) / 10));   // the expression of the cell, see below
    __syncthreads();
    const int p = w0 + lane;
    const int64_t x = x0 + lane;
    const bool valid = x < R.npos;
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    FamAcc A; A.a32 = a32; A.a64 = a64; A.bk = (int (*)[2][NBUCKETS][64])bk; A.R = &R; A.x = x; A.lane = lane; A.my_ref = (valid ? (int)R.refsym[x] : 0);
    const LdsCounts<64> con = { &con_s[(PASS == 5 && DG) ? 0 : wv][0][(PASS == 5 && DG) ? 0 : lane] }, mmm = { &mmm_s[((PASS == 5) != DG) ? wv : 0][0][((PASS == 5) != DG) ? lane : 0] };
    const bool padded_ignored_w = (P.microadjust_padded_deletion_flag & (proton ? 0x2 : 0x1)) != 0;
    // the unit records of this wave, 64 at a time: one per lane, those that reach the window picked by ballot (see k_fam_p4d)
    for (int kb = lo + wv; kb < hi; kb += 4 * 64) {
        int ur[16];
        const int kmine = kb + 4 * lane;
        if (kmine < hi) {
            const int4 *q4 = (const int4 *)(R.fss + R.generic_sorted[kmine]);
#pragma unroll
            for (int i = 0; i < 4; i++) { const int4 t = q4[i]; ur[4 * i] = t.x; ur[4 * i + 1] = t.y; ur[4 * i + 2] = t.z; ur[4 * i + 3] = t.w; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) ur[i] = 0;
        }
        unsigned long long todo = __ballot(kmine < hi && ur[3] > w0 && ur[2] < w0 + 64);
      while (todo) {
        const int uj = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        FsRec u;
        u.frag_beg = bcast(ur[0], uj); u.frag_end = bcast(ur[1], uj); u.beg = bcast(ur[2], uj); u.end = bcast(ur[3], uj);
        u.strand = bcast(ur[4], uj); u.dflag = bcast(ur[5], uj); u.fam = bcast(ur[6], uj); u.generic = bcast(ur[7], uj);
        u.work_off = (int64_t)(((unsigned long long)(unsigned)bcast(ur[9], uj) << 32) | (unsigned long long)(unsigned)bcast(ur[8], uj));
        u.l2r_end_median = bcast(ur[10], uj); u.r2l_end_median = bcast(ur[11], uj); u.nsb_min = bcast(ur[12], uj); u.nsb_max = bcast(ur[13], uj);
        u.other_fs = bcast(ur[14], uj); u.pad_ = 0;
        if (!(valid && p >= u.beg && p < u.end)) continue;
        const int strand = u.strand;
        if (PASS == 4) {
            if (DG) {
                // one pass over the unit's fragments serves P4, P5 and the duplex pass: P5 needs the BQ-sum consensus (mmm) and the vote
                // counts of its symbol, the duplex pass the vote consensus with thresholds {1, 1}; both are left here per (unit, position)
                unit_counts<true>(R, P, u, p, proton, con, mmm);
                uint32_t dg[8]; dg[6] = 0; dg[7] = 0;
                for (int vi = 0; vi < 2; vi++) {
                    const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
                    int cs, csum, tsum;
                    fill_consensus(mmm, cs, csum, tsum, st, false, false);
                    int tot_nfrags = 0;
                    const int sb = (st == 0 ? UVC_BASE_A : UVC_LINK_M), se = (st == 0 ? UVC_BASE_NN : UVC_LINK_NN);
                    for (int sy = sb; sy <= se; sy++) tot_nfrags += con[sy];
                    dg[3 * vi] = (uint32_t)cs | ((uint32_t)imin(con[cs], 16383) << 4) | ((uint32_t)imin(tot_nfrags, 16383) << 18);
                    dg[3 * vi + 1] = (uint32_t)csum; dg[3 * vi + 2] = (uint32_t)tsum;
                    int ds, dc, dt;
                    fill_consensus(con, ds, dc, dt, st, false, st == UVC_BASE_SYMBOL && padded_ignored_w);
                    const int adj = imax(dc * 2, dt) - dt;
                    dg[6] |= ((uint32_t)ds | ((adj >= 1) ? 16u : 0u)) << (8 * vi);
                }
                uint4 *dst = (uint4 *)(R.fam_digest + 8 * (u.work_off + (int64_t)(p - u.beg)));
                dst[0] = make_uint4(dg[0], dg[1], dg[2], dg[3]); dst[1] = make_uint4(dg[4], dg[5], dg[6], dg[7]);
            } else unit_counts<false>(R, P, u, p, proton, con, con);
            for (int vi = 0; vi < 2; vi++) {
                const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
                int cs, cc, ct;
                fill_consensus(con, cs, cc, ct, st, false, false);
                if (0 == ct) continue;
                const bool is_fam_good = ((P.fam_thres_dup1add <= ct) && (cc * 100 >= ct * P.fam_thres_dup1perc) && ((u.dflag & 0x1) || (P.fam_flag & 0x2)));
                A.fap(strand, UVC_FAM_cDP12, cs, 1);
                if (1 == ct) A.fap(strand, UVC_FAM_cDP21, cs, 1);
                if (!P.inferred_is_vcf_generated) continue;
                if (is_fam_good) {
                    A.fap(strand, UVC_FAM_cDP2, cs, 1);
                    int rbeg = imin(u.nsb_min, p), rend = imax(u.nsb_max, p);
                    const bool nonconf_middle = (u.l2r_end_median <= (u.r2l_end_median + P.indel_adj_tracklen_dist));
                    if (nonconf_middle && p < u.r2l_end_median) rend = imax(imin(u.l2r_end_median, imin(u.r2l_end_median, rend)), p);
                    if (nonconf_middle && u.l2r_end_median < p) rbeg = imin(imax(u.l2r_end_median, imax(u.r2l_end_median, rbeg)), p);
                    const bool isGap = (UVC_LINK_SYMBOL == st);
                    const int bq = 90, dist = 1024 * 1024;
                    if (((!isGap) && bq >= P.bias_thres_highBQ) || (isGap && dist >= P.bias_thres_highBQ)) {
                        const bool tier2 = (isGap || bq >= P.bias_thres_highBQ);
                        const int l_nb = (int)nnminus(p + 1, rbeg), r_nb = (int)nnminus(rend, p);
                        const int _LPxT = TH(R, UVC_T_aLPxT, x), RPxT = TH(R, UVC_T_aRPxT, x);
                        const int LPxT = (isGap ? _LPxT : imin(_LPxT, RPxT));
                        // indel_len = majority COUNT of one inserted sequence among the unit's fragments (main.hpp:3239-3243): see fam2_ins_len
                        const int indel_len = (is_ins(cs) ? fam2_ins_len(R, P, u, p, con[cs]) : 0);
                        const bool far = (l_nb + (is_ins(cs) ? (int)nnminus(indel_len, P.microadjust_nobias_pos_indel_maxlen) : 0) >= LPxT) && (r_nb >= RPxT);
                        if (far) {
                            int LP1 = 0, LP2 = 0, RP1 = 0, RP2 = 0; long long LPL = 0, RPL = 0;
                            bidir(LP1, LP2, RP1, RP2, LPL, RPL, TH(R, UVC_T_aLP1t, x), TH(R, UVC_T_aLP2t, x), TH(R, UVC_T_aRP1t, x), TH(R, UVC_T_aRP2t, x), l_nb, r_nb, true, 0);
                            if (LP1) A.fi(UVC_FI_c2LP1, cs, LP1);
                            if (LP2) A.fi(UVC_FI_c2LP2, cs, LP2);
                            if (RP1) A.fi(UVC_FI_c2RP1, cs, RP1);
                            if (RP2) A.fi(UVC_FI_c2RP2, cs, RP2);
                            A.fi(UVC_FI_c2LPL, cs, (int)LPL); A.fi(UVC_FI_c2RPL, cs, (int)RPL);
                        }
                        if ((int)nnminus(p + 1, u.nsb_min) >= P.bias_thres_strict_c2LRP0) A.fi(UVC_FI_c2LP0, cs, 1);
                        if ((int)nnminus(u.nsb_max, p) >= P.bias_thres_strict_c2LRP0) A.fi(UVC_FI_c2RP0, cs, 1);
                        const long long baq_last = R.end - 1;
                        const int seg_l_baq = (int)(BAQ1(R, p) - BAQ1(R, lmax((long long)rbeg, nnminus(p, MAX_STR_N_BASES))) + 1);
                        const long long rr = lmin((long long)rend - 1, lmin((long long)p + MAX_STR_N_BASES, baq_last));
                        const int _seg_r_baq = (int)(BAQ1(R, rr) - BAQ1(R, p) + 1);
                        const int seg_r_baq = (isGap ? (int)lmin((long long)_seg_r_baq, BAQ2(R, rr) - BAQ2(R, p) + 7) : _seg_r_baq);
                        const int thres_highBAQ = P.bias_thres_highBAQ + (isGap ? 0 : 3);
                        if (seg_l_baq >= thres_highBAQ && seg_r_baq >= thres_highBAQ) {
                            int LB1 = 0, LB2 = 0, RB1 = 0, RB2 = 0; long long LBL = 0, RBL = 0;
                            bidir(LB1, LB2, RB1, RB2, LBL, RBL, P.bias_thres_BAQ1, P.bias_thres_BAQ2, P.bias_thres_BAQ1, P.bias_thres_BAQ2, seg_l_baq, seg_r_baq, tier2, 0);
                            if (LB1) A.fi(UVC_FI_c2LB1, cs, LB1);
                            if (LB2) A.fi(UVC_FI_c2LB2, cs, LB2);
                            if (RB1) A.fi(UVC_FI_c2RB1, cs, RB1);
                            if (RB2) A.fi(UVC_FI_c2RB2, cs, RB2);
                            A.fi64(UVC_FI64_c2LBL, cs, LBL); A.fi64(UVC_FI64_c2RBL, cs, RBL);
                        }
                        A.fi(UVC_FI_c2BQ2, cs, 1);
                    }
                }
                if (P.fam_thres_dup2add <= ct && (cc * 100 >= ct * P.fam_thres_dup2perc)) A.fap(strand, UVC_FAM_cDP3, cs, 1);
                const int flat = (is_subst(cs) ? P.fam_thres_emperr_all_flat_snv : P.fam_thres_emperr_all_flat_indel);
                const int perc = (is_subst(cs) ? P.fam_thres_emperr_con_perc_snv : P.fam_thres_emperr_con_perc_indel);
                if (ct < flat) continue;
                if (cc * 100 < ct * perc) continue;
                const int sb = (st == 0 ? UVC_BASE_A : UVC_LINK_M), se = (st == 0 ? UVC_BASE_NN : UVC_LINK_NN);
                int m = 0, M = 0;
                for (int s = sb; s <= se; s++) if (s != cs) { m += con[s]; M += ct; }
                if (m) A.fap(strand, UVC_FAM_cDPm, cs, m);
                A.fap(strand, UVC_FAM_cDPM, cs, M);
            }
        } else {
            const bool is_duplex_fam = (0x2 == (u.dflag & 0x2));
            const bool other_present = (u.other_fs >= 0);
            const bool will_inc_dscs = is_duplex_fam && other_present;
            const bool will_inc_sscs = is_duplex_fam && !other_present;
            uint4 d0 = make_uint4(0, 0, 0, 0), d1 = make_uint4(0, 0, 0, 0);
            if (DG) { const uint4 *src = (const uint4 *)(R.fam_digest + 8 * (u.work_off + (int64_t)(p - u.beg))); d0 = src[0]; d1 = src[1]; }
            else unit_counts<true>(R, P, u, p, proton, con, mmm);
            for (int vi = 0; vi < 2; vi++) {
                const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
                int cs, con_sumBQs, tot_sumBQs, con_nfrags, tot_nfrags = 0;
                if (DG) {
                    const uint32_t a = (vi == 0 ? d0.x : d0.w);
                    cs = (int)(a & 15u); con_nfrags = (int)((a >> 4) & 16383u); tot_nfrags = (int)(a >> 18);
                    con_sumBQs = (int)(vi == 0 ? d0.y : d1.x); tot_sumBQs = (int)(vi == 0 ? d0.z : d1.y);
                    if (0 == tot_sumBQs) continue;
                } else {
                    fill_consensus(mmm, cs, con_sumBQs, tot_sumBQs, st, false, false);
                    if (0 == tot_sumBQs) continue;
                    con_nfrags = con[cs];
                    const int sb = (st == 0 ? UVC_BASE_A : UVC_LINK_M), se = (st == 0 ? UVC_BASE_NN : UVC_LINK_NN);
                    for (int s = sb; s <= se; s++) tot_nfrags += con[s];
                }
                A.fap(strand, UVC_FAM_cDP1, cs, 1);
                if (will_inc_sscs && (!will_inc_dscs) && (tot_nfrags >= P.fam_thres_dup1add) && (con_nfrags * 100 >= tot_nfrags * P.fam_thres_dup1perc))
                    A.fap(strand, UVC_FAM_cDPD, cs, 1);
                const int avgBQ = ((0 == tot_nfrags) ? 1 : (con_sumBQs / tot_nfrags));
                const int majorcount = FAP(R, strand, UVC_FAM_cDPM, cs, x), minorcount = FAP(R, strand, UVC_FAM_cDPm, cs, x);   // complete: P4 ran before
                // realphred = -10 log10(prob) enters through round(k * realphred) and round(realphred) only.  A single-precision evaluation
                // (error of realphred < 1e-5) decides both roundings unless a product lies within its error bound of a half: only then (one
                // wave iteration in ten) the fp64 expression of the reference is evaluated.  Two fp64 divisions and a log per cell and symbol
                // type were most of this kernel.
                const int kfr = (con_nfrags * 2) - tot_nfrags;
                int indep_frag_phred, round_realphred;
                {
                    const float wf = __builtin_amdgcn_rcpf((float)minorcount + 1.0f);
                    const float p2pf = ((unsigned)avgBQ < 128u) ? (float)p2p_s[avgBQ] : 0.0f;
                    const float probf = ((float)minorcount + wf) * __builtin_amdgcn_rcpf((float)majorcount + (float)minorcount + wf * __builtin_amdgcn_rcpf(p2pf));
                    const float rf = -3.0102999566f * __builtin_amdgcn_logf(probf);   // v_log_f32 is log2
                    const float v1 = (float)kfr * rf;
                    const float tol1 = 5.0e-4f + 4.0e-5f * fabsf((float)kfr) + 1.0e-6f * fabsf(v1), tol0 = 5.0e-4f;
                    const bool safe = ((unsigned)avgBQ < 128u) && (majorcount + minorcount < (1 << 22)) && rf > 0.0f && rf < 1000.0f
                                      && fabsf(v1 - floorf(v1) - 0.5f) > tol1 && fabsf(rf - floorf(rf) - 0.5f) > tol0;
                    if (safe) { indep_frag_phred = (int)roundf(v1); round_realphred = (int)roundf(rf); }
                    else {
                        const double prior_weight = 1.0 / (minorcount + 1.0);
                        const double p2p = ((unsigned)avgBQ < 128u) ? p2p_s[avgBQ] : pow(10.0, (double)(-((float)avgBQ) / 10));   // phred2prob's float cast, main_conversion.hpp:885-888
                        const double prob = (minorcount + prior_weight) / (majorcount + minorcount + prior_weight / p2p);
                        const double realphred = -10 * log(prob) / log(10.0);
                        indep_frag_phred = (int)round(kfr * realphred); round_realphred = (int)round(realphred);
                    }
                }
                int confam_qual;
                if (UVC_LINK_SYMBOL == st) confam_qual = imax(1, imin(indep_frag_phred, P.fam_phred_indel_inc_before_barcode_labeling + round_realphred));
                else confam_qual = imax(1, imin(indep_frag_phred, (con_sumBQs * 2) - tot_sumBQs));
                const int max_qual = sscs_phred(P, R.refsym[x], cs) + (!P.tumor_vcf_is_provided ? 0 : 4);
                const int confam_qual2 = imin(confam_qual, max_qual);
                if (tot_nfrags >= P.fam_thres_dup1add) {
                    const int pbucket = (max_qual - confam_qual2 + 2) / 4;
                    if (pbucket >= 0 && pbucket < NBUCKETS) A.bucket(strand, cs, pbucket);
                }
            }
        }
      }
    }
    __syncthreads();
    // one add per non-zero (field, dense symbol, position) of the window
    for (int i = threadIdx.x; i < 2 * FAMW_SLOTS * 64; i += 256) {
        const int v = (&a32[0][0][0])[i];
        if (!v) continue;
        const int ln = i & 63, slot = (i >> 6) % FAMW_SLOTS, d = (i >> 6) / FAMW_SLOTS;
        const int64_t xx = x0 + ln;
        const int sym = (d == 0 ? (int)R.refsym[xx] : UVC_LINK_M);
        if (slot < 2 * UVC_NFAM) add_own(&FAP(R, slot / UVC_NFAM, slot % UVC_NFAM, sym, xx), v);
        else { mark_fi(R, sym, xx); add_own(&FIP(R, slot - 2 * UVC_NFAM, sym, xx), v); }
    }
    for (int i = threadIdx.x; i < 2 * UVC_NFAMINFO64 * 64; i += 256) {
        const unsigned long long v = (&a64[0][0][0])[i];
        if (!v) continue;
        const int ln = i & 63, f = (i >> 6) % UVC_NFAMINFO64, d = (i >> 6) / UVC_NFAMINFO64;
        const int64_t xx = x0 + ln;
        mark_fi(R, (d == 0 ? (int)R.refsym[xx] : UVC_LINK_M), xx);
        add64_own(&FI64P(R, f, (d == 0 ? (int)R.refsym[xx] : UVC_LINK_M), xx), (long long)v);
    }
    if (PASS == 5) for (int i = threadIdx.x; i < 2 * 2 * NBUCKETS * 64; i += 256) {
        const int v = (&bk[0][0][0][0])[i];
        if (!v) continue;
        const int ln = i & 63, b = (i >> 6) % NBUCKETS, strand = ((i >> 6) / NBUCKETS) % 2, d = (i >> 6) / (2 * NBUCKETS);
        const int64_t xx = x0 + ln;
        add_own(&BKP(R, strand, (d == 0 ? (int)R.refsym[xx] : UVC_LINK_M), b, xx), v);
    }
}

// the consensus numbers of a (unit, position) that has a fragment of the general kind: out of line, so that its registers do not count
// against the occupancy of k_fam_p4d (the call is rare)
DEV void general_unit(const RegionDev &R, const UvcParams &P, const FsRec &u, int p, bool proton, bool padded_ignored,
                                                int *vcs, int *vcc, int *vct, int *mcs, int *msum, int *mtot, int *mcon, int *dcs, int *dadj) {
    int con_l[NSYM], mmm_l[NSYM];
    unit_counts<true>(R, P, u, p, proton, con_l, mmm_l);
    for (int vi = 0; vi < 2; vi++) {
        const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
        fill_consensus(con_l, vcs[vi], vcc[vi], vct[vi], st, false, false);
        fill_consensus(mmm_l, mcs[vi], msum[vi], mtot[vi], st, false, false);
        mcon[vi] = con_l[mcs[vi]];
        int dc, dt;
        fill_consensus(con_l, dcs[vi], dc, dt, st, false, st == UVC_BASE_SYMBOL && padded_ignored);
        dadj[vi] = imax(dc * 2, dt) - dt;
    }
}

// k_fam_p4d: P4 of the generic units on deep data, writing the digest (see k_fam_win).  Same window structure, but the votes (con) and the
// BQ sums (mmm) of a (unit, position) live in registers: a fragment of <= 2 simple alignments can only vote for LINK_M and for one of
// A C G T N, so the updates are a fixed LINK_M add and a select chain over five symbols.  Without the two per-wave LDS count arrays the
// block needs 17 KiB of LDS instead of 45, and the read-modify-write chains through LDS are gone.  A lane that meets a fragment of the
// general kind (InDel next to the position, > 2 alignments) redoes its unit through unit_counts<true> on local arrays (rare).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) k_fam_p4d(RegionDev R, UvcParams P) {
    __shared__ int a32[2][FAMW_SLOTS][64];
    __shared__ unsigned long long a64[2][UVC_NFAMINFO64][64];
    __shared__ long long baq_s[2][P4_BAQ_WIN];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t x0 = (int64_t)xcd_block() * 64;
    if (x0 >= R.npos) return;
    const int w0 = R.beg + (int)x0;
    const int lo = win_lo(R, 7, (int)(x0 >> 6)), hi = win_hi(R, 7, (int)(x0 >> 6));
    if (lo >= hi) return;   // block-uniform
    for (int i = threadIdx.x; i < 2 * FAMW_SLOTS * 64; i += 256) (&a32[0][0][0])[i] = 0;
    for (int i = threadIdx.x; i < 2 * UVC_NFAMINFO64 * 64; i += 256) (&a64[0][0][0])[i] = 0ull;
    // the BAQ prefix sums the position-bias tests of this window can ask for (p4_apply: within MAX_STR_N_BASES of the cell), staged once:
    // six dependent global gathers per (unit, window) otherwise
    const int lb_lo = w0 - MAX_STR_N_BASES - 1;
    for (int i = threadIdx.x; i < P4_BAQ_WIN; i += 256) {
        const long long q = lmin(lmax((long long)lb_lo + i, (long long)R.beg), (long long)R.beg + R.npos - 1);
        baq_s[0][i] = BAQ1(R, q); baq_s[1][i] = BAQ2(R, q);
    }
    __syncthreads();
    const int p = w0 + lane;
    const int64_t x = x0 + lane;
    const bool valid = x < R.npos;
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    const bool padded_ignored = (P.microadjust_padded_deletion_flag & (proton ? 0x2 : 0x1)) != 0;
    FamAcc A; A.a32 = a32; A.a64 = a64; A.bk = nullptr; A.R = &R; A.x = x; A.lane = lane; A.my_ref = (valid ? (int)R.refsym[x] : 0);
    const int noindel80 = ((valid && x > 0) ? imin(80, imin(RTRP(R, UVC_RTR_indelphred, x - 1), RTRP(R, UVC_RTR_indelphred, x))) : 80);
    const __amdgpu_buffer_rsrc_t rs = bq_rsrc(R);
    P4Pos Q = { 0, 0, 0, 0, 0, 0, 0, 0, &baq_s[0][0], &baq_s[1][0], lb_lo };
    if (valid) { Q.LPxT = TH(R, UVC_T_aLPxT, x); Q.RPxT = TH(R, UVC_T_aRPxT, x); Q.LP1t = TH(R, UVC_T_aLP1t, x); Q.LP2t = TH(R, UVC_T_aLP2t, x); Q.RP1t = TH(R, UVC_T_aRP1t, x); Q.RP2t = TH(R, UVC_T_aRP2t, x);
                 Q.baq1 = BAQ1(R, p); Q.baq2 = BAQ2(R, p); }
    // the unit records of this wave, 64 at a time: one per lane, those that reach the window picked by ballot, their fields broadcast --
    // a unit that ends in front of the window costs no memory round trip of its own
    for (int kb = lo + wv; kb < hi; kb += 4 * 64) {
        int ur[16];
        const int kmine = kb + 4 * lane;
        if (kmine < hi) {
            const int4 *q4 = (const int4 *)(R.fss + R.generic_sorted[kmine]);
#pragma unroll
            for (int i = 0; i < 4; i++) { const int4 t = q4[i]; ur[4 * i] = t.x; ur[4 * i + 1] = t.y; ur[4 * i + 2] = t.z; ur[4 * i + 3] = t.w; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) ur[i] = 0;
        }
        unsigned long long todo = __ballot(kmine < hi && ur[3] > w0 && ur[2] < w0 + 64);   // FsRec::end, FsRec::beg
      while (todo) {
        const int uj = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        FsRec u;
        u.frag_beg = bcast(ur[0], uj); u.frag_end = bcast(ur[1], uj); u.beg = bcast(ur[2], uj); u.end = bcast(ur[3], uj);
        u.strand = bcast(ur[4], uj); u.dflag = bcast(ur[5], uj); u.fam = bcast(ur[6], uj); u.generic = bcast(ur[7], uj);
        u.work_off = (int64_t)(((unsigned long long)(unsigned)bcast(ur[9], uj) << 32) | (unsigned long long)(unsigned)bcast(ur[8], uj));
        u.l2r_end_median = bcast(ur[10], uj); u.r2l_end_median = bcast(ur[11], uj); u.nsb_min = bcast(ur[12], uj); u.nsb_max = bcast(ur[13], uj);
        u.other_fs = bcast(ur[14], uj); u.pad_ = 0;
        const bool mine = (valid && p >= u.beg && p < u.end);
        // votes / BQ sums of LINK_M and of A C G T N (registers); the other eight symbols only through the general path
        int cL = 0, mL = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0;
        bool general = false;
        // the fragment records of the unit: one per lane (12 dwords), fields of record j broadcast with v_readlane, the base / quality
        // bytes of record j + 1 requested before record j is evaluated -- three dependent memory round trips per unit instead of three per fragment
        const int nfr = u.frag_end - u.frag_beg;
        for (int f0 = 0; f0 < nfr; f0 += 64) {
            int c[12];
            if (f0 + lane < nfr) {
                const int4 *q4 = (const int4 *)(R.ffast_u + u.frag_beg + f0 + lane);
#pragma unroll
                for (int i = 0; i < 3; i++) { const int4 t = q4[i]; c[4 * i] = t.x; c[4 * i + 1] = t.y; c[4 * i + 2] = t.z; c[4 * i + 3] = t.w; }
            } else {
#pragma unroll
                for (int i = 0; i < 12; i++) c[i] = 0;
            }
            const int n = imin(64, nfr - f0);
            int bq0n = 0, bq1n = 0;
            auto issue = [&](int j) {
                const int fl = bcast(c[3], j);
                bq0n = bq_load(rs, bcast(c[8], j) + p);
                if (((fl >> 3) & 0xF) == 2) bq1n = bq_load(rs, bcast(c[9], j) + p);
            };
            issue(0);
            for (int j = 0; j < n; j++) {
                const int bq0 = bq0n, bq1 = bq1n;
                if (j + 1 < n) issue(j + 1);
                const int fbeg = bcast(c[0], j), fend = bcast(c[1], j), flags = bcast(c[3], j);
                if (!(mine && p >= fbeg && p < fend)) continue;
                if ((flags & 0x101) != 0 || proton) { general = true; continue; }
                const int pos0 = bcast(c[4], j), rend0 = bcast(c[5], j), pos1 = bcast(c[6], j), rend1 = bcast(c[7], j);
                const int nogap0 = bcast(c[10], j), nogap1 = bcast(c[11], j);
                const bool has2 = (((flags >> 3) & 0xF) == 2);
                const bool in0 = (p >= pos0 && p < rend0), in1 = (has2 && p >= pos1 && p < rend1);
                const int lv0 = ((in0 && p > pos0) ? imax(noindel80 - nogap0, 0) + 1 : 0), lv1 = ((in1 && p > pos1) ? imax(noindel80 - nogap1, 0) + 1 : 0);
                const int lv = imax(lv0, lv1);
                if (lv > 0) { cL += 1; mL += lv; }
                if (in0 || in1) {
                    const int b0 = bq0 & 0xFF, b1 = bq1 & 0xFF;
                    const int v0 = ((bq0 >> 8) & 0xFF) + P.bq_phred_added_misma, v1 = ((bq1 >> 8) & 0xFF) + P.bq_phred_added_misma;
                    const int Aq = (in0 ? v0 : 0), Bq = (in1 ? v1 : 0);
                    const bool diff = (in0 && in1 && b0 != b1);
                    const bool first = (v0 > v1) || (v0 == v1 && b0 < b1);
                    const int cc = imax(Aq, Bq), ct = (diff ? Aq + Bq : cc);
                    const int cs = ((in0 && (!diff || first)) ? b0 : b1);
                    int cs4 = cs, cc4 = cc, ct4 = ct;
                    if (padded_ignored) {
                        const int A4 = ((in0 && b0 <= UVC_BASE_T) ? v0 : 0), B4 = ((in1 && b1 <= UVC_BASE_T) ? v1 : 0);
                        const bool first4 = (A4 > B4) || (A4 == B4 && b0 < b1);
                        cc4 = imax(A4, B4); ct4 = (diff ? A4 + B4 : cc4);
                        cs4 = ((A4 == 0 && B4 == 0) ? UVC_BASE_T : (diff ? (first4 ? b0 : b1) : cs));
                    }
                    if (cs > UVC_BASE_N || cs4 > UVC_BASE_N) { general = true; continue; }   // cannot happen for packed base codes 0..4; kept as a guard
                    const int adj = imax(cc4 * 2, ct4) - ct4;
                    const int vote = ((adj >= P.fam_thres_highBQ_snv && adj > 0) ? 1 : 0);
                    c0 += (cs4 == 0 ? vote : 0); c1 += (cs4 == 1 ? vote : 0); c2 += (cs4 == 2 ? vote : 0); c3 += (cs4 == 3 ? vote : 0); c4 += (cs4 == 4 ? vote : 0);
                    const int adj5 = imax(imax(cc * 2, ct) - ct, 0);
                    m0 += (cs == 0 ? adj5 : 0); m1 += (cs == 1 ? adj5 : 0); m2 += (cs == 2 ? adj5 : 0); m3 += (cs == 3 ? adj5 : 0); m4 += (cs == 4 ? adj5 : 0);
                }
            }
        }
        if (!mine) continue;
        // per symbol type: vote consensus (cs, cc, ct), BQ-sum consensus (ms, msum, mtot) with the votes of its symbol, duplex vote
        int vcs[2], vcc[2], vct[2], mcs[2], msum[2], mtot[2], mcon[2], dcs[2], dadj[2];
        if (!general) {
            // LINK: only LINK_M can be non-zero; an empty type gives the last symbol of the type (fill_consensus starts from it)
            vcs[0] = (cL > 0 ? UVC_LINK_M : UVC_LINK_NN); vcc[0] = cL; vct[0] = cL;
            mcs[0] = (mL > 0 ? UVC_LINK_M : UVC_LINK_NN); msum[0] = mL; mtot[0] = mL; mcon[0] = (mL > 0 ? cL : 0);
            dcs[0] = vcs[0]; dadj[0] = cL;   // imax(2 * cc, ct) - ct with cc == ct
            // BASE: first maximum in symbol order A C G T N (NN is zero)
            int bs = UVC_BASE_NN, bc = 0;
            if (c0 > bc) { bs = 0; bc = c0; } if (c1 > bc) { bs = 1; bc = c1; } if (c2 > bc) { bs = 2; bc = c2; } if (c3 > bc) { bs = 3; bc = c3; } if (c4 > bc) { bs = 4; bc = c4; }
            vcs[1] = bs; vcc[1] = bc; vct[1] = c0 + c1 + c2 + c3 + c4;
            int qs = UVC_BASE_NN, qc = 0;
            if (m0 > qc) { qs = 0; qc = m0; } if (m1 > qc) { qs = 1; qc = m1; } if (m2 > qc) { qs = 2; qc = m2; } if (m3 > qc) { qs = 3; qc = m3; } if (m4 > qc) { qs = 4; qc = m4; }
            mcs[1] = qs; msum[1] = qc; mtot[1] = m0 + m1 + m2 + m3 + m4;
            mcon[1] = (qs == 0 ? c0 : qs == 1 ? c1 : qs == 2 ? c2 : qs == 3 ? c3 : qs == 4 ? c4 : 0);
            // the duplex vote looks at A..T only when padded deletions are ignored (fill_consensus(..., ignore_padded_del)): then the start
            // symbol is T and N does not take part
            if (padded_ignored) {
                int ds = UVC_BASE_T, dc = 0;
                if (c0 > dc) { ds = 0; dc = c0; } if (c1 > dc) { ds = 1; dc = c1; } if (c2 > dc) { ds = 2; dc = c2; } if (c3 > dc) { ds = 3; dc = c3; }
                const int dt = c0 + c1 + c2 + c3;
                dcs[1] = ds; dadj[1] = imax(dc * 2, dt) - dt;
            } else { dcs[1] = bs; dadj[1] = imax(bc * 2, vct[1]) - vct[1]; }
        } else continue;   // a cell under a fragment of the general kind: k_fam_p4d_rest writes its digest and its increments
        uint32_t dg6 = 0;
        uint32_t dga[2];
        for (int vi = 0; vi < 2; vi++) {
            dga[vi] = (uint32_t)mcs[vi] | ((uint32_t)imin(mcon[vi], 16383) << 4) | ((uint32_t)imin(vct[vi], 16383) << 18);
            dg6 |= ((uint32_t)dcs[vi] | ((dadj[vi] >= 1) ? 16u : 0u)) << (8 * vi);
        }
        uint4 *dst = (uint4 *)(R.fam_digest + 8 * (u.work_off + (int64_t)(p - u.beg)));
        dst[0] = make_uint4(dga[0], (uint32_t)msum[0], (uint32_t)mtot[0], dga[1]); dst[1] = make_uint4((uint32_t)msum[1], (uint32_t)mtot[1], dg6, 0u);
#ifndef UVC_ABLATE_P4APPLY
        P4Cell W;
        p4_cell(W, R, P, u, p, Q);
        if (vct[0]) { const int d = A.dense(vcs[0]); if (d >= 0) p4_apply_dense<true>(A, d, P, u, vcc[0], vct[0], Q, W); else p4_apply<true>(A, R, P, u, p, x, UVC_LINK_SYMBOL, vcs[0], vcc[0], vct[0], Q); }
        if (vct[1]) { const int d = A.dense(vcs[1]); if (d >= 0) p4_apply_dense<false>(A, d, P, u, vcc[1], vct[1], Q, W); else p4_apply<true>(A, R, P, u, p, x, UVC_BASE_SYMBOL, vcs[1], vcc[1], vct[1], Q); }
#endif
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * FAMW_SLOTS * 64; i += 256) {
        const int v = (&a32[0][0][0])[i];
        if (!v) continue;
        const int ln = i & 63, slot = (i >> 6) % FAMW_SLOTS, d = (i >> 6) / FAMW_SLOTS;
        const int64_t xx = x0 + ln;
        const int sym = (d == 0 ? (int)R.refsym[xx] : UVC_LINK_M);
        if (slot < 2 * UVC_NFAM) add_own(&FAP(R, slot / UVC_NFAM, slot % UVC_NFAM, sym, xx), v);
        else { mark_fi(R, sym, xx); add_own(&FIP(R, slot - 2 * UVC_NFAM, sym, xx), v); }
    }
    for (int i = threadIdx.x; i < 2 * UVC_NFAMINFO64 * 64; i += 256) {
        const unsigned long long v = (&a64[0][0][0])[i];
        if (!v) continue;
        const int ln = i & 63, f = (i >> 6) % UVC_NFAMINFO64, d = (i >> 6) / UVC_NFAMINFO64;
        const int64_t xx = x0 + ln;
        mark_fi(R, (d == 0 ? (int)R.refsym[xx] : UVC_LINK_M), xx);
        add64_own(&FI64P(R, f, (d == 0 ? (int)R.refsym[xx] : UVC_LINK_M), xx), (long long)v);
    }
}

// The cells k_fam_p4d leaves out: positions of a unit under a fragment of the general kind (an InDel read, more than two alignments).
// One wave per generic unit that has such a fragment (FsRec::pad_, set by k_fam_stat), lanes over its positions; the consensus comes from
// the general walk (unit_counts<true>), the increments are global atomics.  A few per cent of the units on panel data.
__global__ void __launch_bounds__(64) k_fam_p4d_rest(RegionDev R, UvcParams P) {
    if ((int)blockIdx.x >= R.n_generic_fs) return;
    const FsRec u = R.fss[R.generic_fs[blockIdx.x]];
    if (!u.pad_) return;
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    const bool padded_ignored = (P.microadjust_padded_deletion_flag & (proton ? 0x2 : 0x1)) != 0;
    for (int p = u.beg + (int)threadIdx.x; p < u.end; p += 64) {
        const int64_t x = (int64_t)p - R.beg;
        if (x < 0 || x >= R.npos) continue;
        bool general = false;
        for (int f = u.frag_beg; f < u.frag_end && !general; f++) {
            const FragUnit &ff = R.ffast_u[f];
            general = (p >= ff.v[0] && p < ff.v[1] && (((ff.v[3] & 0x101) != 0) || proton));
        }
        if (!general) continue;
        int vcs[2], vcc[2], vct[2], mcs[2], msum[2], mtot[2], mcon[2], dcs[2], dadj[2];
        general_unit(R, P, u, p, proton, padded_ignored, vcs, vcc, vct, mcs, msum, mtot, mcon, dcs, dadj);
        uint32_t dg6 = 0;
        uint32_t dga[2];
        for (int vi = 0; vi < 2; vi++) {
            dga[vi] = (uint32_t)mcs[vi] | ((uint32_t)imin(mcon[vi], 16383) << 4) | ((uint32_t)imin(vct[vi], 16383) << 18);
            dg6 |= ((uint32_t)dcs[vi] | ((dadj[vi] >= 1) ? 16u : 0u)) << (8 * vi);
        }
        uint4 *dst = (uint4 *)(R.fam_digest + 8 * (u.work_off + (int64_t)(p - u.beg)));
        dst[0] = make_uint4(dga[0], (uint32_t)msum[0], (uint32_t)mtot[0], dga[1]); dst[1] = make_uint4((uint32_t)msum[1], (uint32_t)mtot[1], dg6, 0u);
        FamAcc A; A.a32 = nullptr; A.a64 = nullptr; A.bk = nullptr; A.R = &R; A.x = x; A.lane = 0; A.my_ref = -1;
        P4Pos Q;
        Q.LPxT = TH(R, UVC_T_aLPxT, x); Q.RPxT = TH(R, UVC_T_aRPxT, x); Q.LP1t = TH(R, UVC_T_aLP1t, x); Q.LP2t = TH(R, UVC_T_aLP2t, x); Q.RP1t = TH(R, UVC_T_aRP1t, x); Q.RP2t = TH(R, UVC_T_aRP2t, x);
        Q.baq1 = BAQ1(R, p); Q.baq2 = BAQ2(R, p);
        for (int vi = 0; vi < 2; vi++) {
            if (0 == vct[vi]) continue;
            p4_apply(A, R, P, u, p, x, (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL), vcs[vi], vcc[vi], vct[vi], Q);
        }
    }
}

// duplex consensus (main.hpp:3427-3433, 3523-3550): one thread per (strand-0 unit of a duplex family with both strands, position)
__global__ void __launch_bounds__(256) k_duplex(RegionDev R, UvcParams P, const int32_t *dup_units, int n_dup, const int64_t *dup_off, int64_t n_work) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_work) return;
    int lo = 0, hi = n_dup;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (dup_off[mid] <= w) lo = mid; else hi = mid; }
    const FsRec u0 = R.fss[dup_units[lo]];
    const FsRec u1 = R.fss[u0.other_fs];
    const int dbeg = imin(u0.beg, u1.beg);
    const int p = dbeg + (int)(w - dup_off[lo]);
    const int64_t x = p - R.beg;
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    const bool padded_ignored = (P.microadjust_padded_deletion_flag & (proton ? 0x2 : 0x1)) != 0;
    __shared__ int con_s[NSYM][256], dup_s[NSYM][256];
    const LdsCounts<256> con = { &con_s[0][threadIdx.x] }, dup = { &dup_s[0][threadIdx.x] };
    for (int s = 0; s < NSYM; s++) dup[s] = 0;
    for (int k = 0; k < 2; k++) {
        const FsRec &u = (k == 0 ? u0 : u1);
        if (p < u.beg || p >= u.end) continue;
        unit_counts<false>(R, P, u, p, proton, con, con);
        for (int st = 0; st < 2; st++) {   // updateByFiltering<true,false,false> with thresholds {1,1}
            int cs, cc, ct;
            fill_consensus(con, cs, cc, ct, st, false, st == UVC_BASE_SYMBOL && padded_ignored);
            const int adj = imax(cc * 2, ct) - ct;
            if (adj >= 1 && adj > 0) dup[cs] += 1;
        }
    }
    for (int st = 0; st < 2; st++) {
        int cs, cc, ct;
        fill_consensus(dup, cs, cc, ct, st, false, false);
        if (0 < ct) { mark_dup(R, cs, x); atomicAdd(&DUP(R, UVC_DUPLEX_dDP1, cs, x), 1); }
        if (1 < ct) atomicAdd(&DUP(R, UVC_DUPLEX_dDP2, cs, x), 1);
    }
}

// the duplex pass from the digests P4 left (k_fam_win<4, true>): the votes of the two strand units of a duplex family at a position
__global__ void __launch_bounds__(64) k_duplex_d(RegionDev R, const int32_t *dup_units, int n_dup, const int64_t *dup_off, int64_t n_work) {
    // one wave per duplex family (its strand-0 unit), lanes over the positions of the two units' common span: no search for the family of a
    // (family, position) cell, the two unit records are read once
    if ((int)blockIdx.x >= n_dup) return;
    const FsRec u0 = R.fss[dup_units[blockIdx.x]];
    const FsRec u1 = R.fss[u0.other_fs];
    const int pbeg = imin(u0.beg, u1.beg);
    const int span = (int)(((int)blockIdx.x + 1 < n_dup ? dup_off[blockIdx.x + 1] : n_work) - dup_off[blockIdx.x]);   // the family's cells in the old (family, position) numbering
    for (int i = (int)threadIdx.x; i < span; i += 64) {
        const int p = pbeg + i;
        const int64_t x = p - R.beg;
        uint32_t v[2] = { 0, 0 };
        if (p >= u0.beg && p < u0.end) v[0] = R.fam_digest[8 * (u0.work_off + (int64_t)(p - u0.beg)) + 6];
        if (p >= u1.beg && p < u1.end) v[1] = R.fam_digest[8 * (u1.work_off + (int64_t)(p - u1.beg)) + 6];
        for (int vi = 0; vi < 2; vi++) {   // fill_consensus over at most two votes: the larger count wins, the smaller symbol on a tie
            const uint32_t a = (v[0] >> (8 * vi)) & 31u, b = (v[1] >> (8 * vi)) & 31u;
            const bool va = (a & 16u) != 0, vb = (b & 16u) != 0;
            if (!va && !vb) continue;
            const int sa = (int)(a & 15u), sb = (int)(b & 15u);
            const int ct = (va ? 1 : 0) + (vb ? 1 : 0);
            const int cs = (va && vb) ? ((sa == sb) ? sa : imin(sa, sb)) : (va ? sa : sb);
            mark_dup(R, cs, x);
            atomicAdd(&DUP(R, UVC_DUPLEX_dDP1, cs, x), 1);
            if (1 < ct) atomicAdd(&DUP(R, UVC_DUPLEX_dDP2, cs, x), 1);   // tot_count of the two votes, as fill_consensus sums it (main.hpp:3540-3546)
        }
    }
}

// P5b (main.hpp:3552-3591): one thread per (position, strand)
__global__ void __launch_bounds__(256) k_p5b(RegionDev R, UvcParams P) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= R.npos * 2) return;
    if (!R.p5flag[t]) return;   // no P5 bucket was filled at this (strand, position): every output is 0 (the flag index is strand * npos + x)
    const int strand = (int)(t / R.npos);
    const int64_t x = t % R.npos;
    const int qIAQ = (strand ? UVC_VQ_cIAQr : UVC_VQ_cIAQf), qIAD = (strand ? UVC_VQ_cIADr : UVC_VQ_cIADf), qIDQ = (strand ? UVC_VQ_cIDQr : UVC_VQ_cIDQf);
    const int ref_symbol = R.refsym[x];
    for (int st = 0; st < 2; st++) {
        const int sb = (st == 0 ? UVC_BASE_A : UVC_LINK_M), se = (st == 0 ? UVC_BASE_NN : UVC_LINK_NN);
        int totDP = 0;
        for (int s = sb; s <= se; s++) totDP += FAP(R, strand, UVC_FAM_cDP1, s, x);
        if (totDP == 0) continue;
        for (int s = sb; s <= se; s++) {
            if (0 == FAP(R, strand, UVC_FAM_cDP1, s, x)) continue;   // k_fam_p5 fills a bucket only next to a cDP1 increment of the same symbol: empty histogram, all outputs 0
            const int max_qual = sscs_phred(P, ref_symbol, s) + (!P.tumor_vcf_is_provided ? 0 : 4);
            int mv, ad, bq;
            infer_max_qual(mv, ad, bq, max_qual, 4, totDP, [&](int b) { return BKP(R, strand, s, b, x); });
            if (mv | ad | bq) { VQP(R, qIAQ, s, x) += mv; VQP(R, qIAD, s, x) += ad; VQP(R, qIDQ, s, x) += bq; }
            // the bucket planes are transient: leave them zero so that the next accumulate need not clear them
            for (int b = 0; b < NBUCKETS; b++) if (BKP(R, strand, s, b, x)) BKP(R, strand, s, b, x) = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// InDel allele tables.  The reference keeps, beside FRAG_bDP / FAM_cDP12.. / FAM_cDP2 / FAM_cDPD + DUPLEX_dDP2, maps
// position -> inserted sequence (or deleted length) -> count (main.hpp:2710-2717, 3327-3336, 3196-3206, 3458-3469, 3535-3546), and
// scoring splits every InDel symbol into its alleles with them (fill_by_indel_info / indel_get_majority, main.hpp:5350-5455).
// Here: k_p2_slow<false> leaves one AlnGap event per I / D op; the events are sorted by (family, position); k_gap_alleles takes
// one (family, position) per thread, redoes the fragment / family / duplex consensus of the LINK symbol type at that position
// and emits one increment per map update; the increments are sorted by (position, symbol, allele) and k_gap_rows sums the runs.
// ------------------------------------------------------------------------------------------------
DEV int base_text_rank(int b) { return b == UVC_BASE_N ? 3 : (b == UVC_BASE_T ? 4 : b); }   // order of the characters "ACGTN": A < C < G < N < T
// order of two alleles of the same symbol: std::string resp. integer key order of the reference's maps
DEV int gap_cmp(const RegionDev &R, const AlnGap &a, const AlnGap &b) {
    if (is_del(a.sym)) return (a.len > b.len) - (a.len < b.len);
    const uint8_t *sa = R.bases + R.alns[a.aln].seq_off + a.qpos, *sb = R.bases + R.alns[b.aln].seq_off + b.qpos;
    const int n = imin(a.len, b.len);
    for (int i = 0; i < n; i++) { const int ra = base_text_rank(sa[i]), rb = base_text_rank(sb[i]); if (ra != rb) return ra < rb ? -1 : 1; }
    return (a.len > b.len) - (a.len < b.len);
}
// 35-bit allele code: exact for deletions and for insertions of up to 13 bases (base-5 digits behind a leading 1 < 5^14 < 2^33), a hash
// with bit 34 set for longer insertions
DEV unsigned long long gap_code(const RegionDev &R, const AlnGap &e) {
    if (is_del(e.sym)) return (unsigned long long)e.len & ((1ull << 33) - 1ull);
    const uint8_t *s = R.bases + R.alns[e.aln].seq_off + e.qpos;
    if (e.len <= 13) { unsigned long long c = 1; for (int i = 0; i < e.len; i++) c = c * 5ull + s[i]; return c; }
    unsigned long long h = 1469598103934665603ull ^ (unsigned long long)e.len;
    for (int i = 0; i < e.len; i++) { h ^= s[i]; h *= 1099511628211ull; }
    return (1ull << 34) | (h >> 30);
}

__global__ void __launch_bounds__(256) k_gap_keys(RegionDev R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R.gap.n_ev) return;
    const AlnGap e = R.gap.ev[i];
    unsigned long long key = ~0ull;
    if (e.sym >= 0) key = ((unsigned long long)R.fss[R.alns[e.aln].fs].fam << 26) | (unsigned long long)(e.epos - R.beg);
    R.gap.ckey[i] = key; R.gap.cval[i] = (unsigned long long)i;
}

// The events of one (family, position) run as k_gap_alleles reads them.  GapEvGlobal: straight from the event array (any run).  GapEvLds: from
// the block's LDS copy, where every lane has loaded one event and worked out its allele code -- the consensus arithmetic below is O(n^2)
// in the events of the run and, read from global memory, a chain of thousands of dependent loads for one family of a 2000x panel (one
// thread's chain was the kernel's whole 2.6 ms).  Runs that leave the block's 128-event window, or hold an insertion longer than 13 bases
// (hashed code: equality needs the sequences), take the global form.
struct GapEvL { int32_t sym, aln, weight, mark, id, pad_; unsigned long long code; };
struct GapEvGlobal {
    const RegionDev &R; AlnGap *ev; const unsigned long long *order; int i0;
    DEV int id(int k) const { return (int)order[i0 + k]; }
    DEV int sym(int k) const { return ev[id(k)].sym; }
    DEV int aln(int k) const { return ev[id(k)].aln; }
    DEV int weight(int k) const { return ev[id(k)].weight; }
    DEV int mark(int k) const { return ev[id(k)].mark; }
    DEV void set_mark(int k, int v) const { ev[id(k)].mark = v; }
    DEV bool same(int a, int b) const { return 0 == gap_cmp(R, ev[id(a)], ev[id(b)]); }
    DEV int cmp(int a, int b) const { return gap_cmp(R, ev[id(a)], ev[id(b)]); }
    DEV unsigned long long code(int k) const { return gap_code(R, ev[id(k)]); }
};
struct GapEvLds {
    const RegionDev &R; AlnGap *ev; GapEvL *L;
    DEV int id(int k) const { return L[k].id; }
    DEV int sym(int k) const { return L[k].sym; }
    DEV int aln(int k) const { return L[k].aln; }
    DEV int weight(int k) const { return L[k].weight; }
    DEV int mark(int k) const { return L[k].mark; }
    DEV void set_mark(int k, int v) const { L[k].mark = v; }
    DEV bool same(int a, int b) const { return L[a].code == L[b].code; }   // (callers compare events of one symbol; exact codes only)
    DEV int cmp(int a, int b) const { return gap_cmp(R, ev[L[a].id], ev[L[b].id]); }   // ties between different alleles: rare
    DEV unsigned long long code(int k) const { return L[k].code; }
};

// one (family, position): events [0, nrun) of E, the first of them at sorted index i0
template <class EV>
DEV void gap_alleles_run(const RegionDev &R, const UvcParams &P, const EV &E, const int i0, const int nrun) {
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    AlnGap *ev = R.gap.ev;
    const int epos = ev[E.id(0)].epos;
    const long long x = epos - R.beg;
    for (int k = 0; k < nrun; k++) E.set_mark(k, 0);
    auto emit = [&](int strand, int level, int sym, int k) {
        const int q = atomicAdd(R.gap.n_inc, 1);
        if (q >= R.gap.inc_cap) { atomicExch(R.err, UVCGPU_EDEVICE); return; }   // cannot happen: the capacity is 7 per event + 8
        R.gap.ikey[q] = ((unsigned long long)x << 38) | ((unsigned long long)(sym - UVC_LINK_D3P) << 35) | E.code(k);
        R.gap.ival[q] = ((unsigned long long)E.id(k) << 8) | (unsigned long long)(strand * 4 + level);
    };
    // the event among `cand(k)` whose allele has the largest total weight; ties go to the larger allele (indelToData_getMajority, main.hpp:50-63)
    auto majority = [&](auto cand, auto weight) -> int {
        int best = -1; long long best_w = 0;
        for (int k = 0; k < nrun; k++) {
            if (!cand(k)) continue;
            long long w = 0;
            for (int j = 0; j < nrun; j++) if (cand(j) && E.same(k, j)) w += weight(j);
            // two steps on purpose: written as one expression `best < 0 || w > best_w || (w == best_w && gap_cmp(...) > 0)` the tie arm was
            // never taken in the code hipcc 7.2 generated at -O3 (found with device printf; tests/test_gpu_indel_alleles.py covers it)
            bool take = (best < 0 || w > best_w);
            if (!take && w == best_w) { const int c = E.cmp(k, best); take = (c > 0); }
            if (take) { best = k; best_w = w; }
        }
        return best;
    };
    const int fs_first = R.alns[E.aln(0)].fs;
    const FsRec u_first = R.fss[fs_first];
    int units[2] = { -1, -1 };
    units[u_first.strand] = fs_first; units[1 - u_first.strand] = u_first.other_fs;
    const bool is_duplex_fam = (0x2 == (u_first.dflag & 0x2));
    const bool will_inc_dscs = is_duplex_fam && units[0] >= 0 && units[1] >= 0;
    const bool will_inc_sscs = is_duplex_fam && !will_inc_dscs;
    int con[2][NSYM], mmm[2][NSYM], cnt[NSYM];
    for (int s = 0; s < 2; s++) for (int k = 0; k < NSYM; k++) { con[s][k] = 0; mmm[s][k] = 0; }
    R.gap.maj[2 * i0] = 0; R.gap.maj[2 * i0 + 1] = 0;
    // fragments: P3 consensus (main.hpp:2650-2717) and their votes for the unit (updateByFiltering / updateByMajorMinusMinor, main.hpp:1659-1725)
    for (int s = 0; s < 2; s++) {
        if (units[s] < 0) continue;
        const FsRec u = R.fss[units[s]];
        for (int fi = u.frag_beg; fi < u.frag_end; fi++) {
            const FragRec f = R.frags[fi];
            if (epos < f.beg || epos >= f.end) continue;
            frag_counts(R, P, f, epos, proton, cnt);
            int cs, cc, ct;
            fill_consensus(cnt, cs, cc, ct, UVC_LINK_SYMBOL, true, false);
            if (0 == ct) continue;
            const int adj = imax(cc * 2, ct) - ct;
            if (adj > 0) { con[s][cs] += 1; mmm[s][cs] += adj; }
            if (!(is_ins(cs) || is_del(cs))) continue;
            const int e = majority([&](int q) { return E.sym(q) == cs && E.aln(q) >= f.aln_beg && E.aln(q) < f.aln_end; }, [&](int q) { return (long long)E.weight(q); });
            if (e < 0) { atomicExch(R.err, UVCGPU_EDEVICE); continue; }   // an InDel consensus without an InDel event cannot happen
            E.set_mark(e, 0x10000 | (s << 8) | cs);
            emit(s, 0, cs, e);
        }
    }
    // family x strand units: P4 (main.hpp:3180-3336) and P5 (main.hpp:3448-3469)
    int dup[NSYM], dup_allele[2] = { -1, -1 }, dup_sym[2] = { -1, -1 };
    for (int k = 0; k < NSYM; k++) dup[k] = 0;
    for (int s = 0; s < 2; s++) {
        if (units[s] < 0) continue;
        const FsRec u = R.fss[units[s]];
        auto unit_allele = [&](int sym) { return majority([&](int q) { return E.mark(q) == (0x10000 | (s << 8) | sym); }, [&](int q) { return 1LL; }); };
        {   // the count read_family_con_ampl_getMajority_ins returns: the inserted sequence most fragments of the unit agree on, over all three
            // insertion symbols (main.hpp:188-198); the FAM2 position-bias test of the family kernels reads it (main.hpp:3239-3246)
            // (the map is fed by every fragment whose consensus is an insertion symbol, main.hpp:1670-1676: the marked events of the unit)
            int m = 0;
            for (int k = 0; k < nrun; k++) {
                const int mk = E.mark(k);
                if ((mk >> 8) != (0x100 | s) || !is_ins(mk & 0xFF)) continue;
                int w = 0;
                for (int j = 0; j < nrun; j++) if (E.mark(j) == mk && E.same(k, j)) w++;
                m = imax(m, w);
            }
            R.gap.maj[2 * i0 + s] = m;
        }
        int cs, cc, ct;
        fill_consensus(con[s], cs, cc, ct, UVC_LINK_SYMBOL, false, false);
        if (ct > 0 && (is_ins(cs) || is_del(cs))) {
            const int e = unit_allele(cs);
            if (e < 0) atomicExch(R.err, UVCGPU_EDEVICE);
            else {
                const bool is_fam_good = ((P.fam_thres_dup1add <= ct) && (cc * 100 >= ct * P.fam_thres_dup1perc) && ((u.dflag & 0x1) || (P.fam_flag & 0x2)));
                emit(s, 1, cs, e);
                if (is_fam_good) emit(s, 2, cs, e);
            }
        }
        if (will_inc_dscs) {   // updateByFiltering<true, false, false>(con, {1, 1}), main.hpp:3429-3432
            const int adj = imax(cc * 2, ct) - ct;
            if (adj >= 1) dup[cs] += 1;
            if (is_ins(cs) || is_del(cs)) { dup_allele[s] = unit_allele(cs); dup_sym[s] = cs; }
        }
        int c5, cc5, ct5;
        fill_consensus(mmm[s], c5, cc5, ct5, UVC_LINK_SYMBOL, false, false);
        if (ct5 > 0 && (is_ins(c5) || is_del(c5))) {
            const int con_nfrags = con[s][c5];
            int tot_nfrags = 0;
            for (int k = UVC_LINK_M; k <= UVC_LINK_NN; k++) tot_nfrags += con[s][k];
            if (will_inc_sscs && (!will_inc_dscs) && (tot_nfrags >= P.fam_thres_dup1add) && (con_nfrags * 100 >= tot_nfrags * P.fam_thres_dup1perc)) {
                const int e = unit_allele(c5);
                if (e < 0) atomicExch(R.err, UVCGPU_EDEVICE); else emit(s, 3, c5, e);
            }
        }
    }
    if (will_inc_dscs) {   // main.hpp:3523-3548
        int cs, cc, ct;
        fill_consensus(dup, cs, cc, ct, UVC_LINK_SYMBOL, false, false);
        if (1 < ct && (is_ins(cs) || is_del(cs))) {
            int e = -1;
            const bool h0 = (dup_sym[0] == cs && dup_allele[0] >= 0), h1 = (dup_sym[1] == cs && dup_allele[1] >= 0);
            if (h0 && h1) e = (E.cmp(dup_allele[0], dup_allele[1]) >= 0 ? dup_allele[0] : dup_allele[1]);
            else if (h0) e = dup_allele[0];
            else if (h1) e = dup_allele[1];
            if (e < 0) atomicExch(R.err, UVCGPU_EDEVICE);
            else { emit(0, 3, cs, e); emit(1, 3, cs, e); }
        }
    }
}

#define GAP_WIN 128
__global__ void __launch_bounds__(64) k_gap_alleles(RegionDev R, UvcParams P) {
    __shared__ GapEvL arena[GAP_WIN];
    const int n = R.gap.n_ev, b0 = blockIdx.x * 64, lane = threadIdx.x;
    const unsigned long long *order = R.gap.cval_s;
    AlnGap *ev = R.gap.ev;
    // every lane brings two events of the block's window into LDS, with their allele codes
    for (int h = 0; h < GAP_WIN / 64; h++) {
        const int i = b0 + h * 64 + lane;
        GapEvL g; g.sym = -1; g.aln = 0; g.weight = 0; g.mark = 0; g.id = 0; g.pad_ = 0; g.code = 0;
        if (i < n && R.gap.ckey_s[i] != ~0ull) {
            const int e = (int)order[i];
            const AlnGap a = ev[e];
            g.sym = a.sym; g.aln = a.aln; g.weight = a.weight; g.id = e; g.code = gap_code(R, a);
        }
        arena[h * 64 + lane] = g;
    }
    __syncthreads();
    const int i0 = b0 + lane;
    if (i0 >= n) return;
    const unsigned long long key = R.gap.ckey_s[i0];
    if (key == ~0ull || (i0 > 0 && R.gap.ckey_s[i0 - 1] == key)) return;   // one thread per (family, position): the head of the run
    int i1 = i0 + 1;
    while (i1 < n && R.gap.ckey_s[i1] == key) i1++;
    bool in_lds = (i1 - b0 <= GAP_WIN);
    for (int i = i0; in_lds && i < i1; i++) if ((arena[i - b0].code >> 34) & 1ull) in_lds = false;   // (an insertion of more than 13 bases)
    if (in_lds) { const GapEvLds E{ R, ev, arena + lane }; gap_alleles_run(R, P, E, i0, i1 - i0); }
    else { const GapEvGlobal E{ R, ev, order, i0 }; gap_alleles_run(R, P, E, i0, i1 - i0); }
}

// The sorted increments -> one GapRow per (position, symbol, allele).  At 2000x a site's allele has thousands of increments, so a run is
// not summed by one thread: k_gap_rows opens a row per run (the head of the run), k_gap_rows_add lets every increment find its run's head
// (binary search in the sorted keys) and add itself to the row's counters, k_gap_rows_fin fills in the allele of each row from its
// representative event (the smallest event index of the run).  `run_row` (the unsorted key array, free once the sort has run) holds the
// row of each run at the index of its head.
__global__ void __launch_bounds__(256) k_gap_rows(RegionDev R) {
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = imin(*R.gap.n_inc, R.gap.inc_cap);
    if (i0 >= n) return;
    const unsigned long long key = R.gap.ikey_s[i0];
    if (i0 > 0 && R.gap.ikey_s[i0 - 1] == key) return;
    int32_t *run_row = (int32_t *)R.gap.ikey;
    const bool hashed = ((key >> 34) & 1ull) != 0;
    if (!hashed) {   // exact codes: the whole run is one row
        const int r = atomicAdd(R.gap.n_rows, 1);
        GapRow row;
        for (int k = 0; k < 8; k++) row.cnt[k] = 0;
        row.x = (int)(key >> 38); row.sym = -1; row.len = 0; row.ev = INT32_MAX; row.seq_off = -1;
        R.gap.rows[r] = row;
        run_row[i0] = r;
        return;
    }
    // The allele code of an insertion longer than 13 bases is a hash: equal keys need not be equal sequences.  Such a run is split by
    // comparing the sequences themselves (one pass when they all agree, which is the case unless two long insertions of one site collide
    // in 34 bits); bit 63 of a value marks an increment that has found its row.  Long insertions are rare: this thread does the run alone.
    run_row[i0] = -1;
    int i1 = i0;
    while (i1 < n && R.gap.ikey_s[i1] == key) i1++;
    for (int ib = i0; ib < i1; ib++) {
        if (R.gap.ival_s[ib] >> 63) continue;
        const AlnGap first = R.gap.ev[(int)((R.gap.ival_s[ib] & ~(1ull << 63)) >> 8)];
        GapRow row;
        for (int k = 0; k < 8; k++) row.cnt[k] = 0;
        int rep = INT32_MAX;
        for (int i = ib; i < i1; i++) {
            const unsigned long long v = R.gap.ival_s[i];
            if (v >> 63) continue;
            if (i != ib && 0 != gap_cmp(R, first, R.gap.ev[(int)(v >> 8)])) continue;   // another sequence under the same hash: a later row
            row.cnt[v & 0xFF] += 1;
            rep = imin(rep, (int)(v >> 8));
            R.gap.ival_s[i] = v | (1ull << 63);
        }
        row.x = (int)(key >> 38); row.sym = -1; row.len = 0; row.ev = rep; row.seq_off = -1;
        R.gap.rows[atomicAdd(R.gap.n_rows, 1)] = row;
    }
}
__global__ void __launch_bounds__(256) k_gap_rows_add(RegionDev R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = imin(*R.gap.n_inc, R.gap.inc_cap);
    if (i >= n) return;
    const unsigned long long key = R.gap.ikey_s[i];
    if ((key >> 34) & 1ull) return;   // hashed alleles: done by the head of the run
    int lo = 0, hi = i;   // first index with this key
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (R.gap.ikey_s[mid] < key) lo = mid + 1; else hi = mid; }
    const int r = ((const int32_t *)R.gap.ikey)[lo];
    const unsigned long long v = R.gap.ival_s[i];
    atomicAdd(&R.gap.rows[r].cnt[v & 0xFF], 1);
    atomicMin(&R.gap.rows[r].ev, (int)(v >> 8));
}
__global__ void __launch_bounds__(256) k_gap_rows_fin(RegionDev R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= *R.gap.n_rows) return;
    GapRow &row = R.gap.rows[r];
    const AlnGap e = R.gap.ev[row.ev];
    row.sym = e.sym; row.len = e.len;
    if (is_ins(e.sym)) {
        const long long off = (long long)atomicAdd(R.gap.seq_len, (unsigned long long)e.len);
        if (off + e.len <= R.gap.seq_cap) {
            const uint8_t *src = R.bases + R.alns[e.aln].seq_off + e.qpos;
            for (int k = 0; k < e.len; k++) R.gap.seq[off + k] = src[k];
            row.seq_off = off;
        } else atomicExch(R.err, UVCGPU_EDEVICE);
    }
}

// ------------------------------------------------------------------------------------------------
// Haplotype links.  The reference strings together, per fragment (P3) and per family-strand unit (P5), the positions whose
// consensus symbol is a high-quality mutation (main.hpp:2720-2737, 3490-3521) and counts equal strings per strand
// (mutform2count4map_bq / _fq / _f2q); updateHapMap (main.hpp:3596-3663) turns the maps into the links FORMAT/bHap, cHap, c2Hap print.
// Only objects with at least two such positions matter, and an object cannot have more of them than its alignments have mismatching
// bases + InDel ops (AlnRec::n_mutc).  So: pick the candidates, reserve their slots, then one wave per candidate re-derives its consensus
// along its span (the generic per-position forms: this is O(candidates), off the accumulate path, run when the links are asked for).
// The maps, the sort and the link arithmetic are host work on a few thousand short lists (uvc_hap.cpp).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_hap_cand(RegionDev R, HapWork H, int units) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = units ? R.n_fs : R.n_frags;
    if (i >= n) return;
    int a0, a1;
    if (units) { const FsRec &u = R.fss[i]; a0 = R.frags[u.frag_beg].aln_beg; a1 = R.frags[u.frag_end - 1].aln_end; }
    else { a0 = R.frags[i].aln_beg; a1 = R.frags[i].aln_end; }
    int bound = 0;
    for (int k = a0; k < a1; k++) bound += R.alns[k].n_mutc;
    if (bound < 2) return;
    const int span = units ? (R.fss[i].end - R.fss[i].beg) : (R.frags[i].end - R.frags[i].beg);
    bound = imin(bound, 2 * span);
    const int slot = atomicAdd(H.n_cand, 1);
    H.cand[slot] = i; H.cand_cap[slot] = bound;
    H.cand_off[slot] = (int)atomicAdd(H.total, (unsigned long long)((units ? 2 : 1) * (bound + 2)));
}
// appends this lane's (up to two) events in lane order behind `count` events already written; LINK before BASE at one position
DEV int hap_append(int32_t *dst, int cap, int count, bool evL, int vL, bool evB, int vB, int lane, int32_t *err) {
    const unsigned long long mL = __ballot(evL), mB = __ballot(evB), below = (lane == 0 ? 0ull : (~0ull >> (64 - lane)));
    const int at = count + __popcll(mL & below) + __popcll(mB & below);
    if (evL) { if (at < cap) dst[2 + at] = vL; else atomicExch(err, UVCGPU_EDEVICE); }
    if (evB) { const int a2 = at + (evL ? 1 : 0); if (a2 < cap) dst[2 + a2] = vB; else atomicExch(err, UVCGPU_EDEVICE); }
    return count + __popcll(mL) + __popcll(mB);
}
__global__ void __launch_bounds__(64) k_hap_frags(RegionDev R, UvcParams P, HapWork H, int n_cand) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= n_cand) return;
    const int fi = H.cand[t];
    const FragRec &f = R.frags[fi];
    int32_t *dst = H.events + H.cand_off[t];
    const int cap = H.cand_cap[t];
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    int count = 0;
    int cnt[NSYM];
    for (int i0 = 0; i0 < f.end - f.beg; i0 += 64) {
        const int p = f.beg + i0 + lane;
        bool ev[2] = { false, false }; int val[2] = { 0, 0 };
        if (p < f.end) {
            frag_counts(R, P, f, p, proton, cnt);
            const int refsymbol = R.refsym[p - R.beg];
            for (int vi = 0; vi < 2; vi++) {   // SYMBOL_TYPES_IN_VCF_ORDER: LINK, then BASE
                const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
                int cs, cc, ct;
                fill_consensus(cnt, cs, cc, ct, st, st == UVC_LINK_SYMBOL, false);
                if (0 == ct) continue;
                const int con_qual = cc * 2 - ct;
                const bool highBQ = (proton ? (UVC_BASE_SYMBOL == st || con_qual + 3 >= P.bias_thres_highBQ) : (UVC_LINK_SYMBOL == st || con_qual >= P.bias_thres_highBQ));
                if (symbols_mutated(refsymbol, cs) && highBQ) { ev[vi] = true; val[vi] = ((p - R.beg) << 4) | cs; }
            }
        }
        count = hap_append(dst, cap, count, ev[0], val[0], ev[1], val[1], lane, R.err);
    }
    if (lane == 0) { dst[0] = f.strand | (cap << 8); dst[1] = imin(count, cap); }   // kind 0 (bq); the slot length lets the host walk the buffer
}
__global__ void __launch_bounds__(64) k_hap_units(RegionDev R, UvcParams P, HapWork H, int n_cand) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= n_cand) return;
    const FsRec u = R.fss[H.cand[t]];
    const int cap = H.cand_cap[t];
    int32_t *dst1 = H.events + H.cand_off[t], *dst2 = dst1 + cap + 2;   // fq, f2q
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    const int strand = u.strand;
    __shared__ int con_s[NSYM][64], mmm_s[NSYM][64];
    const LdsCounts<64> con = { &con_s[0][lane] }, mmm = { &mmm_s[0][lane] };
    int count1 = 0, count2 = 0;
    for (int i0 = 0; i0 < u.end - u.beg; i0 += 64) {
        const int p = u.beg + i0 + lane;
        bool ev1[2] = { false, false }, ev2[2] = { false, false }; int val[2] = { 0, 0 };
        if (p < u.end) {
            const int64_t x = p - R.beg;
            unit_counts<true>(R, P, u, p, proton, con, mmm);
            for (int vi = 0; vi < 2; vi++) {
                const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
                int cs, con_sumBQs, tot_sumBQs;
                fill_consensus(mmm, cs, con_sumBQs, tot_sumBQs, st, false, false);
                if (0 == tot_sumBQs) continue;
                const int con_nfrags = con[cs];
                int tot_nfrags = 0;
                const int sb = (st == 0 ? UVC_BASE_A : UVC_LINK_M), se = (st == 0 ? UVC_BASE_NN : UVC_LINK_NN);
                for (int s2 = sb; s2 <= se; s2++) tot_nfrags += con[s2];
                // the empirical family quality of P5 (main.hpp:3472-3488), as in k_fam_p5
                const int avgBQ = ((0 == tot_nfrags) ? 1 : (con_sumBQs / tot_nfrags));
                const int majorcount = FAP(R, strand, UVC_FAM_cDPM, cs, x), minorcount = FAP(R, strand, UVC_FAM_cDPm, cs, x);
                const double prior_weight = 1.0 / (minorcount + 1.0);
                const double p2p = pow(10.0, (double)(-((float)avgBQ) / 10));
                const double prob = (minorcount + prior_weight) / (majorcount + minorcount + prior_weight / p2p);
                const double realphred = -10 * log(prob) / log(10.0);
                const int indep_frag_phred = (int)round(((con_nfrags * 2) - tot_nfrags) * realphred);
                int confam_qual;
                if (UVC_LINK_SYMBOL == st) confam_qual = imax(1, imin(indep_frag_phred, P.fam_phred_indel_inc_before_barcode_labeling + (int)round(realphred)));
                else confam_qual = imax(1, imin(indep_frag_phred, (con_sumBQs * 2) - tot_sumBQs));
                const bool highBQ = (proton ? (UVC_BASE_SYMBOL == st || imax(confam_qual + 3, avgBQ) >= P.bias_thres_highBQ) : (UVC_LINK_SYMBOL == st || confam_qual >= P.bias_thres_highBQ));
                if (symbols_mutated(R.refsym[x], cs) && highBQ) {
                    ev1[vi] = true; val[vi] = ((int)x << 4) | cs;
                    int cs1, cc1, ct1;
                    fill_consensus(con, cs1, cc1, ct1, st, false, false);
                    if (cs == cs1 && P.fam_thres_dup1add <= ct1 && (cc1 * 100 >= ct1 * P.fam_thres_dup1perc)) ev2[vi] = true;
                }
            }
        }
        count1 = hap_append(dst1, cap, count1, ev1[0], val[0], ev1[1], val[1], lane, R.err);
        count2 = hap_append(dst2, cap, count2, ev2[0], val[0], ev2[1], val[1], lane, R.err);
    }
    if (lane == 0) { dst1[0] = strand | (1 << 1) | (cap << 8); dst1[1] = imin(count1, cap); dst2[0] = strand | (2 << 1) | (cap << 8); dst2[1] = imin(count2, cap); }
}
extern "C" void uvc_launch_hap_cand(const RegionDev *R, const HapWork *H, int units, hipStream_t s) {
    const int n = units ? R->n_fs : R->n_frags;
    if (n > 0) hipLaunchKernelGGL(k_hap_cand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *R, *H, units);
}
extern "C" void uvc_launch_hap_events(const RegionDev *R, const UvcParams *P, const HapWork *H, int units, int n_cand, hipStream_t s) {
    if (n_cand <= 0) return;
    if (units) hipLaunchKernelGGL(k_hap_units, dim3((unsigned)n_cand), dim3(64), 0, s, *R, *P, *H, n_cand);
    else hipLaunchKernelGGL(k_hap_frags, dim3((unsigned)n_cand), dim3(64), 0, s, *R, *P, *H, n_cand);
}

extern "C" int uvc_gap_sort(void *tmp, size_t tmp_bytes, const unsigned long long *kin, unsigned long long *kout, const unsigned long long *vin, unsigned long long *vout,
                            size_t n, int end_bit, hipStream_t s);
static inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }
static inline int imin_h(int a, int b) { return a < b ? a : b; }

// optional per-kernel HIP-event timing on the handle's own stream (bench.py roofline leg)
struct UvcProf { int on; int n; const char *name[32]; hipEvent_t ev[32][2]; };
#define TIMED(prof, kname, ...) do { \
        UvcProf *p_ = (prof); int i_ = -1; \
        if (p_ && p_->on && p_->n < 32) { i_ = p_->n++; p_->name[i_] = kname; if (!p_->ev[i_][0]) { hipEventCreate(&p_->ev[i_][0]); hipEventCreate(&p_->ev[i_][1]); } hipEventRecord(p_->ev[i_][0], s); } \
        __VA_ARGS__; \
        if (i_ >= 0) hipEventRecord(p_->ev[i_][1], s); \
    } while (0)

// apply_bq_err_correction3 (grouping.cpp:459-543): quality increment and cap, tail penalty behind a long soft clip / homopolymer at
// the 3' end, and the poly-G penalty.  One thread per alignment; the base codes are mapped to the BAM 4-bit codes the reference
// compares (A=1, C=2, G=4, T=8, N=15; "no base yet" = 0).
__global__ void __launch_bounds__(256) k_correct_bq(RegionDev R, int bq_max, int bq_inc) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= R.n_alns) return;
    const AlnRec &a = R.alns[id];
    const int l = a.l_qseq;
    if (0 == l || (a.flag & 0x4)) return;
    uint8_t *q = (uint8_t *)R.quals + a.seq_off;
    const uint8_t *b = R.bases + a.seq_off;
    const uint32_t *cigar = R.cigars + a.cigar_off;
    auto code = [&](int i) -> int { const int v = b[i]; return v < 4 ? (1 << v) : 15; };
    for (int i = 0; i < l; i++) q[i] = (uint8_t)imin((int)q[i] + bq_inc, bq_max);
    const int isrc = ((a.flag & 0x10) ? 1 : 0);
    int inclu_beg_poss[2] = { 0, l - 1 };
    int exclu_end_poss[2] = { l, 0 - 1 };
    int end_clip_len = 0;
    if (a.n_cigar > 0) {
        uint32_t c1 = cigar[0];
        if (cig_op(c1) == C_SOFT_CLIP) {
            if (0 == isrc) inclu_beg_poss[0] += cig_len(c1);
            else { exclu_end_poss[1] += cig_len(c1); end_clip_len = cig_len(c1); }
        }
        c1 = cigar[a.n_cigar - 1];
        if (cig_op(c1) == C_SOFT_CLIP) {
            if (1 == isrc) inclu_beg_poss[1] -= cig_len(c1);
            else { exclu_end_poss[0] -= cig_len(c1); end_clip_len = cig_len(c1); }
        }
    }
    const int inc = (isrc ? -1 : 1), ibeg = inclu_beg_poss[isrc], eend = exclu_end_poss[isrc];
    // a read whose soft clips leave no aligned base would send the reference's loops out of the arrays (undefined there): skip it
    if ((isrc ? (ibeg <= eend) : (ibeg >= eend)) || ibeg < 0 || ibeg >= l || eend < -1 || eend > l) return;
    {
        int prev_b = 0, distinct_cnt = 0;
        int termpos = eend - inc;
        for (; termpos != ibeg - inc; termpos -= inc) {
            const int bb = code(termpos), qq = q[termpos];
            if (bb != prev_b && qq >= 20) { prev_b = bb; distinct_cnt += 1; if (2 == distinct_cnt) break; }
        }
        const int homopol_tracklen = abs(termpos - (eend - inc));
        const int tail_penal = (end_clip_len >= 20 ? 1 : 0) + (homopol_tracklen >= 15 ? 2 : (homopol_tracklen >= 10 ? 1 : 0));
        if (tail_penal > 0)
            for (int pos = eend - inc; pos != (ibeg - inc) && pos != termpos; pos -= inc) q[pos] = (uint8_t)(imax((int)q[pos], tail_penal + 1) - tail_penal);
    }
    {
        int homopol_len = 0, prev_b = 0;
        for (int pos = ibeg; pos != eend; pos += inc) {
            const int bb = code(pos);
            if (bb == prev_b) { homopol_len++; if (homopol_len >= 4 && bb == 4 /* G */) q[pos] = (uint8_t)(imax((int)q[pos], 1 + 1) - 1); }
            else { prev_b = bb; homopol_len = 1; }
        }
    }
}
extern "C" void uvc_launch_correct_bq(const RegionDev *R, int bq_max, int bq_inc, hipStream_t s) {
    if (R->n_alns) hipLaunchKernelGGL(k_correct_bq, dim3((unsigned)((R->n_alns + 255) / 256)), dim3(256), 0, s, *R, bq_max, bq_inc);
}

// (a base code above 4 is refused here, where the one-byte-per-base column enters: k_aln_bm compares eight bases per load and relies on every
// byte being below 0x80, the per-symbol tables index with the code)
__global__ void __launch_bounds__(256) k_pack_bq1(const uint8_t *bases, const uint8_t *quals, uint16_t *bq, int64_t n, int32_t *bad) {
    int any = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        bq[i] = (uint16_t)(bases[i] | (quals[i] << 8));
        any |= (bases[i] > 4);
    }
    if (bad && __any(any) && (threadIdx.x & 63) == 0) *bad = 2;
}
// eight read bases per thread: two 8-byte loads, one 16-byte store (a byte per thread ran at 1.7 TB/s of the 1.2 GB it moves)
__global__ void __launch_bounds__(256) k_pack_bq(const uint8_t *bases, const uint8_t *quals, uint16_t *bq, int64_t n, int32_t *bad) {
    const int64_t n8 = n >> 3;
    unsigned long long over = 0;   // bits 3..7 of any byte set <=> a code above 7; codes 5..7: bit 2 with bit 0 or 1
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long b = ((const unsigned long long *)bases)[i], q = ((const unsigned long long *)quals)[i];
        over |= (b & 0xF8F8F8F8F8F8F8F8ull) | (b & 0x0404040404040404ull & (((b | (b >> 1)) & 0x0101010101010101ull) << 2));
        uint4 o;
        o.x = (uint32_t)((b & 0xFF) | ((q & 0xFF) << 8) | (((b >> 8) & 0xFF) << 16) | (((q >> 8) & 0xFF) << 24));
        o.y = (uint32_t)(((b >> 16) & 0xFF) | (((q >> 16) & 0xFF) << 8) | (((b >> 24) & 0xFF) << 16) | (((q >> 24) & 0xFF) << 24));
        o.z = (uint32_t)(((b >> 32) & 0xFF) | (((q >> 32) & 0xFF) << 8) | (((b >> 40) & 0xFF) << 16) | (((q >> 40) & 0xFF) << 24));
        o.w = (uint32_t)(((b >> 48) & 0xFF) | (((q >> 48) & 0xFF) << 8) | (((b >> 56) & 0xFF) << 16) | (((q >> 56) & 0xFF) << 24));
        ((uint4 *)bq)[i] = o;
    }
    if (blockIdx.x == 0) for (int64_t i = (n8 << 3) + threadIdx.x; i < n; i += blockDim.x) { bq[i] = (uint16_t)(bases[i] | (quals[i] << 8)); over |= (bases[i] > 4); }   // the last n % 8
    if (bad && __any(over != 0) && (threadIdx.x & 63) == 0) *bad = 2;
}
extern "C" void uvc_launch_pack_bq(const uint8_t *bases, const uint8_t *quals, uint16_t *bq, int64_t n, int32_t *bad, hipStream_t s) {
    if (n <= 0) return;
    // the wide form needs 8- / 16-byte aligned columns (device allocations are; a caller's sub-array of set_reads_device may not be)
    const bool aligned = !(((uintptr_t)bases | (uintptr_t)quals) & 7) && !((uintptr_t)bq & 15);
    if (aligned) hipLaunchKernelGGL(k_pack_bq, dim3((unsigned)std::min<int64_t>(((n >> 3) + 255) / 256 + 1, 16384)), dim3(256), 0, s, bases, quals, bq, n, bad);
    else hipLaunchKernelGGL(k_pack_bq1, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65536)), dim3(256), 0, s, bases, quals, bq, n, bad);
}
extern "C" void uvc_launch_build_p2list(const RegionDev *R, const int32_t *fast_rank, const int32_t *aln, const int32_t *cbeg, const int32_t *cend, const int32_t *qb, hipStream_t s) {
    if (R->n_fast2) hipLaunchKernelGGL(k_build_p2list, dim3(nblk(R->n_fast2, 256)), dim3(256), 0, s, *R, fast_rank, aln, cbeg, cend, qb);
}
extern "C" void uvc_launch_prelude(const RegionDev *R, const RawReads *W, const UvcParams *P, hipStream_t s) {
    if (R->n_alns && R->n_fast) hipLaunchKernelGGL(k_aln_bm, dim3(imin_h((int)nblk(R->n_alns, 256), 16384)), dim3(256), 0, s, *R, *W);
    if (R->n_alns) hipLaunchKernelGGL(k_aln_prelude, dim3(nblk(R->n_alns, 256)), dim3(256), 0, s, *R, *W, *P);
}
// The sequential, low-occupancy kernels of the InDel reads and the per-fragment statistics need P1 / P1b only, not P2: they run on
// a side stream underneath the two issue-bound P2 kernels and join before anything reads their outputs.
#define TIMED2(prof, kname, ...) do { hipStream_t s = s2; TIMED(prof, kname, __VA_ARGS__); } while (0)
#define TIMED3(prof, kname, ...) do { hipStream_t s = s3; TIMED(prof, kname, __VA_ARGS__); } while (0)
// Zero fill of the plane slab in front of an accumulate, without the parts the last accumulate did not write (RegionDev::dirty): a block
// per (plane, block of 4 096 positions).  A 1 Mb tile's slab is 5.6 KB per position; what every tile writes (the per-position planes, the
// planes of A C G T N and LINK_M) is 2.5 KB of it, the rest a few per cent.
struct ZeroPlane { unsigned long long off; int32_t elem; int16_t fam, sym; };   // fam < 0: always filled
__global__ void __launch_bounds__(256) k_zero_state(char *slab, const ZeroPlane *planes, const uint8_t *dirty, int ndblk, int64_t npos) {
    const ZeroPlane zp = planes[blockIdx.x];
    const int b = (int)blockIdx.y;
    if (zp.fam >= 0 && !dirty[((size_t)zp.fam * NSYM + zp.sym) * ndblk + b]) return;
    const int64_t x0 = (int64_t)b << UVC_DIRTY_SHIFT, x1 = (x0 + ((int64_t)1 << UVC_DIRTY_SHIFT) < npos ? x0 + ((int64_t)1 << UVC_DIRTY_SHIFT) : npos);
    char *p0 = slab + zp.off + x0 * zp.elem, *p1 = slab + zp.off + x1 * zp.elem;
    char *a0 = (char *)(((uintptr_t)p0 + 15) & ~(uintptr_t)15), *a1 = (char *)((uintptr_t)p1 & ~(uintptr_t)15);
    if (a0 >= a1) { for (char *q = p0 + threadIdx.x; q < p1; q += 256) *q = 0; return; }
    for (char *q = p0 + threadIdx.x; q < a0; q += 256) *q = 0;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (uint4 *q = (uint4 *)a0 + threadIdx.x; q < (uint4 *)a1; q += 256) *q = z;
    for (char *q = a1 + threadIdx.x; q < p1; q += 256) *q = 0;
}
extern "C" void uvc_launch_zero_state(char *slab, const void *planes, int n_planes, uint8_t *dirty, int ndblk, int64_t npos, hipStream_t s) {
    if (n_planes <= 0 || ndblk <= 0) return;
    hipLaunchKernelGGL(k_zero_state, dim3((unsigned)n_planes, (unsigned)ndblk), dim3(256), 0, s, slab, (const ZeroPlane *)planes, dirty, ndblk, npos);
}
// uvcgpu_region_check_presence, second half: a (family, symbol, block) that is not marked holds only zeros
__global__ void __launch_bounds__(256) k_check_dirty(RegionDev R, unsigned long long *n_bad) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = (int)blockIdx.y;
    if (x >= R.npos) return;
    const int b = (int)(x >> UVC_DIRTY_SHIFT);
    long long core = 0, fi = 0, du = 0;
    for (int f = 0; f < UVC_NSEG32; f++) core |= S32(R, f, s, x);
    for (int f = 0; f < UVC_NSEG64; f++) core |= S64(R, f, s, x);
    for (int f = 0; f < UVC_NVQ; f++) core |= VQP(R, f, s, x);
    core |= BQS(R, s, x);
    for (int st = 0; st < 2; st++) { for (int f = 0; f < UVC_NFRAG; f++) core |= FRP(R, st, f, s, x); for (int f = 0; f < UVC_NFAM; f++) core |= FAP(R, st, f, s, x); }
    for (int f = 0; f < UVC_NFAMINFO32; f++) fi |= FIP(R, f, s, x);
    for (int f = 0; f < UVC_NFAMINFO64; f++) fi |= FI64P(R, f, s, x);
    for (int f = 0; f < UVC_NDUPLEX; f++) du |= DUP(R, f, s, x);
    int bad = 0;
    if (core && !sym_always_filled(s) && !R.dirty[(size_t)s * R.ndblk + b]) bad++;
    if (fi && !R.dirty[((size_t)NSYM + s) * R.ndblk + b]) bad++;
    if (du && !R.dirty[((size_t)2 * NSYM + s) * R.ndblk + b]) bad++;
    if (bad) atomicAdd(n_bad, (unsigned long long)bad);
}
extern "C" void uvc_launch_check_dirty(const RegionDev *R, unsigned long long *d_n_bad, hipStream_t s) {
    hipLaunchKernelGGL(k_check_dirty, dim3((unsigned)((R->npos + 255) / 256), NSYM), dim3(256), 0, s, *R, d_n_bad);
}

extern "C" void uvc_launch_accumulate(const RegionDev *R, const UvcParams *P, int half_ratio_phred,
                                      const int32_t *dup_units, int n_dup, const int64_t *dup_off, int64_t n_dup_work, hipStream_t s, UvcProf *prof,
                                      hipStream_t side, hipEvent_t e_fork, hipEvent_t e_join, hipEvent_t e_fork2, hipStream_t side3, hipEvent_t e_join3, hipEvent_t e_stat, hipEvent_t e_alleles) {
    const unsigned nwin = nblk(R->npos, 256);   // 4 waves x 64 positions per block
    // fewer windows than four per SIMD, and many reads on each (at least sixteen chunks of 64: at 300x a wave of the split form would get fewer
    // than two and pay its prologue for them -- twice as slow on the 1 Mb tile): a block per window in the kernels that can share a window's
    // reads among its waves (UVCGPU_SPLIT=0 / 1 forces)
    const char *sp_env = getenv("UVCGPU_SPLIT");
    const long long per_window = (long long)R->n_fast2 * (R->max_p2_span + 64) / (R->npos > 0 ? R->npos : 1);
    const bool split_windows = sp_env ? (atoi(sp_env) != 0) : (R->nwin < 4 * 1024 && per_window >= 1024);
    if (prof) prof->n = 0;
    hipStream_t s2 = (side ? side : s);
    hipStream_t s3 = ((side && side3) ? side3 : s2);   // the two CIGAR walks of the InDel reads are independent, one wave per 64 reads and long: a stream each
    {   // the window index of the alignment lists (the fragment list's follows k_fragstat_fast, which writes it)
        const int64_t n = 2 * (int64_t)R->nwin * 5;
        TIMED(prof, "k_win_index", hipLaunchKernelGGL(k_win_index, dim3(nblk(n, 256)), dim3(256), 0, s, *R, 0, 5));
        if (R->n_generic_fs) hipLaunchKernelGGL(k_win_index, dim3(nblk(2 * (int64_t)R->nwin, 256)), dim3(256), 0, s, *R, 7, 8);
    }
    if (P->inferred_is_vcf_generated && R->n_fast) TIMED(prof, "k_prep_sums", hipLaunchKernelGGL(k_prep_sums, dim3(nblk(R->npos, PSUM_TILE)), dim3(256), 0, s, *R, *P));
    if (P->inferred_is_vcf_generated) {
        // P1: the InDel reads (one wave per read) underneath the position-centric pass
        if (side && R->n_complex) { hipEventRecord(e_fork, s); hipStreamWaitEvent(s2, e_fork, 0); }
        if (R->n_complex) TIMED2(prof, "k_prep_slow", hipLaunchKernelGGL(k_prep_slow, dim3(R->n_complex), dim3(64), 0, s, *R, *P));
        if (side && R->n_complex) hipEventRecord(e_join, s2);
        if (split_windows) TIMED(prof, "k_prep_fast", hipLaunchKernelGGL(k_prep_fast<true>, dim3(R->nwin), dim3(256), 0, s, *R, *P));
        else TIMED(prof, "k_prep_fast", hipLaunchKernelGGL(k_prep_fast<false>, dim3(nwin), dim3(256), 0, s, *R, *P));
        if (side && R->n_complex) hipStreamWaitEvent(s, e_join, 0);
    }
    // P1b is part of updateByAlns3UsingBQ too (main.hpp:3691-3702): on a FASTQ-only run the thresholds stay zero and rtr.indelphred unedited
    if (P->inferred_is_vcf_generated) TIMED(prof, "k_thres", hipLaunchKernelGGL(k_thres, dim3(nblk(R->npos, 256)), dim3(256), 0, s, *R, *P, half_ratio_phred));
    if (side) { hipEventRecord(e_fork, s); hipStreamWaitEvent(s2, e_fork, 0); if (s3 != s2) hipStreamWaitEvent(s3, e_fork, 0); }
    // ---- side streams.  s2: the walk that produces the P2 items.  s3: the walk that fills the contribution table, then the per-fragment
    // statistics, the fragment window index and the interval sums (they need P1 / P1b only), then the queued mismatches of the base pass.
    if (P->inferred_is_vcf_generated && R->n_complex) TIMED2(prof, "k_p2_slow_walk", hipLaunchKernelGGL(k_p2_slow<true>, dim3(nblk(R->n_complex, 64)), dim3(64), 0, s, *R, *P));
    if (R->n_complex) TIMED3(prof, "k_p2_slow_table", hipLaunchKernelGGL(k_p2_slow<false>, dim3(nblk(R->n_complex, 64)), dim3(64), 0, s, *R, *P));
    TIMED3(prof, "k_fragstat_fast", hipLaunchKernelGGL(k_fragstat_fast, dim3(nblk(R->n_frags, 256)), dim3(256), 0, s, *R, *P));
    TIMED3(prof, "k_win_index_frag", hipLaunchKernelGGL(k_win_index, dim3(nblk(2 * (int64_t)R->nwin * 2, 256)), dim3(256), 0, s, *R, 5, 7));
    if (R->n_sweep) TIMED3(prof, "k_fragstat_sweep", hipLaunchKernelGGL(k_fragstat_sweep, dim3(R->n_sweep), dim3(64), 0, s, *R, *P, R->sweep_frags, (const int32_t *)nullptr, R->n_sweep));
    // fragments whose event list overflowed (device-side list; the grid covers the worst case, surplus threads exit)
    TIMED3(prof, "k_fragstat_overflow", hipLaunchKernelGGL(k_fragstat_sweep, dim3(imin_h(R->n_frags, 65535)), dim3(64), 0, s, *R, *P, (const int32_t *)R->overflow_frags, (const int32_t *)R->n_overflow, 0));
    if (UVC_PLATFORM_IONTORRENT != P->inferred_sequencing_platform)   // (there every fragment takes k_frag_generic)
        TIMED3(prof, "k_frag_sums", hipLaunchKernelGGL(k_frag_sums, dim3(nblk(R->npos, FSUM_TILE), 2), dim3(256), 0, s, *R, *P));
    // ---- main stream: the base symbols first, so that the queued mismatches (rare symbols, atomics: disjoint from the planes the
    // LINK_M pass stores to) are applied on a side stream while the LINK_M pass runs
    if (P->inferred_is_vcf_generated) {
        // no IonTorrent values, no amplicon-flagged family, no primer length, median read length at or above microadjust_median_readlen_thres:
        // the specialisation without those arms
        const bool plain = (UVC_PLATFORM_IONTORRENT != P->inferred_sequencing_platform) && !R->any_amplicon && !(P->primerlen > 0 && !(0x2 & P->primer_flag))
                           && (P->central_readlen >= P->microadjust_median_readlen_thres);
        if (split_windows && plain) TIMED(prof, "k_p2_fast_base", hipLaunchKernelGGL((k_p2_fast_split<false, true, true>), dim3(R->nwin), dim3(256), 0, s, *R, *P));
        else if (split_windows) TIMED(prof, "k_p2_fast_base", hipLaunchKernelGGL((k_p2_fast_split<false, true, false>), dim3(R->nwin), dim3(256), 0, s, *R, *P));
        else if (plain) TIMED(prof, "k_p2_fast_base", hipLaunchKernelGGL((k_p2_fast<false, true, true>), dim3(nwin), dim3(256), 0, s, *R, *P));
        else TIMED(prof, "k_p2_fast_base", hipLaunchKernelGGL((k_p2_fast<false, true, false>), dim3(nwin), dim3(256), 0, s, *R, *P));
        if (side) { hipEventRecord(e_fork2, s); hipStreamWaitEvent(s3, e_fork2, 0); }
        TIMED3(prof, "k_p2_mism", hipLaunchKernelGGL(k_p2_mism, dim3(2048), dim3(256), 0, s, *R, *P));
        if (split_windows && plain) TIMED(prof, "k_p2_fast_link", hipLaunchKernelGGL((k_p2_fast_split<true, false, true>), dim3(R->nwin), dim3(256), 0, s, *R, *P));
        else if (split_windows) TIMED(prof, "k_p2_fast_link", hipLaunchKernelGGL((k_p2_fast_split<true, false, false>), dim3(R->nwin), dim3(256), 0, s, *R, *P));
        else if (plain) TIMED(prof, "k_p2_fast_link", hipLaunchKernelGGL((k_p2_fast<true, false, true>), dim3(nwin), dim3(256), 0, s, *R, *P));
        else TIMED(prof, "k_p2_fast_link", hipLaunchKernelGGL((k_p2_fast<true, false, false>), dim3(nwin), dim3(256), 0, s, *R, *P));
    }
    if (side) {
        hipEventRecord(e_join, s2); hipStreamWaitEvent(s, e_join, 0);
        if (s3 != s2) { hipEventRecord(e_join3, s3); hipStreamWaitEvent(s, e_join3, 0); hipStreamWaitEvent(s2, e_join3, 0); }   // (the allele tables below read the contribution table)
    }
    // InDel allele tables: they read the contribution table (side stream) and the reads only, so the pipeline stays on the side stream
    // underneath the fragment / family kernels; the main stream picks it up at the end
    const GapWork &G = R->gap;
    hipMemsetAsync(G.n_inc, 0, 16, s2);   // n_inc, n_rows, seq_len
    if (P->inferred_is_vcf_generated && G.n_ev > 0) {
        TIMED2(prof, "k_gap", {
            hipLaunchKernelGGL(k_gap_keys, dim3(nblk(G.n_ev, 256)), dim3(256), 0, s, *R);
            uvc_gap_sort(G.sort_tmp, G.sort_tmp_bytes, G.ckey, G.ckey_s, G.cval, G.cval_s, (size_t)G.n_ev, 58, s);
            hipMemsetAsync(G.ikey, 0xFF, sizeof(unsigned long long) * (size_t)G.inc_cap, s);
            hipLaunchKernelGGL(k_gap_alleles, dim3(nblk(G.n_ev, 64)), dim3(64), 0, s, *R, *P);
            if (side) hipEventRecord(e_alleles, s);   // what the family kernels read of this chain (fam2_ins_len) is complete here
            uvc_gap_sort(G.sort_tmp, G.sort_tmp_bytes, G.ikey, G.ikey_s, G.ival, G.ival_s, (size_t)G.inc_cap, 64, s);
            hipLaunchKernelGGL(k_gap_rows, dim3(nblk(G.inc_cap, 256)), dim3(256), 0, s, *R);
            hipLaunchKernelGGL(k_gap_rows_add, dim3(nblk(G.inc_cap, 256)), dim3(256), 0, s, *R);
            hipLaunchKernelGGL(k_gap_rows_fin, dim3(nblk(G.n_ev, 256)), dim3(256), 0, s, *R);   // (rows <= events)
        });
    }
    if (side && !(P->inferred_is_vcf_generated && G.n_ev > 0)) hipEventRecord(e_alleles, s2);
    // per-unit statistics of the generic family units (medians, first / last consensus position, general-kind flag): they need the contribution
    // table and the fragment records only: on s3 behind the queued mismatches, beside the allele tables on s2, underneath k_frag
    if (R->n_generic_fs) TIMED3(prof, "k_fam_stat", hipLaunchKernelGGL(k_fam_stat, dim3(nblk(R->n_generic_fs, 64)), dim3(64), 0, s, *R, *P));
    if (side) { hipEventRecord(e_stat, s3); hipEventRecord(e_fork2, s2); }
    if (P->inferred_is_vcf_generated && R->n_complex) TIMED(prof, "k_p2_items", hipLaunchKernelGGL(k_p2_items, dim3(R->n_complex), dim3(64), 0, s, *R, *P));
    {
        const bool proton = (UVC_PLATFORM_IONTORRENT == P->inferred_sequencing_platform);
        const int n_gen = proton ? R->n_frags : R->n_sweep;
        if (n_gen) TIMED(prof, "k_frag_generic", hipLaunchKernelGGL(k_frag_generic, dim3(imin_h(n_gen, 1 << 20), imin_h((R->max_frag_span + 63) / 64, 16)), dim3(64), 0, s, *R, *P, proton ? (const int32_t *)nullptr : R->sweep_frags, n_gen));
    }
    {
        const bool plain = P->inferred_is_vcf_generated && (UVC_PLATFORM_IONTORRENT != P->inferred_sequencing_platform) && !(0x1 & P->fam_flag) && !(P->microadjust_padded_deletion_flag & 0x1);
        const bool h16 = (R->max_frag_depth < 65536) && !R->frag32;
        if (split_windows && h16 && plain) TIMED(prof, "k_frag", hipLaunchKernelGGL(k_frag16_split<true>, dim3(R->nwin), dim3(256), 0, s, *R, *P));
        else if (split_windows && h16) TIMED(prof, "k_frag", hipLaunchKernelGGL(k_frag16_split<false>, dim3(R->nwin), dim3(256), 0, s, *R, *P));
        else if (plain && h16) TIMED(prof, "k_frag", hipLaunchKernelGGL(k_frag16<true>, dim3(nwin), dim3(256), 0, s, *R, *P));
        else if (plain) TIMED(prof, "k_frag", hipLaunchKernelGGL(k_frag<true>, dim3(nwin), dim3(256), 0, s, *R, *P));
        else if (h16) TIMED(prof, "k_frag", hipLaunchKernelGGL(k_frag16<false>, dim3(nwin), dim3(256), 0, s, *R, *P));
        else TIMED(prof, "k_frag", hipLaunchKernelGGL(k_frag<false>, dim3(nwin), dim3(256), 0, s, *R, *P));
    }
    if (R->n_generic_fs) {
        if (side) { hipStreamWaitEvent(s, e_stat, 0); hipStreamWaitEvent(s, e_alleles, 0); }   // k_fam_stat's unit records; fam2_ins_len reads what k_gap_alleles left (both ran under k_frag)
        // shallow data: one thread per (unit, position); deep data (many units per position, e.g. UMI panels): the window kernel, whose
        // LDS collection removes most of the atomics that bound the per-thread form
        const bool deep = (R->fam_path == 1 ? false : (R->fam_path == 2 ? true : (R->n_generic_work > 8 * R->npos)));
        const bool digest = deep && R->fam_digest && P->inferred_is_vcf_generated;   // one walk over the fragments of a unit instead of three
        if (digest) TIMED(prof, "k_fam_p4", { hipLaunchKernelGGL(k_fam_p4d, dim3(nblk(R->npos, 64)), dim3(256), 0, s, *R, *P);
                                               hipLaunchKernelGGL(k_fam_p4d_rest, dim3(R->n_generic_fs), dim3(64), 0, s, *R, *P); });
        else if (deep) TIMED(prof, "k_fam_p4", hipLaunchKernelGGL((k_fam_win<4, false>), dim3(nblk(R->npos, 64)), dim3(256), 0, s, *R, *P));
        else TIMED(prof, "k_fam_p4", hipLaunchKernelGGL(k_fam_p4, dim3(nblk(R->n_generic_work, 256)), dim3(256), 0, s, *R, *P));
        if (P->inferred_is_vcf_generated) {
            if (digest) TIMED(prof, "k_fam_p5", hipLaunchKernelGGL((k_fam_win<5, true>), dim3(nblk(R->npos, 64)), dim3(256), 0, s, *R, *P));
            else TIMED(prof, "k_fam_p5", hipLaunchKernelGGL(k_fam_p5, dim3(nblk(R->n_generic_work, 256)), dim3(256), 0, s, *R, *P));
            if (n_dup && digest) TIMED(prof, "k_duplex", hipLaunchKernelGGL(k_duplex_d, dim3(n_dup), dim3(64), 0, s, *R, dup_units, n_dup, dup_off, n_dup_work));
            else if (n_dup) TIMED(prof, "k_duplex", hipLaunchKernelGGL(k_duplex, dim3(nblk(n_dup_work, 256)), dim3(256), 0, s, *R, *P, dup_units, n_dup, dup_off, n_dup_work));
        }
    }
    if (P->inferred_is_vcf_generated) TIMED(prof, "k_p5b", hipLaunchKernelGGL(k_p5b, dim3(nblk(R->npos * 2, 256)), dim3(256), 0, s, *R, *P));
    if (side) hipStreamWaitEvent(s, e_fork2, 0);   // the allele tables
}
