// uvc_device.h -- device-side data layout of the MI355X UVC hot path (gfx950 only).
//
// HBM layout: every per-position quantity is a plane [field][symbol][pos]
// with the position fastest, so that a wavefront whose 64 lanes own 64 consecutive positions
// reads and writes 256 contiguous bytes per plane.  Reads are kept as 1 B/base + 1 B/qual SoA;
// a lane that owns position p reads byte (qbase + p) of every alignment that covers p, so the
// 64 lanes of a wave read 64 consecutive bytes of the same read (coalesced), and everything else
// about the read is wave-uniform and comes through scalar loads.
#ifndef UVC_DEVICE_H
#define UVC_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "uvcgpu.h"

#define NSYM UVC_NUM_SYMBOLS
#define NBUCKETS 16          // NUM_BUCKETS, main_conversion.hpp:920
#define SQR_QUAL_DIV 32      // main_conversion.hpp:20
#define MAX_INSERT_SIZE 2000 // common.hpp:64
#define MAX_STR_N_BASES 100  // common.hpp:63
#define UVC_MAXEV 8

enum { C_MATCH = 0, C_INS = 1, C_DEL = 2, C_REF_SKIP = 3, C_SOFT_CLIP = 4, C_HARD_CLIP = 5, C_PAD = 6, C_EQUAL = 7, C_DIFF = 8 };

// One alignment.  "simple" = [one clip op] one M/=/X op [one clip op]; everything else is "complex"
// and goes through the sequential per-read kernels + the contribution table.
struct AlnRec {
    int32_t pos, rend, mpos, isize;
    int32_t flag, mapq, dflag, l_qseq;
    int64_t seq_off;            // first query base in bases[] / quals[]
    int64_t qbase;              // simple: seq_off + lclip_q - pos, so that query byte of ref position p is qbase + p
    int64_t cigar_off;
    int64_t table_off;          // complex: first row of the contribution table (row = p - pos), else -1
    int64_t item_off;           // complex: first slot of the P2 item list, else -1
    int64_t gap_off;            // complex: first slot of its InDel events (one per I / D op, in CIGAR order), else -1
    int32_t n_cigar, kind;      // kind: 0 simple, 1 complex
    int32_t xm1500, go1500;
    int32_t bm1500[5];          // per base symbol (main.hpp:1860-1863); LINK/NN symbols are always 0
    int32_t clip_cnt;
    int32_t nogap_penal, indel_penal;   // micro_nogap_penal / micro_indel_penal, main.hpp:1882-1885
    int32_t lclip_len, rclip_len;       // soft-clip lengths at either end (0 if none), main.hpp:1880-1881
    int32_t lclip_oplen, rclip_oplen;   // S or H clip op adjacent to the M op (0 if none)
    int32_t lclip_q;                    // query offset of the first aligned base
    int32_t m_index;                    // cigar index of the M op (simple)
    int32_t frag, fs;                   // owning fragment / family-strand unit
    int32_t id;                         // index in RegionDev::alns
    int32_t n_mutc;                     // bases that differ from the reference + I / D ops: an upper bound of the mutated (position, symbol) pairs the
                                        // alignment can add to a haplotype string (k_hap_*)
    int64_t baq_pos, baq_last, baq2_last;   // baq[pos], baq[rend-1], baq2[rend-1] (region constants, main.hpp:1394-1396)
};

// 64-byte digest of a simple alignment for the position-centric kernels, stored pos-sorted.  A wave loads 64 consecutive
// records (one per lane, coalesced) and broadcasts the fields of record j with v_readlane, so the per-read scalars never
// cost a dependent memory round trip inside the loop.
struct FastRec {
    int32_t pos, rend, qb_lo, aln;          // [pos, rend): covered positions (the whole read, or one M run of an InDel read); qb_lo: low word of
                                            // (index of the base at position p) - p; aln: index into alns[]
    int32_t fmd, isize, mpos, xm1500;       // fmd = flag | mapq << 16 | dflag << 24
    int32_t bmv, xbv, bm4c, clips;          // bmv: a2BM2 increment (<= 100) of base symbols 0..3, one byte each; xbv: that of symbol 4 | a2XM2 increment << 8;
                                            // bm4c: clip_cnt << 16 | (nogap_penal & 0xFFFF); clips = lclip_oplen | rclip_oplen << 16
    int32_t baq_pos, baq_last, baq2_last, ext;   // ext: (pos - read start) | (read end - rend) << 16, zero for a whole read
};

struct FragRec {
    int32_t aln_beg, aln_end;   // alignments are stored fragment-major
    int32_t beg, end;           // span with the reference's cumulative "+1 per alignment" (fillTidBegEndFromAlns1, main.hpp:658-673), clamped to the region
    int32_t fs, strand, dflag, normMQ;
    int32_t n_cov, n_near;      // b10xSeqTlen / b10xSeqTNevents, main.hpp:2747-2756 (filled by k_fragstat)
    int32_t singleton;          // 1: the only fragment of its family-strand unit and the fast P4/P5 identities apply
    int32_t stat_kind;          // 0: n_cov/n_near from mutation events (<= 2 simple alignments, no primer gating); 1: per-fragment sweep
};

struct FsRec {                  // family x strand unit (alns2 of main.hpp:2869)
    int32_t frag_beg, frag_end;
    int32_t beg, end;           // fillTidBegEndFromAlns2 span, clamped
    int32_t strand, dflag, fam, generic;   // generic: handled by the per-(unit, position) family kernels
    int64_t work_off;           // prefix offset of this unit's positions in the generic work list
    int32_t l2r_end_median, r2l_end_median, nsb_min, nsb_max;   // main.hpp:2939-2940, 2959-2998
    int32_t other_fs;           // the opposite-strand unit of the same family, or -1
    int32_t pad_;
};
static_assert(sizeof(FsRec) == 64, "k_fam_p4d fetches a unit record as four 16-byte words per lane");

// One P2 update of an InDel read: "add value `val` of symbol `sym` at position `epos` and run dealwith_segbias with these
// arguments".  The sequential CIGAR walk (k_p2_slow<true>) only produces items; k_p2_items applies them in parallel.
struct Item { int32_t epos; uint8_t sym, flags /* bit0 isGap, bits 1..4 cigar op */; uint16_t val; uint16_t dist, indel_len; int32_t pad2; };
static_assert(sizeof(Item) == 16, "Item");

// A base of a simple alignment that differs from the reference: its P2 update goes to a non-dense symbol, so k_p2_fast
// queues it here and k_p2_mism applies it (one lane per item).  symval = sym | value << 8.
struct MisItem { int32_t rank, epos, symval; };

// compact per-fragment record for k_frag, stored in beg-sorted order (written by k_fragstat_fast).  A fragment of <= 2
// alignments, each with at most one InDel, is described by up to two M runs per alignment (run B empty for a simple alignment)
// plus, per alignment, the "special" positions around its InDel, which k_frag leaves to k_frag_generic.
// The first 12 dwords of a FragFast record again, indexed by fragment number (the fragments of a family-strand unit are consecutive):
// k_fam_p4d reaches the records of a unit without the frag_rank indirection.
struct FragUnit { int32_t v[12]; };
// What a plain fragment (<= 2 simple alignments, flags & 0x301 == 0) adds to LINK_M and to "some base" at a position does not depend on the
// position's bases: a count by micro_nogap_penal class (1..5) for the bucket histogram and sums of per-fragment constants.  k_frag_sums builds
// them as interval sums (O(1) per fragment and plane), k_frag reads them back per position instead of adding them per (fragment, position).
enum { UVC_FSUM_LCNT = 0 /* .. 4: LINK_M by class */, UVC_FSUM_LTA = 5, UVC_FSUM_LTB, UVC_FSUM_LSING, UVC_FSUM_LMQ, UVC_FSUM_BDP, UVC_FSUM_BTA, UVC_FSUM_BTB, UVC_FSUM_BMQ, UVC_FSUM_N };
struct FragFast {
    int32_t beg, end, fi, flags;        // flags: bit0 generic path only, bit1 strand, bit2 singleton, bits 3..6 number of alignments, bit8 has runs B / special ranges, bit9 a micro_nogap_penal outside 1..5 (k_frag adds its LINK_M per position)
    int32_t pos0, rend0, pos1, rend1;   // run A of alignment 0 / 1: [pos, rend)
    int32_t qb0, qb1, nogap0, nogap1;   // low word of the query offset of run A; micro_nogap_penal of the alignment
    int32_t sq, n_cov, n_near, pad_;    // sq = normMQ^2 / SQR_QUAL_DIV
    int32_t bpos0, brend0, bqb0, sp0;   // run B of alignment 0, its query offset; sp = special range (beg - fragment beg) | length << 16
    int32_t bpos1, brend1, bqb1, sp1;
};

// One I / D op of an alignment as the BASE_QUALITY_MAX walk sees it (incIns / incDel, main.hpp:2101-2113, 2216): the allele-keyed
// counters of the reference (pos2iseq2data / pos2dlen2data) are rebuilt from these by k_gap_alleles.
struct AlnGap {
    int32_t epos, sym;          // sym < 0: gated by the primer window or closer than indel_filter_edge_dist to a read end
    int32_t len, qpos;          // op length; query offset of the first inserted base
    int32_t weight, aln;        // max(1, incvalue2) resp. max(1, incvalue); owning alignment
    int32_t mark, pad_;         // scratch of k_gap_alleles
};
// allele-level increment: key = (epos - beg) << 38 | (sym - LINK_D3P) << 35 | allele code; val = representative event << 8 | strand * 4 + level
struct GapRow { int32_t x, sym, len, ev; int64_t seq_off; int32_t cnt[8]; };   // cnt[strand * 4 + level], level: 0 fragments, 1 families, 2 cDP2, 3 c2dDP
struct GapWork {
    AlnGap *ev; int32_t n_ev;
    unsigned long long *ckey, *ckey_s, *cval, *cval_s;       // candidate sort: (family, position) -> event
    unsigned long long *ikey, *ikey_s, *ival, *ival_s;       // allele increments and their sorted copies
    int32_t inc_cap; int32_t *n_inc;
    GapRow *rows; int32_t *n_rows; uint8_t *seq; unsigned long long *seq_len; int64_t seq_cap;
    int32_t *maj;               // [2 * n_ev], at the head of each (family, position) run of the sorted candidates: per strand the largest number of
                                // fragments of the unit that agree on one inserted sequence (read_family_con_ampl_getMajority_ins, main.hpp:188-198)
    void *sort_tmp; size_t sort_tmp_bytes;
};

// contribution of one alignment at one reference position under BASE_QUALITY_MAX (main.hpp:1980, 1924, 2077, 2192, 2223)
// BASE_QUALITY_MAX contributions of an InDel read at one reference position: its base symbol (if any) and one slot per LINK symbol, so that
// any pile of LINK symbols at one position of one read fits (an insertion in front of a deletion behind a padded deletion ...); 16-bit
// values: reads whose NM tag is far below their InDel lengths get negative penalties, i.e. values beyond 255 (main.hpp:1882-1885)
struct Contrib { uint8_t bsym_p1 /* base symbol + 1, 0 = none */, lmask /* bit s - LINK_M: LINK symbol s present */; uint16_t bval; uint16_t lval[8]; uint16_t pad_[2]; };
static_assert(sizeof(Contrib) == 24, "Contrib rows are zero-filled as bytes and indexed as 24-byte records");

// haplotype links: the candidates (fragments / family-strand units with at least two possible mutations), where each one's
// events go in the event buffer, and the buffer.  An event list: [strand | kind << 1, count, (x << 4 | symbol) ...], kind 0 bq, 1 fq, 2 f2q.
struct HapWork { int32_t *cand, *cand_off, *cand_cap; int32_t *n_cand; unsigned long long *total; int32_t *events; };

struct DevParams { UvcParams P; };

struct RegionDev {
    int32_t beg, end; int64_t npos;
    const uint8_t *refsym;          // [npos + 1]
    int32_t *rtr;                   // [UVC_NRTR][npos]
    const int64_t *baq;             // [2][npos]
    int32_t *prep32; int64_t *prep64; int32_t *thres;
    int32_t *seg32; int64_t *seg64; int32_t *vq; int32_t *bqsum;
    int32_t *frag; int32_t *fam; int32_t *faminfo32; int64_t *faminfo64; int32_t *duplex;
    int32_t *bucket;                // [2][NSYM][NBUCKETS][npos] dedup_ampDistr, main.hpp:2377
    const uint8_t *bases; const uint8_t *quals; const uint32_t *cigars;
    const uint16_t *bq; uint32_t bq_bytes;   // base | qual << 8 per read base: one bounds-checked buffer load per (alignment, position)
    AlnRec *alns; int32_t n_alns;
    int32_t n_fast;                 // number of simple alignments
    FastRec *frec;                  // [n_fast] their digests, sorted by pos (FastRec::aln: the index into alns[])
    int32_t p2_off[5];              // frec2 is four pos-sorted sub-lists [p2_off[c], p2_off[c + 1]), c = is-reverse | bam_get_strand << 1
    FastRec *frec2; int32_t n_fast2; int32_t max_p2_span;   // P2 work list, sorted by (class, pos): simple alignments + the M runs of InDel reads whose
                                    // InDels are all high-quality (kind 2), which behave like simple alignments in P2 (see k_p2_fast)
    const int32_t *complex_ids; int32_t n_complex;
    FragRec *frags; int32_t n_frags;
    int32_t *frag_nmut; int32_t *frag_mut;          // mutation events per fragment: count + up to UVC_MAXEV positions
    const int32_t *sweep_frags; int32_t n_sweep;    // fragments that need the sequential sweep (host list)
    int32_t *overflow_frags; int32_t *n_overflow;   // fragments whose event list overflowed (device list)
    const int32_t *frag_sorted;     // fragment ids sorted by FragRec::beg
    const int32_t *frag_rank;       // inverse permutation of frag_sorted
    int32_t frag_off[3];            // ffast = the strand-0 fragments sorted by beg, then the strand-1 fragments sorted by beg
    FragFast *ffast;                // [n_frags] in (strand, beg)-sorted order
    FragUnit *ffast_u;              // [n_frags] by fragment number (see FragUnit)
    int32_t *win; int32_t nwin;     // window index [8 lists][lo, hi][nwin = ceil(npos / 64)] (k_win_index)
    int32_t *fsum;                  // [2 strands][UVC_FSUM_N][npos]: interval sums of the plain fragments (k_frag_sums), read by k_frag
    FsRec *fss; int32_t n_fs;
    const int32_t *generic_fs; int32_t n_generic_fs; int64_t n_generic_work;
    const int32_t *generic_sorted; int32_t max_unit_span;   // the generic units ordered by FsRec::beg (window kernels k_fam_win)
    uint8_t *p5flag;                // [2][npos]: a P5 bucket of this (strand, position) was filled
    uint8_t *dirty; int32_t ndblk;  // [3][NSYM][ndblk]: which (plane family, symbol, block of 4 096 positions) the accumulate wrote outside the part that is zero-filled
                                    // anyway -- family 0: a rare symbol (BASE_NN, every LINK symbol but LINK_M) in the SEG / VQ / BQSUM / FRAG / FAM planes, 1: any symbol in
                                    // FAMINFO32 / 64, 2: any symbol in DUPLEX.  The fill in front of the next accumulate skips what is not marked (uvc_launch_zero_state)
    uint32_t *occ;                  // [npos] bit s: somebody wrote a cell of (s, position) outside the dense-symbol paths (occ_mark); zeroed with the planes, read by the scoring gate
    uint32_t *fam_digest;           // [n_generic_work][8] or NULL: what P4 leaves per (unit, position) for P5 and the duplex pass (k_fam_win<4> / k_fam_win5d / k_duplex_d)
    Contrib *table;
    int32_t *ir_list;               // per InDel read: the positions of its low-quality InDels (k_p2_slow's cursor, main.hpp:1817-1859), [gap_off + 2 * rank .. ) with sentinels
    Item *items; int32_t *item_cnt;     // per complex alignment (indexed like complex_ids)
    MisItem *mis; int32_t *mis_cnt; int32_t mis_cap;   // mismatch queue of k_p2_fast, sized from the exact count below
    unsigned long long *mis_total;  // number of read bases of simple alignments that differ from the reference (k_aln_prelude)
    int32_t max_aln_span, max_frag_span;
    int32_t any_amplicon;           // some family carries the amplicon flag (fam_dflag & 0x4)
    int32_t frag32;                 // UVCGPU_FRAG32=1: k_frag with 32-bit buckets although the depth would allow the packed form (tests compare the two)
    int32_t fam_path;               // 0: the family kernels are chosen from the data; 1 / 2: UVCGPU_FAM_PATH=generic / window (tests compare the three forms)
    int32_t max_frag_depth;         // upper bound of the number of fragments that cover one position
    int32_t *err;                   // device error flag (unsupported CIGAR shapes etc.)
    GapWork gap;                    // InDel allele tables
};

#define DEV __device__ __forceinline__

DEV int imin(int a, int b) { return a < b ? a : b; }
DEV int imax(int a, int b) { return a > b ? a : b; }
DEV int64_t lmin(int64_t a, int64_t b) { return a < b ? a : b; }
DEV int64_t lmax(int64_t a, int64_t b) { return a > b ? a : b; }
DEV int64_t nnminus(int64_t a, int64_t b) { return a > b ? a - b : 0; }     // non_neg_minus, common.hpp:195-200
DEV int ibetween(int v, int a, int b) { return imin(imax(a, v), b); }       // BETWEEN, main_conversion.hpp:124-128
DEV bool is_ins(int s) { return s == UVC_LINK_I3P || s == UVC_LINK_I2 || s == UVC_LINK_I1; }
DEV bool is_del(int s) { return s == UVC_LINK_D3P || s == UVC_LINK_D2 || s == UVC_LINK_D1; }
DEV bool is_subst(int s) { return s <= UVC_BASE_NN; }
DEV bool symbols_mutated(int ref, int alt) {      // areSymbolsMutated, main_conversion.hpp:364-371
    if (alt <= UVC_BASE_NN) return ref != alt && ref < UVC_BASE_N && alt < UVC_BASE_N;
    return alt != UVC_LINK_M && alt != UVC_LINK_NN;
}
DEV int cig_op(uint32_t c) { return (int)(c & 0xF); }
DEV int cig_len(uint32_t c) { return (int)(c >> 4); }

// plane addressing
#define P32(R, f, x) ((R).prep32[(size_t)(f) * (R).npos + (x)])
#define P64(R, f, x) ((R).prep64[(size_t)(f) * (R).npos + (x)])
#define TH(R, f, x) ((R).thres[(size_t)(f) * (R).npos + (x)])
#define S32(R, f, s, x) ((R).seg32[((size_t)(f) * NSYM + (s)) * (R).npos + (x)])
#define S64(R, f, s, x) ((R).seg64[((size_t)(f) * NSYM + (s)) * (R).npos + (x)])
#define VQP(R, f, s, x) ((R).vq[((size_t)(f) * NSYM + (s)) * (R).npos + (x)])
#define BQS(R, s, x) ((R).bqsum[(size_t)(s) * (R).npos + (x)])
#define FRP(R, st, f, s, x) ((R).frag[(((size_t)(st) * UVC_NFRAG + (f)) * NSYM + (s)) * (R).npos + (x)])
#define FAP(R, st, f, s, x) ((R).fam[(((size_t)(st) * UVC_NFAM + (f)) * NSYM + (s)) * (R).npos + (x)])
#define FIP(R, f, s, x) ((R).faminfo32[((size_t)(f) * NSYM + (s)) * (R).npos + (x)])
#define FI64P(R, f, s, x) ((R).faminfo64[((size_t)(f) * NSYM + (s)) * (R).npos + (x)])
#define DUP(R, f, s, x) ((R).duplex[((size_t)(f) * NSYM + (s)) * (R).npos + (x)])
#define BKP(R, st, s, b, x) ((R).bucket[((((size_t)(st) * NSYM + (s)) * NBUCKETS) + (b)) * (R).npos + (x)])
#define RTRP(R, f, x) ((R).rtr[(size_t)(f) * (R).npos + (x)])
#define BAQ1(R, p) ((R).baq[(p) - (R).beg])
#define BAQ2(R, p) ((R).baq[(R).npos + (p) - (R).beg])

// (see RegionDev::dirty)
#define UVC_DIRTY_SHIFT 12
DEV bool sym_always_filled(int s) { return s < UVC_BASE_NN || s == UVC_LINK_M; }   // A C G T N (a reference base) and LINK_M: written at nearly every position
// RegionDev::occ: every writer of a cell that does not belong to one of the position's two dense symbols says so here (agent scope: kernels
// on the side streams mark the same words), so that scoring knows the symbols of a position without reading their planes
// (a plain look first: the P2 kernels have usually marked the symbol long before the fragment and family kernels come by; a stale look costs an OR)
DEV void occ_mark(const RegionDev &R, int s, int64_t x) { if (s != UVC_LINK_M && s != (int)R.refsym[x] && !((R.occ[x] >> s) & 1u)) atomicOr(&R.occ[x], 1u << s); }
DEV void mark_sym(const RegionDev &R, int s, int64_t x) { occ_mark(R, s, x); if (!sym_always_filled(s)) R.dirty[(size_t)s * R.ndblk + (x >> UVC_DIRTY_SHIFT)] = 1; }
DEV void mark_fi(const RegionDev &R, int s, int64_t x) { R.dirty[((size_t)NSYM + s) * R.ndblk + (x >> UVC_DIRTY_SHIFT)] = 1; }
DEV void mark_dup(const RegionDev &R, int s, int64_t x) { occ_mark(R, s, x); R.dirty[((size_t)2 * NSYM + s) * R.ndblk + (x >> UVC_DIRTY_SHIFT)] = 1; }

DEV void add64(int64_t *p, int64_t v) { atomicAdd((unsigned long long *)p, (unsigned long long)v); }
// Adds to a cell that only one workgroup touches during the kernel (a window kernel owns its 64 positions): an L2 atomic of workgroup
// scope.  atomicAdd() has agent scope, which on this part is carried out at the memory side (the L2s of the eight XCDs are not coherent
// with each other) and costs a memory round trip per add.
DEV void add_own(int32_t *p, int v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEV void add64_own(int64_t *p, int64_t v) { __hip_atomic_fetch_add((unsigned long long *)p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// per-thread symbol-count array kept in LDS ([symbol][thread], conflict-free): dynamic indexing by symbol without scratch memory
template <int STRIDE> struct LdsCounts {
    int *p;   // &array[0][threadIdx.x]
    DEV int &operator[](int s) const { return p[s * STRIDE]; }
};

// GenericSymbol2Count::_fillConsensusCounts<TIsRefCountedOnlyOnce>, main.hpp:374-402
template <class Arr>
DEV void fill_consensus(const Arr &c, int &argmax, int &cmax, int &csum, int st, bool ref_once_in_link, bool ignore_padded_del) {
    const int b = (st == UVC_BASE_SYMBOL ? UVC_BASE_A : UVC_LINK_M);
    const int e = (st == UVC_BASE_SYMBOL ? (ignore_padded_del ? UVC_BASE_T : UVC_BASE_NN) : UVC_LINK_NN);
    const bool once = (st == UVC_LINK_SYMBOL && ref_once_in_link);
    argmax = e; cmax = 0; csum = 0;
    for (int s = b; s <= e; s++) {
        const int v = c[s];
        if (once) {
            if (cmax < v || (UVC_LINK_M == argmax && (0 < v))) { argmax = s; cmax = v; csum = cmax; }
        } else {
            if (cmax < v) { argmax = s; cmax = v; }
            csum += v;
        }
    }
}

// PhredMutationTable::toPhredErrRate, main.hpp:213-262
DEV int sscs_phred(const UvcParams &P, int con_symbol, int alt_symbol) {
    int raw;
    if (is_ins(con_symbol) || is_del(con_symbol)) raw = P.fam_phred_sscs_indel_open;
    else if (con_symbol == UVC_LINK_M) {
        if (UVC_LINK_D1 == alt_symbol || UVC_LINK_I1 == alt_symbol) raw = P.fam_phred_sscs_indel_open;
        else if (UVC_LINK_D2 == alt_symbol || UVC_LINK_I2 == alt_symbol) raw = P.fam_phred_sscs_indel_open + P.fam_phred_sscs_indel_ext;
        else raw = P.fam_phred_sscs_indel_open + P.fam_phred_sscs_indel_ext * 2;
    } else if ((con_symbol == UVC_BASE_C && alt_symbol == UVC_BASE_T) || (con_symbol == UVC_BASE_G && alt_symbol == UVC_BASE_A)) raw = P.fam_phred_sscs_transition_CG_TA;
    else if ((con_symbol == UVC_BASE_A && alt_symbol == UVC_BASE_G) || (con_symbol == UVC_BASE_T && alt_symbol == UVC_BASE_C)) raw = P.fam_phred_sscs_transition_AT_GC;
    else if ((con_symbol == UVC_BASE_C && alt_symbol == UVC_BASE_A) || (con_symbol == UVC_BASE_G && alt_symbol == UVC_BASE_T)) raw = P.fam_phred_sscs_transversion_CG_AT;
    else raw = P.fam_phred_sscs_transversion_other;
    return raw + (P.tumor_vcf_fname_nonempty ? 3 : 0);
}

// infer_max_qual_assuming_independence, main_conversion.hpp:943-974.  fp64 log, result truncated to int.
template <class GetBucket>
DEV void infer_max_qual(int &maxvqual, int &argmaxAD, int &argmaxBQ, int max_qual, int dec_qual, int totDP, GetBucket get) {
    int currAD = 0;
    maxvqual = 0; argmaxAD = 0; argmaxBQ = 0;
    const int n = imin(NBUCKETS, max_qual / dec_qual);
    for (int idx = 0; idx < n; idx++) {
        const int currQD = get(idx);
        if (0 == currQD) continue;
        currAD += currQD;
        const int currBQ = max_qual - (dec_qual * idx);
        const double expBQ = 10.0 / log(10.0) * log(((double)totDP / (double)currAD) + 2.220446049250313e-16);
        const int currvqual = (int)(currAD * (currBQ - expBQ));
        if (currvqual > maxvqual) { argmaxAD = currAD; argmaxBQ = currBQ; maxvqual = currvqual; }
    }
}

// The same with the bucket counts in registers (fully unrolled: static indices), dec_qual == 1, and the fp64 logarithm only where it can
// matter.  The scan keeps the first bucket that attains the largest (int)(currAD * (currBQ - expBQ)) > 0.  A single-precision pass (v_rcp,
// v_log: |error of the product| < 3 while currAD < 65 536) finds the largest approximate value; a bucket more than 2 * 8 + 1 below it is
// more than 1 below the true maximum and can neither win nor tie after truncation, so only the others are evaluated exactly, in order.
// (A wave evaluates a bucket index if any of its lanes needs it: 3-4 of 16 instead of all -- the logarithms were 12 % of k_frag's instructions.)
DEV void infer_max_qual_regs(int &maxvqual, int &argmaxAD, int &argmaxBQ, int max_qual, int totDP, const int (&h)[NBUCKETS]) {
    unsigned cand = 0;
    {
        const bool small = (totDP < 65536);
        const float ft = (float)totDP;
        auto approx = [&](int idx, int ad) {   // (two passes instead of sixteen live values)
            const float e = 3.0102999566f * __builtin_amdgcn_logf(ft * __builtin_amdgcn_rcpf((float)ad));   // 10 log10(totDP / currAD)
            return (float)ad * ((float)(max_qual - idx) - e);
        };
        float amax = -3.0e38f;
        int ad = 0;
#pragma unroll
        for (int idx = 0; idx < NBUCKETS; idx++) if (idx < max_qual && h[idx] != 0) { ad += h[idx]; amax = fmaxf(amax, approx(idx, ad)); }
        ad = 0;
#pragma unroll
        for (int idx = 0; idx < NBUCKETS; idx++) if (idx < max_qual && h[idx] != 0) {
            ad += h[idx];
            const float a = approx(idx, ad);
            if (!small || (a >= amax - 17.0f && a > -16.0f)) cand |= 1u << idx;   // (values <= 0 never win: maxvqual starts at 0)
        }
    }
    int currAD = 0;
    maxvqual = 0; argmaxAD = 0; argmaxBQ = 0;
#pragma unroll
    for (int idx = 0; idx < NBUCKETS; idx++) {
        if (idx < max_qual) currAD += h[idx];
        if ((cand >> idx) & 1u) {
            const int currBQ = max_qual - idx;
            const double expBQ = 10.0 / log(10.0) * log(((double)totDP / (double)currAD) + 2.220446049250313e-16);
            const int currvqual = (int)(currAD * (currBQ - expBQ));
            if (currvqual > maxvqual) { argmaxAD = currAD; argmaxBQ = currBQ; maxvqual = currvqual; }
        }
    }
}

#endif
