// TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT (see oracle_common.hpp header).
// C ABI of the oracle (liboracle.so), shaped exactly like include/uvcgpu.h so that the same test
// harness drives the oracle and the HIP library.  Symbols are prefixed uvc_oracle_.
#include "oracle_common.hpp"

using namespace uvco;

namespace uvco { i32 oracle_sscs_phred(const UvcParams &P, int con_symbol, int alt_symbol);
const char *score_trace_names();
void test_segbias(State &S, bool isGap, i32 bq, i32 rpos, int sym, i32 a_pos, i32 a_endpos, i32 a_mpos, i32 a_isize, i32 a_flag, i32 a_mapq,
                  i32 xm1500, i32 bm1500, int cigar_op, i32 indel_len, i32 dist, int dflag, i32 clip_cnt, const i32 *thres, i64 *out); }

static thread_local std::string g_err;

extern "C" {

const char *uvc_oracle_last_error(void) { return g_err.c_str(); }

void uvc_oracle_params_default(UvcParams *p) {
    memset(p, 0, sizeof(*p));
    p->struct_size = (int32_t)sizeof(UvcParams);
#define UVC_PI(name, dflt) p->name = (int32_t)(dflt);
#define UVC_PD(name, dflt) p->name = (double)(dflt);
#include "uvc_params.def"
#undef UVC_PI
#undef UVC_PD
}

int uvc_oracle_create(void **out, const UvcParams *params, int32_t tid, int32_t beg, int32_t end, const char *refseq) {
    if (!out || !params || !refseq || end <= beg || params->struct_size != (int32_t)sizeof(UvcParams)) { g_err = "bad argument"; return UVCGPU_EINVAL; }
    State *S = new State();
    S->tid = tid; S->beg = beg; S->end = end + 1; S->npos = (i64)end - beg + 1;   // Symbol2CountCoverageSet(tid, beg, end + 1), main.cpp:569
    S->refstring.assign(refseq, (size_t)(end - beg));
    S->P = *params;
    build_side_arrays(*S);
    *out = S;
    return 0;
}

int uvc_oracle_set_reads(void *h, const UvcReadSoA *r) {
    State &S = *(State *)h;
    if (!r || r->n_reads < 0) { g_err = "bad reads"; return UVCGPU_EINVAL; }
    if (r->struct_size != (int32_t)sizeof(UvcReadSoA)) { g_err = "UvcReadSoA::struct_size mismatch"; return UVCGPU_EINVAL; }
    S.bases.assign(r->bases, r->bases + r->n_bases);
    S.quals.assign(r->quals, r->quals + r->n_bases);
    S.cigars.assign(r->cigars, r->cigars + r->n_cigar_ops);
    S.alns.clear(); S.frags.clear(); S.fams.clear();
    S.fams.resize(r->n_fams);
    for (auto &f : S.fams) { f.fs[0].frag_beg = f.fs[0].frag_end = f.fs[1].frag_beg = f.fs[1].frag_end = 0; f.dflag = 0; }
    for (int i = 0; i < r->n_fams; i++) S.fams[i].dflag = r->fam_dflag[i];
    int prev_fam = -1, prev_strand = -1, prev_frag = -1;
    std::vector<char> seen((size_t)r->n_fams * 2, 0);
    for (i64 i = 0; i < r->n_reads; i++) {
        Aln a;
        a.pos = r->pos[i]; a.mpos = r->mpos[i]; a.isize = r->isize[i]; a.flag = r->flag[i]; a.mapq = r->mapq[i]; a.nm = r->nm[i];
        a.l_qseq = r->l_qseq[i]; a.n_cigar = r->n_cigar[i];
        if (r->seq_off[i] < 0 || r->seq_off[i] + a.l_qseq > r->n_bases || r->cigar_off[i] < 0 || r->cigar_off[i] + a.n_cigar > r->n_cigar_ops) { g_err = "read offsets out of range"; return UVCGPU_EINVAL; }
        a.bases = S.bases.data() + r->seq_off[i]; a.quals = S.quals.data() + r->seq_off[i]; a.cigar = S.cigars.data() + r->cigar_off[i];
        a.frag = r->frag_id[i]; a.fam = r->fam_id[i]; a.strand = r->fam_strand[i];
        i32 e = a.pos; i64 q = 0;   // bam_endpos: pos + sum of reference-consuming op lengths (htslib 1.11 sam.c; SAM spec)
        for (int k = 0; k < a.n_cigar; k++) { int op = cig_op(a.cigar[k]); i32 len = (i32)cig_len(a.cigar[k]);
            if (op == C_MATCH || op == C_DEL || op == C_REF_SKIP || op == C_EQUAL || op == C_DIFF) e += len;
            if (op == C_MATCH || op == C_INS || op == C_SOFT_CLIP || op == C_EQUAL || op == C_DIFF) q += len; }
        if (e == a.pos) e = a.pos + 1;   // bam_endpos returns pos + 1 for reads without reference-consuming ops
        a.endpos = e;
        if (q != a.l_qseq) { g_err = "CIGAR query length != l_qseq"; return UVCGPU_EINVAL; }
        if (a.fam < 0 || a.fam >= r->n_fams || a.strand > 1) { g_err = "fam_id / fam_strand out of range"; return UVCGPU_EINVAL; }
        if (a.pos < S.beg || a.endpos > S.end - 1) { g_err = "read outside region"; return UVCGPU_EINVAL; }
        const bool new_fs = (a.fam != prev_fam || a.strand != prev_strand);
        if (new_fs) {
            if (seen[(size_t)a.fam * 2 + a.strand]) { g_err = "reads of one (fam_id, fam_strand) are not contiguous"; return UVCGPU_EINVAL; }
            seen[(size_t)a.fam * 2 + a.strand] = 1;
            S.fams[a.fam].fs[a.strand].frag_beg = (int)S.frags.size();
        }
        if (new_fs || a.frag != prev_frag) { Frag f; f.aln_beg = (int)S.alns.size(); f.aln_end = f.aln_beg; S.frags.push_back(f); }
        S.alns.push_back(a);
        S.frags.back().aln_end = (int)S.alns.size();
        S.fams[a.fam].fs[a.strand].frag_end = (int)S.frags.size();
        prev_fam = a.fam; prev_strand = a.strand; prev_frag = a.frag;
    }
    S.accumulated = false;
    return 0;
}

// apply_bq_err_correction3, grouping.cpp:459-543, on the oracle's copy of the reads.  The reference works on BAM 4-bit base codes
// (A=1, C=2, G=4, T=8, N=15) with "no base yet" = 0; the SoA holds 0..4, mapped here before the comparisons.
static void correct_bq_one(u8 *q, const u8 *b, int l, int flag, const u32 *cigar, int n_cigar, int bq_max, int bq_inc) {
    if ((0 == l) || (flag & 0x4)) return;
    auto code = [&](int i) -> int { const int v = b[i]; return v < 4 ? (1 << v) : 15; };
    for (int i = 0; i < l; i++) { const int bq = q[i]; q[i] = (u8)std::min(bq + bq_inc, bq_max); }   // :462-465
    const int isrc = ((flag & 0x10) ? 1 : 0);
    int inclu_beg_poss[2] = { 0, l - 1 };
    int exclu_end_poss[2] = { l, 0 - 1 };
    int end_clip_len = 0;
    if (n_cigar > 0) {                                                                               // :473-491
        u32 c1 = cigar[0];
        if (cig_op(c1) == C_SOFT_CLIP) {
            if (0 == isrc) inclu_beg_poss[0] += (int)cig_len(c1);
            else { exclu_end_poss[1] += (int)cig_len(c1); end_clip_len = (int)cig_len(c1); }
        }
        c1 = cigar[n_cigar - 1];
        if (cig_op(c1) == C_SOFT_CLIP) {
            if (1 == isrc) inclu_beg_poss[1] -= (int)cig_len(c1);
            else { exclu_end_poss[0] -= (int)cig_len(c1); end_clip_len = (int)cig_len(c1); }
        }
    }
    const int pos_incs[2] = { 1, -1 };
    const int inc = pos_incs[isrc], ibeg = inclu_beg_poss[isrc], eend = exclu_end_poss[isrc];
    // no aligned base left between the clips: the reference's loops would leave the arrays (undefined behaviour); skipped, as in the HIP kernel
    if ((isrc ? (ibeg <= eend) : (ibeg >= eend)) || ibeg < 0 || ibeg >= l || eend < -1 || eend > l) return;
    {                                                                                                // :494-522
        int prev_b = 0; unsigned distinct_cnt = 0;
        int termpos = eend - inc;
        for (; termpos != ibeg - inc; termpos -= inc) {
            const int bb = code(termpos); const int qq = q[termpos];
            if (bb != prev_b && qq >= 20) { prev_b = bb; distinct_cnt += 1; if (2 == distinct_cnt) break; }
        }
        const int homopol_tracklen = abs(termpos - (eend - inc));
        const int tail_penal = (end_clip_len >= 20 ? 1 : 0) + (homopol_tracklen >= 15 ? 2 : (homopol_tracklen >= 10 ? 1 : 0));
        if (tail_penal > 0)
            for (int pos = eend - inc; pos != (ibeg - inc) && pos != termpos; pos -= inc) q[pos] = (u8)(std::max((int)q[pos], tail_penal + 1) - tail_penal);
    }
    {                                                                                                // :523-539
        int homopol_len = 0, prev_b = 0;
        for (int pos = ibeg; pos != eend; pos += inc) {
            const int bb = code(pos);
            if (bb == prev_b) { homopol_len++; if (homopol_len >= 4 && bb == 4) q[pos] = (u8)(std::max((int)q[pos], 1 + 1) - 1); }
            else { prev_b = bb; homopol_len = 1; }
        }
    }
}

int uvc_oracle_region_correct_bq(void *h) {
    State &S = *(State *)h;
    for (auto &a : S.alns) correct_bq_one(S.quals.data() + (a.quals - S.quals.data()), a.bases, a.l_qseq, a.flag, a.cigar, a.n_cigar, S.P.assay_sequencing_BQ_max, S.P.assay_sequencing_BQ_inc);
    S.accumulated = false;
    return 0;
}

int uvc_oracle_region_read_quals(void *h, uint8_t *dst, int64_t n) {
    State &S = *(State *)h;
    if ((size_t)n != S.quals.size()) { g_err = "n must equal n_bases"; return UVCGPU_EINVAL; }
    memcpy(dst, S.quals.data(), (size_t)n);
    return 0;
}

// test hook: one dealwith_segbias call (args[17] = isGap bq rpos sym pos endpos mpos isize flag mapq xm1500 bm1500 cigar_op indel_len dist dflag clip_cnt)
int uvc_oracle_test_segbias(void *h, const int32_t *args, const int32_t *thres, int64_t *out) {
    State &S = *(State *)h;
    if (args[2] < S.beg || args[2] >= S.end || args[4] < S.beg || args[5] > S.end || args[5] <= args[4]) { g_err = "segbias hook: positions outside the region"; return UVCGPU_EINVAL; }
    test_segbias(S, args[0] != 0, args[1], args[2], args[3], args[4], args[5], args[6], args[7], args[8], args[9], args[10], args[11], args[12], args[13], args[14], args[15], args[16], thres, out);
    return 0;
}
int uvc_oracle_accumulate(void *h) { State &S = *(State *)h; return accumulate(S, g_err); }

// transposes the position-major internal storage into the plane layout [field][symbol][pos] of uvcgpu.h
static const void *group_ptr(State &S, int g, i64 &bytes, std::vector<i32> &tmp32, std::vector<i64> &tmp64) {
    const i64 n = S.npos;
    auto t32 = [&](size_t planes) { tmp32.assign(planes * (size_t)n, 0); bytes = (i64)tmp32.size() * 4; };
    auto t64 = [&](size_t planes) { tmp64.assign(planes * (size_t)n, 0); bytes = (i64)tmp64.size() * 8; };
    switch (g) {
        case UVC_F_PREP32: t32(UVC_NPREP32); for (i64 i = 0; i < n; i++) for (int f = 0; f < UVC_NPREP32; f++) tmp32[(size_t)f * n + i] = S.p32(f, i); return tmp32.data();
        case UVC_F_PREP64: t64(UVC_NPREP64); for (i64 i = 0; i < n; i++) for (int f = 0; f < UVC_NPREP64; f++) tmp64[(size_t)f * n + i] = S.p64(f, i); return tmp64.data();
        case UVC_F_THRES: t32(UVC_NTHRES); for (i64 i = 0; i < n; i++) for (int f = 0; f < UVC_NTHRES; f++) tmp32[(size_t)f * n + i] = S.th(f, i); return tmp32.data();
        case UVC_F_SEG32: t32((size_t)UVC_NSEG32 * NSYM); for (i64 i = 0; i < n; i++) for (int s = 0; s < NSYM; s++) for (int f = 0; f < UVC_NSEG32; f++) tmp32[((size_t)f * NSYM + s) * n + i] = S.s32(f, s, i); return tmp32.data();
        case UVC_F_SEG64: t64((size_t)UVC_NSEG64 * NSYM); for (i64 i = 0; i < n; i++) for (int s = 0; s < NSYM; s++) for (int f = 0; f < UVC_NSEG64; f++) tmp64[((size_t)f * NSYM + s) * n + i] = S.s64(f, s, i); return tmp64.data();
        case UVC_F_VQ: t32((size_t)UVC_NVQ * NSYM); for (i64 i = 0; i < n; i++) for (int s = 0; s < NSYM; s++) for (int f = 0; f < UVC_NVQ; f++) tmp32[((size_t)f * NSYM + s) * n + i] = S.VQ(f, s, i); return tmp32.data();
        case UVC_F_BQSUM: t32(NSYM); for (i64 i = 0; i < n; i++) for (int s = 0; s < NSYM; s++) tmp32[(size_t)s * n + i] = S.BQS(s, i); return tmp32.data();
        case UVC_F_FRAG: t32((size_t)2 * UVC_NFRAG * NSYM); for (i64 i = 0; i < n; i++) for (int d = 0; d < 2; d++) for (int s = 0; s < NSYM; s++) for (int f = 0; f < UVC_NFRAG; f++) tmp32[(((size_t)d * UVC_NFRAG + f) * NSYM + s) * n + i] = S.FR(d, f, s, i); return tmp32.data();
        case UVC_F_FAM: t32((size_t)2 * UVC_NFAM * NSYM); for (i64 i = 0; i < n; i++) for (int d = 0; d < 2; d++) for (int s = 0; s < NSYM; s++) for (int f = 0; f < UVC_NFAM; f++) tmp32[(((size_t)d * UVC_NFAM + f) * NSYM + s) * n + i] = S.FA(d, f, s, i); return tmp32.data();
        case UVC_F_FAMINFO32: t32((size_t)UVC_NFAMINFO32 * NSYM); for (i64 i = 0; i < n; i++) for (int s = 0; s < NSYM; s++) for (int f = 0; f < UVC_NFAMINFO32; f++) tmp32[((size_t)f * NSYM + s) * n + i] = S.FI(f, s, i); return tmp32.data();
        case UVC_F_FAMINFO64: t64((size_t)UVC_NFAMINFO64 * NSYM); for (i64 i = 0; i < n; i++) for (int s = 0; s < NSYM; s++) for (int f = 0; f < UVC_NFAMINFO64; f++) tmp64[((size_t)f * NSYM + s) * n + i] = S.FI64(f, s, i); return tmp64.data();
        case UVC_F_DUPLEX: t32((size_t)UVC_NDUPLEX * NSYM); for (i64 i = 0; i < n; i++) for (int s = 0; s < NSYM; s++) for (int f = 0; f < UVC_NDUPLEX; f++) tmp32[((size_t)f * NSYM + s) * n + i] = S.DU(f, s, i); return tmp32.data();
        case UVC_F_RTR: {
            t32(UVC_NRTR);
            for (i64 i = 0; i < n; i++) { const Rtr &r = S.rtr[i];
                const i32 v[UVC_NRTR] = { r.begpos, r.tracklen, r.unitlen, r.indelphred, r.anyTR_begpos, r.anyTR_tracklen, r.anyTR_unitlen };
                for (int f = 0; f < UVC_NRTR; f++) tmp32[(size_t)f * n + i] = v[f]; }
            return tmp32.data(); }
        case UVC_F_BAQ: t64(2); for (i64 i = 0; i < n; i++) { tmp64[i] = S.baq[i]; tmp64[n + i] = S.baq2[i]; } return tmp64.data();
    }
    bytes = -1; return NULL;
}

int64_t uvc_oracle_field_bytes(void *h, int32_t g) {
    State &S = *(State *)h; i64 b; std::vector<i32> t32; std::vector<i64> t64;
    if (g != UVC_F_RTR && g != UVC_F_BAQ && !S.accumulated) return -1;
    group_ptr(S, g, b, t32, t64); return b;
}
int uvc_oracle_fetch(void *h, int32_t g, void *dst, int64_t dst_bytes) {
    State &S = *(State *)h; i64 b; std::vector<i32> t32; std::vector<i64> t64;
    if (g != UVC_F_RTR && g != UVC_F_BAQ && !S.accumulated) { g_err = "fetch before accumulate"; return UVCGPU_ESTATE; }
    const void *p = group_ptr(S, g, b, t32, t64);
    if (!p || b != dst_bytes) { g_err = "bad field group or size"; return UVCGPU_EINVAL; }
    memcpy(dst, p, (size_t)b); return 0;
}

int uvc_oracle_score(void *h, const UvcScoreRequest *req, UvcScoreOut *out) {
    State &S = *(State *)h;
    std::vector<std::vector<i32>> recs;
    int rc = score(S, req, recs, g_err);
    if (rc) return rc;
    out->n_records = (i64)recs.size();
    if (out->n_records > out->capacity) { g_err = "score output capacity too small"; return UVCGPU_ENOMEM; }
    for (i64 i = 0; i < out->n_records; i++) for (int f = 0; f < UVC_NUM_SCORE_FIELDS; f++) out->fields[(size_t)f * out->capacity + i] = recs[i][f];
    return 0;
}

// test hook: the inputs of calc_DPv / calc_qual of every record the request scores, in record order (names: uvc_oracle_score_trace_names,
// ';'-separated); *n_values = records x names.  UVCGPU_ENOMEM with *n_values set when `cap` is too small.
const char *uvc_oracle_score_trace_names(void) { return score_trace_names(); }
int uvc_oracle_score_trace(void *h, const UvcScoreRequest *req, double *buf, int64_t cap, int64_t *n_values, double *buf2 /* [records][6] or NULL */) {
    State &S = *(State *)h;
    std::vector<double> t, t2; std::vector<std::vector<i32>> recs;
    S.trace = &t; S.trace2 = &t2;
    const int rc = score(S, req, recs, g_err);
    S.trace = nullptr; S.trace2 = nullptr;
    if (rc) return rc;
    *n_values = (i64)t.size();
    if ((i64)t.size() > cap) { g_err = "trace capacity too small"; return UVCGPU_ENOMEM; }
    if (!t.empty()) memcpy(buf, t.data(), t.size() * sizeof(double));
    if (buf2 && !t2.empty()) memcpy(buf2, t2.data(), t2.size() * sizeof(double));   // 6 per record, same record order
    return 0;
}

int uvc_oracle_region_indel_alleles(void *h, UvcGapRow *rows, int64_t row_capacity, int64_t *n_rows, uint8_t *seq, int64_t seq_capacity, int64_t *seq_bytes) {
    State &S = *(State *)h;
    if (!S.accumulated) { g_err = "indel_alleles before accumulate"; return UVCGPU_ESTATE; }
    std::vector<UvcGapRow> r; std::vector<u8> q;
    indel_allele_rows(S, r, q);
    if (n_rows) *n_rows = (i64)r.size();
    if (seq_bytes) *seq_bytes = (i64)q.size();
    if ((i64)r.size() > row_capacity || (i64)q.size() > seq_capacity) { g_err = "allele table capacity too small"; return UVCGPU_ENOMEM; }
    if (!r.empty()) memcpy(rows, r.data(), r.size() * sizeof(UvcGapRow));
    if (!q.empty()) memcpy(seq, q.data(), q.size());
    return 0;
}

// The text of the records the request writes: per record "<CHROM..INFO> \x1e <tier2 flag> \x1e <FORMAT field lines>", records joined by \x1d.
// The test streams the field lines through the reference's own streamAppendBcfFormat (oracle/_ref/libref_vcf.so).
int uvc_oracle_region_vcf(void *h, const UvcScoreRequest *req, const char *tname, char *dst, int64_t cap, int64_t *len) {
    State &S = *(State *)h;
    VcfSink sink; sink.tname = tname;
    S.vcf_sink = &sink;
    std::vector<std::vector<i32>> recs;
    int rc = score(S, req, recs, g_err);
    S.vcf_sink = nullptr;
    if (rc) return rc;
    std::string out;
    for (size_t i = 0; i < sink.fixed.size(); i++) { if (i) out += '\x1d'; out += sink.fixed[i]; out += '\x1e'; out += (sink.tier2[i] < 0 ? "-1" : sink.tier2[i] ? "1" : "0"); out += '\x1e'; out += sink.spec[i]; }
    *len = (int64_t)out.size();
    if (!dst || cap < (int64_t)out.size()) { g_err = "destination too small"; return UVCGPU_ENOMEM; }
    memcpy(dst, out.data(), out.size());
    return 0;
}

// hap_bq / hap_fq / hap_f2q of updateByRegion3Aln, as uvcgpu_region_hap_links returns them
int uvc_oracle_region_hap_links(void *h, UvcHapLink *links, int64_t link_capacity, int64_t *n_links, int32_t *muts, int64_t mut_capacity, int64_t *n_mut_ints) {
    State &S = *(State *)h;
    if (!S.accumulated) { g_err = "haplotype links before accumulate"; return UVCGPU_ESTATE; }
    int64_t nl = 0, nm = 0;
    for (int w = 0; w < 3; w++) for (const State::HapLink &l : S.haplinks[w]) { nl++; nm += 2 * (int64_t)l.form.size(); }
    if (n_links) *n_links = nl;
    if (n_mut_ints) *n_mut_ints = nm;
    if (nl > link_capacity || nm > mut_capacity) { g_err = "haplotype link capacity too small"; return UVCGPU_ENOMEM; }
    int64_t li = 0, mi = 0;
    for (int w = 0; w < 3; w++) for (const State::HapLink &l : S.haplinks[w]) {
        UvcHapLink &o = links[li++];
        o.which = w; o.n_muts = (int32_t)l.form.size(); o.mut_off = mi; o.fr_cnt[0] = l.fr[0]; o.fr_cnt[1] = l.fr[1]; o.other_cnt[0] = l.other[0]; o.other_cnt[1] = l.other[1];
        for (const auto &ps : l.form) { muts[mi++] = ps.first; muts[mi++] = ps.second; }
    }
    return 0;
}

void uvc_oracle_destroy(void *h) { delete (State *)h; }

// ---- unit-test hooks for the math primitives (tests/test_oracle_math.py) ----
double uvc_oracle_calc_binom_10log10_likeratio(double prob, double a, double b, int bidirectional, int set_max_prob_to_one) {
    return calc_binom_10log10_likeratio(prob, a, b, bidirectional != 0, set_max_prob_to_one != 0);
}
double uvc_oracle_prob2odds(double p) { return p / (1.0 - p); }          // main_conversion.hpp:191-196
double uvc_oracle_odds2prob(double odds) { return odds / (odds + 1.0); } // main_conversion.hpp:198-203
void uvc_oracle_dp4_to_pcFA(double *out2, int bidir, int overseq_disabled, double overseq_frac, double aADpass, double aADfail, double aDPpass, double aDPfail,
                            double pl_exponent, double n_nats, double aADavgKeyVal, double aDPavgKeyVal, double priorAD, double priorDP) {
    dp4_to_pcFA(out2, bidir != 0, overseq_disabled != 0, overseq_frac, aADpass, aADfail, aDPpass, aDPfail, pl_exponent, n_nats, aADavgKeyVal, aDPavgKeyVal, priorAD, priorDP);
}
void uvc_oracle_infer_max_qual(int32_t *out3, int32_t max_qual, int32_t dec_qual, const int32_t *distr16, int32_t totDP) {
    infer_max_qual_assuming_independence(out3[0], out3[1], out3[2], max_qual, dec_qual, distr16, totDP);
}
int32_t uvc_oracle_indel_phred(double ampfact, int32_t rs, int32_t rn) { return indel_phred(ampfact, rs, rn); }
int32_t uvc_oracle_indel_len_rusize_phred(int32_t len, int32_t rs) { return indel_len_rusize_phred(len, rs); }
int32_t uvc_oracle_sscs_phred(const UvcParams *p, int32_t con, int32_t alt) { return oracle_sscs_phred(*p, con, alt); }

}  // extern "C"
