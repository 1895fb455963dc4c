// TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/.
//
// CPU restatement of the family-assignment arithmetic of genetronhealth/uvc: a sequential walk with std::map /
// std::set exactly as bamfname_to_strand_to_familyuid_to_reads does it (grouping.cpp:608-997), minus htslib: the alignments arrive as
// plain columns, read names and UMIs as the hash pairs of include/uvcgroup.h.
// PARITY UNPINNED: the reference has no test or golden vector for this path and cannot be built here (htslib is absent); the
// restatement is pinned only by the parameter defaults (tests/test_params.py) and by independent re-implementations in tests/.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <tuple>
#include <vector>
#include "uvcgpu.h"
#include "uvcgroup.h"

namespace {
const int MAX_INSERT = 2000;         // MAX_INSERT_SIZE, common.hpp:64
const int ARRPOS_MARGIN = MAX_INSERT, ARRPOS_OUTER_RANGE = 10, ARRPOS_INNER_RANGE = 3;   // grouping.cpp:22-24
typedef uint64_t u64;
inline int64_t nnminus(int64_t a, int64_t b) { return a > b ? a - b : 0; }

// Hash.hpp:6-39
u64 strnhash(const char *s, size_t n, u64 base) { u64 r = 0; for (size_t i = 0; i < n && s[i]; i++) r = r * base + (u64)s[i]; return r; }
u64 hash2hash(u64 a, u64 b) { return a * ((1UL << 31UL) - 1UL) + b; }

struct Pre { int reason; bool isrc, isr2; int tBeg, tEnd; int isize; };

// fill_isrc_isr2_beg_end_with_aln, grouping.cpp:347-415.  The call sites pass (kept_aln_min_aln_len, kept_aln_min_mapqual) into the
// parameters (min_mapqual, min_aln_len) -- swapped (grouping.cpp:672-673 vs :351-352); reproduced as called.
Pre prefilter(const UvcGroupParams &P, int flag, int mapq, int pos, int endpos, int mpos, int isize_raw) {
    Pre r; r.reason = UVC_FR_NOT_FILTERED; r.isrc = r.isr2 = false; r.tBeg = r.tEnd = 0;
    const int isize = (std::abs(isize_raw) >= MAX_INSERT ? 0 : isize_raw);   // NORM_INSERT_SIZE, common.hpp:75
    r.isize = isize;
    const int min_mapqual = P.kept_aln_min_aln_len, min_aln_len = P.kept_aln_min_mapqual;
    const bool merge = (P.pair_end_merge == 0);
    if (flag & 0x4) { r.reason = UVC_FR_NOT_MAPPED; return r; }
    if ((flag & 0x900) != 0) { r.reason = UVC_FR_NOT_PRIMARY_ALN; return r; }
    if (mapq < min_mapqual) { r.reason = UVC_FR_LOW_MAPQ; return r; }
    if ((endpos - pos) < min_aln_len) { r.reason = UVC_FR_LOW_ALN_LEN; return r; }
    if (0 == isize) { if (P.kept_aln_is_zero_isize_discarded) { r.reason = UVC_FR_ZERO_ISIZE; return r; } }
    else {
        if (std::abs(isize) < P.kept_aln_min_isize) { r.reason = UVC_FR_LOW_ISIZE; return r; }
        if (std::abs(isize) > P.kept_aln_max_isize) { r.reason = UVC_FR_HIGH_ISIZE; return r; }
    }
    r.isrc = ((flag & 0x10) == 0x10);
    r.isr2 = ((flag & 0x80) == 0x80 && (flag & 0x1) == 0x1);
    if (!merge) r.isr2 = false;
    const int begpos = pos, endp = endpos - 1;
    if (!merge || ((flag & 0x1) == 0) || (flag & 0x8) || (0 == isize) || (std::abs(isize) >= ARRPOS_MARGIN)) {
        r.tBeg = (r.isrc ? endp : begpos); r.tEnd = (r.isrc ? begpos : endp);
    } else {
        const int tBegP1 = std::min(begpos, mpos), tEndP1 = tBegP1 + std::abs(isize) - 1;
        const bool strand = (((flag & 0x81) == 0x81) ? ((flag & 0x20) != 0) : ((flag & 0x10) != 0));   // bam_get_strand, common.hpp:90
        r.tBeg = (strand ? tEndP1 : tBegP1); r.tEnd = (strand ? tBegP1 : tEndP1);
    }
    const int oB = std::min(r.tBeg, r.tEnd), oE = std::max(r.tBeg, r.tEnd);
    if (oB + (ARRPOS_MARGIN - ARRPOS_OUTER_RANGE) <= P.fetch_tbeg || P.fetch_tend - 1 + (ARRPOS_MARGIN - ARRPOS_OUTER_RANGE) <= oE) { r.reason = UVC_FR_OUT_OF_RANGE; return r; }
    if (P.end2end && !(oB <= P.fetch_tbeg && oE >= P.fetch_tend)) { r.reason = UVC_FR_NOT_END_TO_END; return r; }
    return r;
}

// poscounter_to_pos2pcenter, grouping.cpp:422-442
void pos2pcenter(std::vector<int> &center, const std::vector<int> &count, double mult) {
    for (int lo = ARRPOS_INNER_RANGE; lo < (int)count.size() - ARRPOS_INNER_RANGE; lo++) {
        const int locnt = count[lo];
        center[lo] = lo;
        int maxc = locnt;
        for (int hi = lo - ARRPOS_INNER_RANGE; hi < lo + ARRPOS_INNER_RANGE + 1; hi++) {
            const int hicnt = count[hi];
            const u64 d = (u64)(lo > hi ? lo - hi : hi - lo);
            if ((hicnt > maxc) && ((hicnt + 1) > (locnt + 1) * pow(mult, (double)d))) { center[lo] = hi; maxc = hicnt; }
        }
    }
}

// MolecularBarcode::createKey (MolecularID.hpp:20-52) with the strings replaced by their hash pairs (an absent UMI is the empty string: 0, 0)
typedef std::tuple<int, int, int, int, u64, u64, u64, u64, int, int> Key;
Key molecular_key(int begtid, int beg, int endtid, int end, u64 q31, u64 q17, u64 u31, u64 u17, int dflag, int idflag) {
    std::pair<int, int> bp(begtid, beg), ep(endtid, end), kb(-1, -1), ke(-1, -1);
    if (0x3 == (0x3 & idflag)) { kb = std::min(bp, ep); ke = std::max(bp, ep); }
    else if (0x1 & idflag) kb = bp;
    else if (0x2 & idflag) ke = ep;
    return Key(kb.first, kb.second, ke.first, ke.second, (0x4 & idflag) ? q31 : 0, (0x4 & idflag) ? q17 : 0, (0x8 & idflag) ? u31 : 0, (0x8 & idflag) ? u17 : 0, dflag, idflag);
}
}  // namespace

extern "C" {
// the family key as ten 64-bit values (test hook: checked against the reference's createKey / operator< in tests/test_ref_molid.py)
void uvc_oracle_molecular_key(int begtid, int beg, int endtid, int end, uint64_t q31, uint64_t q17, uint64_t u31, uint64_t u17, int dflag, int idflag, int64_t *out10) {
    const Key k = molecular_key(begtid, beg, endtid, end, q31, q17, u31, u17, dflag, idflag);
    out10[0] = std::get<0>(k); out10[1] = std::get<1>(k); out10[2] = std::get<2>(k); out10[3] = std::get<3>(k);
    out10[4] = (int64_t)std::get<4>(k); out10[5] = (int64_t)std::get<5>(k); out10[6] = (int64_t)std::get<6>(k); out10[7] = (int64_t)std::get<7>(k);
    out10[8] = std::get<8>(k); out10[9] = std::get<9>(k);
}

void uvc_oracle_group_params_default(UvcGroupParams *p) {
    memset(p, 0, sizeof(*p));
    p->struct_size = (int32_t)sizeof(UvcGroupParams);
#define UVC_GI(name, dflt) p->name = (int32_t)(dflt);
#define UVC_GD(name, dflt) p->name = (double)(dflt);
#include "uvc_group_params.def"
#undef UVC_GI
#undef UVC_GD
    p->inferred_sequencing_platform = 1;
}
uint64_t uvc_oracle_strnhash(const char *s, size_t n, uint64_t base) { return strnhash(s, n, base); }
uint64_t uvc_oracle_hash2hash(uint64_t a, uint64_t b) { return hash2hash(a, b); }

// grouping.cpp:763-786
int uvc_oracle_qname_digest(const char *qname, int molecule_tag, int disable_duplex, uint64_t *q31, uint64_t *q17, uint64_t *u31, uint64_t *u17) {
    *q31 = strnhash(qname, SIZE_MAX, 31UL); *q17 = strnhash(qname, SIZE_MAX, 17UL);
    const size_t qname_len = strlen(qname);
    const char *umi_beg1 = strchr(qname, '#');
    const char *umi_beg = ((NULL != umi_beg1) ? (umi_beg1 + 1) : (qname + qname_len));
    const char *umi_end1 = strchr(umi_beg, '#');
    const char *umi_end = ((NULL != umi_end1) ? umi_end1 : (qname + qname_len));
    const int found = ((umi_beg + 1 < umi_end) && (1 /* MOLECULE_TAG_NONE */ != molecule_tag));
    *u31 = *u17 = 0;
    if (!found) return 0;
    const size_t umi_len = umi_end - umi_beg, umi_half = (umi_end - umi_beg - 1) / 2;
    *u31 = strnhash(umi_beg, umi_len, 31UL); *u17 = strnhash(umi_beg, umi_len, 17UL);
    const bool duplex = ((umi_len % 2 == 1) && ('+' == umi_beg[umi_half]) && (!disable_duplex));
    return 1 | (duplex ? 2 : 0);
}
int uvc_oracle_qname_digest_batch(const char *names, const int64_t *off, int64_t n, int molecule_tag, int disable_duplex, uint64_t *q31, uint64_t *q17, uint64_t *u31, uint64_t *u17, uint8_t *umi_kind) {
    for (int64_t i = 0; i < n; i++) umi_kind[i] = (uint8_t)uvc_oracle_qname_digest(names + off[i], molecule_tag, disable_duplex, &q31[i], &q17[i], &u31[i], &u17[i]);
    return 0;
}


// bam2umihash (grouping.cpp:569-606, called :787-792): single-end reads without a UMI in their name are searched for an in-read UMI
// pattern (environment ONE_STEP_UMI_STRUCT of the reference, main.cpp:1224-1225; letters as seq_nt16_table codes, N = any base = a UMI
// letter), forward at the first five offsets, then reverse-complemented from the read's end.  A hit sets the "UMI found" bit of umi_kind;
// the hash of the UMI letters is returned too, although the reference's family key never reads it (its umistring stays empty there:
// umi_beg / umi_len come from the read name, grouping.cpp:929).  Bases arrive as the codes of UvcBamBatch (0..3 = ACGT, 4 = anything
// else, which the reference holds as its 4-bit code: an ambiguity letter in the READ therefore only matches an N of the pattern).
static int oracle_nt16_of_char(char c) {   // seq_nt16_table of htslib (SAM specification, section 4.2.3: "=ACMGRSVTWYHKDBN")
    switch (c) { case '=': return 0; case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'M': case 'm': return 3; case 'G': case 'g': return 4;
                 case 'R': case 'r': return 5; case 'S': case 's': return 6; case 'V': case 'v': return 7; case 'T': case 't': return 8; case 'W': case 'w': return 9;
                 case 'Y': case 'y': return 10; case 'H': case 'h': return 11; case 'K': case 'k': return 12; case 'D': case 'd': return 13; case 'B': case 'b': return 14; default: return 15; }
}
int uvc_oracle_umi_in_read_batch(const char *umi_struct, const uint8_t *bases, const int64_t *seq_off, const int32_t *l_qseq, const uint16_t *flag, int64_t n,
                           uint8_t *umi_kind, uint64_t *umi_hash) {
    if (n < 0 || (n > 0 && (!bases || !seq_off || !l_qseq || !flag || !umi_kind))) return UVCGPU_EINVAL;
    if (!umi_struct || !*umi_struct) return 0;
    int pat[256]; int np = 0;
    for (const char *c = umi_struct; *c && np < 256; c++) pat[np++] = oracle_nt16_of_char(*c);
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

int uvc_oracle_group_families(const UvcGroupParams *Pp, const UvcGroupInput *in, UvcGroupOut *out) {
    const UvcGroupParams &P = *Pp;
    const int64_t n = in->n_alns;
    const int fetch_size = P.fetch_tend - P.fetch_tbeg + (ARRPOS_MARGIN + ARRPOS_OUTER_RANGE) * 2;
    std::array<std::vector<int>, 4> begc, endc;
    for (int c = 0; c < 4; c++) { begc[c].assign(fetch_size, 0); endc[c].assign(fetch_size, 0); }
    std::set<std::pair<u64, u64>> visited;
    std::vector<Pre> pre((size_t)n);
    for (int64_t i = 0; i < n; i++) {                                           // first scan, grouping.cpp:662-694
        const Pre r = pre[i] = prefilter(P, in->flag[i], in->mapq[i], in->pos[i], in->endpos[i], in->mpos[i], in->isize[i]);
        out->isize_norm[i] = r.isize;
        if (r.reason != UVC_FR_NOT_FILTERED) continue;
        const int c = r.isrc * 2 + r.isr2;
        const int bi = r.tBeg + ARRPOS_MARGIN - P.fetch_tbeg, ei = r.tEnd + ARRPOS_MARGIN - P.fetch_tbeg;
        if (bi >= 0 && bi < fetch_size) begc[c][bi] += 1;
        if (ei >= 0 && ei < fetch_size) endc[c][ei] += 1;
        const int mn = std::min(r.tBeg, r.tEnd), mx = std::max(r.tBeg, r.tEnd) + 2;
        if (!((mx <= P.fetch_tbeg) || (P.fetch_tend <= mn))) visited.insert(std::make_pair(in->qname_hash31[i], in->qname_hash17[i]));
    }
    std::array<std::vector<int64_t>, 4> border;                                // :696-705
    std::array<std::vector<int>, 4> b2c, e2c;
    for (int c = 0; c < 4; c++) {
        border[c].assign((size_t)fetch_size + 1, 0);
        int64_t bs = 0, es = 0;
        for (int i = 0; i < fetch_size; i++) { bs += begc[c][i]; es += endc[c][i]; border[c][i + 1] = bs + es; }
        b2c[c].assign(fetch_size, 0); e2c[c].assign(fetch_size, 0);
        pos2pcenter(b2c[c], begc[c], P.dedup_center_mult);
        pos2pcenter(e2c[c], endc[c], P.dedup_center_mult);
    }
    struct Fam { std::array<std::map<u64, std::vector<int64_t>>, 2> strands; int dflag, idflag; };
    std::map<Key, Fam> fams;
    out->extended_inclu_beg_pos = INT32_MAX; out->extended_exclu_end_pos = 0; out->n_amplicon = 0;
    for (int64_t i = 0; i < n; i++) {                                           // second scan, :732-985
        out->filter_reason[i] = pre[i].reason;
        if (in->pos[i] < nnminus(P.fetch_tbeg, MAX_INSERT + 1) || in->endpos[i] > (P.fetch_tend + MAX_INSERT + 1)) { out->filter_reason[i] = UVC_FR_NOT_IN_WINDOW; continue; }
        if (!visited.count(std::make_pair(in->qname_hash31[i], in->qname_hash17[i]))) { out->filter_reason[i] = UVC_FR_QNAME_NOT_VISITED; continue; }
        const Pre &r = pre[i];
        if (r.reason != UVC_FR_NOT_FILTERED) continue;
        out->extended_inclu_beg_pos = std::min(out->extended_inclu_beg_pos, in->pos[i]);
        out->extended_exclu_end_pos = std::max(out->extended_exclu_end_pos, in->endpos[i]);
        const int flag = in->flag[i], isize = r.isize;
        const bool is_umi_found = (in->umi_kind[i] & 1), is_duplex_found = (in->umi_kind[i] & 2);
        const int c = r.isrc * 2 + r.isr2;
        const int beg1 = r.tBeg + ARRPOS_MARGIN - P.fetch_tbeg, end1 = r.tEnd + ARRPOS_MARGIN - P.fetch_tbeg;
        const int beg2 = b2c[c][beg1], end2 = e2c[c][end1];
        const int64_t beg2count = begc[c][beg2], end2count = endc[c][end2];
        const int iL = std::min(beg2 + 6, end2), iR = std::max(beg2, (int)nnminus(end2, 6));
        const int64_t tot = border[c][iR] - border[c][iL];
        const double begratio = (double)(beg2count * (iR - iL) + 1) / (double)(tot + (iR - iL) + 1);
        const double endratio = (double)(end2count * (iR - iL) + 1) / (double)(tot + (iR - iL) + 1);
        const bool b_amp = (begratio > P.dedup_amplicon_border_to_insert_cov_weak_avgDP_ratio && (beg2count >= P.dedup_amplicon_border_weak_minDP) && (beg2count >= tot * P.dedup_amplicon_border_to_insert_cov_weak_totDP_ratio));
        const bool e_amp = (endratio > P.dedup_amplicon_border_to_insert_cov_weak_avgDP_ratio && (end2count >= P.dedup_amplicon_border_weak_minDP) && (end2count >= tot * P.dedup_amplicon_border_to_insert_cov_weak_totDP_ratio));
        const bool b_str = (begratio > P.dedup_amplicon_border_to_insert_cov_strong_avgDP_ratio && (beg2count >= P.dedup_amplicon_border_strong_minDP) && (beg2count >= tot * P.dedup_amplicon_border_to_insert_cov_strong_totDP_ratio));
        const bool e_str = (endratio > P.dedup_amplicon_border_to_insert_cov_strong_avgDP_ratio && (end2count >= P.dedup_amplicon_border_strong_minDP) && (end2count >= tot * P.dedup_amplicon_border_to_insert_cov_strong_totDP_ratio));
        const bool amplicon = (b_str || e_str || (b_amp && e_amp));
        out->n_amplicon += amplicon;
        int idflag = 0;                                                         // :857-886
        if (P.dedup_flag != 0) idflag = P.dedup_flag;
        else if (2 /* IONTORRENT */ == P.inferred_sequencing_platform) idflag = (is_umi_found ? 0x9 : (amplicon ? 0x7 : 0x3));
        else {
            if (is_umi_found) {
                if (b_str && e_amp && beg2count > end2count * P.dedup_amplicon_end2end_ratio) idflag = 0x9;
                else if (e_str && b_amp && end2count > beg2count * P.dedup_amplicon_end2end_ratio) idflag = 0xA;
                else idflag = 0xB;
            } else idflag = (amplicon ? 0x7 : 0x3);
        }
        const bool preserved = ((flag & 0x1) && (!(flag & 0x4)) && (!(flag & 0x8)) && (std::abs(isize) >= (MAX_INSERT * 3 / 4) || isize == 0));
        const int begtid = ((!(flag & 0x4)) ? in->tid[i] : (INT32_MAX - 1));
        const int endtid = (((flag & 0x1) && !(flag & 0x8)) ? in->mtid[i] : (INT32_MAX - 1));
        const int beg3 = (preserved ? in->pos[i] : (beg2 - ARRPOS_MARGIN + P.fetch_tbeg));
        const int end3 = (preserved ? in->mpos[i] : (end2 - ARRPOS_MARGIN + P.fetch_tbeg));
        const int strand = (((flag & 0x81) == 0x81) ? ((flag & 0x20) != 0) : ((flag & 0x10) != 0));
        const int dflag = (is_umi_found ? 0x1 : 0) + (is_duplex_found ? 0x2 : 0) + (amplicon ? 0x4 : 0) + (preserved ? 0x8 : 0);
        const Key key = molecular_key(begtid, beg3, endtid, end3, in->qname_hash31[i], in->qname_hash17[i], is_umi_found ? in->umi_hash31[i] : 0, is_umi_found ? in->umi_hash17[i] : 0, dflag, idflag);
        Fam &f = fams[key];
        f.dflag = dflag; f.idflag = idflag;
        f.strands[strand][in->qname_hash17[i]].push_back(i);
    }
    int64_t k = 0; int fam = 0, frag = 0;
    for (auto &kv : fams) {                                                     // alns3 order, grouping.cpp:545-566
        out->fam_dflag[fam] = (uint8_t)kv.second.dflag; out->fam_idflag[fam] = (uint8_t)kv.second.idflag;
        for (int s = 0; s < 2; s++) for (auto &fr : kv.second.strands[s]) {
            for (int64_t i : fr.second) { out->order[k] = (int32_t)i; out->fam_id[k] = fam; out->frag_id[k] = frag; out->fam_strand[k] = (uint8_t)s; k++; }
            frag++;
        }
        fam++;
    }
    out->n_kept = k; out->n_fams = fam; out->n_frags = frag; out->n_visited_qnames = (int64_t)visited.size();
    return 0;
}

}  // extern "C"
