// TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT (see oracle_common.hpp header).
//
// CPU restatement of the accumulate half of the hot path:
//   region side arrays   main.hpp:699-874, main.cpp:400-429
//   P1   update_seg_format_prep_sets_by_aln          main.hpp:924-1204
//   P1b  update_seg_format_thres_from_prep_sets      main.hpp:1206-1299
//   P2   updateByAln<SYMBOL_COUNT_SUM, bias=true> + dealwith_segbias   main.hpp:1360-1595, 1762-2296
//   P3   updateByAlns3UsingBQ body + P3b             main.hpp:2620-2830
//   P4/P5/P5b updateByAlns3UsingFQ                   main.hpp:2832-3594
// Same visiting order as the reference (family -> strand -> fragment -> alignment), scalar code.
// Out of scope here exactly as in SURVEY.md section 2: consensus FASTQ (C6), haplotype maps (a12).
#include "oracle_common.hpp"
#include <map>
#include <tuple>

namespace uvco {

// ------------------------------------------------------------------------------------------------
// math primitives
// ------------------------------------------------------------------------------------------------

// is_indel_context_more_STR, main.hpp:699-721
bool is_indel_context_more_STR(i32 rulen1, i32 rc1, i32 rulen2, i32 rc2, i32 indel_str_repeatsize_max) {
    if (rulen2 * rc2 == 0) return true;
    if (rulen1 > indel_str_repeatsize_max || rulen2 > indel_str_repeatsize_max) {
        return (rulen1 < rulen2 || (rulen1 == rulen2 && rc1 > rc2));
    }
    int rank1 = (rc1 <= 1 ? (-(int)rc1 * rulen1) : ((int)(rc1 - 1) * rulen1));
    int rank2 = (rc2 <= 1 ? (-(int)rc2 * rulen1) : ((int)(rc2 - 1) * rulen2));  // sic: rulen1 in the first arm (main.hpp:709)
    if (0 == rc1 || 0 == rulen1) rank1 = -100;
    if (0 == rc2 || 0 == rulen2) rank2 = -100;
    return rank1 > rank2;
}

// indelpos_to_context, main.hpp:733-755 (returns the unit length instead of the unit string)
void indelpos_to_context(i32 &repeatunit_len, i32 &max_repeatnum, const std::string &refstring, i32 refpos, i32 indel_str_repeatsize_max) {
    max_repeatnum = 0;
    repeatunit_len = 0;
    const i32 n = (i32)refstring.size();
    if (refpos >= n) return;
    i32 rs_at_max = 0;
    for (i32 rs = 1; rs <= indel_str_repeatsize_max; rs++) {
        i32 q = refpos;
        while ((q + rs < n) && refstring[q] == refstring[q + rs]) q++;
        i32 rn = (q - refpos) / rs + 1;
        if (is_indel_context_more_STR(rs, rn, rs_at_max, max_repeatnum, indel_str_repeatsize_max)) {
            max_repeatnum = rn;
            rs_at_max = rs;
        }
    }
    repeatunit_len = (i32)refstring.substr(refpos, rs_at_max).size();
}

// a_dp == 0 (an insertion right after the last aligned base of the only covering read) makes the reference convert
// round(-inf) to an integer and subtract 3 from it (main.hpp:2038-2044, 2147-2150): undefined behaviour, whose outcome
// depends on the compiler.  Both this restatement and the HIP path DEFINE that case as "no bonus" (a very negative phredinc).
static const i32 PHREDINC_NO_DEPTH = -1000000;

// indel_len_rusize_phred, main.hpp:757-790
i32 indel_len_rusize_phred(i32 indel_len, i32 repeatunit_size) {
    static const i32 n_units_to_phred[19] = { 0, 0, 3, 5, 6, 7, 8, 8, 9, 10, 10, 10, 11, 11, 11, 12, 12, 12, 13 };
    if (0 == (indel_len % repeatunit_size)) {
        i32 n_units = indel_len / repeatunit_size;
        return n_units_to_phred[min_(n_units, 18)];
    }
    return n_units_to_phred[min_(indel_len, 18)];
}

// indel_phred, main.hpp:794-801 (prob2phred = floor(-10 log10 p), main_conversion.hpp:890-893)
i32 indel_phred(double ampfact, i32 rs, i32 rn) {
    i32 region_size = rs * rn;
    double num_slips = (region_size > 64 ? (double)(region_size - 8) : log1p(exp((double)region_size - (double)8))) * ampfact / ((double)(rs * rs));
    double p = (1.0 - DBL_EPSILON) / (num_slips + 1.0);
    return (i32)floor(-10 * log(p) / log(10));
}

// infer_max_qual_assuming_independence, main_conversion.hpp:943-974
void infer_max_qual_assuming_independence(i32 &maxvqual, i32 &argmaxAD, i32 &argmaxBQ, i32 max_qual, i32 dec_qual, const i32 *qual_distr, i32 totDP) {
    i32 currvqual = 0, currAD = 0;
    maxvqual = 0; argmaxAD = 0; argmaxBQ = 0;
    for (i32 idx = 0; idx < min_(NBUCKETS, max_qual / dec_qual); idx++) {
        const i32 currQD = qual_distr[idx];
        if (0 == currQD) continue;
        currAD += currQD;
        i32 currBQ = max_qual - (dec_qual * idx);
        double expBQ = 10.0 / log(10.0) * log(((double)totDP / (double)currAD) + DBL_EPSILON);
        currvqual = (i32)(currAD * (currBQ - expBQ));
        if (currvqual > maxvqual) { argmaxAD = currAD; argmaxBQ = currBQ; maxvqual = currvqual; }
    }
}

// PhredMutationTable::toPhredErrRate, main.hpp:213-262
static i32 sscs_phred(const UvcParams &P, int con_symbol, int alt_symbol) {
    i32 raw;
    if (is_ins(con_symbol) || is_del(con_symbol)) raw = P.fam_phred_sscs_indel_open;
    else if (con_symbol == UVC_LINK_M) {
        if (UVC_LINK_D1 == alt_symbol || UVC_LINK_I1 == alt_symbol) raw = P.fam_phred_sscs_indel_open;
        else if (UVC_LINK_D2 == alt_symbol || UVC_LINK_I2 == alt_symbol) raw = P.fam_phred_sscs_indel_open + P.fam_phred_sscs_indel_ext * 1;
        else raw = P.fam_phred_sscs_indel_open + P.fam_phred_sscs_indel_ext * 2;
    } else if ((con_symbol == UVC_BASE_C && alt_symbol == UVC_BASE_T) || (con_symbol == UVC_BASE_G && alt_symbol == UVC_BASE_A)) raw = P.fam_phred_sscs_transition_CG_TA;
    else if ((con_symbol == UVC_BASE_A && alt_symbol == UVC_BASE_G) || (con_symbol == UVC_BASE_T && alt_symbol == UVC_BASE_C)) raw = P.fam_phred_sscs_transition_AT_GC;
    else if ((con_symbol == UVC_BASE_C && alt_symbol == UVC_BASE_A) || (con_symbol == UVC_BASE_G && alt_symbol == UVC_BASE_T)) raw = P.fam_phred_sscs_transversion_CG_AT;
    else raw = P.fam_phred_sscs_transversion_other;
    return raw + (P.tumor_vcf_fname_nonempty ? 3 : 0);   // all_mutation_inc, main.hpp:236 + :2564
}
i32 oracle_sscs_phred(const UvcParams &P, int con_symbol, int alt_symbol) { return sscs_phred(P, con_symbol, alt_symbol); }

// ------------------------------------------------------------------------------------------------
// region side arrays (C10)
// ------------------------------------------------------------------------------------------------
static int char_to_symbol(char c) {  // CHAR_TO_SYMBOL, main_conversion.hpp:473-488 (base letters only)
    switch (c) { case 'A': case 'a': return UVC_BASE_A; case 'C': case 'c': return UVC_BASE_C;
                 case 'G': case 'g': return UVC_BASE_G; case 'T': case 't': return UVC_BASE_T;
                 case 'I': case 'i': return UVC_LINK_M; case '-': case '_': return UVC_LINK_D1; default: return UVC_BASE_N; }
}

void build_side_arrays(State &S) {
    const UvcParams &P = S.P;
    const std::string &ref = S.refstring;
    const i32 n = (i32)ref.size();
    S.refsym.assign(S.npos + 1, 0);
    for (i32 i = 0; i < n; i++) S.refsym[i] = (u8)char_to_symbol(ref[i]);
    // refstring2repeatvec, main.hpp:803-874
    std::vector<Rtr> v(n);
    for (auto &r : v) r.indelphred = P.indel_BQ_max;
    const i32 strmax = P.indel_str_repeatsize_max, vntrmax = P.indel_vntr_repeatsize_max;
    for (i32 refpos = 0; refpos < n;) {
        i32 rs_at_max = 0, max_rn = 0, repeat_endpos = refpos;
        i32 a_rs_at_max = 0, a_max_rn = 0, a_repeat_endpos = refpos;
        for (i32 rs = 1; rs <= vntrmax; rs++) {
            i32 q = refpos;
            while (q + rs < n && ref[q] == ref[q + rs]) q++;
            i32 rn = (q - refpos) / rs + 1;
            if (rs <= strmax && is_indel_context_more_STR(rs, rn, rs_at_max, max_rn, strmax)) { rs_at_max = rs; max_rn = rn; repeat_endpos = q + rs; }
            if (is_indel_context_more_STR(rs, rn, a_rs_at_max, a_max_rn, vntrmax)) { a_rs_at_max = rs; a_max_rn = rn; a_repeat_endpos = q + rs; }
        }
        {
            i32 tl = min_(repeat_endpos, n) - refpos;
            const i32 decphred = indel_phred(P.indel_polymerase_slip_rate * P.indel_del_to_ins_err_ratio, rs_at_max, tl / rs_at_max);
            for (i32 i = refpos; i != min_(repeat_endpos, n); i++) {
                if (tl > v[i].tracklen) {
                    v[i].begpos = refpos; v[i].tracklen = tl; v[i].unitlen = rs_at_max;
                    v[i].indelphred = P.indel_BQ_max - min_(P.indel_BQ_max - 1, decphred);
                }
            }
        }
        {
            i32 atl = min_(a_repeat_endpos, n) - refpos;
            for (i32 i = refpos; i != min_(a_repeat_endpos, n); i++) {
                if (atl > v[i].anyTR_tracklen) { v[i].anyTR_begpos = refpos; v[i].anyTR_tracklen = atl; v[i].anyTR_unitlen = a_rs_at_max; }
            }
        }
        const i32 nbases_to_next = strmax + rs_at_max;
        refpos += max_(rs_at_max * max_rn, nbases_to_next + 1) - nbases_to_next;
    }
    v.push_back(v.back());   // main.hpp:872
    S.rtr = v;
    // region_repeatvec_to_baq_offsetarr<false/true>, main.cpp:400-429
    for (int any = 0; any < 2; any++) {
        std::vector<i64> &out = any ? S.baq2 : S.baq;
        out.assign(S.npos, 0);
        i64 prefix = 0;
        for (i64 i = 0; i < S.npos; i++) {
            const Rtr &r = S.rtr[i];
            const i32 tl2 = any ? r.anyTR_tracklen : r.tracklen;
            if (tl2 / r.unitlen >= 3 || (tl2 / r.unitlen >= 2 && tl2 >= (i32)round(P.indel_polymerase_size))) {
                prefix += (P.indel_str_phred_per_region * 10) / tl2 + 1;
            } else {
                prefix += P.indel_nonSTR_phred_per_base * 10;
            }
            out[i] = prefix;
        }
        for (i64 i = 0; i < S.npos; i++) out[i] /= 10;
    }
}

// ------------------------------------------------------------------------------------------------
// P1: update_seg_format_prep_sets_by_aln, main.hpp:924-1204
// ------------------------------------------------------------------------------------------------
static int p1_prep_by_aln(State &S, const Aln &a, int dflag, std::string &err) {
    const UvcParams &P = S.P;
    const i32 off = S.beg;
    const i32 rend = a.endpos;
    i32 nge_cnt = 0, ngo_cnt = 0, insbaq_sum = 0, delbaq_sum = 0, inslen_sum = 0, dellen_sum = 0;
    i32 qpos = 0, rpos = a.pos;
    auto BAQ = [&](i64 p) -> i64 { return S.baq[p - off]; };
    const i64 baq_last = S.end - 1;   // baq_offsetarr.getExcluEndPosition() - 1
    for (i32 i = 0; i < a.n_cigar; i++) {
        const int op = cig_op(a.cigar[i]); const i32 len = (i32)cig_len(a.cigar[i]);
        if (C_INS == op || C_DEL == op) {
            nge_cnt += len; ngo_cnt++;
            if (C_INS == op) {
                insbaq_sum += (i32)(BAQ(min_((i64)rpos + len, baq_last)) - BAQ(rpos));
                inslen_sum += len; qpos += len;
            } else {
                delbaq_sum += (i32)(BAQ(min_((i64)rpos + len, baq_last)) - BAQ(rpos));
                dellen_sum += len; rpos += len;
            }
        } else if (op == C_MATCH || op == C_EQUAL || op == C_DIFF) { qpos += len; rpos += len; }
        else if (op == C_REF_SKIP) rpos += len;
        else if (op == C_SOFT_CLIP) qpos += len;
        else if (op == C_HARD_CLIP || op == C_PAD) {}
        else { err = "unsupported CIGAR op (process_cigar throws, main_conversion.hpp:902-916)"; return UVCGPU_EUNSUPPORTED; }
    }
    const i32 nm_cnt = (a.nm >= 0 ? a.nm : nge_cnt);
    const i32 xm_cnt = nm_cnt - nge_cnt;
    const i32 qlen = rend - a.pos;
    const i32 xm1500 = xm_cnt * 1500 / qlen;
    const i32 go1500 = ngo_cnt * 1500 / qlen;
    const i32 avg_gaplen = nge_cnt / max_(1, ngo_cnt);
    const i32 frag_pos_L = min_(a.pos, a.mpos);
    const i32 frag_pos_R = frag_pos_L + abs(a.isize);
    const bool isrc = a.isrc();
    const i32 pcr_dp_inc = ((dflag & 0x4) ? 1 : 0);
    const i32 umi_dp_inc = ((dflag & 0x1) ? 1 : 0);
    qpos = 0; rpos = a.pos;
    const i32 atd = P.indel_adj_tracklen_dist;
    const i32 nrtr = (i32)S.rtr.size();
    for (i32 i = 0; i < a.n_cigar; i++) {
        const int op = cig_op(a.cigar[i]); const i32 len = (i32)cig_len(a.cigar[i]);
        if (op == C_MATCH || op == C_EQUAL || op == C_DIFF) {
            for (i32 j = 0; j < len; j++) {
                const i64 x = rpos - off;
                S.p32(UVC_P_a_pcr_dp, x) += pcr_dp_inc;
                S.p32(UVC_P_a_umi_dp, x) += umi_dp_inc;
                S.p32(UVC_P_a_dp, x) += 1;
                S.p32(UVC_P_a_qlen, x) += qlen;
                S.p32(UVC_P_a_XM1500, x) += xm1500;
                S.p32(UVC_P_a_GO1500, x) += go1500;
                S.p32(UVC_P_a_GAPLEN, x) += avg_gaplen;
                if (a.isize != 0) {
                    if (isrc) { S.p64(UVC_P_a_LI, x) += min_(rpos - frag_pos_L + 1, MAX_INSERT_SIZE_); S.p32(UVC_P_a_LIDP, x) += 1; }
                    else      { S.p64(UVC_P_a_RI, x) += min_(frag_pos_R - rpos, MAX_INSERT_SIZE_);     S.p32(UVC_P_a_RIDP, x) += 1; }
                }
                // SNV / DNV run detection, main.hpp:1025-1046
                int refsymbol = UVC_BASE_NN, readsymbol = UVC_NUM_SYMBOLS;
                i32 nq = qpos, nr = rpos;
                while (refsymbol != readsymbol && nq < a.l_qseq && nr < rend) {
                    refsymbol = S.refsym[nr - off];
                    readsymbol = a.bases[nq];
                    nq++; nr++;
                }
                if (nr == rpos + 2) for (i32 r = max_(a.pos, rpos - 1); r < min_(nr, rend); r++) S.p32(UVC_P_a_snv_dp, r - off) += 1;
                if (nr > rpos + 2)  for (i32 r = max_(a.pos, rpos - 1); r < min_(nr, rend); r++) S.p32(UVC_P_a_dnv_dp, r - off) += 1;
                if (a.quals[qpos] >= P.bias_thres_highBQ) {
                    S.p32(UVC_P_a_l_dist_sum, x) += rpos - a.pos + 1;
                    S.p32(UVC_P_a_r_dist_sum, x) += rend - rpos;
                    S.p32(UVC_P_a_inslen_sum, x) += inslen_sum;
                    S.p32(UVC_P_a_dellen_sum, x) += dellen_sum;
                    const i32 lbaq = (i32)(BAQ(rpos) - BAQ(a.pos) + 1);
                    const i32 rbaq = (i32)(BAQ(rend - 1) - BAQ(rpos) + 1);
                    S.p64(UVC_P_a_l_BAQ_sum, x) += lbaq;
                    S.p64(UVC_P_a_r_BAQ_sum, x) += rbaq;
                    S.p64(UVC_P_a_insBAQ_sum, x) += insbaq_sum;
                    S.p64(UVC_P_a_delBAQ_sum, x) += delbaq_sum;
                    S.p32(UVC_P_a_highBQ_dp, x) += 1;
                }
                qpos++; rpos++;
            }
        } else if (op == C_INS) {
            const Rtr &rtr1 = S.rtr[max_(atd, rpos - off) - atd];
            const Rtr &rtr2 = S.rtr[min_(rpos - off + atd, nrtr - 1)];
            const i32 unitlen2 = max_(1, (rtr1.tracklen > rtr2.tracklen) ? rtr1.unitlen : rtr2.unitlen);
            const i32 nbases = (i32)((u32)len * (u32)P.indel_adj_indellen_perc / 100u);
            for (i32 r2 = max_(rpos - nbases, a.pos); r2 < min_(rpos + nbases, rend); r2++) {
                const i64 x = r2 - off;
                S.p32(UVC_P_a_near_ins_dp, x) += 1;
                S.p64(UVC_P_a_near_ins_pow2len, x) += (i64)((u32)len * (u32)len);
                S.p64(UVC_P_a_near_ins_l_pow2len, x) += (i64)(r2 + 1 - (rpos - nbases)) * (r2 + 1 - (rpos - nbases));
                S.p64(UVC_P_a_near_ins_r_pow2len, x) += (i64)((rpos + nbases) - r2) * ((rpos + nbases) - r2);
                S.p32(UVC_P_a_near_ins_inv100len, x) += (i32)(100u / ((0 == (u32)len % (u32)unitlen2) ? ((u32)len / (u32)unitlen2) : 4u));
            }
            for (i32 r2 = max_((off + rtr1.begpos) - atd, a.pos); r2 < min_((off + rtr2.begpos + rtr2.tracklen) + atd, rend); r2++) {
                S.p32(UVC_P_a_near_RTR_ins_dp, r2 - off) += 1;
            }
            S.p32(UVC_P_a_at_ins_dp, rpos - off) += 1;
            qpos += len;
        } else if (op == C_DEL) {
            const Rtr &rtr1 = S.rtr[max_(atd, rpos - off) - atd];
            const Rtr &rtr2 = S.rtr[min_(rpos - off + atd, nrtr - 1)];
            for (i32 r2 = rpos; r2 < rpos + len; r2++) {
                const i64 x = r2 - off;
                S.p32(UVC_P_a_pcr_dp, x) += pcr_dp_inc;
                S.p32(UVC_P_a_umi_dp, x) += umi_dp_inc;
                S.p32(UVC_P_a_dp, x) += 1;
                S.p32(UVC_P_a_qlen, x) += qlen;
                S.p32(UVC_P_a_highBQ_dp, x) += 1;
                S.p32(UVC_P_a_XM1500, x) += xm1500;
                S.p32(UVC_P_a_GO1500, x) += go1500;
                S.p32(UVC_P_a_GAPLEN, x) += avg_gaplen;
                if (a.isize != 0) {   // sic: uses rpos (the deletion start), not r2 (main.hpp:1137-1145)
                    if (isrc) { S.p64(UVC_P_a_LI, x) += min_(rpos - frag_pos_L + 1, MAX_INSERT_SIZE_); S.p32(UVC_P_a_LIDP, x) += 1; }
                    else      { S.p64(UVC_P_a_RI, x) += min_(frag_pos_R - rpos, MAX_INSERT_SIZE_);     S.p32(UVC_P_a_RIDP, x) += 1; }
                }
                S.p32(UVC_P_a_l_dist_sum, x) += rpos - a.pos + 1;
                S.p32(UVC_P_a_r_dist_sum, x) += rend - rpos;
                S.p32(UVC_P_a_inslen_sum, x) += inslen_sum;
                S.p32(UVC_P_a_dellen_sum, x) += dellen_sum;
                const i32 lbaq = (i32)(BAQ(rpos) - BAQ(a.pos) + 1);
                const i32 rbaq = (i32)(BAQ(rend - 1) - BAQ(rpos) + 1);
                S.p64(UVC_P_a_l_BAQ_sum, rpos - off) += lbaq;   // sic: written at rpos, not r2 (main.hpp:1156-1157)
                S.p64(UVC_P_a_r_BAQ_sum, rpos - off) += rbaq;
                S.p64(UVC_P_a_insBAQ_sum, x) += insbaq_sum;
                S.p64(UVC_P_a_delBAQ_sum, x) += delbaq_sum;
            }
            const i32 unitlen2 = max_(1, (rtr1.tracklen > rtr2.tracklen) ? rtr1.unitlen : rtr2.unitlen);
            const i32 nbases_l = (i32)((u32)len * (u32)(P.indel_adj_indellen_perc - 100) / 100u);
            const i32 nbases_r = (i32)((u32)len * (u32)P.indel_adj_indellen_perc / 100u);
            const i32 lpos = max_(rpos - nbases_l, a.pos);
            const i32 rpos_r = min_(rpos + nbases_r, rend) - 1;
            for (i32 r2 = lpos; r2 <= rpos_r; r2++) {
                const i64 x = r2 - off;
                S.p32(UVC_P_a_near_del_dp, x) += 1;
                S.p64(UVC_P_a_near_del_pow2len, x) += (i64)((u32)len * (u32)len);
                S.p64(UVC_P_a_near_del_l_pow2len, x) += (i64)(r2 - lpos + 1) * (r2 - lpos + 1);
                S.p64(UVC_P_a_near_del_r_pow2len, x) += (i64)(rpos_r - r2 + 1) * (rpos_r - r2 + 1);
                S.p32(UVC_P_a_near_del_inv100len, x) += (i32)(100u / ((0 == (u32)len % (u32)unitlen2) ? ((u32)len / (u32)unitlen2) : 4u));
            }
            for (i32 r2 = max_((off + rtr1.begpos) - atd, a.pos); r2 < min_((off + rtr2.begpos + rtr2.tracklen) + atd, rend); r2++) {
                S.p32(UVC_P_a_near_RTR_del_dp, r2 - off) += 1;
            }
            S.p32(UVC_P_a_at_del_dp, rpos - off) += 1;
            rpos += len;
        } else {
            const i32 rpos_delta = ((0 == i) ? 0 : -1);
            if ((C_SOFT_CLIP == op || C_HARD_CLIP == op) && pcr_dp_inc) {
                for (i32 r2 = rpos + rpos_delta - P.microadjust_near_clip_dist; r2 <= rpos + rpos_delta + P.microadjust_near_clip_dist; r2++) {
                    if (S.beg <= r2 && r2 < S.end) S.p32(UVC_P_a_near_pcr_clip_dp, r2 - off) += pcr_dp_inc;
                }
            }
            if ((C_SOFT_CLIP == op || C_HARD_CLIP == op) && (0 == pcr_dp_inc) && (len >= P.microadjust_alignment_clip_min_len)) {
                S.p32(UVC_P_a_near_long_clip_dp, rpos + rpos_delta - off) += 1;
            }
            if (op == C_REF_SKIP) rpos += len; else if (op == C_SOFT_CLIP) qpos += len;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// P1b: update_seg_format_thres_from_prep_sets, main.hpp:1206-1299
// ------------------------------------------------------------------------------------------------
static void p1b_thres(State &S) {
    const UvcParams &P = S.P;
    const bool is_normal = P.tumor_vcf_is_provided;
    for (i64 x = 0; x < S.npos; x++) {
        const i32 segLIDP = max_(S.p32(UVC_P_a_LIDP, x), 1), segRIDP = max_(S.p32(UVC_P_a_RIDP, x), 1);
        const i32 ins_dp = S.p32(UVC_P_a_near_ins_dp, x), del_dp = S.p32(UVC_P_a_near_del_dp, x);
        const double ins_l = ceil(sqrt((double)(S.p64(UVC_P_a_near_ins_l_pow2len, x) / max_(ins_dp, 1))));
        const double del_l = ceil(sqrt((double)(S.p64(UVC_P_a_near_del_l_pow2len, x) / max_(del_dp, 1))));
        const double ins_r = ceil(sqrt((double)(S.p64(UVC_P_a_near_ins_r_pow2len, x) / max_(ins_dp, 1))));
        const double del_r = ceil(sqrt((double)(S.p64(UVC_P_a_near_del_r_pow2len, x) / max_(del_dp, 1))));
        const int dnv_border_len = ((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform && (S.p32(UVC_P_a_dnv_dp, x) * 2 > S.p32(UVC_P_a_snv_dp, x))) ? 10 : 0);
        const double max_l = max_(ins_l, max_(del_l, (double)dnv_border_len));
        const double max_r = max_(ins_r, max_(del_r, (double)dnv_border_len));
        S.th(UVC_T_aLPxT, x) = (i32)(max_l + P.bias_thres_aLPxT_add);
        S.th(UVC_T_aRPxT, x) = (i32)(max_r + P.bias_thres_aLPxT_add);
        Rtr &rtr = S.rtr[x];
        const i32 half = (i32)round(numstates2phred(P.indel_del_to_ins_err_ratio)) / 2;
        if (ins_dp * P.indel_del_to_ins_err_ratio < del_dp) rtr.indelphred += half;
        if (del_dp * P.indel_del_to_ins_err_ratio < ins_dp) rtr.indelphred -= half;
        const i32 pc_inc1 = (i32)(3 * 100 * max_(1, ins_dp + del_dp) / (max_(1, S.p32(UVC_P_a_near_ins_inv100len, x) + S.p32(UVC_P_a_near_del_inv100len, x)))) - 3;
        rtr.indelphred += between_(pc_inc1, 0, 6);
        rtr.indelphred = max_(rtr.indelphred, 0);

        const i32 aLRI1T_perc = (is_normal ? P.bias_thres_aLRI1NT_perc : P.bias_thres_aLRI1T_perc);
        const i32 aLRI1t_perc = (is_normal ? P.bias_thres_aLRI1Nt_perc : P.bias_thres_aLRI1t_perc);
        const i64 LI = S.p64(UVC_P_a_LI, x), RI = S.p64(UVC_P_a_RI, x);
        S.th(UVC_T_aLI1T, x) = (i32)(LI * aLRI1T_perc / (segLIDP * 100) + P.bias_thres_aLRI1T_add);
        S.th(UVC_T_aLI2T, x) = (i32)(LI * P.bias_thres_aLRI2T_perc / (segLIDP * 100) + P.bias_thres_aLRI2T_add);
        S.th(UVC_T_aLI1t, x) = (i32)(LI * aLRI1t_perc / (segLIDP * 100));
        S.th(UVC_T_aLI2t, x) = (i32)(LI * P.bias_thres_aLRI2t_perc / (segLIDP * 100));
        S.th(UVC_T_aRI1T, x) = (i32)(RI * aLRI1T_perc / (segRIDP * 100) + P.bias_thres_aLRI1T_add);
        S.th(UVC_T_aRI2T, x) = (i32)(RI * P.bias_thres_aLRI2T_perc / (segRIDP * 100) + P.bias_thres_aLRI2T_add);
        S.th(UVC_T_aRI1t, x) = (i32)(RI * aLRI1t_perc / (segRIDP * 100));
        S.th(UVC_T_aRI2t, x) = (i32)(RI * P.bias_thres_aLRI2t_perc / (segRIDP * 100));

        const i32 aLRP1t_perc = (is_normal ? P.bias_thres_aLRP1Nt_avgmul_perc : P.bias_thres_aLRP1t_avgmul_perc);
        const i32 aLRP2t_perc = P.bias_thres_aLRP2t_avgmul_perc;
        const i32 aLRB1t_perc = (is_normal ? P.bias_thres_aLRB1Nt_avgmul_perc : P.bias_thres_aLRB1t_avgmul_perc);
        const i32 aLRB2t_perc = P.bias_thres_aLRB2t_avgmul_perc;
        const i32 hb = S.p32(UVC_P_a_highBQ_dp, x);
        const i64 den = max_(1, hb * 100);
        const i64 lds = S.p32(UVC_P_a_l_dist_sum, x), rds = S.p32(UVC_P_a_r_dist_sum, x);
        S.th(UVC_T_aLP1t, x) = (i32)nnminus(lds * aLRP1t_perc / den, P.bias_thres_aLRP1t_minus);
        S.th(UVC_T_aLP2t, x) = (i32)nnminus(lds * aLRP2t_perc / den, P.bias_thres_aLRP2t_minus);
        S.th(UVC_T_aRP1t, x) = (i32)nnminus(rds * aLRP1t_perc / den, P.bias_thres_aLRP1t_minus);
        S.th(UVC_T_aRP2t, x) = (i32)nnminus(rds * aLRP2t_perc / den, P.bias_thres_aLRP2t_minus);
        const i64 pdel = S.p64(UVC_P_a_delBAQ_sum, x) / max_(1, hb);
        const i64 lb = S.p64(UVC_P_a_l_BAQ_sum, x), rb = S.p64(UVC_P_a_r_BAQ_sum, x);
        S.th(UVC_T_aLB1t, x) = (i32)nnminus(lb * aLRB1t_perc / den, P.bias_thres_aLRB1t_minus + pdel);
        S.th(UVC_T_aLB2t, x) = (i32)nnminus(lb * aLRB2t_perc / den, P.bias_thres_aLRB2t_minus);
        S.th(UVC_T_aRB1t, x) = (i32)nnminus(rb * aLRB1t_perc / den, P.bias_thres_aLRB1t_minus + pdel);
        S.th(UVC_T_aRB2t, x) = (i32)nnminus(rb * aLRB2t_perc / den, P.bias_thres_aLRB2t_minus);
    }
}

// ------------------------------------------------------------------------------------------------
// dealwith_segbias<isGap>, main.hpp:1360-1595 (COMPILATION_ENABLE_XMGOT == 0)
// ------------------------------------------------------------------------------------------------
static void bidir_bias(i32 &LP1, i32 &LP2, i32 &RP1, i32 &RP2, i64 &LPL, i64 &RPL, i32 L1, i32 L2, i32 R1, i32 R2, i64 nl, i64 nr, bool tier2, i32 n_indel) {
    // update_bidirectional_bias, main.hpp:1318-1358
    if (nl + n_indel >= L1) LP1 += 1;
    if ((nl + n_indel >= L2) && tier2) LP2 += 1;
    if (nr >= R1) RP1 += 1;
    if ((nr >= R2) && tier2) RP2 += 1;
    LPL += nl; RPL += nr;
}

static void segbias(State &S, bool isGap, i32 bq, i32 rpos, int sym, const Aln &a, i32 xm1500, i32 bm1500, int cigar_op, i32 indel_len, i32 dist_to_interfering_indel, int dflag, i32 clip_cnt) {
    const UvcParams &P = S.P;
    const i64 x = rpos - S.beg;
    const bool is_assay_amplicon = ((dflag & 0x4) || ((P.primerlen > 0) && !(0x2 & P.primer_flag)));
    const bool is_normal_used_to_filter_vars_on_primers = (P.tn_is_paired && (0x1 & P.primer_flag));
    const bool is_assay_UMI = (dflag & 0x1);
    const i32 rend = a.endpos;
    auto BAQ = [&](i64 p) -> i64 { return S.baq[p - S.beg]; };
    auto BAQ2 = [&](i64 p) -> i64 { return S.baq2[p - S.beg]; };
    const i32 seg_l_baq1 = (i32)(BAQ(rpos) - BAQ(a.pos) + 1);
    const i32 _seg_r_baq = (i32)(BAQ(rend - 1) - BAQ(rpos) + 1);
    const i32 seg_r_baq1 = (isGap ? (i32)min_((i64)_seg_r_baq, BAQ2(rend - 1) - BAQ2(rpos) + 7) : _seg_r_baq);
    const i32 seg_l_nbases = rpos - a.pos + 1;
    const i32 seg_r_nbases = rend - rpos;
    const bool is_high_readlen = (P.central_readlen >= P.microadjust_median_readlen_thres);
    const i32 seg_l_baq = (is_high_readlen ? seg_l_baq1 : max_(seg_l_baq1, seg_l_nbases * P.microadjust_BAQ_per_base_x1024 / 1024));
    const i32 seg_r_baq = (is_high_readlen ? seg_r_baq1 : max_(seg_r_baq1, seg_r_nbases * P.microadjust_BAQ_per_base_x1024 / 1024));
    const i32 frag_pos_L = min_(a.pos, a.mpos);
    const i32 frag_pos_R = frag_pos_L + abs(a.isize);
    const i32 frag_l_nbases2 = ((a.isize != 0) ? min_(rpos - frag_pos_L + 1, MAX_INSERT_SIZE_) : MAX_INSERT_SIZE_);
    const i32 frag_r_nbases2 = ((a.isize != 0) ? min_(frag_pos_R - rpos + 0, MAX_INSERT_SIZE_) : MAX_INSERT_SIZE_);
    const bool is_normal = ((a.isize != 0) || (0 == (a.flag & 0x1)));
    const bool isrc = a.isrc();
    const bool strand = a.bam_strand();

    S.VQ(isrc ? UVC_VQ_a1BQr : UVC_VQ_a1BQf, sym, x) += bq;
    S.VQ(isrc ? UVC_VQ_a2BQr : UVC_VQ_a2BQf, sym, x) += bq * bq / SQR_QUAL_DIV_;
    S.s32(UVC_S_aMQs, sym, x) += a.mapq;
    S.s32(strand ? (isrc ? UVC_S_aDPrr : UVC_S_aDPrf) : (isrc ? UVC_S_aDPfr : UVC_S_aDPff), sym, x) += 1;
    if (min_(dist_to_interfering_indel, min_(seg_l_nbases, seg_r_nbases)) >= P.bias_thres_interfering_indel) S.s32(UVC_S_aP3, sym, x) += 1;
    if (0 == clip_cnt) S.s32(UVC_S_aNC, sym, x) += 1;
    if (isrc) S.s64(UVC_S64_aLIT, sym, x) += ((a.isize != 0) ? frag_l_nbases2 : 0);
    else      S.s64(UVC_S64_aRIT, sym, x) += ((a.isize != 0) ? frag_r_nbases2 : 0);

    const i32 _LPxT = S.th(UVC_T_aLPxT, x), RPxT = S.th(UVC_T_aRPxT, x);
    const i32 LPxT = (isGap ? _LPxT : min_(_LPxT, RPxT));
    const bool is_far_from_edge = (seg_l_nbases + ((C_INS == cigar_op) ? (i32)nnminus(indel_len, P.microadjust_nobias_pos_indel_maxlen) : 0) >= LPxT) && (seg_r_nbases >= RPxT);
    const i32 thres_highBAQ = P.bias_thres_highBAQ + (isGap ? 0 : 3);
    const bool is_unaffected_by_edge = (seg_l_baq >= thres_highBAQ && seg_r_baq >= thres_highBAQ);
    const i32 min_dist2iend = ((a.flag & 0x1) ? min_(frag_l_nbases2, frag_r_nbases2) : (isrc ? seg_r_nbases : seg_l_nbases));
    if (is_far_from_edge && is_unaffected_by_edge && (min_dist2iend > P.primerlen2 || !is_assay_amplicon)) S.s32(UVC_S_aP1, sym, x) += 1;
    if (is_assay_UMI || !is_assay_amplicon) S.s32(UVC_S_aP2, sym, x) += 1;

    i32 ampfact2 = 100;
    if (bq < P.bias_thres_PFBQ1) ampfact2 = 100 * (bq * bq) / (P.bias_thres_PFBQ1 * P.bias_thres_PFBQ1);
    S.s32(UVC_S_aPF1, sym, x) += (isGap ? min_(100, ampfact2) : (100 * ampfact2 / 100));
    ampfact2 = 100;
    if (bq < P.bias_thres_PFBQ2) ampfact2 = 100 * (bq * bq) / (P.bias_thres_PFBQ2 * P.bias_thres_PFBQ2);
    S.s32(UVC_S_aPF2, sym, x) += (isGap ? min_(100, ampfact2) : (100 * ampfact2 / 100));
    if (!isGap) {
        S.s32(UVC_S_a2XM2, sym, x) += (xm1500 > 20 ? (100 * (20 * 20) / (xm1500 * xm1500)) : 100);
        S.s32(UVC_S_a2BM2, sym, x) += (bm1500 > 20 ? (100 * (20 * 20) / (bm1500 * bm1500)) : 100);
    }
    if (((!isGap) && bq >= P.bias_thres_highBQ) || (isGap && dist_to_interfering_indel >= P.bias_thres_interfering_indel)) {
        const bool tier2 = (isGap || bq >= P.bias_thres_highBQ);
        if (is_far_from_edge) {
            i64 LPL = S.s32(UVC_S_aLPL, sym, x), RPL = S.s32(UVC_S_aRPL, sym, x);
            bidir_bias(S.s32(UVC_S_aLP1, sym, x), S.s32(UVC_S_aLP2, sym, x), S.s32(UVC_S_aRP1, sym, x), S.s32(UVC_S_aRP2, sym, x), LPL, RPL,
                       S.th(UVC_T_aLP1t, x), S.th(UVC_T_aLP2t, x), S.th(UVC_T_aRP1t, x), S.th(UVC_T_aRP2t, x), seg_l_nbases, seg_r_nbases, tier2, indel_len);
            S.s32(UVC_S_aLPL, sym, x) = (i32)LPL; S.s32(UVC_S_aRPL, sym, x) = (i32)RPL;
        }
        if (is_unaffected_by_edge) {
            bidir_bias(S.s32(UVC_S_aLB1, sym, x), S.s32(UVC_S_aLB2, sym, x), S.s32(UVC_S_aRB1, sym, x), S.s32(UVC_S_aRB2, sym, x),
                       S.s64(UVC_S64_aLBL, sym, x), S.s64(UVC_S64_aRBL, sym, x),
                       P.bias_thres_BAQ1, P.bias_thres_BAQ2, P.bias_thres_BAQ1, P.bias_thres_BAQ2, seg_l_baq, seg_r_baq, tier2, 0);
        }
        S.s32(UVC_S_aBQ2, sym, x) += 1;
    }
    const bool mate_ok = ((0 == (a.flag & 0x8)) || (0 == (a.flag & 0x1)));
    const bool is_l_nonbiased = (mate_ok && seg_l_nbases > seg_r_nbases);
    const bool is_r_nonbiased = (mate_ok && seg_l_nbases < seg_r_nbases);
    const bool pos_good = ((!is_assay_amplicon) || (!is_normal_used_to_filter_vars_on_primers) || (is_far_from_edge && is_unaffected_by_edge));
    if (isrc) {
        const i32 d = frag_l_nbases2;
        if ((d >= S.th(UVC_T_aLI1t, x)) && (d <= S.th(UVC_T_aLI1T, x) || isGap) && (is_normal || (isGap && is_l_nonbiased))) S.s32(UVC_S_aLI1, sym, x) += 1;
        if ((d >= S.th(UVC_T_aLI2t, x)) && (d <= S.th(UVC_T_aLI2T, x) || isGap) && (is_normal || (isGap && is_l_nonbiased))) { if (pos_good) S.s32(UVC_S_aLI2, sym, x) += 1; }
        if (pos_good) S.s32(UVC_S_aLIr, sym, x) += 1;
    } else {
        const i32 d = frag_r_nbases2;
        if ((d >= S.th(UVC_T_aRI1t, x)) && (d <= S.th(UVC_T_aRI1T, x) || isGap) && (is_normal || (isGap && is_r_nonbiased))) S.s32(UVC_S_aRI1, sym, x) += 1;
        if ((d >= S.th(UVC_T_aRI2t, x)) && (d <= S.th(UVC_T_aRI2T, x) || isGap) && (is_normal || (isGap && is_r_nonbiased))) { if (pos_good) S.s32(UVC_S_aRI2, sym, x) += 1; }
        if (pos_good) S.s32(UVC_S_aRIf, sym, x) += 1;
    }
}

// Test hook (tests/test_segbias_cpu.py): ONE call of dealwith_segbias on a zeroed cell of the handle's planes, with the position's
// SegFormatThresSet given by the caller; out = the cell's SegFormatInfoSet i32 fields, its four i64 fields, then a1BQf a1BQr a2BQf a2BQr.
void test_segbias(State &S, bool isGap, i32 bq, i32 rpos, int sym, i32 a_pos, i32 a_endpos, i32 a_mpos, i32 a_isize, i32 a_flag, i32 a_mapq,
                  i32 xm1500, i32 bm1500, int cigar_op, i32 indel_len, i32 dist, int dflag, i32 clip_cnt, const i32 *thres, i64 *out) {
    const i64 x = rpos - S.beg;
    if (S.seg32.empty()) {   // the planes are allocated by accumulate(): a handle without reads gets them here
        S.thres.assign((size_t)S.npos * UVC_NTHRES, 0); S.seg32.assign((size_t)S.npos * NSYM * UVC_NSEG32, 0); S.seg64.assign((size_t)S.npos * NSYM * UVC_NSEG64, 0);
        S.vq.assign((size_t)S.npos * NSYM * UVC_NVQ, 0);
    }
    for (int f = 0; f < UVC_NTHRES; f++) S.th(f, x) = thres[f];
    for (int f = 0; f < UVC_NSEG32; f++) S.s32(f, sym, x) = 0;
    for (int f = 0; f < UVC_NSEG64; f++) S.s64(f, sym, x) = 0;
    for (int f = 0; f < UVC_NVQ; f++) S.VQ(f, sym, x) = 0;
    Aln a; memset(&a, 0, sizeof(a));
    a.pos = a_pos; a.endpos = a_endpos; a.mpos = a_mpos; a.isize = a_isize; a.flag = a_flag; a.mapq = a_mapq;
    segbias(S, isGap, bq, rpos, sym, a, xm1500, bm1500, cigar_op, indel_len, dist, dflag, clip_cnt);
    int k = 0;
    for (int f = 0; f < UVC_NSEG32; f++) out[k++] = S.s32(f, sym, x);
    for (int f = 0; f < UVC_NSEG64; f++) out[k++] = S.s64(f, sym, x);
    out[k++] = S.VQ(UVC_VQ_a1BQf, sym, x); out[k++] = S.VQ(UVC_VQ_a1BQr, sym, x); out[k++] = S.VQ(UVC_VQ_a2BQf, sym, x); out[k++] = S.VQ(UVC_VQ_a2BQr, sym, x);
}

// ------------------------------------------------------------------------------------------------
// ref_to_phredvalue, main.hpp:876-922.  NOTE: n_units is an OUT parameter bound to inslen/dellen
// at the call sites (main.hpp:2025-2026, 2134-2135), so it overwrites the indel length there.
// ------------------------------------------------------------------------------------------------
static i32 ref_to_phredvalue(i32 &n_units, i32 &max_rn, i32 &rs_at_max, const State &S, i32 refpos, i32 max_phred, double ampfact, i32 oplen, int op) {
    const UvcParams &P = S.P;
    const i32 n = (i32)S.refstring.size();
    max_rn = 0; rs_at_max = 0;
    for (i32 rs = 1; rs <= P.indel_str_repeatsize_max; rs++) {
        i32 q = refpos;
        while (q + rs < n && S.refsym[q] == S.refsym[q + rs]) q++;
        i32 rn = (q - refpos) / rs + 1;
        if (is_indel_context_more_STR(rs, rn, rs_at_max, max_rn, P.indel_str_repeatsize_max)) { max_rn = rn; rs_at_max = rs; }
    }
    if (oplen == rs_at_max && op == C_DEL) ampfact *= P.indel_del_to_ins_err_ratio;
    i32 decphred = indel_phred(ampfact, rs_at_max, max_rn);
    if (rs_at_max * (max_rn - 1) >= 6 - 1) n_units = ((0 == oplen % rs_at_max) ? (oplen / rs_at_max) : ((1 == oplen) ? 1 : 0));
    else n_units = 1 + (oplen / 6);
    return max_phred - min_(max_phred, decphred) + indel_len_rusize_phred(oplen, rs_at_max);
}

// proton_cigarlen2phred, main_conversion.hpp:922-941
static i32 proton_cigarlen2phred(i32 cigarlen) {
    static const i32 t[13] = { 0, 0, 9, 14, 18, 21, 23, 25, 27, 29, 30, 31, 32 };
    return t[min_(cigarlen, 12)];
}

// per-fragment / per-family temporary Symbol2CountCoverage (main.hpp:1597-1602): counts + the
// insertion-sequence / deletion-length side maps (pos2iseq2data / pos2dlen2data, main.hpp:529-530)
struct Cov {
    i32 beg = 0, end = 0;
    std::vector<i32> d;   // [(epos-beg)*NSYM + sym]
    std::map<i32, std::map<std::string, i32>> iseq[3];   // index: I1=0, I2=1, I3P=2 (main.hpp:579-583)
    std::map<i32, std::map<i32, i32>> dlen[3];           // index: D1=0, D2=1, D3P=2 (main.hpp:574-578)
    void init(i32 b, i32 e) { beg = b; end = e; d.assign((size_t)(e - b) * NSYM, 0); for (int i = 0; i < 3; i++) { iseq[i].clear(); dlen[i].clear(); } }
    inline i32 &at(i32 epos, int s) { return d[(size_t)(epos - beg) * NSYM + s]; }
    inline const i32 *row(i32 epos) const { return &d[(size_t)(epos - beg) * NSYM]; }
};
static inline int ins_idx(int s) { return (UVC_LINK_I1 == s ? 0 : ((UVC_LINK_I2 == s) ? 1 : 2)); }
static inline int del_idx(int s) { return (UVC_LINK_D1 == s ? 0 : ((UVC_LINK_D2 == s) ? 1 : 2)); }

// indelToData_getMajority, main.hpp:50-63 (ties -> larger key)
template <class K> static std::pair<i32, K> get_majority(const std::map<K, i32> &m) {
    i32 maxcnt = 0; K arg = K();
    for (const auto &ic : m) if (ic.second > maxcnt || ((ic.second == maxcnt) && (ic.first > arg))) { maxcnt = ic.second; arg = ic.first; }
    return std::make_pair(maxcnt, arg);
}
// posToIndelToCount_updateByConsensus, main.hpp:83-95
template <class K> static void indel_update_by_consensus(std::map<i32, std::map<K, i32>> &dst, const std::map<i32, std::map<K, i32>> &src, i32 epos, i32 inc) {
    auto it = src.find(epos);
    if (it == src.end() || it->second.empty()) return;   // the reference asserts this cannot happen
    const K k = (it->second.size() > 1 ? get_majority(it->second).second : it->second.begin()->first);
    dst[epos][k] += inc;
}

// ------------------------------------------------------------------------------------------------
// updateByAln<TIsProton, TUpdateType, TIsBiasUpdated>, main.hpp:1762-2296
//   sum_mode = true  : SYMBOL_COUNT_SUM into S.bqsum  (P2, bias updated)
//   sum_mode = false : BASE_QUALITY_MAX into tmp      (P3/P4/P5, no bias)
// ------------------------------------------------------------------------------------------------
static int update_by_aln(State &S, const Aln &a, int dflag, bool bias, Cov *tmp, std::string &err) {
    const UvcParams &P = S.P;
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    const i32 off = S.beg;
    const bool is_assay_amplicon = ((dflag & 0x4) || ((P.primerlen > 0) && !(0x2 & P.primer_flag)));
    const i32 n_cigar = a.n_cigar;
    const u32 *cigar = a.cigar;
    const i32 rend = a.endpos;
    const i32 addPhred[2] = { P.bq_phred_added_misma, P.bq_phred_added_indel };
    auto Q = [&](i32 q) -> i32 { return (i32)a.quals[min_(max_(q, 0), a.l_qseq - 1)]; };  // clamp only guards the reference's own out-of-range reads at read ends
    auto inc = [&](i32 epos, int sym, i32 v) {
        if (tmp) { i32 &c = tmp->at(epos, sym); c = max_(c, v); }   // incSymbolCount<BASE_QUALITY_MAX>, update_max_inc = 0 (main.hpp:339-349)
        else S.BQS(sym, epos - off) += v;
    };
    auto BAQ = [&](i64 p) -> i64 { return S.baq[p - off]; };

    i32 nge_cnt = 0, ngo_cnt = 0, clip_cnt = 0;
    for (i32 i = 0; i < n_cigar; i++) {
        const int op = cig_op(cigar[i]);
        if (C_INS == op || C_DEL == op) { nge_cnt += (i32)cig_len(cigar[i]); ngo_cnt++; }
        if (C_SOFT_CLIP == op || C_HARD_CLIP == op) clip_cnt++;
    }
    const i32 nm_cnt = (a.nm >= 0 ? a.nm : nge_cnt);
    const i32 xm_cnt = nm_cnt - nge_cnt;
    const i32 xm1500 = xm_cnt * 1500 / (rend - a.pos);
    const i32 go1500 = ngo_cnt * 1500 / (rend - a.pos);

    std::vector<i32> indel_rposs; indel_rposs.push_back(0);
    size_t indel_rposs_idx = 0;
    i32 bm_cnts[NSYM] = { 0 };
    {
        i32 qpos = 0, rpos = a.pos;
        for (i32 i = 0; i < n_cigar; i++) {
            const int op = cig_op(cigar[i]); const i32 len = (i32)cig_len(cigar[i]);
            if (op == C_MATCH || op == C_EQUAL || op == C_DIFF) {
                for (i32 i2 = 0; i2 < len; i2++) {
                    const int b = a.bases[qpos];
                    if (S.refsym[rpos - off] != b) bm_cnts[b] += 1;
                    qpos++; rpos++;
                }
            } else if (op == C_INS) {
                bool is_lowBQ = false;
                for (i32 q2 = qpos - min_(qpos, 1); q2 < min_(qpos + len + 1, rend); q2++) {   // sic: query index bounded by rend (main.hpp:1841)
                    if (Q(q2) < P.bias_thres_interfering_indel_BQ) is_lowBQ = true;
                }
                if (is_lowBQ) indel_rposs.push_back(rpos);
                qpos += len;
            } else if (op == C_DEL) {
                bool is_lowBQ = (min_(Q(max_(1, qpos) - 1), Q(qpos)) <= P.bias_thres_interfering_indel_BQ);
                if (is_lowBQ) indel_rposs.push_back(rpos);
                rpos += len;
            } else if (op == C_REF_SKIP) rpos += len;
            else if (op == C_SOFT_CLIP) qpos += len;
            else if (op == C_HARD_CLIP || op == C_PAD) {}
            else { err = "unsupported CIGAR op"; return UVCGPU_EUNSUPPORTED; }
        }
        indel_rposs.push_back(INT32_MAX);
    }
    i32 bm1500s[NSYM];
    for (int i = 0; i < NSYM; i++) bm1500s[i] = bm_cnts[i] * 1500 / (rend - a.pos);

    const bool isrc = a.isrc();
    const bool normal_filter_primers = (P.tn_is_paired && (0x1 & P.primer_flag));
    const i32 ibeg = ((a.isize != 0) ? (min_(a.pos, a.mpos) + P.primerlen) : ((isrc && (0x0 == (0x1 & a.flag))) ? 0 : (a.pos + P.primerlen)));
    const i32 iend = ((a.isize != 0) ? (i32)nnminus(min_(a.pos, a.mpos) + abs(a.isize), P.primerlen)
                                     : ((isrc && (0x0 == (0x1 & a.flag))) ? (i32)nnminus(rend, P.primerlen) : INT32_MAX));
    i32 qpos = 0, rpos = a.pos;
    i32 incvalue = 1;
    const i32 lclip_len = ((n_cigar > 0 && cig_op(cigar[0]) == C_SOFT_CLIP) ? (i32)cig_len(cigar[0]) : 0);
    const i32 rclip_len = ((n_cigar > 0 && cig_op(cigar[n_cigar - 1]) == C_SOFT_CLIP) ? (i32)cig_len(cigar[n_cigar - 1]) : 0);
    const i32 penal_by_clip = max_(lclip_len, rclip_len) / 6;
    const i32 penal_by_nm = (xm1500 + go1500) / 30;
    const i32 micro_indel_penal = min_(1, penal_by_nm + penal_by_clip);
    const i32 micro_nogap_penal = min_(4, penal_by_nm + penal_by_clip) + 1;
    const i32 atd = P.indel_adj_tracklen_dist;
    const i32 nrtr = (i32)S.rtr.size();

    for (i32 i = 0; i < n_cigar; i++) {
        const int op = cig_op(cigar[i]); const i32 len = (i32)cig_len(cigar[i]);
        if (op == C_MATCH || op == C_EQUAL || op == C_DIFF) {
            for (i32 i2 = 0; i2 < len; i2++) {
                if ((normal_filter_primers || !is_assay_amplicon) || (ibeg <= rpos && rpos < iend)) {
                    i32 dist_to_interfering_indel = 10000;
                    if (bias && nge_cnt > 0) {
                        if (indel_rposs[indel_rposs_idx] <= rpos) indel_rposs_idx++;
                        const i32 prev_indel_rpos = indel_rposs[indel_rposs_idx - 1];
                        const i32 next_indel_rpos = indel_rposs[indel_rposs_idx];
                        const Rtr &rtr1 = S.rtr[max_(rpos - off, atd) - atd];
                        const Rtr &rtr2 = S.rtr[min_(rpos - off + atd, nrtr - 1)];
                        const i64 prevlen = nnminus(rpos - prev_indel_rpos, max_(rpos - (off + rtr1.begpos), S.th(UVC_T_aLP1t, rpos - off)));
                        const i64 nextlen = nnminus((i64)next_indel_rpos - rpos, max_((off + rtr2.begpos + rtr2.tracklen) - rpos, S.th(UVC_T_aRP1t, rpos - off)));
                        dist_to_interfering_indel = (i32)min_(prevlen, nextlen);
                    }
                    if (i2 > 0) {
                        const i32 noindel_phredvalue = min_(S.rtr[rpos - off - 1].indelphred, S.rtr[rpos - off].indelphred);
                        const i32 qfromBQ2 = (proton ? min_(Q(qpos - 1), Q(qpos)) : 80);
                        incvalue = (i32)nnminus(min_(qfromBQ2, noindel_phredvalue), micro_nogap_penal) + 1;
                        inc(rpos, UVC_LINK_M, incvalue);
                        if (bias) segbias(S, true, incvalue, rpos, UVC_LINK_M, a, xm1500, bm1500s[UVC_LINK_M], op, 0, dist_to_interfering_indel, dflag, clip_cnt);
                    }
                    const int symbol = a.bases[qpos];
                    if (proton && ((0 == i2) || (len - 1 == i2))) {
                        // prev/next_cigar are compared as packed words against op codes in the reference (main.hpp:1953-1956)
                        const i64 prev_cigar = (0 < i ? (i64)cigar[i - 1] : -1);
                        const i64 next_cigar = (i + 1 < n_cigar ? (i64)cigar[i + 1] : -1);
                        const bool next_gap = ((len - 1 == i2) && (C_MATCH != next_cigar) && (C_EQUAL != next_cigar) && (C_DIFF != next_cigar));
                        const bool prev_gap = ((0 == i2) && (C_MATCH != prev_cigar) && (C_EQUAL != prev_cigar) && (C_DIFF != prev_cigar));
                        if (next_gap || prev_gap) {
                            const bool isrc2 = (0 != i2);
                            i32 prev_base_phred = 1;
                            if (isrc2 && (qpos + 1 < a.l_qseq)) prev_base_phred = Q(qpos + 1);
                            if ((!isrc2) && (qpos > 0)) prev_base_phred = Q(qpos - 1);
                            i32 adj_gap_cigarlen = 100;
                            if (next_gap) adj_gap_cigarlen = min_(adj_gap_cigarlen, ((i + 1 < n_cigar) ? (i32)cig_len(cigar[i + 1]) : 100));
                            if (prev_gap) adj_gap_cigarlen = min_(adj_gap_cigarlen, ((0 < i) ? (i32)cig_len(cigar[i - 1]) : 100));
                            if (adj_gap_cigarlen < 3) incvalue = min_(Q(qpos), prev_base_phred) + min_(addPhred[0], addPhred[1]);
                            else incvalue = min_(Q(qpos), prev_base_phred) + addPhred[0];
                        } else incvalue = Q(qpos) + addPhred[0];
                    } else incvalue = Q(qpos) + addPhred[0];
                    inc(rpos, symbol, incvalue);
                    if (bias) segbias(S, false, incvalue, rpos, symbol, a, xm1500, bm1500s[symbol], op, 0, dist_to_interfering_indel, dflag, clip_cnt);
                }
                rpos += 1; qpos += 1;
            }
        } else if (op == C_INS) {
            if ((normal_filter_primers || !is_assay_amplicon) || (ibeg <= rpos && rpos < iend)) {
                const i32 nbases2end = min_(qpos, a.l_qseq - (qpos + len));
                const bool at_read_end = (nbases2end <= 0);
                i32 inslen = len;
                if (at_read_end) {
                    incvalue = (0 != qpos ? Q(qpos - 1) : ((qpos + len < a.l_qseq) ? Q(qpos + len) : 1)) + addPhred[1];
                } else {
                    i32 max_rn, rs_at_max;
                    i32 phredvalue = ref_to_phredvalue(inslen, max_rn, rs_at_max, S, rpos - off, P.indel_BQ_max, P.indel_polymerase_slip_rate, len, op);
                    const i64 x = rpos - off;
                    const i32 adp = S.p32(UVC_P_a_dp, x);
                    const i32 phredinc = (adp > 0 ? (i32)round(2 * numstates2phred((double)adp / (double)(1.0 + nnminus(adp, S.p32(UVC_P_a_at_ins_dp, x) + S.p32(UVC_P_a_at_del_dp, x))))) : PHREDINC_NO_DEPTH);
                    const i32 ratiothres = (!P.tumor_vcf_is_provided ? 2 : 4);
                    const bool is_multiallelic_ins = (S.p64(UVC_P_a_near_ins_pow2len, x) * ratiothres > (i64)max_(1, S.p32(UVC_P_a_near_ins_dp, x)) * (i64)((u32)len * 3u));
                    if (1 == inslen && !is_multiallelic_ins) phredvalue += between_(phredinc - 3, 0, 4);
                    const i32 thisdp = S.p32(UVC_P_a_at_ins_dp, x);
                    const i32 neardp = max_(S.p32(UVC_P_a_near_ins_dp, x), S.p32(UVC_P_a_near_RTR_ins_dp, x));
                    i32 insbase_minphred = 80;
                    for (i32 q2 = qpos; q2 < qpos + len; q2++) insbase_minphred = min_(insbase_minphred, Q(q2));
                    i32 ancbase_minphred = 80;
                    if (qpos > 0) ancbase_minphred = min_(ancbase_minphred, Q(qpos - 1));
                    if (qpos + len + 1 < a.l_qseq) ancbase_minphred = min_(ancbase_minphred, Q(qpos + len + 1));   // sic: +1 (main.hpp:2055-2056)
                    i32 minq = 80;
                    if (proton && (1 == len) && (1 == rs_at_max) && (1 < max_rn)) {
                        for (i32 qinc = 0; (qinc < max_rn + 2) && (qpos + qinc) < a.l_qseq; qinc++)
                            if (a.bases[qpos + qinc] == a.bases[qpos]) minq = min_(minq, Q(qpos + qinc));
                    }
                    const i32 qfromBQ1 = (proton ? min_(ancbase_minphred, minq) : min_(ancbase_minphred, insbase_minphred));
                    const i32 qfromBQ2 = ((thisdp * ratiothres <= neardp || (1 == len &&
                                (xm1500 >= P.microadjust_xm
                                 || ((lclip_len + P.microadjust_cliplen >= rpos - a.pos) && isrc)
                                 || ((rclip_len + P.microadjust_cliplen >= rend - a.pos) && !isrc))))
                            ? qfromBQ1 : (proton ? min_(qfromBQ1 + proton_cigarlen2phred(len), max_(3, qfromBQ1) * len) : 80));
                    incvalue = (i32)nnminus(min_(qfromBQ2, phredvalue + addPhred[1]), micro_indel_penal) + 1;
                }
                if (nbases2end >= P.indel_filter_edge_dist) {
                    const int symbol = ins_len_to_symbol(inslen);
                    inc(rpos, symbol, max_(1, incvalue));
                    if (bias) segbias(S, true, max_(1, incvalue), rpos, symbol, a, xm1500, bm1500s[symbol], op, len, 10000, dflag, clip_cnt);
                    if (tmp) {   // incIns, main.hpp:2101-2113
                        std::string iseq; i32 incvalue2 = incvalue;
                        for (i32 i2 = 0; i2 < len; i2++) { iseq.push_back("ACGTN"[a.bases[qpos + i2]]); incvalue2 = min_(incvalue2, Q(qpos + i2) + addPhred[1]); }
                        tmp->iseq[ins_idx(symbol)][rpos][iseq] += max_(1, incvalue2);
                    }
                }
            }
            qpos += len;
        } else if (op == C_DEL) {
            if ((normal_filter_primers || !is_assay_amplicon) || (ibeg <= rpos && rpos < iend)) {
                const i32 nbases2end = min_(qpos, a.l_qseq - qpos);
                const bool at_read_end = (nbases2end <= 0);
                i32 dellen = len;
                if (at_read_end) {
                    incvalue = (0 != qpos ? Q(qpos - 1) : ((qpos < a.l_qseq) ? Q(qpos) : 1)) + addPhred[1];
                } else {
                    i32 max_rn, rs_at_max;
                    i32 phredvalue = ref_to_phredvalue(dellen, max_rn, rs_at_max, S, rpos - off, P.indel_BQ_max, P.indel_polymerase_slip_rate, len, op);
                    const i64 x = rpos - off;
                    const i32 adp = S.p32(UVC_P_a_dp, x);
                    const i32 phredinc = (adp > 0 ? (i32)round(2 * numstates2phred((double)adp / (double)(1.0 + nnminus(adp, S.p32(UVC_P_a_at_ins_dp, x) + S.p32(UVC_P_a_at_del_dp, x))))) : PHREDINC_NO_DEPTH);
                    if (1 == dellen) phredvalue += between_(phredinc - 3, 0, 4);
                    const i32 thisdp = S.p32(UVC_P_a_at_del_dp, x);
                    const i32 neardp = max_(S.p32(UVC_P_a_near_del_dp, x), S.p32(UVC_P_a_near_RTR_del_dp, x));
                    i32 minq = 80;
                    if (proton && (1 == len) && (1 == rs_at_max) && (1 < max_rn)) {
                        for (i32 qinc = 0; qinc < (max_rn + 2) && (qpos + qinc) < a.l_qseq; qinc++)
                            if (a.bases[qpos + qinc] == a.bases[qpos]) minq = min_(minq, Q(qpos + qinc));
                    }
                    const i32 qfromBQ1 = min_(Q(qpos), min_(Q(qpos - 1), minq));
                    const i32 ratiothres = (!P.tumor_vcf_is_provided ? 2 : 4);
                    const i32 qfromBQ2 = ((thisdp * ratiothres <= neardp) ? (i32)nnminus(qfromBQ1, 1)
                                          : (proton ? min_(qfromBQ1 + proton_cigarlen2phred(len), max_(3, qfromBQ1) * len) : 80));
                    const double delFA = ((double)(thisdp + 0.5) / (double)(adp + 1));
                    const i32 delFAQ = max_(0, P.microadjust_delFAQmax + (i32)round(P.powlaw_exponent * numstates2phred(delFA)));
                    i32 prev_cidx = i, prev_rpos = rpos;
                    while ((0 != prev_cidx) && (C_INS != cig_op(cigar[prev_cidx]) || (u32)len != cig_len(cigar[prev_cidx]))) {
                        prev_cidx--;
                        const int o = cig_op(cigar[prev_cidx]);
                        if (C_MATCH == o || C_EQUAL == o || C_DIFF == o || C_DEL == o || C_REF_SKIP == o) prev_rpos -= (i32)cig_len(cigar[prev_cidx]);
                    }
                    i32 next_cidx = i, next_rpos = rpos + len;
                    while ((n_cigar - 1 != next_cidx) && (C_INS != cig_op(cigar[next_cidx]) || (u32)len != cig_len(cigar[next_cidx]))) {
                        next_cidx++;
                        const int o = cig_op(cigar[next_cidx]);
                        if (C_MATCH == o || C_EQUAL == o || C_DIFF == o || C_DEL == o || C_REF_SKIP == o) next_rpos += (i32)cig_len(cigar[next_cidx]);
                    }
                    const i32 qfromBAQl = (i32)(BAQ(rpos) - BAQ(prev_rpos));
                    const i32 qfromBAQr = (i32)(BAQ(next_rpos) - BAQ(rpos + len));
                    const i32 qfromBAQ = max_(delFAQ, max_(qfromBQ1, min_(qfromBAQl, qfromBAQr)));
                    incvalue = (i32)nnminus(min_(qfromBQ2, min_(qfromBAQ, phredvalue + addPhred[1])), micro_indel_penal) + 1;
                }
                if (nbases2end >= P.indel_filter_edge_dist) {
                    const int symbol = del_len_to_symbol(dellen);
                    inc(rpos, symbol, max_(1, incvalue));
                    if (bias) segbias(S, true, max_(1, incvalue), rpos, symbol, a, xm1500, bm1500s[symbol], op, len, 10000, dflag, clip_cnt);
                    if (tmp) tmp->dlen[del_idx(symbol)][rpos][len] += max_(1, incvalue);   // incDel, main.hpp:2216
                    for (i32 r2 = rpos; r2 < min_(rpos + len, rend); r2++) {   // padded deletion, main.hpp:2219-2253
                        for (int k = 0; k < 2; k++) {
                            const int s = (k == 0 ? UVC_BASE_NN : UVC_LINK_NN);
                            const i32 p = ((UVC_BASE_NN == s) ? r2 : (r2 + 1));
                            if (p >= rend) continue;
                            inc(p, s, max_(1, incvalue));
                            if (bias) {
                                if (indel_rposs[indel_rposs_idx] <= rpos) indel_rposs_idx++;
                                const u32 prev_indel_rpos = (u32)indel_rposs[indel_rposs_idx - 1];
                                const u32 next_indel_rpos = (u32)indel_rposs[indel_rposs_idx];
                                const u32 d = min_((u32)rpos - prev_indel_rpos, next_indel_rpos - (u32)rpos);
                                segbias(S, true, max_(1, incvalue), p, s, a, xm1500, bm1500s[s], op, len, (i32)d, dflag, clip_cnt);
                            }
                        }
                    }
                }
            }
            rpos += len;
        } else {
            if (op == C_REF_SKIP) rpos += len; else if (op == C_SOFT_CLIP) qpos += len;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// GenericSymbol2Count consensus primitives, main.hpp:374-520
// ------------------------------------------------------------------------------------------------
static void fill_consensus(const i32 *c, int &argmax, i32 &cmax, i32 &csum, int st, bool ref_once_in_link = false, bool ignore_padded_del = false) {
    const int b = st_beg(st);
    const int e = (st == UVC_BASE_SYMBOL ? (ignore_padded_del ? UVC_BASE_T : UVC_BASE_NN) : UVC_LINK_NN);
    const bool once = (st == UVC_LINK_SYMBOL && ref_once_in_link);
    argmax = e; cmax = 0; csum = 0;
    for (int s = b; s <= e; s++) {
        if (once) {
            if (cmax < c[s] || (UVC_LINK_M == argmax && (0 < c[s]))) { argmax = s; cmax = c[s]; csum = cmax; }
        } else {
            if (cmax < c[s]) { argmax = s; cmax = c[s]; }
            csum += c[s];
        }
    }
}

// GenericSymbol2CountCoverage::updateByFiltering<TIsIndelCounted=true, ., TIsRefCountedOnceInLink>, main.hpp:1659-1692 + :466-495
static void cov_update_by_filtering(Cov &dst, const Cov &src, i32 thres_base, i32 thres_link, bool padded_del_ignored, bool ref_once) {
    for (i32 epos = src.beg; epos < src.end; epos++) {
        int con[2];
        for (int st = 0; st < 2; st++) {
            int cs; i32 cc, ct;
            if (st == UVC_LINK_SYMBOL) fill_consensus(src.row(epos), cs, cc, ct, st, ref_once);
            else fill_consensus(src.row(epos), cs, cc, ct, st, false, padded_del_ignored);
            const i32 adj = max_(cc * 2, ct) - ct;
            if (adj >= (st == 0 ? thres_base : thres_link) && adj > 0) dst.at(epos, cs) += 1;
            con[st] = cs;
        }
        if (is_ins(con[1])) indel_update_by_consensus(dst.iseq[ins_idx(con[1])], src.iseq[ins_idx(con[1])], epos, 1);
        else if (is_del(con[1])) indel_update_by_consensus(dst.dlen[del_idx(con[1])], src.dlen[del_idx(con[1])], epos, 1);
    }
}
// updateByMajorMinusMinor, main.hpp:1694-1725 + :497-520
static void cov_update_by_mmm(Cov &dst, const Cov &src) {
    for (i32 epos = src.beg; epos < src.end; epos++) {
        for (int st = 0; st < 2; st++) {
            int cs; i32 cc, ct;
            fill_consensus(src.row(epos), cs, cc, ct, st, st == UVC_LINK_SYMBOL);
            const i32 adj = max_(cc * 2, ct) - ct;
            if (adj > 0) dst.at(epos, cs) += adj;
        }
    }
}

// fillTidBegEndFromAlns1, main.hpp:658-673 (the "+ 1" is applied once per alignment, cumulatively)
static void span_add_alns(const State &S, const Frag &f, i32 &beg, i32 &end) {
    for (int k = f.aln_beg; k < f.aln_end; k++) { beg = min_(beg, S.alns[k].pos); end = max_(end, S.alns[k].endpos) + 1; }
}

// ------------------------------------------------------------------------------------------------
// updateByAlns3UsingBQ, main.hpp:2543-2830  (P1, P1b, P2, P3, P3b)
// ------------------------------------------------------------------------------------------------
static int using_bq(State &S, std::string &err) {
    const UvcParams &P = S.P;
    int rc;
    for (const Family &fam : S.fams) for (int strand = 0; strand < 2; strand++)
        for (int f = fam.fs[strand].frag_beg; f < fam.fs[strand].frag_end; f++)
            for (int k = S.frags[f].aln_beg; k < S.frags[f].aln_end; k++)
                if ((rc = p1_prep_by_aln(S, S.alns[k], fam.dflag, err))) return rc;
    p1b_thres(S);
    for (const Family &fam : S.fams) for (int strand = 0; strand < 2; strand++)
        for (int f = fam.fs[strand].frag_beg; f < fam.fs[strand].frag_end; f++)
            for (int k = S.frags[f].aln_beg; k < S.frags[f].aln_end; k++)
                if ((rc = update_by_aln(S, S.alns[k], fam.dflag, true, NULL, err))) return rc;
    Cov tmp;
    std::vector<int8_t> cov_mut;
    std::vector<int> cm_base, cm_link;
    for (const Family &fam : S.fams) for (int strand = 0; strand < 2; strand++) {
        for (int f = fam.fs[strand].frag_beg; f < fam.fs[strand].frag_end; f++) {
            const Frag &fr = S.frags[f];
            i32 beg2 = INT32_MAX, end2 = 0;
            span_add_alns(S, fr, beg2, end2);
            end2 = min_(end2, S.end);
            tmp.init(beg2, end2);
            i32 normMQ = 0;
            for (int k = fr.aln_beg; k < fr.aln_end; k++) {
                if ((rc = update_by_aln(S, S.alns[k], fam.dflag, false, &tmp, err))) return rc;
                normMQ = max_(normMQ, S.alns[k].mapq);
            }
            const size_t tlen = (size_t)(end2 - beg2);
            cov_mut.assign(tlen, 0); cm_base.assign(tlen, UVC_NUM_SYMBOLS); cm_link.assign(tlen, UVC_NUM_SYMBOLS);
            State::MutForm pos_symbol_string;   // main.hpp:2650, 2734-2737
            for (i32 epos = beg2; epos < end2; epos++) {
                const i64 x = epos - S.beg;
                for (int vi = 0; vi < 2; vi++) {
                    const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);   // SYMBOL_TYPES_IN_VCF_ORDER, main_conversion.hpp:461
                    const int refsymbol = S.refsym[x];
                    int con_symbol; i32 con_count, tot_count;
                    fill_consensus(tmp.row(epos), con_symbol, con_count, tot_count, st, st == UVC_LINK_SYMBOL);
                    if (0 == tot_count) continue;
                    const i32 max_qual = 8 + S.BQS(con_symbol, x) / max_(1, S.seg_ad(con_symbol, x));   // get_avgBQ, main_conversion.hpp:791-796
                    const i32 con_qual = con_count * 2 - tot_count;
                    i32 phredlike;
                    if (0x1 & P.fam_flag) phredlike = min_(con_qual, min_(max_qual, sscs_phred(P, refsymbol, con_symbol)));
                    else phredlike = min_(con_qual, max_qual);
                    int pbucket = max_(0, max_qual - phredlike);
                    if (pbucket < NBUCKETS) S.BK(0, con_symbol, pbucket, x) += 1;
                    S.FR(strand, UVC_FRAG_bDP, con_symbol, x) += 1;
                    S.VQ(UVC_VQ_bMQ, con_symbol, x) += (normMQ * normMQ) / SQR_QUAL_DIV_;
                    if (is_ins(con_symbol)) indel_update_by_consensus(S.gap_frag[strand].iseq[ins_idx(con_symbol)], tmp.iseq[ins_idx(con_symbol)], epos, 1);   // main.hpp:2710-2717
                    if (is_del(con_symbol)) indel_update_by_consensus(S.gap_frag[strand].dlen[del_idx(con_symbol)], tmp.dlen[del_idx(con_symbol)], epos, 1);
                    cov_mut[epos - beg2] |= 0x1;
                    const bool is_var_of_highBQ = ((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform)
                            ? (UVC_BASE_SYMBOL == st || con_qual + 3 >= P.bias_thres_highBQ)
                            : (UVC_LINK_SYMBOL == st || con_qual >= P.bias_thres_highBQ));
                    if (symbols_mutated(refsymbol, con_symbol) && is_var_of_highBQ) { pos_symbol_string.push_back(std::make_pair(epos, con_symbol)); cov_mut[epos - beg2] |= 0x2; }
                    if (UVC_LINK_SYMBOL == st) cm_link[epos - beg2] = con_symbol; else cm_base[epos - beg2] = con_symbol;
                }
            }
            if (pos_symbol_string.size() > 1) { auto &c = S.hapmap[0][pos_symbol_string]; c[strand]++; }   // (a fresh entry is value-initialised to {0, 0})
            for (size_t i = 0; i < tlen; i++) if (cov_mut[i] & 0x2) {
                for (int j = (int)i - (int)P.syserr_mut_region_n_bases; j < (int)(i + P.syserr_mut_region_n_bases + 1); j++)
                    if ((0 <= j) && (j < (int)tlen)) cov_mut[j] |= 0x4;
            }
            i32 n_cov = 0, n_near = 0;
            for (size_t i = 0; i < tlen; i++) if (cov_mut[i] & 0x1) { n_cov++; if (cov_mut[i] & 0x4) n_near++; }
            for (i32 epos = beg2; epos < end2; epos++) {
                const size_t iv = epos - beg2;
                const int two[2] = { cm_base[iv], cm_link[iv] };
                for (int t = 0; t < 2; t++) if (two[t] != UVC_NUM_SYMBOLS) {
                    S.FR(strand, UVC_FRAG_bTA, two[t], epos - S.beg) += n_cov;
                    S.FR(strand, UVC_FRAG_bTB, two[t], epos - S.beg) += n_near;
                }
            }
        }
    }
    // P3b, main.hpp:2801-2828
    for (i64 x = 0; x < S.npos; x++) {
        for (int st = 0; st < 2; st++) {
            i32 totDP = 0;
            for (int k = 0; k < ST_NSYMBOLS[st]; k++) totDP += S.FR(0, UVC_FRAG_bDP, ST_SYMBOLS[st][k], x) + S.FR(1, UVC_FRAG_bDP, ST_SYMBOLS[st][k], x);
            for (int k = 0; k < ST_NSYMBOLS[st]; k++) {
                const int symbol = ST_SYMBOLS[st][k];
                const i32 max_qual = 8 + S.BQS(symbol, x) / max_(1, S.seg_ad(symbol, x));
                i32 distr[NBUCKETS];
                for (int b = 0; b < NBUCKETS; b++) distr[b] = S.BK(0, symbol, b, x);
                i32 mv, ad, bq;
                infer_max_qual_assuming_independence(mv, ad, bq, max_qual, 1, distr, totDP);
                S.VQ(UVC_VQ_bIAQb, symbol, x) += mv; S.VQ(UVC_VQ_bIADb, symbol, x) += ad; S.VQ(UVC_VQ_bIDQb, symbol, x) += bq;
            }
        }
        for (int s = 0; s < NSYM; s++) for (int b = 0; b < NBUCKETS; b++) S.BK(0, s, b, x) = 0;   // clearSymbolBucketCount
    }
    return 0;
}

// median helper: MEDIAN(v) of main_conversion.hpp:24-28 is applied to the vector AS FILLED (not sorted)
static i32 median_as_is(const std::vector<i32> &v) { return (v[(v.size() - 1) / 2] + v[v.size() / 2]) / 2; }

// ------------------------------------------------------------------------------------------------
// updateByAlns3UsingFQ, main.hpp:2832-3594  (P4, P5, P5b; consensus-FASTQ emission left out)
// ------------------------------------------------------------------------------------------------
static int using_fq(State &S, std::string &err) {
    const UvcParams &P = S.P;
    const bool proton = (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform);
    const bool padded_del_ignored = (P.microadjust_padded_deletion_flag & (proton ? 0x2 : 0x1)) != 0;
    int rc;
    Cov tmp, con, mmm, dup;
    // ---- P4 ----
    for (const Family &fam : S.fams) for (int strand = 0; strand < 2; strand++) {
        const FamStrand &fs = fam.fs[strand];
        const int nfrags = fs.frag_end - fs.frag_beg;
        if (nfrags == 0) continue;
        i32 beg2 = INT32_MAX, end2 = 0;
        for (int f = fs.frag_beg; f < fs.frag_end; f++) span_add_alns(S, S.frags[f], beg2, end2);
        end2 = min_(end2, S.end);
        con.init(beg2, end2);
        for (int f = fs.frag_beg; f < fs.frag_end; f++) {
            i32 b1 = INT32_MAX, e1 = 0;
            span_add_alns(S, S.frags[f], b1, e1);
            e1 = min_(e1, S.end);
            tmp.init(b1, e1);
            for (int k = S.frags[f].aln_beg; k < S.frags[f].aln_end; k++) if ((rc = update_by_aln(S, S.alns[k], fam.dflag, false, &tmp, err))) return rc;
            cov_update_by_filtering(con, tmp, P.fam_thres_highBQ_snv, 0, padded_del_ignored, true);
        }
        std::vector<i32> l2r_end_poss, r2l_end_poss;
        i32 qseqlen_sum = 0, n_qseqs = 0;
        for (int f = fs.frag_beg; f < fs.frag_end; f++) for (int k = S.frags[f].aln_beg; k < S.frags[f].aln_end; k++) {
            const Aln &a = S.alns[k];
            if (a.isrc()) r2l_end_poss.push_back(a.pos); else l2r_end_poss.push_back(a.endpos);
            qseqlen_sum += a.l_qseq; n_qseqs += 1;
        }
        const i32 l2r_end_median_pos = (l2r_end_poss.size() > 0 ? median_as_is(l2r_end_poss) : con.end);
        const i32 r2l_end_median_pos = (r2l_end_poss.size() > 0 ? median_as_is(r2l_end_poss) : con.beg);
        const bool fam_has_nonconf_middle = (l2r_end_median_pos <= (r2l_end_median_pos + P.indel_adj_tracklen_dist));
        i32 nsb_min = con.end, nsb_max = con.beg;
        if ((nfrags >= P.fam_thres_dup1add) && (qseqlen_sum >= n_qseqs * P.fam_thres_qseqlen)) {
            i32 poss[2] = { con.end, con.beg };
            for (int i = 0; i < 2; i++) {
                i64 b = (i ? (con.end - 1) : con.beg), e = (i ? ((i64)con.beg - 1) : con.end), step = (i ? -1 : 1);
                for (i64 epos = b; epos != e; epos += step) {
                    int cs; i32 cc, ct;
                    fill_consensus(con.row((i32)epos), cs, cc, ct, UVC_BASE_SYMBOL);
                    if (0 == ct) continue;
                    const bool good = ((P.fam_thres_dup1add <= ct) && (cc * 100 >= ct * P.fam_thres_dup1perc) && ((fam.dflag & 0x1) || (P.fam_flag & 0x2)));
                    if (good && (UVC_BASE_N != cs) && (UVC_BASE_NN != cs)) { poss[i] = (i32)epos; break; }
                }
            }
            nsb_min = poss[0]; nsb_max = poss[1];
        }
        for (i32 epos = con.beg; epos < con.end; epos++) {
            const i64 x = epos - S.beg;
            const i32 *row = con.row(epos);
            for (int vi = 0; vi < 2; vi++) {
                const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
                int con_symbol; i32 con_count, tot_count;
                fill_consensus(row, con_symbol, con_count, tot_count, st);
                const bool is_fam_good = ((P.fam_thres_dup1add <= tot_count) && (con_count * 100 >= tot_count * P.fam_thres_dup1perc)
                                          && ((fam.dflag & 0x1) || (P.fam_flag & 0x2)));
                if (0 == tot_count) continue;
                S.FA(strand, UVC_FAM_cDP12, con_symbol, x) += 1;
                if (1 == tot_count) S.FA(strand, UVC_FAM_cDP21, con_symbol, x) += 1;
                if (!P.inferred_is_vcf_generated) continue;
                if (is_fam_good) {
                    S.FA(strand, UVC_FAM_cDP2, con_symbol, x) += 1;
                    if (is_ins(con_symbol)) indel_update_by_consensus(S.gap_c2[strand].iseq[ins_idx(con_symbol)], con.iseq[ins_idx(con_symbol)], epos, 1);   // main.hpp:3196-3206
                    if (is_del(con_symbol)) indel_update_by_consensus(S.gap_c2[strand].dlen[del_idx(con_symbol)], con.dlen[del_idx(con_symbol)], epos, 1);
                    // FAM2 bias, main.hpp:3208-3319
                    const i32 rpos = epos;
                    i32 rbeg = min_(nsb_min, epos), rend = max_(nsb_max, epos);
                    if (fam_has_nonconf_middle && epos < r2l_end_median_pos) rend = max_(min_(l2r_end_median_pos, min_(r2l_end_median_pos, rend)), epos);
                    if (fam_has_nonconf_middle && l2r_end_median_pos < epos) rbeg = min_(max_(l2r_end_median_pos, max_(r2l_end_median_pos, rbeg)), epos);
                    const bool isGap = (UVC_LINK_SYMBOL == st);
                    const i32 bq = 90; const i32 dist = 1024 * 1024;
                    if (((!isGap) && bq >= P.bias_thres_highBQ) || (isGap && dist >= P.bias_thres_highBQ)) {
                        const bool tier2 = (isGap || bq >= P.bias_thres_highBQ);
                        const i32 l_nb = (i32)nnminus(epos + 1, rbeg), r_nb = (i32)nnminus(rend, epos);
                        const i32 _LPxT = S.th(UVC_T_aLPxT, x), RPxT = S.th(UVC_T_aRPxT, x);
                        const i32 LPxT = (isGap ? _LPxT : min_(_LPxT, RPxT));
                        i32 indel_len = 0;   // sic: .first of getMajority is the COUNT (main.hpp:3239-3243)
                        if (is_ins(con_symbol)) {
                            std::map<std::string, i32> m; m[""] = 0;
                            const int order[3] = { UVC_LINK_I1, UVC_LINK_I2, UVC_LINK_I3P };   // INS_SYMBOLS, main_conversion.hpp:357
                            for (int o = 0; o < 3; o++) { auto it = con.iseq[ins_idx(order[o])].find(epos); if (it != con.iseq[ins_idx(order[o])].end()) m.insert(it->second.begin(), it->second.end()); }
                            indel_len = get_majority(m).first;
                        } else if (is_del(con_symbol)) {
                            std::map<i32, i32> m; m[0] = 0;
                            const int order[3] = { UVC_LINK_D1, UVC_LINK_D2, UVC_LINK_D3P };
                            for (int o = 0; o < 3; o++) { auto it = con.dlen[del_idx(order[o])].find(epos); if (it != con.dlen[del_idx(order[o])].end()) m.insert(it->second.begin(), it->second.end()); }
                            indel_len = get_majority(m).first;
                        }
                        const bool far = (l_nb + (is_ins(con_symbol) ? (i32)nnminus(indel_len, P.microadjust_nobias_pos_indel_maxlen) : 0) >= LPxT) && (r_nb >= RPxT);
                        if (far) {
                            i64 LPL = S.FI(UVC_FI_c2LPL, con_symbol, x), RPL = S.FI(UVC_FI_c2RPL, con_symbol, x);
                            bidir_bias(S.FI(UVC_FI_c2LP1, con_symbol, x), S.FI(UVC_FI_c2LP2, con_symbol, x), S.FI(UVC_FI_c2RP1, con_symbol, x), S.FI(UVC_FI_c2RP2, con_symbol, x), LPL, RPL,
                                       S.th(UVC_T_aLP1t, x), S.th(UVC_T_aLP2t, x), S.th(UVC_T_aRP1t, x), S.th(UVC_T_aRP2t, x), l_nb, r_nb, true, 0);
                            S.FI(UVC_FI_c2LPL, con_symbol, x) = (i32)LPL; S.FI(UVC_FI_c2RPL, con_symbol, x) = (i32)RPL;
                        }
                        if ((i32)nnminus(epos + 1, nsb_min) >= P.bias_thres_strict_c2LRP0) S.FI(UVC_FI_c2LP0, con_symbol, x) += 1;
                        if ((i32)nnminus(nsb_max, epos) >= P.bias_thres_strict_c2LRP0) S.FI(UVC_FI_c2RP0, con_symbol, x) += 1;
                        auto BAQ = [&](i64 p) -> i64 { return S.baq[p - S.beg]; };
                        auto BAQ2 = [&](i64 p) -> i64 { return S.baq2[p - S.beg]; };
                        const i64 baq_last = S.end - 1;
                        const i32 seg_l_baq = (i32)(BAQ(rpos) - BAQ(max_((i64)rbeg, nnminus(rpos, MAX_STR_N_BASES_))) + 1);
                        const i32 _seg_r_baq = (i32)(BAQ(min_((i64)rend - 1, min_((i64)rpos + MAX_STR_N_BASES_, baq_last))) - BAQ(rpos) + 1);
                        const i32 seg_r_baq = (isGap ? (i32)min_((i64)_seg_r_baq, BAQ2(min_((i64)rend - 1, min_((i64)rpos + MAX_STR_N_BASES_, baq_last))) - BAQ2(rpos) + 7) : _seg_r_baq);
                        const i32 thres_highBAQ = P.bias_thres_highBAQ + (isGap ? 0 : 3);
                        if (seg_l_baq >= thres_highBAQ && seg_r_baq >= thres_highBAQ) {
                            bidir_bias(S.FI(UVC_FI_c2LB1, con_symbol, x), S.FI(UVC_FI_c2LB2, con_symbol, x), S.FI(UVC_FI_c2RB1, con_symbol, x), S.FI(UVC_FI_c2RB2, con_symbol, x),
                                       S.FI64(UVC_FI64_c2LBL, con_symbol, x), S.FI64(UVC_FI64_c2RBL, con_symbol, x),
                                       P.bias_thres_BAQ1, P.bias_thres_BAQ2, P.bias_thres_BAQ1, P.bias_thres_BAQ2, seg_l_baq, seg_r_baq, tier2, 0);
                        }
                        S.FI(UVC_FI_c2BQ2, con_symbol, x) += 1;
                    }
                }
                if (P.fam_thres_dup2add <= tot_count && (con_count * 100 >= tot_count * P.fam_thres_dup2perc)) S.FA(strand, UVC_FAM_cDP3, con_symbol, x) += 1;
                if (is_ins(con_symbol)) indel_update_by_consensus(S.gap_fam[strand].iseq[ins_idx(con_symbol)], con.iseq[ins_idx(con_symbol)], epos, 1);   // main.hpp:3327-3336
                if (is_del(con_symbol)) indel_update_by_consensus(S.gap_fam[strand].dlen[del_idx(con_symbol)], con.dlen[del_idx(con_symbol)], epos, 1);
                const i32 flat = (is_subst(con_symbol) ? P.fam_thres_emperr_all_flat_snv : P.fam_thres_emperr_all_flat_indel);
                const i32 perc = (is_subst(con_symbol) ? P.fam_thres_emperr_con_perc_snv : P.fam_thres_emperr_con_perc_indel);
                if (tot_count < flat) continue;
                if (con_count * 100 < tot_count * perc) continue;
                for (int k = 0; k < ST_NSYMBOLS[st]; k++) {
                    const int symbol = ST_SYMBOLS[st][k];
                    if (con_symbol != symbol) {
                        S.FA(strand, UVC_FAM_cDPm, con_symbol, x) += row[symbol];
                        S.FA(strand, UVC_FAM_cDPM, con_symbol, x) += tot_count;
                    }
                }
            }
        }
    }
    if (!P.inferred_is_vcf_generated) return 0;
    // ---- P5 ----
    for (const Family &fam : S.fams) {
        i32 dbeg = INT32_MAX, dend = 0;
        for (int strand = 0; strand < 2; strand++)
            for (int f = fam.fs[strand].frag_beg; f < fam.fs[strand].frag_end; f++) span_add_alns(S, S.frags[f], dbeg, dend);
        dend = min_(dend, S.end);
        const int n0 = fam.fs[0].frag_end - fam.fs[0].frag_beg, n1 = fam.fs[1].frag_end - fam.fs[1].frag_beg;
        bool will_inc_dscs = false, will_inc_sscs = false;
        if ((0x2 == (fam.dflag & 0x2)) && n0 > 0 && n1 > 0) will_inc_dscs = true;
        else if ((0x2 == (fam.dflag & 0x2)) && (n0 <= 0 || n1 <= 0)) will_inc_sscs = true;
        if (will_inc_dscs) dup.init(dbeg, dend);
        for (int strand = 0; strand < 2; strand++) {
            const FamStrand &fs = fam.fs[strand];
            if (fs.frag_end == fs.frag_beg) continue;
            i32 beg2 = INT32_MAX, end2 = 0;
            for (int f = fs.frag_beg; f < fs.frag_end; f++) span_add_alns(S, S.frags[f], beg2, end2);
            end2 = min_(end2, S.end);
            con.init(beg2, end2); mmm.init(beg2, end2);
            for (int f = fs.frag_beg; f < fs.frag_end; f++) {
                i32 b1 = INT32_MAX, e1 = 0;
                span_add_alns(S, S.frags[f], b1, e1);
                e1 = min_(e1, S.end);
                tmp.init(b1, e1);
                for (int k = S.frags[f].aln_beg; k < S.frags[f].aln_end; k++) if ((rc = update_by_aln(S, S.alns[k], fam.dflag, false, &tmp, err))) return rc;
                cov_update_by_filtering(con, tmp, P.fam_thres_highBQ_snv, 0, padded_del_ignored, true);
                cov_update_by_mmm(mmm, tmp);
            }
            if (will_inc_dscs) cov_update_by_filtering(dup, con, 1, 1, padded_del_ignored, false);   // <true,false,false>, main.hpp:3429-3432
            State::MutForm pos_symbol_string, pos_symbol_string_confam;   // main.hpp:3434-3435
            for (i32 epos = mmm.beg; epos < mmm.end; epos++) {
                const i64 x = epos - S.beg;
                for (int vi = 0; vi < 2; vi++) {
                    const int st = (vi == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
                    int con_symbol; i32 con_sumBQs, tot_sumBQs;
                    fill_consensus(mmm.row(epos), con_symbol, con_sumBQs, tot_sumBQs, st);
                    if (0 == tot_sumBQs) continue;
                    const i32 con_nfrags = con.row(epos)[con_symbol];
                    i32 tot_nfrags = 0;
                    for (int s = st_beg(st); s <= st_end(st); s++) tot_nfrags += con.row(epos)[s];
                    S.FA(strand, UVC_FAM_cDP1, con_symbol, x) += 1;
                    if (will_inc_sscs && (!will_inc_dscs) && (tot_nfrags >= P.fam_thres_dup1add) && (con_nfrags * 100 >= tot_nfrags * P.fam_thres_dup1perc))
                    {
                        S.FA(strand, UVC_FAM_cDPD, con_symbol, x) += 1;
                        if (is_ins(con_symbol)) indel_update_by_consensus(S.gap_c2d[strand].iseq[ins_idx(con_symbol)], con.iseq[ins_idx(con_symbol)], epos, 1);   // main.hpp:3458-3469
                        if (is_del(con_symbol)) indel_update_by_consensus(S.gap_c2d[strand].dlen[del_idx(con_symbol)], con.dlen[del_idx(con_symbol)], epos, 1);
                    }
                    const i32 avgBQ = ((0 == tot_nfrags) ? 1 : (con_sumBQs / tot_nfrags));
                    const i32 majorcount = S.FA(strand, UVC_FAM_cDPM, con_symbol, x);
                    const i32 minorcount = S.FA(strand, UVC_FAM_cDPm, con_symbol, x);
                    const double prior_weight = 1.0 / (minorcount + 1.0);
                    const double phred2prob_avgBQ = pow(10, -((float)avgBQ) / 10);   // phred2prob: float cast, main_conversion.hpp:885-888
                    const double prob = (minorcount + prior_weight) / (majorcount + minorcount + prior_weight / phred2prob_avgBQ);
                    const double realphred = -10 * log(prob) / log(10);               // prob2realphred, main_conversion.hpp:895-898
                    const i32 indep_frag_phred = (i32)round(((con_nfrags * 2) - tot_nfrags) * realphred);
                    i32 confam_qual;
                    if (UVC_LINK_SYMBOL == st) confam_qual = max_(1, min_(indep_frag_phred, (i32)P.fam_phred_indel_inc_before_barcode_labeling + (i32)round(realphred)));
                    else confam_qual = max_(1, min_(indep_frag_phred, (con_sumBQs * 2) - tot_sumBQs));
                    const int ref_symbol = S.refsym[x];
                    const bool is_var_of_highBQ = (proton ? (UVC_BASE_SYMBOL == st || max_(confam_qual + 3, avgBQ) >= P.bias_thres_highBQ)
                                                          : (UVC_LINK_SYMBOL == st || confam_qual >= P.bias_thres_highBQ));   // main.hpp:3490-3505
                    if (symbols_mutated(ref_symbol, con_symbol) && is_var_of_highBQ) {
                        pos_symbol_string.push_back(std::make_pair(epos, con_symbol));
                        int con_symbol1; i32 con_count1, tot_count1;
                        fill_consensus(con.row(epos), con_symbol1, con_count1, tot_count1, st);
                        if (con_symbol == con_symbol1 && P.fam_thres_dup1add <= tot_count1 && (con_count1 * 100 >= tot_count1 * P.fam_thres_dup1perc)) pos_symbol_string_confam.push_back(std::make_pair(epos, con_symbol));
                    }
                    const i32 max_qual = sscs_phred(P, ref_symbol, con_symbol) + (!P.tumor_vcf_is_provided ? 0 : 4);
                    const i32 confam_qual2 = min_(confam_qual, max_qual);
                    if (tot_nfrags >= P.fam_thres_dup1add) {
                        const int pbucket = (max_qual - confam_qual2 + 2) / 4;
                        if (pbucket >= 0 && pbucket < NBUCKETS) S.BK(strand, con_symbol, pbucket, x) += 1;   // the reference's .at() would throw beyond 15
                    }
                }
            }
            if (pos_symbol_string.size() > 1) { auto &c = S.hapmap[1][pos_symbol_string]; c[strand]++; }               // main.hpp:3514-3517
            if (pos_symbol_string_confam.size() > 1) { auto &c = S.hapmap[2][pos_symbol_string_confam]; c[strand]++; }   // main.hpp:3518-3521
        }
        if (will_inc_dscs) {
            for (i32 epos = dup.beg; epos < dup.end; epos++) for (int st = 0; st < 2; st++) {
                int cs; i32 cc, ct;
                fill_consensus(dup.row(epos), cs, cc, ct, st);
                if (0 < ct) S.DU(UVC_DUPLEX_dDP1, cs, epos - S.beg) += 1;
                if (1 < ct) {
                    S.DU(UVC_DUPLEX_dDP2, cs, epos - S.beg) += 1;
                    for (int strand = 0; strand < 2; strand++) {   // main.hpp:3535-3546
                        if (is_ins(cs)) indel_update_by_consensus(S.gap_c2d[strand].iseq[ins_idx(cs)], dup.iseq[ins_idx(cs)], epos, 1);
                        if (is_del(cs)) indel_update_by_consensus(S.gap_c2d[strand].dlen[del_idx(cs)], dup.dlen[del_idx(cs)], epos, 1);
                    }
                }
            }
        }
    }
    // ---- P5b ----
    for (int strand = 0; strand < 2; strand++) {
        const int qIAQ = (strand ? UVC_VQ_cIAQr : UVC_VQ_cIAQf), qIAD = (strand ? UVC_VQ_cIADr : UVC_VQ_cIADf), qIDQ = (strand ? UVC_VQ_cIDQr : UVC_VQ_cIDQf);
        for (i64 x = 0; x < S.npos; x++) {
            const int ref_symbol = S.refsym[x];
            for (int st = 0; st < 2; st++) {
                i32 totDP = 0;
                for (int k = 0; k < ST_NSYMBOLS[st]; k++) totDP += S.FA(strand, UVC_FAM_cDP1, ST_SYMBOLS[st][k], x);
                for (int k = 0; k < ST_NSYMBOLS[st]; k++) {
                    const int symbol = ST_SYMBOLS[st][k];
                    const i32 max_qual = sscs_phred(P, ref_symbol, symbol) + (!P.tumor_vcf_is_provided ? 0 : 4);
                    i32 distr[NBUCKETS];
                    for (int b = 0; b < NBUCKETS; b++) distr[b] = S.BK(strand, symbol, b, x);
                    i32 mv, ad, bq;
                    infer_max_qual_assuming_independence(mv, ad, bq, max_qual, 4, distr, totDP);
                    S.VQ(qIAQ, symbol, x) += mv; S.VQ(qIAD, symbol, x) += ad; S.VQ(qIDQ, symbol, x) += bq;
                }
            }
        }
    }
    return 0;
}

// updateHapMap, main.hpp:3596-3663.  (tsum_depth only lends its range to the per-position counter tsum_depth_2.)
static void update_hap_map(State &S, int which) {
    const UvcParams &P = S.P;
    typedef std::tuple<i32, State::MutForm, std::array<i32, 2>> Row;
    std::vector<Row> v;
    for (const auto &it : S.hapmap[which]) v.push_back(Row(it.second[0] + it.second[1], it.first, it.second));
    std::sort(v.rbegin(), v.rend());
    const size_t num_dst = min_((size_t)P.phasing_haplotype_max_detail_cnt, v.size());
    std::vector<i32> inc_fw(num_dst, 0), inc_rv(num_dst, 0);
    for (size_t i = 0; i < num_dst; i++) {
        const State::MutForm &dst = std::get<1>(v[i]);
        for (size_t j = i + 1; j < v.size(); j++) {
            const State::MutForm &src = std::get<1>(v[j]);
            bool skipped = false;
            for (const auto &al : dst) if (std::find(src.begin(), src.end(), al) == src.end()) { skipped = true; break; }
            if (!skipped) { inc_fw[i] += std::get<2>(v[j])[0]; inc_rv[i] += std::get<2>(v[j])[1]; }
        }
    }
    std::vector<i32> tsum((size_t)S.npos + 1, 0);
    for (size_t i = 0; i < v.size(); i++) {
        const State::MutForm &form = std::get<1>(v[i]);
        const std::array<i32, 2> &cnt = std::get<2>(v[i]);
        if ((cnt[0] + cnt[1]) < (P.phasing_haplotype_min_ad + (i32)form.size())) continue;
        i32 haplo_totDP = 0;
        for (const auto &sm : form) { tsum[(size_t)(sm.first - S.beg)] += 1; haplo_totDP += tsum[(size_t)(sm.first - S.beg)]; }
        if ((i64)haplo_totDP > (i64)P.phasing_haplotype_max_count * (i64)form.size()) continue;
        State::HapLink h; h.form = form; h.fr[0] = cnt[0]; h.fr[1] = cnt[1];
        h.other[0] = (i >= num_dst ? -1 : inc_fw[i]); h.other[1] = (i >= num_dst ? -1 : inc_rv[i]);
        S.haplinks[which].push_back(h);
    }
}
// mutform2count4vec_to_simplemut2indices (main.cpp:82-97) + mutform2count4map_to_phase (main.hpp:5380-5404) for one (refpos, symbol)
std::string hap_phase_string(const State &S, int which, i32 refpos, int symbol) {
    static const char *const DESC[] = { "A", "C", "G", "T", "N", "*", "<LR>", "<LD3P>", "<LD2>", "<LD1>", "<LI3P>", "<LI2>", "<LI1>", "*" };
    std::string out;
    for (const State::HapLink &h : S.haplinks[which]) {
        if (h.fr[0] + h.fr[1] < 2) continue;
        if (std::find(h.form.begin(), h.form.end(), std::make_pair(refpos, symbol)) == h.form.end()) continue;
        if (!((h.fr[0] + h.fr[1]) > 1)) continue;
        out += "(";
        for (const auto &ps : h.form) out += std::string("(") + std::to_string(ps.first + (ps.second <= UVC_BASE_NN ? 1 : 0)) + "&" + DESC[ps.second] + ")";
        const std::string add = ((-1 < h.other[0]) ? ("&&" + std::to_string(h.other[0] + h.fr[0]) + "&" + std::to_string(h.other[1] + h.fr[1])) : std::string());
        out += std::string("&") + std::to_string(h.fr[0]) + "&" + std::to_string(h.fr[1]) + add + ")";
    }
    return out;
}

// updateByRegion3Aln, main.hpp:3665-3742
int accumulate(State &S, std::string &err) {
    if (S.alns.empty()) { err = "no reads"; return UVCGPU_ENOREADS; }
    auto zero32 = [&](std::vector<i32> &v, size_t n) { v.assign(n, 0); };
    zero32(S.prep32, (size_t)UVC_NPREP32 * S.npos); S.prep64.assign((size_t)UVC_NPREP64 * S.npos, 0);
    zero32(S.thres, (size_t)UVC_NTHRES * S.npos);
    zero32(S.seg32, (size_t)UVC_NSEG32 * NSYM * S.npos); S.seg64.assign((size_t)UVC_NSEG64 * NSYM * S.npos, 0);
    zero32(S.vq, (size_t)UVC_NVQ * NSYM * S.npos); zero32(S.bqsum, (size_t)NSYM * S.npos);
    zero32(S.frag, (size_t)2 * UVC_NFRAG * NSYM * S.npos); zero32(S.fam, (size_t)2 * UVC_NFAM * NSYM * S.npos);
    zero32(S.faminfo32, (size_t)UVC_NFAMINFO32 * NSYM * S.npos); S.faminfo64.assign((size_t)UVC_NFAMINFO64 * NSYM * S.npos, 0);
    zero32(S.duplex, (size_t)UVC_NDUPLEX * NSYM * S.npos);
    for (int s = 0; s < 2; s++) zero32(S.bucket[s], (size_t)NSYM * NBUCKETS * S.npos);
    for (int st = 0; st < 2; st++) { S.gap_frag[st].clear(); S.gap_fam[st].clear(); S.gap_c2[st].clear(); S.gap_c2d[st].clear(); }
    for (int k = 0; k < 3; k++) { S.hapmap[k].clear(); S.haplinks[k].clear(); }
    build_side_arrays(S);   // rtr.indelphred is mutated by P1b, so rebuild on every accumulate
    int rc;
    if (S.P.inferred_is_vcf_generated) { if ((rc = using_bq(S, err))) return rc; update_hap_map(S, 0); }
    if ((rc = using_fq(S, err))) return rc;
    update_hap_map(S, 1); update_hap_map(S, 2);
    S.accumulated = true;
    return 0;
}

}  // namespace uvco
