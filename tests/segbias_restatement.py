"""An independent Python restatement of dealwith_segbias<isGap> and update_bidirectional_bias, written from the reference text
(/root/reference/main.hpp:1316-1595; COMPILATION_ENABLE_XMGOT == 0, common.hpp:4), not from oracle/ or the kernels.  The CPU suite holds
the oracle against it call by call on fuzzed arguments (tests/test_segbias_cpu.py).  Test infrastructure."""

MAX_INSERT_SIZE = 2000   # common.hpp:64
SQR_QUAL_DIV = 32        # main_conversion.hpp:20
BAM_CINS = 1


def cdiv(a, b):
    """C++ integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def non_neg_minus(a, b):   # common.hpp
    return a - b if a > b else 0


def update_bidirectional_bias(info, kL1, kL2, kR1, kR2, kLL, kRL, L1, L2, R1, R2, nl, nr, is_BQ_high_enough_for_tier2, n_indel):
    """main.hpp:1316-1358"""
    if nl + n_indel >= L1:
        info[kL1] += 1
    if nl + n_indel >= L2 and is_BQ_high_enough_for_tier2:
        info[kL2] += 1
    if nr >= R1:
        info[kR1] += 1
    if nr >= R2 and is_BQ_high_enough_for_tier2:
        info[kR2] += 1
    info[kLL] += nl
    info[kRL] += nr


SEG_FIELDS = ("a2XM2 a2BM2 aPF1 aPF2 aBQ2 aMQs aP1 aP2 aP3 aNC aDPff aDPfr aDPrf aDPrr aLP1 aLP2 aLPL aRP1 aRP2 aRPL aLB1 aLB2 aRB1 aRB2 "
              "aLI1 aLI2 aRI1 aRI2 aRIf aLIr aLBL aRBL aLIT aRIT a1BQf a1BQr a2BQf a2BQr").split()   # UVC_S_*, UVC_S64_*, then the four VQ sums


def dealwith_segbias(isGap, bq, rpos, thres, aln, xm1500, bm1500, baq, baq2, region_beg, cigar_op, indel_len_arg, dist_to_interfering_indel, dflag, clip_cnt, P):
    """One call on an empty SegFormatInfoSet / VQFormatTagSet; returns {field: increment}.  `thres` = SegFormatThresSet of rpos as a dict
    (aLPxT aRPxT aLI1T aLI2T aRI1T aRI2T aLI1t aLI2t aRI1t aRI2t aLP1t aLP2t aRP1t aRP2t ...), `aln` = dict(pos, endpos, mpos, isize, flag, qual),
    baq / baq2 = the two prefix-sum arrays indexed by position - region_beg."""
    info = {k: 0 for k in SEG_FIELDS}
    is_assay_amplicon = bool(dflag & 0x4) or (P.primerlen > 0 and not (0x2 & P.primer_flag))
    is_normal_used_to_filter_vars_on_primers = bool(P.tn_is_paired and (0x1 & P.primer_flag))
    is_assay_UMI = bool(dflag & 0x1)
    indel_len = int(indel_len_arg)
    bias_thres_veryhighBQ = P.bias_thres_highBQ
    rend = aln["endpos"]

    def B(p): return int(baq[p - region_beg])
    def B2(p): return int(baq2[p - region_beg])
    seg_l_baq1 = B(rpos) - B(aln["pos"]) + 1
    _seg_r_baq = B(rend - 1) - B(rpos) + 1
    seg_r_baq1 = min(_seg_r_baq, B2(rend - 1) - B2(rpos) + 7) if isGap else _seg_r_baq
    seg_l_nbases = rpos - aln["pos"] + 1
    seg_r_nbases = rend - rpos
    is_high_readlen = P.central_readlen >= P.microadjust_median_readlen_thres
    seg_l_baq = seg_l_baq1 if is_high_readlen else max(seg_l_baq1, cdiv(seg_l_nbases * P.microadjust_BAQ_per_base_x1024, 1024))
    seg_r_baq = seg_r_baq1 if is_high_readlen else max(seg_r_baq1, cdiv(seg_r_nbases * P.microadjust_BAQ_per_base_x1024, 1024))
    frag_pos_L = min(aln["pos"], aln["mpos"])
    frag_pos_R = frag_pos_L + abs(aln["isize"])
    frag_l_nbases2 = min(rpos - frag_pos_L + 1, MAX_INSERT_SIZE) if aln["isize"] != 0 else MAX_INSERT_SIZE
    frag_r_nbases2 = min(frag_pos_R - rpos + 0, MAX_INSERT_SIZE) if aln["isize"] != 0 else MAX_INSERT_SIZE
    flag = aln["flag"]
    is_normal = (aln["isize"] != 0) or (0 == (flag & 0x1))
    isrc = (flag & 0x10) == 0x10
    strand = bool(flag & 0x20) if (flag & 0x81) == 0x81 else bool(flag & 0x10)   # bam_get_strand, common.hpp:90

    info["a1BQr" if isrc else "a1BQf"] += bq
    info["a2BQr" if isrc else "a2BQf"] += cdiv(bq * bq, SQR_QUAL_DIV)
    info["aMQs"] += aln["qual"]
    info[("aDPrr" if isrc else "aDPrf") if strand else ("aDPfr" if isrc else "aDPff")] += 1
    if min(dist_to_interfering_indel, seg_l_nbases, seg_r_nbases) >= P.bias_thres_interfering_indel:
        info["aP3"] += 1
    if 0 == clip_cnt:
        info["aNC"] += 1
    if isrc:
        info["aLIT"] += frag_l_nbases2 if aln["isize"] != 0 else 0
    else:
        info["aRIT"] += frag_r_nbases2 if aln["isize"] != 0 else 0

    _const_LPxT = thres["aLPxT"]
    const_RPxT = thres["aRPxT"]
    const_LPxT = _const_LPxT if isGap else min(_const_LPxT, const_RPxT)
    is_far_from_edge = (seg_l_nbases + (non_neg_minus(indel_len, P.microadjust_nobias_pos_indel_maxlen) if BAM_CINS == cigar_op else 0) >= const_LPxT) and (seg_r_nbases >= const_RPxT)
    bias_thres_highBAQ = P.bias_thres_highBAQ + (0 if isGap else 3)
    is_unaffected_by_edge = seg_l_baq >= bias_thres_highBAQ and seg_r_baq >= bias_thres_highBAQ
    min_dist2iend = min(frag_l_nbases2, frag_r_nbases2) if (flag & 0x1) else (seg_r_nbases if isrc else seg_l_nbases)
    if is_far_from_edge and is_unaffected_by_edge and (min_dist2iend > P.primerlen2 or not is_assay_amplicon):
        info["aP1"] += 1
    if is_assay_UMI or not is_assay_amplicon:
        info["aP2"] += 1

    def sq(v): return v * v
    if isGap:
        ampfact1 = 100
        ampfact2 = 100
        if bq < P.bias_thres_PFBQ1:
            ampfact2 = cdiv(100 * sq(bq), sq(P.bias_thres_PFBQ1))
        info["aPF1"] += min(ampfact1, ampfact2)
        ampfact1 = 100
        ampfact2 = 100
        if bq < P.bias_thres_PFBQ2:
            ampfact2 = cdiv(100 * sq(bq), sq(P.bias_thres_PFBQ2))
        info["aPF2"] += min(ampfact1, ampfact2)
    else:
        ampfact1 = 100
        ampfact2 = 100
        if bq < P.bias_thres_PFBQ1:
            ampfact2 = cdiv(100 * sq(bq), sq(P.bias_thres_PFBQ1))
        info["aPF1"] += cdiv(ampfact1 * ampfact2, 100)
        ampfact1 = 100
        ampfact2 = 100
        if bq < P.bias_thres_PFBQ2:
            ampfact2 = cdiv(100 * sq(bq), sq(P.bias_thres_PFBQ2))
        info["aPF2"] += cdiv(ampfact1 * ampfact2, 100)
        info["a2XM2"] += cdiv(100 * sq(20), sq(xm1500)) if xm1500 > 20 else 100
        info["a2BM2"] += cdiv(100 * sq(20), sq(bm1500)) if bm1500 > 20 else 100
    if ((not isGap) and bq >= P.bias_thres_highBQ) or (isGap and dist_to_interfering_indel >= P.bias_thres_interfering_indel):
        is_BQ_high_enough_for_tier2 = isGap or bq >= bias_thres_veryhighBQ
        if is_far_from_edge:
            update_bidirectional_bias(info, "aLP1", "aLP2", "aRP1", "aRP2", "aLPL", "aRPL", thres["aLP1t"], thres["aLP2t"], thres["aRP1t"], thres["aRP2t"],
                                      seg_l_nbases, seg_r_nbases, is_BQ_high_enough_for_tier2, indel_len)
        if is_unaffected_by_edge:
            update_bidirectional_bias(info, "aLB1", "aLB2", "aRB1", "aRB2", "aLBL", "aRBL", P.bias_thres_BAQ1, P.bias_thres_BAQ2, P.bias_thres_BAQ1, P.bias_thres_BAQ2,
                                      seg_l_baq, seg_r_baq, is_BQ_high_enough_for_tier2, 0)
        info["aBQ2"] += 1
    mate_mapped_or_single = (0 == (flag & 0x8)) or (0 == (flag & 0x1))
    is_l_nonbiased = mate_mapped_or_single and seg_l_nbases > seg_r_nbases
    is_r_nonbiased = mate_mapped_or_single and seg_l_nbases < seg_r_nbases
    is_pos_good_for_bias_calc = (not is_assay_amplicon) or (not is_normal_used_to_filter_vars_on_primers) or (is_far_from_edge and is_unaffected_by_edge)
    if isrc:
        dist2iend = frag_l_nbases2
        if dist2iend >= thres["aLI1t"] and (dist2iend <= thres["aLI1T"] or isGap) and (is_normal or (isGap and is_l_nonbiased)):
            info["aLI1"] += 1
        if dist2iend >= thres["aLI2t"] and (dist2iend <= thres["aLI2T"] or isGap) and (is_normal or (isGap and is_l_nonbiased)):
            if is_pos_good_for_bias_calc:
                info["aLI2"] += 1
        if is_pos_good_for_bias_calc:
            info["aLIr"] += 1
    else:
        dist2iend = frag_r_nbases2
        if dist2iend >= thres["aRI1t"] and (dist2iend <= thres["aRI1T"] or isGap) and (is_normal or (isGap and is_r_nonbiased)):
            info["aRI1"] += 1
        if dist2iend >= thres["aRI2t"] and (dist2iend <= thres["aRI2T"] or isGap) and (is_normal or (isGap and is_r_nonbiased)):
            if is_pos_good_for_bias_calc:
                info["aRI2"] += 1
        if is_pos_good_for_bias_calc:
            info["aRIf"] += 1
    return info
