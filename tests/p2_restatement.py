"""Independent Python restatement, written from the reference text (not from oracle/ and not from the kernels), of
GenericSymbol2CountCoverage::updateByAln<TIsProton, SYMBOL_COUNT_SUM, TIsBiasUpdated = true> (main.hpp:1762-2296) with ref_to_phredvalue
(main.hpp:876-922), indel_len_rusize_phred (757-790), proton_cigarlen2phred (main_conversion.hpp:922-941) and insLenToSymbol / delLenToSymbol
(main_conversion.hpp:436-447): the per-read CIGAR walk of P2 -- which (position, symbol) every read updates, with which value,
and the arguments of every dealwith_segbias call.  Together with tests/segbias_restatement.py it yields every SegFormatInfoSet counter, the
four VQ sums and the SYMBOL_COUNT_SUM plane of a region; tests/test_p2_cpu.py holds the oracle's planes against that."""
import math

import numpy as np

from rtr_cases import _indel_phred, _more_str
from segbias_restatement import SEG_FIELDS, cdiv, dealwith_segbias, non_neg_minus

def cround(v):
    """C round(): halves away from zero (Python's round() goes to even)."""
    return math.floor(v + 0.5) if v >= 0 else math.ceil(v - 0.5)


C_MATCH, C_INS, C_DEL, C_REF_SKIP, C_SOFT_CLIP, C_HARD_CLIP, C_PAD, C_EQUAL, C_DIFF = range(9)
BASE_NN, LINK_M, LINK_D3P, LINK_D2, LINK_D1, LINK_I3P, LINK_I2, LINK_I1, LINK_NN = 5, 6, 7, 8, 9, 10, 11, 12, 13
NSYM = 14
INT32_MAX = 2 ** 31 - 1
N_UNITS_TO_PHRED = [0, 0, 3, 5, 6, 7, 8, 8, 9, 10, 10, 10, 11, 11, 11, 12, 12, 12, 13]   # main.hpp:762-781
PROTON_OPLEN2PHRED = [0] + [int(cround(10.0 / math.log(10.0) * math.log(i ** 3))) for i in range(1, 13)]   # main_conversion.hpp:925-939


def u32(v):
    return v & 0xFFFFFFFF


def i32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def indel_len_rusize_phred(indel_len, ru):
    if indel_len % ru == 0:
        return N_UNITS_TO_PHRED[min(indel_len // ru, 18)]
    return N_UNITS_TO_PHRED[min(indel_len, 18)]


def ref_to_phredvalue(codes, refpos, max_phred, ampfact, oplen, op, strmax, del_to_ins):
    """-> (phredvalue, n_units, max_repeatnum, repeatsize_at_max_repeatnum); `codes` = region_symbolvec (its size is the reference's length)."""
    n = len(codes)
    max_rn = rs_at = 0
    for rs in range(1, strmax + 1):
        q = refpos
        while q + rs < n and codes[q] == codes[q + rs]:
            q += 1
        rn = (q - refpos) // rs + 1
        if _more_str(rs, rn, rs_at, max_rn, strmax):
            max_rn, rs_at = rn, rs
    if oplen == rs_at and op == C_DEL:
        ampfact *= del_to_ins
    dec = _indel_phred(ampfact, rs_at, max_rn)
    if rs_at * (max_rn - 1) >= 6 - 1:
        n_units = (oplen // rs_at) if oplen % rs_at == 0 else (1 if oplen == 1 else 0)
    else:
        n_units = 1 + oplen // 6
    return max_phred - min(max_phred, dec) + indel_len_rusize_phred(oplen, rs_at), n_units, max_rn, rs_at


def read_events(reads, i, P, rtr, indelphred, baq, codes, prep, thres, proton, with_bias=True):
    """The updates of alignment i in the order updateByAln makes them: a list of (is_gap, value, position, symbol, cigar_op, indel_len,
    dist_to_interfering_indel) -- one inc<TUpdateType>() each, and with TIsBiasUpdated (with_bias) one dealwith_segbias<is_gap>() call with
    these arguments.  Also returns the per-read constants those calls take and the allele-keyed updates: (events, aln, xm1500, bm1500s, dflag, clip_cnt, gaps)."""
    beg = int(reads["beg"])
    atd = int(P.indel_adj_tracklen_dist)
    n_rtr = rtr.shape[1]
    add_b, add_l = int(P.bq_phred_added_misma), int(P.bq_phred_added_indel)
    normal_filter = bool(P.tn_is_paired and (0x1 & P.primer_flag))
    ratiothres = 2 if not P.tumor_vcf_is_provided else 4
    pos = int(reads["pos"][i]); flag = int(reads["flag"][i]); isize = int(reads["isize"][i]); mpos = int(reads["mpos"][i])
    lq = int(reads["l_qseq"][i]); so = int(reads["seq_off"][i])
    cig = [(int(c) & 0xF, int(c) >> 4) for c in reads["cigars"][int(reads["cigar_off"][i]): int(reads["cigar_off"][i]) + int(reads["n_cigar"][i])]]
    nc = len(cig)
    bs = [int(b) for b in reads["bases"][so: so + lq]]; Q = [int(q) for q in reads["quals"][so: so + lq]]
    dflag = int(reads["fam_dflag"][int(reads["fam_id"][i])])
    rend = pos + sum(l for o, l in cig if o in (C_MATCH, C_EQUAL, C_DIFF, C_DEL, C_REF_SKIP))
    if rend == pos:
        rend = pos + 1
    aln = dict(pos=pos, endpos=rend, mpos=mpos, isize=isize, flag=flag, qual=int(reads["mapq"][i]))
    amplicon = bool(dflag & 0x4) or (P.primerlen > 0 and not (0x2 & P.primer_flag))
    nge = sum(l for o, l in cig if o in (C_INS, C_DEL)); ngo = sum(1 for o, l in cig if o in (C_INS, C_DEL))
    clip_cnt = sum(1 for o, l in cig if o in (C_SOFT_CLIP, C_HARD_CLIP))
    nm = int(reads["nm"][i])
    nm_cnt = nm if nm >= 0 else nge
    xm1500 = cdiv((nm_cnt - nge) * 1500, rend - pos); go1500 = cdiv(ngo * 1500, rend - pos)
    # first walk: mismatches per base symbol, positions of the low-quality InDels
    indel_rposs = [0]
    bm = [0] * NSYM
    qpos, rpos = 0, pos
    for op, ln in cig:
        if op in (C_MATCH, C_EQUAL, C_DIFF):
            for _ in range(ln):
                if int(codes[rpos - beg]) != bs[qpos]:
                    bm[bs[qpos]] += 1
                qpos += 1; rpos += 1
        elif op == C_INS:
            low = False
            for q2 in range(qpos - min(qpos, 1), min(qpos + ln + 1, rend)):     # (sic: the bound is `rend`, a reference position)
                if Q[min(max(q2, 0), lq - 1)] < int(P.bias_thres_interfering_indel_BQ):
                    low = True
            if low:
                indel_rposs.append(rpos)
            qpos += ln
        elif op == C_DEL:
            if min(Q[min(max(max(1, qpos) - 1, 0), lq - 1)], Q[min(qpos, lq - 1)]) <= int(P.bias_thres_interfering_indel_BQ):
                indel_rposs.append(rpos)
            rpos += ln
        elif op == C_REF_SKIP:
            rpos += ln
        elif op == C_SOFT_CLIP:
            qpos += ln
    indel_rposs.append(INT32_MAX)
    bm1500 = [cdiv(c * 1500, rend - pos) for c in bm]
    isrc = (flag & 0x10) == 0x10
    pl = int(P.primerlen)
    if isize != 0:
        ibeg = min(pos, mpos) + pl; iend = non_neg_minus(min(pos, mpos) + abs(isize), pl)
    elif isrc and 0 == (flag & 0x1):
        ibeg = 0; iend = non_neg_minus(rend, pl)
    else:
        ibeg = pos + pl; iend = INT32_MAX
    lclip = cig[0][1] if (nc > 0 and cig[0][0] == C_SOFT_CLIP) else 0
    rclip = cig[-1][1] if (nc > 0 and cig[-1][0] == C_SOFT_CLIP) else 0
    by_clip = cdiv(max(lclip, rclip), 6); by_nm = cdiv(xm1500 + go1500, 30)
    indel_penal = min(1, by_nm + by_clip); nogap_penal = min(4, by_nm + by_clip) + 1
    idx = 0
    incvalue = 1

    ev = []
    gaps = []   # (position, symbol, inserted text | deleted length, weight): the allele-keyed maps of incIns / incDel

    def bias(is_gap, bq, p, sym, op, indel_len, dist):   # one inc<>() + (with TIsBiasUpdated) one dealwith_segbias<is_gap>() of the reference
        ev.append((is_gap, bq, p, sym, op, indel_len, dist))

    def gated(rp):
        return (normal_filter or not amplicon) or (ibeg <= rp < iend)
    qpos, rpos = 0, pos
    for ci, (op, ln) in enumerate(cig):
        if op in (C_MATCH, C_EQUAL, C_DIFF):
            for i2 in range(ln):
                if gated(rpos):
                    dist = 10000
                    if with_bias and nge > 0:
                        if indel_rposs[idx] <= rpos:
                            idx += 1
                        prev_ir, next_ir = indel_rposs[idx - 1], indel_rposs[idx]
                        i1 = max(rpos - beg, atd) - atd; i2r = min(rpos - beg + atd, n_rtr - 1)
                        prevlen = non_neg_minus(rpos - prev_ir, max(rpos - (beg + int(rtr[0][i1])), int(thres["aLP1t"][rpos - beg])))
                        nextlen = non_neg_minus(next_ir - rpos, max((beg + int(rtr[0][i2r]) + int(rtr[1][i2r])) - rpos, int(thres["aRP1t"][rpos - beg])))
                        dist = min(prevlen, nextlen)
                    if i2 > 0:
                        noindel = min(int(indelphred[rpos - beg - 1]), int(indelphred[rpos - beg]))
                        qfromBQ2 = min(Q[qpos - 1], Q[qpos]) if proton else 80
                        incvalue = non_neg_minus(min(qfromBQ2, noindel), nogap_penal) + 1
                        bias(True, incvalue, rpos, LINK_M, op, 0, dist)
                    sym = bs[qpos]
                    if proton and (i2 == 0 or i2 == ln - 1):
                        # the packed neighbouring cigar words are compared with the bare op codes (main.hpp:1953-1956): both "is gap" flags hold at the ends of the op
                        prev_c = ((cig[ci - 1][1] << 4) | cig[ci - 1][0]) if ci > 0 else u32(-1)
                        next_c = ((cig[ci + 1][1] << 4) | cig[ci + 1][0]) if ci + 1 < nc else u32(-1)
                        next_gap = (i2 == ln - 1) and next_c not in (C_MATCH, C_EQUAL, C_DIFF)
                        prev_gap = (i2 == 0) and prev_c not in (C_MATCH, C_EQUAL, C_DIFF)
                        if next_gap or prev_gap:
                            isrc2 = (i2 != 0)
                            pbp = 1
                            if isrc2 and qpos + 1 < lq:
                                pbp = Q[qpos + 1]
                            if (not isrc2) and qpos > 0:
                                pbp = Q[qpos - 1]
                            adj = 100
                            if next_gap:
                                adj = min(adj, cig[ci + 1][1] if ci + 1 < nc else 100)
                            if prev_gap:
                                adj = min(adj, cig[ci - 1][1] if ci > 0 else 100)
                            incvalue = min(Q[qpos], pbp) + (min(add_b, add_l) if adj < 3 else add_b)
                        else:
                            incvalue = Q[qpos] + add_b
                    else:
                        incvalue = Q[qpos] + add_b
                    bias(False, incvalue, rpos, sym, op, 0, dist)
                rpos += 1; qpos += 1
        elif op == C_INS:
            if gated(rpos):
                nb2end = min(qpos, lq - (qpos + ln))
                inslen = ln
                if nb2end <= 0:
                    incvalue = (Q[qpos - 1] if qpos != 0 else (Q[qpos + ln] if qpos + ln < lq else 1)) + add_l
                else:
                    x = rpos - beg
                    phredvalue, inslen, max_rn, rs_at = ref_to_phredvalue(codes, x, int(P.indel_BQ_max), float(P.indel_polymerase_slip_rate), ln, op,
                                                                          int(P.indel_str_repeatsize_max), float(P.indel_del_to_ins_err_ratio))
                    adp = int(prep["a_dp"][x]); at_i = int(prep["a_at_ins_dp"][x]); at_d = int(prep["a_at_del_dp"][x])
                    # a_dp == 0 (an insertion in front of a reference skip) is undefined behaviour in the reference (round(-inf) -> int): "no bonus",
                    # the convention DESIGN.md 7 documents for the library
                    phredinc = int(cround(2 * (10.0 / math.log(10.0)) * math.log(adp / (1.0 + non_neg_minus(adp, at_i + at_d))))) if adp > 0 else -1000000
                    multi = int(prep["a_near_ins_pow2len"][x]) * ratiothres > i32(max(1, int(prep["a_near_ins_dp"][x])) * i32(u32(ln * 3)))
                    if inslen == 1 and not multi:
                        phredvalue += min(max(0, phredinc - 3), 4)
                    thisdp = at_i; neardp = max(int(prep["a_near_ins_dp"][x]), int(prep["a_near_RTR_ins_dp"][x]))
                    ins_min = min([80] + [Q[q2] for q2 in range(qpos, qpos + ln)])
                    anc_min = 80
                    if qpos > 0:
                        anc_min = min(anc_min, Q[qpos - 1])
                    if qpos + ln + 1 < lq:
                        anc_min = min(anc_min, Q[qpos + ln + 1])
                    minq = 80
                    if proton and ln == 1 and rs_at == 1 and max_rn > 1:
                        qinc = 0
                        while qinc < max_rn + 2 and qpos + qinc < lq:
                            if bs[qpos + qinc] == bs[qpos]:
                                minq = min(minq, Q[qpos + qinc])
                            qinc += 1
                    q1 = min(anc_min, minq) if proton else min(anc_min, ins_min)
                    cond = thisdp * ratiothres <= neardp or (ln == 1 and (xm1500 >= int(P.microadjust_xm)
                                                                        or ((lclip + int(P.microadjust_cliplen) >= rpos - pos) and isrc)
                                                                        or ((rclip + int(P.microadjust_cliplen) >= rend - pos) and not isrc)))
                    q2v = q1 if cond else (min(q1 + PROTON_OPLEN2PHRED[min(ln, 12)], max(3, q1) * ln) if proton else 80)
                    incvalue = non_neg_minus(min(q2v, phredvalue + add_l), indel_penal) + 1
                if nb2end >= int(P.indel_filter_edge_dist):
                    sym = LINK_I1 if inslen == 1 else (LINK_I2 if inslen == 2 else LINK_I3P)
                    bias(True, max(1, incvalue), rpos, sym, op, ln, 10000)
                    incvalue2 = incvalue                                   # incIns, main.hpp:2101-2113: the inserted bases as text, the weakest of them caps the weight
                    for q2 in range(qpos, qpos + ln):
                        incvalue2 = min(incvalue2, Q[q2] + add_l)
                    gaps.append((rpos, sym, "".join("ACGTN"[bs[q2]] for q2 in range(qpos, qpos + ln)), max(1, incvalue2)))
            qpos += ln
        elif op == C_DEL:
            if gated(rpos):
                nb2end = min(qpos, lq - qpos)
                dellen = ln
                if nb2end <= 0:
                    incvalue = (Q[qpos - 1] if qpos != 0 else (Q[qpos] if qpos < lq else 1)) + add_l
                else:
                    x = rpos - beg
                    phredvalue, dellen, max_rn, rs_at = ref_to_phredvalue(codes, x, int(P.indel_BQ_max), float(P.indel_polymerase_slip_rate), ln, op,
                                                                          int(P.indel_str_repeatsize_max), float(P.indel_del_to_ins_err_ratio))
                    adp = int(prep["a_dp"][x]); at_i = int(prep["a_at_ins_dp"][x]); at_d = int(prep["a_at_del_dp"][x])
                    phredinc = int(cround(2 * (10.0 / math.log(10.0)) * math.log(adp / (1.0 + non_neg_minus(adp, at_i + at_d))))) if adp > 0 else -1000000
                    if dellen == 1:
                        phredvalue += min(max(0, phredinc - 3), 4)
                    thisdp = at_d; neardp = max(int(prep["a_near_del_dp"][x]), int(prep["a_near_RTR_del_dp"][x]))
                    minq = 80
                    if proton and ln == 1 and rs_at == 1 and max_rn > 1:
                        qinc = 0
                        while qinc < max_rn + 2 and qpos + qinc < lq:
                            if bs[qpos + qinc] == bs[qpos]:
                                minq = min(minq, Q[qpos + qinc])
                            qinc += 1
                    q1 = min(Q[qpos], Q[qpos - 1], minq)
                    q2v = non_neg_minus(q1, 1) if thisdp * ratiothres <= neardp else (min(q1 + PROTON_OPLEN2PHRED[min(ln, 12)], max(3, q1) * ln) if proton else 80)
                    delFA = (thisdp + 0.5) / float(adp + 1)
                    delFAQ = max(0, int(P.microadjust_delFAQmax) + int(cround(float(P.powlaw_exponent) * (10.0 / math.log(10.0)) * math.log(delFA))))
                    pc, prev_rpos = ci, rpos
                    while pc != 0 and (cig[pc][0] != C_INS or cig[pc][1] != ln):
                        pc -= 1
                        if cig[pc][0] in (C_MATCH, C_EQUAL, C_DIFF, C_DEL, C_REF_SKIP):
                            prev_rpos -= cig[pc][1]
                    nx, next_rpos = ci, rpos + ln
                    while nx != nc - 1 and (cig[nx][0] != C_INS or cig[nx][1] != ln):
                        nx += 1
                        if cig[nx][0] in (C_MATCH, C_EQUAL, C_DIFF, C_DEL, C_REF_SKIP):
                            next_rpos += cig[nx][1]
                    baq_l = i32(int(baq[rpos - beg]) - int(baq[prev_rpos - beg])); baq_r = i32(int(baq[next_rpos - beg]) - int(baq[rpos + ln - beg]))
                    q_baq = max(delFAQ, q1, min(baq_l, baq_r))
                    incvalue = non_neg_minus(min(q2v, q_baq, phredvalue + add_l), indel_penal) + 1
                if nb2end >= int(P.indel_filter_edge_dist):
                    sym = LINK_D1 if dellen == 1 else (LINK_D2 if dellen == 2 else LINK_D3P)
                    v = max(1, incvalue)
                    bias(True, v, rpos, sym, op, ln, 10000)
                    gaps.append((rpos, sym, ln, v))                        # incDel, main.hpp:2216
                    for r2 in range(rpos, min(rpos + ln, rend)):            # the padded deletion, main.hpp:2219-2253
                        for s in (BASE_NN, LINK_NN):
                            p = r2 if s == BASE_NN else r2 + 1
                            if p >= rend:
                                continue
                            dist = 0
                            if with_bias:
                                if indel_rposs[idx] <= rpos:
                                    idx += 1
                                prev_ir, next_ir = u32(indel_rposs[idx - 1]), u32(indel_rposs[idx])
                                dist = i32(min(u32(rpos - prev_ir), u32(next_ir - rpos)))
                            bias(True, v, p, s, op, ln, dist)
            rpos += ln
        elif op == C_REF_SKIP:
            rpos += ln
        elif op == C_SOFT_CLIP:
            qpos += ln
    return ev, aln, xm1500, bm1500, dflag, clip_cnt, gaps


def update_by_aln(reads, P, rtr, indelphred, baq, baq2, codes, prep, thres, proton):
    """Every alignment of the region through updateByAln<proton, SYMBOL_COUNT_SUM, true>.  `rtr` [7][npos] (UVC_RTR order; begpos / tracklen
    rows are read), `indelphred` = the track's indelphred AFTER P1b, `prep` / `thres` = the dicts of tests/prep_restatement.py.
    Returns (seg {field: int64 [NSYM][npos]} for SEG_FIELDS, bqsum int64 [NSYM][npos])."""
    beg = int(reads["beg"])
    npos = int(reads["end"]) - beg + 1
    seg = {k: np.zeros((NSYM, npos), dtype=np.int64) for k in SEG_FIELDS}
    bqsum = np.zeros((NSYM, npos), dtype=np.int64)
    for i in range(int(reads["n_reads"])):
        ev, aln, xm1500, bm1500, dflag, clip_cnt, _ = read_events(reads, i, P, rtr, indelphred, baq, codes, prep, thres, proton, True)
        for is_gap, bq, p, sym, op, indel_len, dist in ev:
            bqsum[sym][p - beg] += bq
            th = {k: int(thres[k][p - beg]) for k in thres}
            inc = dealwith_segbias(is_gap, bq, p, th, aln, xm1500, bm1500[sym], baq, baq2, beg, op, indel_len, dist, dflag, clip_cnt, P)
            for k, v in inc.items():
                if v:
                    seg[k][sym][p - beg] += v
    return seg, bqsum
