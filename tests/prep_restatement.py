"""Independent Python restatements, written from the reference text (not from oracle/ and not from the kernels), of

  * update_seg_format_prep_sets_by_aln      main.hpp:924-1204   (the SegFormatPrepSet counters of every position, P1)
  * update_seg_format_thres_from_prep_sets  main.hpp:1206-1301  (row a5: the SegFormatThresSet thresholds and the edit of
                                                                  RegionalTandemRepeat::indelphred, P1b; COMPILATION_ENABLE_XMGOT == 0)

tests/test_prep_cpu.py holds the oracle against them on fuzzed reads (several InDels per read, clips, reference skips, amplicon flags).
C++ semantics that matter are spelled out: int32 / int64 truncating division, unsigned cigar lengths, double -> int truncation."""
import math

import numpy as np

MAX_INSERT_SIZE = 2000   # common.hpp:64
C_MATCH, C_INS, C_DEL, C_REF_SKIP, C_SOFT_CLIP, C_HARD_CLIP, C_PAD, C_EQUAL, C_DIFF = range(9)

PREP32 = ("a_dp a_near_ins_dp a_near_del_dp a_near_RTR_ins_dp a_near_RTR_del_dp a_pcr_dp a_umi_dp a_snv_dp a_dnv_dp a_highBQ_dp "
          "a_near_pcr_clip_dp a_near_long_clip_dp a_at_ins_dp a_at_del_dp a_XM1500 a_GO1500 a_GAPLEN a_qlen a_near_ins_inv100len a_near_del_inv100len "
          "a_LIDP a_RIDP a_l_dist_sum a_r_dist_sum a_inslen_sum a_dellen_sum").split()                     # UVC_P_* order of UVC_F_PREP32
PREP64 = ("a_near_ins_pow2len a_near_del_pow2len a_near_ins_l_pow2len a_near_ins_r_pow2len a_near_del_l_pow2len a_near_del_r_pow2len "
          "a_LI a_RI a_l_BAQ_sum a_r_BAQ_sum a_insBAQ_sum a_delBAQ_sum").split()                            # UVC_F_PREP64
THRES = "aLPxT aRPxT aLI1T aLI2T aRI1T aRI2T aLI1t aLI2t aRI1t aRI2t aLP1t aLP2t aRP1t aRP2t aLB1t aLB2t aRB1t aRB2t".split()   # UVC_F_THRES


def cdiv(a, b):
    """C++ integer division: truncates toward zero."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def i32(v):
    """wrap to int32 (what an int32_t field keeps)."""
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def cround(v):
    """C round(): halves away from zero (Python's round() goes to even)."""
    return math.floor(v + 0.5) if v >= 0 else math.ceil(v - 0.5)


def u32(v):
    return v & 0xFFFFFFFF


def prep_sets(reads, P, rtr, baq, ref_codes):
    """P1 over every alignment.  `rtr` = int [7][npos] in UVC_RTR order (begpos, tracklen, unitlen, ...), `baq` = int64 [npos] (the first BAQ
    prefix-sum array), `ref_codes` = region_symbolvec.  Returns {name: int64 array [npos]} for PREP32 + PREP64 (values unwrapped)."""
    beg = int(reads["beg"])
    npos = int(reads["end"]) - beg + 1
    out = {k: np.zeros(npos, dtype=np.int64) for k in PREP32 + PREP64}
    excl_end = beg + npos                                    # baq_offsetarr.getExcluEndPosition()
    atd = int(P.indel_adj_tracklen_dist)
    n_rtr = rtr.shape[1]
    cig_all = reads["cigars"]

    def B(p):                                                # baq_offsetarr.getByPos
        return int(baq[p - beg])
    for i in range(int(reads["n_reads"])):
        pos = int(reads["pos"][i]); flag = int(reads["flag"][i]); isize = int(reads["isize"][i]); mpos = int(reads["mpos"][i])
        lq = int(reads["l_qseq"][i]); so = int(reads["seq_off"][i])
        cig = [(int(c) & 0xF, int(c) >> 4) for c in cig_all[int(reads["cigar_off"][i]): int(reads["cigar_off"][i]) + int(reads["n_cigar"][i])]]
        bases = reads["bases"][so: so + lq]; quals = reads["quals"][so: so + lq]
        dflag = int(reads["fam_dflag"][int(reads["fam_id"][i])])
        rend = pos + sum(l for o, l in cig if o in (C_MATCH, C_EQUAL, C_DIFF, C_DEL, C_REF_SKIP))   # bam_endpos
        if rend == pos:
            rend = pos + 1
        # first walk: gap statistics of the alignment
        nge = ngo = 0
        insbaq = delbaq = inslen = dellen = 0
        qpos, rpos = 0, pos
        for op, ln in cig:
            if op in (C_INS, C_DEL):
                nge += ln; ngo += 1
                d = B(min(rpos + ln, excl_end - 1)) - B(rpos)
                if op == C_INS:
                    insbaq = i32(insbaq + d); inslen += ln; qpos += ln
                else:
                    delbaq = i32(delbaq + d); dellen += ln; rpos += ln
            elif op in (C_MATCH, C_EQUAL, C_DIFF):
                qpos += ln; rpos += ln
            elif op == C_REF_SKIP:
                rpos += ln
            elif op == C_SOFT_CLIP:
                qpos += ln
        nm = int(reads["nm"][i])
        nm_cnt = nm if nm >= 0 else nge                     # no NM tag: nge_cnt
        xm_cnt = nm_cnt - nge
        xm1500 = cdiv(xm_cnt * 1500, rend - pos)
        go1500 = cdiv(ngo * 1500, rend - pos)
        avg_gaplen = cdiv(nge, max(1, ngo))
        fpl = min(pos, mpos); fpr = fpl + abs(isize)
        isrc = (flag & 0x10) == 0x10
        pcr = 1 if (dflag & 0x4) else 0
        umi = 1 if (dflag & 0x1) else 0
        qpos, rpos = 0, pos
        for ci, (op, ln) in enumerate(cig):
            if op in (C_MATCH, C_EQUAL, C_DIFF):
                for _ in range(ln):
                    x = rpos - beg
                    out["a_pcr_dp"][x] += pcr; out["a_umi_dp"][x] += umi; out["a_dp"][x] += 1
                    out["a_qlen"][x] += rend - pos; out["a_XM1500"][x] += xm1500; out["a_GO1500"][x] += go1500; out["a_GAPLEN"][x] += avg_gaplen
                    if isize != 0:
                        if isrc:
                            out["a_LI"][x] += min(rpos - fpl + 1, MAX_INSERT_SIZE); out["a_LIDP"][x] += 1
                        else:
                            out["a_RI"][x] += min(fpr - rpos, MAX_INSERT_SIZE); out["a_RIDP"][x] += 1
                    refsymbol, readsymbol = 5, 14                        # BASE_NN, END_ALIGNMENT_SYMBOLS
                    nq, nr = qpos, rpos
                    while refsymbol != readsymbol and nq < lq and nr < rend:
                        refsymbol = int(ref_codes[nr - beg]); readsymbol = int(bases[nq]); nq += 1; nr += 1
                    if nr == rpos + 2:
                        for r in range(max(pos, rpos - 1), min(nr, rend)):
                            out["a_snv_dp"][r - beg] += 1
                    if nr > rpos + 2:
                        for r in range(max(pos, rpos - 1), min(nr, rend)):
                            out["a_dnv_dp"][r - beg] += 1
                    if int(quals[qpos]) >= int(P.bias_thres_highBQ):
                        out["a_l_dist_sum"][x] += rpos - pos + 1; out["a_r_dist_sum"][x] += rend - rpos
                        out["a_inslen_sum"][x] += inslen; out["a_dellen_sum"][x] += dellen
                        out["a_l_BAQ_sum"][x] += i32(B(rpos) - B(pos) + 1); out["a_r_BAQ_sum"][x] += i32(B(rend - 1) - B(rpos) + 1)
                        out["a_insBAQ_sum"][x] += insbaq; out["a_delBAQ_sum"][x] += delbaq
                        out["a_highBQ_dp"][x] += 1
                    qpos += 1; rpos += 1
            elif op in (C_INS, C_DEL):
                i1 = max(atd, rpos - beg) - atd; i2 = min(rpos - beg + atd, n_rtr - 1)
                t1_beg, t1_len, t1_unit = int(rtr[0][i1]), int(rtr[1][i1]), int(rtr[2][i1])
                t2_beg, t2_len, t2_unit = int(rtr[0][i2]), int(rtr[1][i2]), int(rtr[2][i2])
                unitlen2 = max(1, t1_unit if t1_len > t2_len else t2_unit)
                inv = cdiv(100, (u32(ln) // unitlen2) if (u32(ln) % unitlen2 == 0) else 4)
                near = "a_near_ins" if op == C_INS else "a_near_del"
                if op == C_INS:
                    nb = i32(u32(u32(ln) * u32(int(P.indel_adj_indellen_perc))) // 100)
                    for r2 in range(max(rpos - nb, pos), min(rpos + nb, rend)):
                        x2 = r2 - beg
                        out["a_near_ins_dp"][x2] += 1; out["a_near_ins_pow2len"][x2] += u32(ln * ln)
                        out["a_near_ins_l_pow2len"][x2] += i32((r2 + 1 - (rpos - nb)) ** 2); out["a_near_ins_r_pow2len"][x2] += i32(((rpos + nb) - r2) ** 2)
                        out["a_near_ins_inv100len"][x2] += inv
                else:
                    for r2 in range(rpos, rpos + ln):      # the deleted positions count as covered
                        x2 = r2 - beg
                        out["a_pcr_dp"][x2] += pcr; out["a_umi_dp"][x2] += umi; out["a_dp"][x2] += 1; out["a_qlen"][x2] += rend - pos
                        out["a_highBQ_dp"][x2] += 1; out["a_XM1500"][x2] += xm1500; out["a_GO1500"][x2] += go1500; out["a_GAPLEN"][x2] += avg_gaplen
                        if isize != 0:
                            if isrc:
                                out["a_LI"][x2] += min(rpos - fpl + 1, MAX_INSERT_SIZE); out["a_LIDP"][x2] += 1
                            else:
                                out["a_RI"][x2] += min(fpr - rpos, MAX_INSERT_SIZE); out["a_RIDP"][x2] += 1
                        out["a_l_dist_sum"][x2] += rpos - pos + 1; out["a_r_dist_sum"][x2] += rend - rpos
                        out["a_inslen_sum"][x2] += inslen; out["a_dellen_sum"][x2] += dellen
                        # sic: the two BAQ sums go to the FIRST deleted position every time (getRefByPos(rpos), main.hpp:1150-1153)
                        out["a_l_BAQ_sum"][rpos - beg] += i32(B(rpos) - B(pos) + 1); out["a_r_BAQ_sum"][rpos - beg] += i32(B(rend - 1) - B(rpos) + 1)
                        out["a_insBAQ_sum"][x2] += insbaq; out["a_delBAQ_sum"][x2] += delbaq
                    perc = int(P.indel_adj_indellen_perc)
                    nb_l = i32(u32(u32(ln) * u32(perc - 100)) // 100); nb_r = i32(u32(u32(ln) * u32(perc)) // 100)
                    lo = max(rpos - nb_l, pos); hi = min(rpos + nb_r, rend) - 1
                    for r2 in range(lo, hi + 1):
                        x2 = r2 - beg
                        out["a_near_del_dp"][x2] += 1; out["a_near_del_pow2len"][x2] += u32(ln * ln)
                        out["a_near_del_l_pow2len"][x2] += i32((r2 - lo + 1) ** 2); out["a_near_del_r_pow2len"][x2] += i32((hi - r2 + 1) ** 2)
                        out["a_near_del_inv100len"][x2] += inv
                for r2 in range(max(beg + t1_beg - atd, pos), min(beg + t2_beg + t2_len + atd, rend)):
                    out[("a_near_RTR_ins_dp" if op == C_INS else "a_near_RTR_del_dp")][r2 - beg] += 1
                out[("a_at_ins_dp" if op == C_INS else "a_at_del_dp")][rpos - beg] += 1
                if op == C_INS:
                    qpos += ln
                else:
                    rpos += ln
            else:
                delta = 0 if ci == 0 else -1
                if op in (C_SOFT_CLIP, C_HARD_CLIP) and pcr:
                    ncd = int(P.microadjust_near_clip_dist)
                    for r2 in range(rpos + delta - ncd, rpos + delta + ncd + 1):
                        if beg <= r2 < excl_end:
                            out["a_near_pcr_clip_dp"][r2 - beg] += pcr
                if op in (C_SOFT_CLIP, C_HARD_CLIP) and pcr == 0 and ln >= int(P.microadjust_alignment_clip_min_len):
                    out["a_near_long_clip_dp"][rpos + delta - beg] += 1
                if op == C_REF_SKIP:
                    rpos += ln
                elif op == C_SOFT_CLIP:
                    qpos += ln
    return out


def thres_sets(prep, indelphred, P, is_normal, iontorrent):
    """P1b per position.  `prep` = the dict of prep_sets (values as the int32 / int64 fields hold them), `indelphred` = the repeat tracks'
    indelphred before the edit.  Returns ({name: int array [npos]}, edited indelphred)."""
    npos = len(indelphred)
    t = {k: np.zeros(npos, dtype=np.int64) for k in THRES}
    ip = np.array(indelphred, dtype=np.int64).copy()
    ratio = float(P.indel_del_to_ins_err_ratio)
    half = int(cdiv(int(cround((10.0 / math.log(10.0)) * math.log(ratio))), 2))   # (uvc1_qual_t)round(numstates2phred(ratio)) / 2
    for x in range(npos):
        p = {k: int(prep[k][x]) for k in prep}
        for k in PREP32:
            p[k] = i32(p[k])
        lidp = max(p["a_LIDP"], 1); ridp = max(p["a_RIDP"], 1)
        ins_dp1 = max(p["a_near_ins_dp"], 1); del_dp1 = max(p["a_near_del_dp"], 1)
        il = math.ceil(math.sqrt(cdiv(p["a_near_ins_l_pow2len"], ins_dp1))); dl = math.ceil(math.sqrt(cdiv(p["a_near_del_l_pow2len"], del_dp1)))
        ir = math.ceil(math.sqrt(cdiv(p["a_near_ins_r_pow2len"], ins_dp1))); dr = math.ceil(math.sqrt(cdiv(p["a_near_del_r_pow2len"], del_dp1)))
        dnv = 10 if (iontorrent and p["a_dnv_dp"] * 2 > p["a_snv_dp"]) else 0
        t["aLPxT"][x] = int(max(il, dl, dnv) + int(P.bias_thres_aLPxT_add))
        t["aRPxT"][x] = int(max(ir, dr, dnv) + int(P.bias_thres_aLPxT_add))
        if p["a_near_ins_dp"] * ratio < p["a_near_del_dp"]:
            ip[x] += half
        if p["a_near_del_dp"] * ratio < p["a_near_ins_dp"]:
            ip[x] -= half
        pc_inc1 = cdiv(3 * 100 * max(1, p["a_near_ins_dp"] + p["a_near_del_dp"]), max(1, p["a_near_ins_inv100len"] + p["a_near_del_inv100len"])) - 3
        ip[x] += min(max(0, pc_inc1), 6)                  # BETWEEN(pc_inc1, 0, 6)
        ip[x] = max(ip[x], 0)
        T1 = int(P.bias_thres_aLRI1NT_perc if is_normal else P.bias_thres_aLRI1T_perc)
        t1 = int(P.bias_thres_aLRI1Nt_perc if is_normal else P.bias_thres_aLRI1t_perc)
        for side, dp in (("L", lidp), ("R", ridp)):
            v = p["a_%sI" % side]
            t["a%sI1T" % side][x] = i32(cdiv(v * T1, dp * 100) + int(P.bias_thres_aLRI1T_add))
            t["a%sI2T" % side][x] = i32(cdiv(v * int(P.bias_thres_aLRI2T_perc), dp * 100) + int(P.bias_thres_aLRI2T_add))
            t["a%sI1t" % side][x] = i32(cdiv(v * t1, dp * 100))
            t["a%sI2t" % side][x] = i32(cdiv(v * int(P.bias_thres_aLRI2t_perc), dp * 100))
        p1 = int(P.bias_thres_aLRP1Nt_avgmul_perc if is_normal else P.bias_thres_aLRP1t_avgmul_perc); p2 = int(P.bias_thres_aLRP2t_avgmul_perc)
        b1 = int(P.bias_thres_aLRB1Nt_avgmul_perc if is_normal else P.bias_thres_aLRB1t_avgmul_perc); b2 = int(P.bias_thres_aLRB2t_avgmul_perc)
        hb = max(1, i32(p["a_highBQ_dp"] * 100))

        def nnm(a, b):                                     # non_neg_minus
            return a - b if a > b else 0
        t["aLP1t"][x] = i32(nnm(cdiv(p["a_l_dist_sum"] * p1, hb), int(P.bias_thres_aLRP1t_minus)))
        t["aLP2t"][x] = i32(nnm(cdiv(p["a_l_dist_sum"] * p2, hb), int(P.bias_thres_aLRP2t_minus)))
        t["aRP1t"][x] = i32(nnm(cdiv(p["a_r_dist_sum"] * p1, hb), int(P.bias_thres_aLRP1t_minus)))
        t["aRP2t"][x] = i32(nnm(cdiv(p["a_r_dist_sum"] * p2, hb), int(P.bias_thres_aLRP2t_minus)))
        pdel = cdiv(p["a_delBAQ_sum"], max(1, p["a_highBQ_dp"]))
        t["aLB1t"][x] = i32(nnm(cdiv(p["a_l_BAQ_sum"] * b1, hb), int(P.bias_thres_aLRB1t_minus) + pdel))
        t["aLB2t"][x] = i32(nnm(cdiv(p["a_l_BAQ_sum"] * b2, hb), int(P.bias_thres_aLRB2t_minus)))
        t["aRB1t"][x] = i32(nnm(cdiv(p["a_r_BAQ_sum"] * b1, hb), int(P.bias_thres_aLRB1t_minus) + pdel))
        t["aRB2t"][x] = i32(nnm(cdiv(p["a_r_BAQ_sum"] * b2, hb), int(P.bias_thres_aLRB2t_minus)))
    return t, ip
