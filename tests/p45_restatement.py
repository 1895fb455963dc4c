"""Independent Python restatement, written from the reference text (not from oracle/ and not from the kernels), of the family passes of
SymbolCountCoverageSet::updateByAlns3UsingFQ (main.hpp:2836-3590; the consensus-FASTQ arms left out):

  P4   per (family, strand) unit: fragments -> updateByFiltering (main.hpp:466-494, 1659-1690) -> FAM_cDP12 / cDP21 / cDP2 / cDP3 / cDPM /
       cDPm, the FAM2 position / BAQ bias counters of FamFormatInfoSet (update_bidirectional_bias, main.hpp:1316-1358), the medians of read
       ends "as filled" (MEDIAN, main_conversion.hpp:24-28), the no-strict-bias window, read_family_con_ampl_getMajority_ins / _del with the
       allele-keyed maps (main.hpp:50-96, 188-212)
  P5   updateByMajorMinusMinor (main.hpp:496-521) -> FAM_cDP1 / cDPD, the empirical family quality and its buckets; the duplex pass
       (DUPLEX_dDP1 / dDP2); infer_max_qual_assuming_independence per strand -> VQ cIAQ / cIAD / cIDQ

tests/test_p45_cpu.py holds the oracle's FAM / FAMINFO32 / FAMINFO64 / DUPLEX planes and the six VQ slots against it."""
import math

import numpy as np

from p2_restatement import read_events
from p3_restatement import (BASE_A, BASE_N, BASE_NN, BASE_T, LINK_M, LINK_NN, NSYM, NUM_BUCKETS, fill_consensus, infer_max_qual, is_del, is_ins,
                            sscs_phred)
from segbias_restatement import non_neg_minus, update_bidirectional_bias

MAX_STR_N_BASES = 100
FAM = "cDP1 cDP12 cDP2 cDP3 cDPM cDPm cDP21 cDPD".split()
FI32 = "c2LP1 c2LP2 c2LPL c2RP1 c2RP2 c2RPL c2LP0 c2RP0 c2LB1 c2LB2 c2RB1 c2RB2 c2BQ2".split()
FI64 = "c2LBL c2RBL".split()


def majority(d):
    """indelToData_getMajority: the largest count, ties to the larger key (std::string / integer order); (0, T()) for an empty map."""
    keys = sorted(d)
    maxcnt, arg = 0, (type(keys[0])() if keys else None)
    for k in keys:
        if d[k] > maxcnt or (d[k] == maxcnt and k > arg):
            maxcnt, arg = d[k], k
    return maxcnt, arg


class Cov:
    """A Symbol2CountCoverage: counts [tlen][NSYM] and the allele-keyed maps per (symbol, position)."""
    def __init__(self, beg, end):
        self.beg, self.end = beg, end
        self.c = np.zeros((end - beg, NSYM), dtype=np.int64)
        self.maps = {}                         # (symbol, position) -> {allele: count}

    def inc_map(self, sym, pos, key, v):
        d = self.maps.setdefault((sym, pos), {})
        d[key] = d.get(key, 0) + v

    def update_map_by_consensus(self, src, sym, pos, v):
        """posToIndelToCount_updateByConsensus: the source's majority allele of (sym, pos) gains v here."""
        d = src.maps[(sym, pos)]
        key = majority(d)[1] if len(d) > 1 else next(iter(d))
        self.inc_map(sym, pos, key, v)


def fragment_cov(reads, idxs, P, rtr, ip, baq, codes, prep, thres, proton):
    evs, gps = [], []
    beg1, end1 = 2 ** 31 - 1, 0
    for k in idxs:
        ev, aln, _, _, _, _, gaps = read_events(reads, k, P, rtr, ip, baq, codes, prep, thres, proton, with_bias=False)
        evs.append(ev); gps.append(gaps)
        beg1 = min(beg1, aln["pos"]); end1 = max(end1, aln["endpos"]) + 1
    f = Cov(beg1, end1)
    for ev in evs:
        for _, v, p, s, _, _, _ in ev:
            f.c[p - beg1][s] = max(f.c[p - beg1][s], v)
    for gaps in gps:
        for p, s, key, w in gaps:
            f.inc_map(s, p, key, w)
    return f


def update_by_filtering(dst, src, thres2, padded_ignored, ref_once=True):
    """GenericSymbol2CountCoverage::updateByFiltering<true, false, ref_once>(src, {thres BASE, thres LINK}, is_padded_del_ignored)."""
    for epos in range(src.beg, src.end):
        row = src.c[epos - src.beg]
        con_link = None
        for st in (0, 1):                      # SYMBOL_TYPE_ARR: BASE, LINK
            if st == 1: con, cc, ct = fill_consensus(row, LINK_M, LINK_NN, ref_once)
            else: con, cc, ct = fill_consensus(row, BASE_A, BASE_T if padded_ignored else BASE_NN, False)
            adj = max(cc * 2, ct) - ct
            if adj >= thres2[st] and adj > 0:
                dst.c[epos - dst.beg][con] += 1
            if st == 1: con_link = con
        if is_ins(con_link) or is_del(con_link):
            dst.update_map_by_consensus(src, con_link, epos, 1)


def update_by_mmm(dst, src):
    for epos in range(src.beg, src.end):
        row = src.c[epos - src.beg]
        for st in (0, 1):
            if st == 1: con, cc, ct = fill_consensus(row, LINK_M, LINK_NN, True)
            else: con, cc, ct = fill_consensus(row, BASE_A, BASE_NN, False)
            adj = max(cc * 2, ct) - ct
            if adj > 0:
                dst.c[epos - dst.beg][con] += adj


def families(reads):
    """alns3 as index lists: [(fam, {strand: [[read indices of one fragment], ...]})] in read order."""
    out, n, i = [], int(reads["n_reads"]), 0
    while i < n:
        fam = int(reads["fam_id"][i]); units = {0: [], 1: []}
        while i < n and int(reads["fam_id"][i]) == fam:
            st, fr = int(reads["fam_strand"][i]), int(reads["frag_id"][i]); j = i
            while j < n and (int(reads["fam_id"][j]), int(reads["fam_strand"][j]), int(reads["frag_id"][j])) == (fam, st, fr):
                j += 1
            units[st].append(list(range(i, j))); i = j
        out.append((fam, units))
    return out


def family_passes(reads, P, rtr, ip, baq, baq2, codes, prep, thres, proton, alleles=None):
    """-> (fam int64 [2][8][NSYM][npos], fi32 {name: [NSYM][npos]}, fi64 {...}, duplex [2][NSYM][npos], vq {cIAQf cIADf cIDQf cIAQr cIADr cIDQr})."""
    beg = int(reads["beg"]); npos = int(reads["end"]) - beg + 1
    famp = np.zeros((2, len(FAM), NSYM, npos), dtype=np.int64)
    fi = {k: np.zeros((NSYM, npos), dtype=np.int64) for k in FI32 + FI64}
    dup = np.zeros((2, NSYM, npos), dtype=np.int64)
    vq = {k: np.zeros((NSYM, npos), dtype=np.int64) for k in "cIAQf cIADf cIDQf cIAQr cIADr cIDQr".split()}
    bucket = np.zeros((2, npos, NSYM, NUM_BUCKETS), dtype=np.int64)
    codes_p = np.append(np.asarray(codes), [4, 4, 4, 4])
    padded_ignored = bool(int(P.microadjust_padded_deletion_flag) & (0x2 if proton else 0x1))
    is_rescued = bool(P.tumor_vcf_fname_nonempty)      # PhredMutationTable's flag: vcf_tumor_fname.size() > 0, true for the default "."
    provided = bool(P.tumor_vcf_is_provided)           # IS_PROVIDED(vcf_tumor_fname)
    excl_end = beg + npos
    F = {k: i for i, k in enumerate(FAM)}
    fams = families(reads)
    # the region's allele-keyed maps per strand (optional): "fq" symbol_to_fam_format_depth_sets_2strand (main.hpp:3326-3336), "c2" pos2*2data_cDP2
    # (main.hpp:3196-3206), "c2d" pos2*2data_c2dDP (main.hpp:3459-3469, 3535-3547)
    amaps = None
    if alleles is not None:
        amaps = {k: (Cov(0, 0), Cov(0, 0)) for k in ("fq", "c2", "c2d")}
        for k, v in amaps.items():
            alleles[k] = (v[0].maps, v[1].maps)

    def B(p): return int(baq[p - beg])
    def B2(p): return int(baq2[p - beg])

    def unit_span(frs):
        b, e = 2 ** 31 - 1, 0
        for fr in frs:
            for k in fr:
                pos = int(reads["pos"][k])
                ev, aln, *_ = read_events(reads, k, P, rtr, ip, baq, codes, prep, thres, proton, with_bias=False)
                b = min(b, pos); e = max(e, aln["endpos"]) + 1
        return b, e
    # ---- P4
    for fam, units in fams:
        dflag = int(reads["fam_dflag"][fam])
        for strand in (0, 1):
            frs = units[strand]
            if not frs:
                continue
            beg2, end2 = unit_span(frs)
            con = Cov(beg2, end2)
            for fr in frs:
                update_by_filtering(con, fragment_cov(reads, fr, P, rtr, ip, baq, codes, prep, thres, proton), (int(P.fam_thres_highBQ_snv), 0), padded_ignored, True)
            l2r_end, r2l_end, qsum, nq = [], [], 0, 0
            for fr in frs:
                for k in fr:
                    ev, aln, *_ = read_events(reads, k, P, rtr, ip, baq, codes, prep, thres, proton, with_bias=False)
                    if aln["flag"] & 0x10: r2l_end.append(aln["pos"])
                    else: l2r_end.append(aln["endpos"])
                    qsum += int(reads["l_qseq"][k]); nq += 1

            def median(v): return (v[(len(v) - 1) // 2] + v[len(v) // 2]) // 2       # as filled, not sorted
            l2r_med = median(l2r_end) if l2r_end else end2
            r2l_med = median(r2l_end) if r2l_end else beg2
            nonconf_middle = l2r_med <= r2l_med + int(P.indel_adj_tracklen_dist)
            nsb_min, nsb_max = end2, beg2
            umi_ok = bool(dflag & 0x1) or bool(int(P.fam_flag) & 0x2)
            if len(frs) >= int(P.fam_thres_dup1add) and qsum >= nq * int(P.fam_thres_qseqlen):
                poss = [end2, beg2]
                for i in (0, 1):
                    rng = range(end2 - 1, beg2 - 1, -1) if i else range(beg2, end2)
                    for epos in rng:
                        cs, cc, ct = fill_consensus(con.c[epos - beg2], BASE_A, BASE_NN, False)
                        if ct == 0:
                            continue
                        good = int(P.fam_thres_dup1add) <= ct and cc * 100 >= ct * int(P.fam_thres_dup1perc) and umi_ok
                        if good and cs != BASE_N and cs != BASE_NN:
                            poss[i] = epos
                            break
                nsb_min, nsb_max = poss
            for epos in range(beg2, end2):
                x = epos - beg
                row = con.c[epos - beg2]
                for st in (1, 0):                                        # SYMBOL_TYPES_IN_VCF_ORDER
                    cs, cc, ct = fill_consensus(row, LINK_M, LINK_NN, False) if st == 1 else fill_consensus(row, BASE_A, BASE_NN, False)
                    good = int(P.fam_thres_dup1add) <= ct and cc * 100 >= ct * int(P.fam_thres_dup1perc) and umi_ok
                    if ct == 0:
                        continue
                    famp[strand][F["cDP12"]][cs][x] += 1
                    if ct == 1:
                        famp[strand][F["cDP21"]][cs][x] += 1
                    if not P.inferred_is_vcf_generated:
                        continue
                    if good:
                        famp[strand][F["cDP2"]][cs][x] += 1
                        if amaps is not None and (is_ins(cs) or is_del(cs)):
                            amaps["c2"][strand].update_map_by_consensus(con, cs, epos, 1)
                        rbeg, rend = min(nsb_min, epos), max(nsb_max, epos)
                        if nonconf_middle and epos < r2l_med:
                            rend = max(min(l2r_med, r2l_med, rend), epos)
                        if nonconf_middle and l2r_med < epos:
                            rbeg = min(max(l2r_med, r2l_med, rbeg), epos)
                        is_gap = (st == 1)
                        if ((not is_gap) and 90 >= int(P.bias_thres_highBQ)) or (is_gap and 1024 * 1024 >= int(P.bias_thres_highBQ)):
                            l_nb = non_neg_minus(epos + 1, rbeg); r_nb = non_neg_minus(rend, epos)
                            _LPxT, RPxT = int(thres["aLPxT"][x]), int(thres["aRPxT"][x])
                            LPxT = _LPxT if is_gap else min(_LPxT, RPxT)
                            indel_len = 0
                            if is_ins(cs):
                                d = {"": 0}
                                for s2 in (12, 11, 10):                      # INS_SYMBOLS: I1, I2, I3P (std::map::insert keeps the first value of a key)
                                    for k2, v2 in con.maps.get((s2, epos), {}).items():
                                        d.setdefault(k2, v2)
                                indel_len = majority(d)[0]
                            elif is_del(cs):
                                d = {0: 0}
                                for s2 in (9, 8, 7):                         # DEL_SYMBOLS: D1, D2, D3P
                                    for k2, v2 in con.maps.get((s2, epos), {}).items():
                                        d.setdefault(k2, v2)
                                indel_len = majority(d)[0]
                            far = (l_nb + (non_neg_minus(indel_len, int(P.microadjust_nobias_pos_indel_maxlen)) if is_ins(cs) else 0) >= LPxT) and r_nb >= RPxT
                            info = {k: 0 for k in FI32 + FI64}
                            if far:
                                update_bidirectional_bias(info, "c2LP1", "c2LP2", "c2RP1", "c2RP2", "c2LPL", "c2RPL", int(thres["aLP1t"][x]), int(thres["aLP2t"][x]),
                                                          int(thres["aRP1t"][x]), int(thres["aRP2t"][x]), l_nb, r_nb, True, 0)
                            if non_neg_minus(epos + 1, nsb_min) >= int(P.bias_thres_strict_c2LRP0): info["c2LP0"] += 1
                            if non_neg_minus(nsb_max, epos) >= int(P.bias_thres_strict_c2LRP0): info["c2RP0"] += 1
                            l_baq = B(epos) - B(max(rbeg, non_neg_minus(epos, MAX_STR_N_BASES))) + 1
                            pe = min(rend - 1, epos + MAX_STR_N_BASES, excl_end - 1)
                            _r_baq = B(pe) - B(epos) + 1
                            r_baq = min(_r_baq, B2(pe) - B2(epos) + 7) if is_gap else _r_baq
                            hb = int(P.bias_thres_highBAQ) + (0 if is_gap else 3)
                            if l_baq >= hb and r_baq >= hb:
                                update_bidirectional_bias(info, "c2LB1", "c2LB2", "c2RB1", "c2RB2", "c2LBL", "c2RBL", int(P.bias_thres_BAQ1), int(P.bias_thres_BAQ2),
                                                          int(P.bias_thres_BAQ1), int(P.bias_thres_BAQ2), l_baq, r_baq, True, 0)
                            info["c2BQ2"] += 1
                            for k2, v2 in info.items():
                                fi[k2][cs][x] += v2
                    if int(P.fam_thres_dup2add) <= ct and cc * 100 >= ct * int(P.fam_thres_dup2perc):
                        famp[strand][F["cDP3"]][cs][x] += 1
                    if amaps is not None and (is_ins(cs) or is_del(cs)):
                        amaps["fq"][strand].update_map_by_consensus(con, cs, epos, 1)
                    subst = cs <= BASE_NN
                    flat = int(P.fam_thres_emperr_all_flat_snv if subst else P.fam_thres_emperr_all_flat_indel)
                    perc = int(P.fam_thres_emperr_con_perc_snv if subst else P.fam_thres_emperr_con_perc_indel)
                    if ct < flat or cc * 100 < ct * perc:
                        continue
                    for s2 in (range(BASE_A, BASE_NN + 1) if st == 0 else range(LINK_M, LINK_NN + 1)):
                        if s2 != cs:
                            famp[strand][F["cDPm"]][cs][x] += int(row[s2]); famp[strand][F["cDPM"]][cs][x] += ct
    # ---- P5
    if P.inferred_is_vcf_generated:
        for fam, units in fams:
            dflag = int(reads["fam_dflag"][fam])
            both = bool(units[0]) and bool(units[1])
            dscs = bool(dflag & 0x2) and both
            sscs = bool(dflag & 0x2) and not both
            b_all, e_all = 2 ** 31 - 1, 0
            for strand in (0, 1):                         # fillTidBegEndFromAlns2 over both strands, the end growing by one per alignment
                for fr in units[strand]:
                    for k in fr:
                        ev, aln, *_ = read_events(reads, k, P, rtr, ip, baq, codes, prep, thres, proton, with_bias=False)
                        b_all = min(b_all, aln["pos"]); e_all = max(e_all, aln["endpos"]) + 1
            duplex = Cov(b_all, e_all)
            for strand in (0, 1):
                frs = units[strand]
                if not frs:
                    continue
                beg2, end2 = unit_span(frs)
                con, mmm = Cov(beg2, end2), Cov(beg2, end2)
                for fr in frs:
                    fc = fragment_cov(reads, fr, P, rtr, ip, baq, codes, prep, thres, proton)
                    update_by_filtering(con, fc, (int(P.fam_thres_highBQ_snv), 0), padded_ignored, True)
                    update_by_mmm(mmm, fc)
                if dscs:
                    update_by_filtering(duplex, con, (1, 1), padded_ignored, False)
                for epos in range(beg2, end2):
                    x = epos - beg
                    for st in (1, 0):
                        mrow, crow = mmm.c[epos - beg2], con.c[epos - beg2]
                        cs, con_sum, tot_sum = fill_consensus(mrow, LINK_M, LINK_NN, False) if st == 1 else fill_consensus(mrow, BASE_A, BASE_NN, False)
                        if tot_sum == 0:
                            continue
                        con_nfrags = int(crow[cs]); tot_nfrags = int(crow[LINK_M:LINK_NN + 1].sum() if st == 1 else crow[BASE_A:BASE_NN + 1].sum())
                        famp[strand][F["cDP1"]][cs][x] += 1
                        if sscs and (not dscs) and tot_nfrags >= int(P.fam_thres_dup1add) and con_nfrags * 100 >= tot_nfrags * int(P.fam_thres_dup1perc):
                            famp[strand][F["cDPD"]][cs][x] += 1
                            if amaps is not None and (is_ins(cs) or is_del(cs)):
                                amaps["c2d"][strand].update_map_by_consensus(con, cs, epos, 1)
                        avgBQ = 1 if tot_nfrags == 0 else int(con_sum) // tot_nfrags
                        major = int(famp[strand][F["cDPM"]][cs][x]); minor = int(famp[strand][F["cDPm"]][cs][x])
                        pw = 1.0 / (minor + 1.0)
                        p_avg = math.pow(10, float(np.float32(-np.float32(avgBQ)) / np.float32(10)))      # phred2prob: pow(10, -((float)phred) / 10)
                        realphred = -10 * math.log((minor + pw) / (major + minor + pw / p_avg)) / math.log(10)
                        indep = int(round_half_away((con_nfrags * 2 - tot_nfrags) * realphred))
                        if st == 1:
                            confam = max(1, min(indep, int(P.fam_phred_indel_inc_before_barcode_labeling) + int(round_half_away(realphred))))
                        else:
                            confam = max(1, min(indep, int(con_sum) * 2 - int(tot_sum)))
                        ref_symbol = int(codes_p[x])
                        max_qual = sscs_phred(P, ref_symbol, cs, is_rescued) + (4 if provided else 0)
                        confam2 = min(confam, max_qual)
                        if tot_nfrags >= int(P.fam_thres_dup1add):
                            pb = cdiv_(max_qual - confam2 + 2, 4)
                            bucket[strand][x][cs][pb] += 1
            if dscs:
                for epos in range(duplex.beg, duplex.end):
                    x = epos - beg
                    for st in (0, 1):
                        row = duplex.c[epos - duplex.beg]
                        cs, cc, ct = fill_consensus(row, LINK_M, LINK_NN, False) if st == 1 else fill_consensus(row, BASE_A, BASE_NN, False)
                        if ct > 0: dup[0][cs][x] += 1
                        if ct > 1:
                            dup[1][cs][x] += 1
                            if amaps is not None and (is_ins(cs) or is_del(cs)):
                                for s2 in (0, 1):
                                    amaps["c2d"][s2].update_map_by_consensus(duplex, cs, epos, 1)
        for strand in (0, 1):
            sfx = "r" if strand else "f"
            for x in range(npos):
                ref_symbol = int(codes_p[x])
                for lo, hi in ((BASE_A, BASE_NN), (LINK_M, LINK_NN)):
                    totDP = int(famp[strand][F["cDP1"]][lo:hi + 1, x].sum())
                    for s in range(lo, hi + 1):
                        mq = sscs_phred(P, ref_symbol, s, is_rescued) + (4 if provided else 0)
                        mv, ad, bq = infer_max_qual(mq, 4, bucket[strand][x][s], totDP)
                        vq["cIAQ" + sfx][s][x] += mv; vq["cIAD" + sfx][s][x] += ad; vq["cIDQ" + sfx][s][x] += bq
    return famp, fi, dup, vq


def round_half_away(v):
    """C round(): halves away from zero."""
    return math.floor(v + 0.5) if v >= 0 else math.ceil(v - 0.5)


def cdiv_(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q
