"""Independent Python restatement, written from the reference text (not from oracle/ and not from the kernels), of the fragment pass of
SymbolCountCoverageSet::updateByAlns3UsingBQ (main.hpp:2620-2830): per fragment the BASE_QUALITY_MAX merge of its alignments
(updateByAln<., BASE_QUALITY_MAX, false>, GenericSymbol2Count::incSymbolCount, main.hpp:339-349), fillTidBegEndFromAlns1 (658-673),
fillConsensusCounts (374-413), get_avgBQ (main_conversion.hpp:791-796), PhredMutationTable (main.hpp:213-262), the quality buckets of
dedup_ampDistr, FRAG_bDP / VQ_bMQ, the covered / mutated / near-mutation positions behind FRAG_bTA / FRAG_bTB, and at the end
infer_max_qual_assuming_independence (main_conversion.hpp:943-978) -> VQ_bIAQb / bIADb / bIDQb.  tests/test_p3_cpu.py holds the oracle's
FRAG planes and VQ slots against it."""
import math

import numpy as np

from p2_restatement import read_events

NSYM, NUM_BUCKETS, SQR_QUAL_DIV = 14, 16, 32
BASE_A, BASE_C, BASE_G, BASE_T, BASE_N, BASE_NN, LINK_M, LINK_D3P, LINK_D2, LINK_D1, LINK_I3P, LINK_I2, LINK_I1, LINK_NN = range(14)
END = 14
DBL_EPSILON = 2.220446049250313e-16


def fill_consensus(c, lo, hi, ref_once):
    """_fillConsensusCounts<TIsRefCountedOnlyOnce> over the symbols lo..hi -> (argmax, max, sum)."""
    argmax, cmax, csum = hi, 0, 0
    for s in range(lo, hi + 1):
        if ref_once:
            if cmax < c[s] or (argmax == LINK_M and 0 < c[s]):
                argmax, cmax, csum = s, c[s], c[s]
        else:
            if cmax < c[s]:
                argmax, cmax = s, c[s]
            csum += c[s]
    return argmax, cmax, csum


def is_ins(s): return s in (LINK_I1, LINK_I2, LINK_I3P)
def is_del(s): return s in (LINK_D1, LINK_D2, LINK_D3P)


def sscs_phred(P, con, alt, is_rescued):
    """PhredMutationTable::toPhredErrRate(con_symbol, alt_symbol) as the fragment pass calls it: (refsymbol, con_symbol).  `is_rescued` is the
    constructor's `vcf_tumor_fname.size() > 0` (main.hpp:2564) -- true for the default "." too (CmdLineArgs.hpp:22, 55), so all_mutation_inc is 3
    unless the name is set to the empty string: UvcParams::tumor_vcf_fname_nonempty, not tumor_vcf_is_provided."""
    if is_ins(con) or is_del(con):
        raw = P.fam_phred_sscs_indel_open
    elif con == LINK_M:
        if alt in (LINK_D1, LINK_I1): raw = P.fam_phred_sscs_indel_open
        elif alt in (LINK_D2, LINK_I2): raw = P.fam_phred_sscs_indel_open + P.fam_phred_sscs_indel_ext
        else: raw = P.fam_phred_sscs_indel_open + P.fam_phred_sscs_indel_ext * 2
    elif (con, alt) in ((BASE_C, BASE_T), (BASE_G, BASE_A)): raw = P.fam_phred_sscs_transition_CG_TA
    elif (con, alt) in ((BASE_A, BASE_G), (BASE_T, BASE_C)): raw = P.fam_phred_sscs_transition_AT_GC
    elif (con, alt) in ((BASE_C, BASE_A), (BASE_G, BASE_T)): raw = P.fam_phred_sscs_transversion_CG_AT
    else: raw = P.fam_phred_sscs_transversion_other
    return int(raw) + (3 if is_rescued else 0)


def symbols_mutated(ref, alt):
    if alt <= BASE_NN:
        return ref != alt and ref < BASE_N and alt < BASE_N
    return alt != LINK_M and alt != LINK_NN


def infer_max_qual(max_qual, dec_qual, distr, totDP):
    maxv = argAD = argBQ = 0
    currAD = 0
    for idx in range(min(NUM_BUCKETS, max_qual // dec_qual) if max_qual >= 0 else 0):
        q = int(distr[idx])
        if q == 0:
            continue
        currAD += q
        currBQ = max_qual - dec_qual * idx
        expBQ = 10.0 / math.log(10.0) * math.log(float(totDP) / float(currAD) + DBL_EPSILON)
        v = int(currAD * (currBQ - expBQ))           # double -> int: toward zero
        if v > maxv:
            argAD, argBQ, maxv = currAD, currBQ, v
    return maxv, argAD, argBQ


def fragment_pass(reads, P, rtr, indelphred, baq, codes, prep, thres, seg, bqsum, proton, alleles=None):
    """-> (frag int64 [2][3][NSYM][npos] (bDP, bTA, bTB), vq {bMQ, bIAQb, bIADb, bIDQb: int64 [NSYM][npos]}).
    alleles (optional dict): gains "bq" = per strand {(symbol, position): {inserted text | deleted length: fragments}}, the allele-keyed maps of
    symbol_to_frag_format_depth_sets (posToIndelToCount_updateByConsensus, main.hpp:2710-2717)."""
    beg = int(reads["beg"])
    npos = int(reads["end"]) - beg + 1
    frag = np.zeros((2, 3, NSYM, npos), dtype=np.int64)
    vq = {k: np.zeros((NSYM, npos), dtype=np.int64) for k in ("bMQ", "bIAQb", "bIADb", "bIDQb")}
    bucket = np.zeros((npos, NSYM, NUM_BUCKETS), dtype=np.int64)
    codes_p = np.append(np.asarray(codes), [4, 4, 4, 4])
    n = int(reads["n_reads"])
    nb = int(P.syserr_mut_region_n_bases)
    highBQ = int(P.bias_thres_highBQ)

    def avgBQ(x, s):
        denom = int(seg["aDPff"][s][x] + seg["aDPfr"][s][x] + seg["aDPrf"][s][x] + seg["aDPrr"][s][x])
        return int(bqsum[s][x]) // max(1, denom)
    i = 0
    while i < n:
        j = i
        while j < n and (reads["fam_id"][j], reads["fam_strand"][j], reads["frag_id"][j]) == (reads["fam_id"][i], reads["fam_strand"][i], reads["frag_id"][i]):
            j += 1
        strand = int(reads["fam_strand"][i])
        # fillTidBegEndFromAlns1: the end grows by one per alignment
        beg2, end2, normMQ = 2 ** 31 - 1, 0, 0
        evs = []
        fmap = {}                                                # the fragment's own allele-keyed maps (incIns / incDel of its reads)
        for k in range(i, j):
            ev, aln, _, _, _, _, gaps = read_events(reads, k, P, rtr, indelphred, baq, codes, prep, thres, proton, with_bias=False)
            evs.append(ev)
            for gp, gs, gkey, gw in gaps:
                d = fmap.setdefault((gs, gp), {}); d[gkey] = d.get(gkey, 0) + gw
            beg2 = min(beg2, aln["pos"]); end2 = max(end2, aln["endpos"]) + 1
            normMQ = max(normMQ, aln["qual"])
        tlen = end2 - beg2
        cnt = np.zeros((tlen, NSYM), dtype=np.int64)
        for ev in evs:
            for _, v, p, s, _, _, _ in ev:
                cnt[p - beg2][s] = max(cnt[p - beg2][s], v)       # incSymbolCount<BASE_QUALITY_MAX>
        cov = [0] * tlen
        base_sym = [END] * tlen; link_sym = [END] * tlen
        for e in range(tlen):
            epos = beg2 + e; x = epos - beg
            for st in (1, 0):                                    # SYMBOL_TYPES_IN_VCF_ORDER: LINK, BASE
                refsymbol = int(codes_p[x])
                if st == 1: con, cc, ct = fill_consensus(cnt[e], LINK_M, LINK_NN, True)
                else: con, cc, ct = fill_consensus(cnt[e], BASE_A, BASE_NN, False)
                if ct == 0:
                    continue
                max_qual = 8 + avgBQ(x, con)
                con_qual = cc * 2 - ct
                if int(P.fam_flag) & 1:
                    phredlike = min(con_qual, max_qual, sscs_phred(P, refsymbol, con, bool(P.tumor_vcf_fname_nonempty)))
                else:
                    phredlike = min(con_qual, max_qual)
                pb = max(0, max_qual - phredlike)
                if pb < NUM_BUCKETS:
                    bucket[x][con][pb] += 1
                frag[strand][0][con][x] += 1
                vq["bMQ"][con][x] += (normMQ * normMQ) // SQR_QUAL_DIV
                if alleles is not None and LINK_M < con < LINK_NN:          # an InDel symbol: the fragment's majority allele gains one fragment
                    d = fmap[(con, epos)]
                    key = max(sorted(d), key=lambda kk: (d[kk], kk)) if len(d) > 1 else next(iter(d))   # indelToData_getMajority: ties to the larger key
                    dst = alleles.setdefault("bq", ({}, {}))[strand].setdefault((con, epos), {})
                    dst[key] = dst.get(key, 0) + 1
                cov[e] |= 1
                high = ((st == 0 or con_qual + 3 >= highBQ) if proton else (st == 1 or con_qual >= highBQ))
                if symbols_mutated(refsymbol, con) and high:
                    cov[e] |= 2
                if st == 1: link_sym[e] = con
                else: base_sym[e] = con
        for e in range(tlen):
            if cov[e] & 2:
                for q in range(e - nb, e + nb + 1):
                    if 0 <= q < tlen:
                        cov[q] |= 4
        n_cov = sum(1 for f in cov if f & 1)
        n_near = sum(1 for f in cov if (f & 1) and (f & 4))
        for e in range(tlen):
            for con in (base_sym[e], link_sym[e]):
                if con != END:
                    frag[strand][1][con][beg2 + e - beg] += n_cov
                    frag[strand][2][con][beg2 + e - beg] += n_near
        i = j
    for x in range(npos):
        for lo, hi in ((BASE_A, BASE_NN), (LINK_M, LINK_NN)):
            totDP = int(frag[0][0][lo:hi + 1, x].sum() + frag[1][0][lo:hi + 1, x].sum())
            for s in range(lo, hi + 1):
                mv, ad, bq = infer_max_qual(8 + avgBQ(x, s), 1, bucket[x][s], totDP)
                vq["bIAQb"][s][x] += mv; vq["bIADb"][s][x] += ad; vq["bIDQb"][s][x] += bq
    return frag, vq
