"""An independent Python restatement of BcfFormat_symbol_calc_DPv (+ dp4_to_pcFA, fmt_bias_push, calc_normFA_from_rawFA_refbias,
BcfFormat_symbol_sum_DPv), written from the reference text (/root/reference/main.hpp:4253-4906, main_conversion.hpp:191-219, 798-849),
not from oracle/ or the kernels.  Inputs: what BcfFormat_symboltype_init / BcfFormat_symbol_init gathered for a record, as the oracle's
trace hook hands it over (uvc_oracle_score_trace).  tests/test_score_cpu.py compares the outputs with the oracle's records.
Test infrastructure."""
import math

DBL_EPSILON = 2.220446049250313e-16
FLT_EPSILON = 1.1920928955078125e-07
BASE_NN, LINK_M, LINK_D3P, LINK_D2, LINK_D1, LINK_I3P, LINK_I2, LINK_I1, LINK_NN = 5, 6, 7, 8, 9, 10, 11, 12, 13
SEQUENCING_PLATFORM_IONTORRENT = 2


def is_subst(s): return s <= BASE_NN            # isSymbolSubstitution, main_conversion.hpp:464-466
def is_ins(s): return s in (LINK_I3P, LINK_I2, LINK_I1)
def is_del(s): return s in (LINK_D3P, LINK_D2, LINK_D1)


def cdiv(a, b):
    """C++ integer division (truncation toward zero)."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def cround(x):
    """C round(): half away from zero."""
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


def trunc(x):
    return int(x)    # (uvc1_readnum100x_t)(double): toward zero


def non_neg_minus(a, b): return a - b if a > b else 0
def between(v, a, b): return min(max(a, v), b)
def prob2odds(p): return p / (1.0 - p)
def phred2nat(x): return (math.log(10.0) / 10.0) * x                      # common.hpp:81
def numstates2phred(x): return (10.0 / math.log(10.0)) * math.log(x)       # common.hpp:85
def numstates2deciphred(x): return cround((100.0 / math.log(10.0)) * math.log(x))   # common.hpp:87


def dp4_to_pcFA(bidir, overseq_disabled, overseq_frac, aADpass, aADfail, aDPpass, aDPfail, pl_exponent=3.0, n_nats=math.log(501.0),
                aADavgKeyVal=-1.0, aDPavgKeyVal=-1.0, priorAD=0.5, priorDP=1.0):
    """main_conversion.hpp:798-849"""
    aADpass, aADfail, aDPpass, aDPfail = float(aADpass), float(aADfail), float(aDPpass), float(aDPfail)
    if not overseq_disabled:
        aDPfail *= overseq_frac; aDPpass *= overseq_frac; aADfail *= overseq_frac; aADpass *= overseq_frac
    aDPfail += priorDP; aDPpass += priorDP; aADfail += priorAD; aADpass += priorAD
    nobiasFA = (aADfail + aADpass) / (aDPfail + aDPpass)
    if (aADpass / aDPpass) >= (aADfail / aDPfail):
        if bidir:
            aDPfail, aDPpass = aDPpass, aDPfail
            aADfail, aADpass = aADpass, aADfail
        else:
            return (aADpass / aDPpass, nobiasFA)
    aBDfail = aDPfail * 2 - aADfail * 1
    aBDpass = aDPpass * 2 - aADpass * 1
    aADpassfrac = aADpass / (aADpass + aADfail)
    aBDpassfrac = aBDpass / (aBDpass + aBDfail)
    if (not bidir) and aADavgKeyVal >= 0 and aDPavgKeyVal >= 0:
        aADpassfrac = aADavgKeyVal / (aADavgKeyVal + aDPavgKeyVal * 0.9)
        aBDpassfrac = 1.0 - aADpassfrac
    infogain = aADfail * math.log((1.0 - aADpassfrac) / (1.0 - aBDpassfrac))
    if bidir:
        infogain += aADpass * math.log(aADpassfrac / aBDpassfrac)
    if infogain <= n_nats:
        return (aADfail / aDPfail, nobiasFA)
    return (max(aADpass / aDPpass, (aADfail / aDPfail) * math.exp((n_nats - infogain) / pl_exponent)), nobiasFA)


def calc_normFA_from_rawFA_refbias(FA, refbias):   # main.hpp:4253-4256
    return (FA + FA * refbias) / (FA + (1.0 - FA) / (1.0 + refbias) + FA * refbias)


def does_fmt_imply_short_frag(f, wgs_min_avg_fragsize):   # main.hpp:170-174
    return (f["APLRI[0]"] + f["APLRI[2]"]) < (f["APLRI[1]"] + f["APLRI[3]"]) * wgs_min_avg_fragsize


class _F:
    """Field access in the reference's spelling: f.X(name, i) for the symbol-type vectors (fmt.NAME[i]), f.a(name) for the allele value
    (fmt.name[a] / LAST(fmt.name))."""
    def __init__(self, d): self.d = d
    def X(self, n, i=0): return int(self.d["%s[%d]" % (n, i)])
    def a(self, n): return int(self.d[n])
    def sumpair(self, n): return self.X(n, 0) + self.X(n, 1)


def calc_DPv(d, P):
    """BcfFormat_symbol_calc_DPv, main.hpp:4274-4844.  d: the traced inputs of one record (dict name -> number); P: UvcParams.
    Returns a dict of what the function writes: nPF, bNMa, bNMb, bNMQ, nNFA, nAFA, nBCFA, FTS (bit mask in push order), FTSpct, tier2,
    AD, bAD, cDP1v/w/x, cDP2v/w/x."""
    f = _F(d)
    out = {}
    tprov = bool(P.tumor_vcf_is_provided)          # IS_PROVIDED(paramset.vcf_tumor_fname)
    rtr1_tracklen, rtr1_unitlen = int(d["rtr1_tracklen"]), int(d["rtr1_unitlen"])
    rtr2_tracklen, rtr2_unitlen, rtr2_anyTR_tracklen = int(d["rtr2_tracklen"]), int(d["rtr2_unitlen"]), int(d["rtr2_anyTR_tracklen"])
    tpfa = float(d["tpfa_dpv"])
    refsymbol = int(d["refsymbol"])
    # seg_format_prep_sets of refpos: the oracle's APDP[] holds them in this order (fill_symboltype_fmt, main.hpp:3745-3793)
    a_dp, a_pcr_dp, a_snv_dp, a_dnv_dp, a_near_pcr_clip_dp, a_umi_dp = f.X("APDP", 0), f.X("APDP", 5), f.X("APDP", 6), f.X("APDP", 7), f.X("APDP", 9), f.X("APDP", 11)

    unbias_ratio = 1.0 if not tprov else math.sqrt(2.0)
    unbias_qualadd = 0 if not tprov else 3
    allbias_allprior = 0 if not tprov else 31
    is_strong_amplicon = a_pcr_dp * 100 > a_dp * 50
    is_weak_amplicon = a_pcr_dp * 100 > a_dp * 30
    is_rescued = tpfa >= 0
    pfa = tpfa if is_rescued else 0.5
    c2altpc = 0.025

    ADP1 = f.X("ADPff") + f.X("ADPfr") + f.X("ADPrf") + f.X("ADPrr")
    aDP1 = f.a("aDPff") + f.a("aDPfr") + f.a("aDPrf") + f.a("aDPrr")
    aDP = aDP1
    ADP = max(ADP1, a_near_pcr_clip_dp)
    cDP1 = f.a("cDP1f") + f.a("cDP1r")
    CDP1 = f.sumpair("CDP1b")
    cFA2 = (f.a("cDP2f") + f.a("cDP2r") + c2altpc) / (f.sumpair("CDP2b") + 1.0)
    cFA3 = (f.a("cDP3f") + f.a("cDP3r") + c2altpc) / (f.sumpair("CDP3b") + 1.0)
    symbol = f.a("symbol")
    gapSa_size = f.a("gapSa_len")                  # LAST(fmt.gapSa).size()

    _counterbias_P_FA = 1e-9
    _counterbias_BQ_FA = 1e-9
    _dir_bias_div = 1.0
    is_nmore_amplicon = is_strong_amplicon if not tprov else is_weak_amplicon
    if (is_nmore_amplicon and (0x2 == (0x2 & P.nobias_flag))) or ((not is_nmore_amplicon) and (0x1 == (0x1 & P.nobias_flag))):
        using_bias_oddsA = prob2odds((aDP - f.a("aP1") + 0.5) / (ADP - f.X("AP1") + 1.0))
        using_nobias_oddsA = prob2odds((f.a("aP1") + 0.5) / (f.X("AP1") + 1.0))
        is_pos_counterbias = ((using_bias_oddsA * P.microadjust_counterbias_pos_odds_ratio < using_nobias_oddsA * (unbias_ratio - DBL_EPSILON))
                              and (f.a("aP1") * (unbias_ratio - DBL_EPSILON) > aDP - f.a("aP1"))
                              and ((ADP - f.X("AP1")) * P.microadjust_counterbias_pos_fold_ratio * (unbias_ratio - DBL_EPSILON) > f.X("AP1"))
                              and ((0 == P.primerlen and 0 != P.primerlen2) or not is_subst(symbol)))
        if is_pos_counterbias:
            _counterbias_P_FA = max(_counterbias_P_FA, (f.a("aP1") + 0.5) / (max(f.X("AP1"), a_near_pcr_clip_dp) + 1.0))
        else:
            _counterbias_P_FA = max(_counterbias_P_FA, 2e-9)
        if is_subst(symbol):
            is_f_good_cov = (f.X("ADPfr") + f.X("ADPrr")) + 150 <= (f.X("ADPff") + f.X("ADPrf")) * 5 * unbias_ratio
            is_r_good_cov = (f.X("ADPff") + f.X("ADPrf")) + 150 <= (f.X("ADPfr") + f.X("ADPrr")) * 5 * unbias_ratio
            avg_f_aBQ = cdiv(f.a("a1BQf"), max(1, f.a("aDPff") + f.a("aDPrf")))
            avg_r_aBQ = cdiv(f.a("a1BQr"), max(1, f.a("aDPfr") + f.a("aDPrr")))
            avg_f_ABQ = cdiv(f.X("A1BQf"), max(1, f.X("ADPff") + f.X("ADPrf")))
            avg_r_ABQ = cdiv(f.X("A1BQr"), max(1, f.X("ADPfr") + f.X("ADPrr")))
            is_f_BQ_counterbias = ((f.a("a1BQf") >= f.a("a1BQr")) and (is_f_good_cov and is_r_good_cov)
                                   and (avg_f_aBQ + unbias_qualadd >= avg_r_ABQ + 14) and (avg_r_ABQ <= 14 + unbias_qualadd))
            if is_f_BQ_counterbias:
                _counterbias_BQ_FA = max(_counterbias_BQ_FA, (f.a("aDPff") + f.a("aDPrf") + 0.5) / (f.X("ADPff") + f.X("ADPrf") + 1.0))
            is_r_BQ_counterbias = ((f.a("a1BQr") >= f.a("a1BQf")) and (is_f_good_cov and is_r_good_cov)
                                   and (avg_r_aBQ + unbias_qualadd >= avg_f_ABQ + 14) and (avg_f_ABQ <= 14 + unbias_qualadd))
            if is_r_BQ_counterbias:
                _counterbias_BQ_FA = max(_counterbias_BQ_FA, (f.a("aDPfr") + f.a("aDPrr") + 0.5) / (f.X("ADPfr") + f.X("ADPrr") + 1.0))
        else:
            _dir_bias_div = 1.0 + cdiv(gapSa_size, P.indel_str_repeatsize_max)
    counterbias_P_FA, counterbias_BQ_FA, dir_bias_div = _counterbias_P_FA, _counterbias_BQ_FA, _dir_bias_div

    aDPgap = non_neg_minus(max(f.X("APDP", 1), f.X("APDP", 2)), f.a("aP3"))
    aDPFAgap = 1.0 if (rtr1_tracklen + rtr2_tracklen < P.indel_str_repeatsize_max) else ((f.a("aP3") + pfa) / (aDPgap + 1.0))
    aDPFA1 = (aDP + pfa) / (ADP + 1.0)
    labelFA = (f.a("aP2") + 1.5 + f.a("aP2")) / (f.X("AP2") + 2.0 + f.a("aP2"))
    aDPFA = min((min(aDPFA1, max(aDPFA1 / 3, aDPFAgap)) if is_subst(symbol) else aDPFA1),
                labelFA * (ADP + 1.0) / (f.X("AP2") + 0.5) * unbias_ratio)
    aDPplus = 0 if is_subst(symbol) else cdiv((aDP + 1) * P.bias_prior_DPadd_perc, 100)
    dp_coef = (max(P.contam_any_mul_frac, 1.0 - max(rtr1_tracklen, rtr2_tracklen) / (max(1, f.X("ALPL"), f.X("ARPL")) / max(1.0 / 150.0, f.X("ABQ2"))))
               if symbol == LINK_M else 1.0)
    _aPpriorfreq = float(P.bias_priorfreq_pos)
    _aBpriorfreq = _aPpriorfreq
    is_in_indel_read = (f.X("APXM", 1)) / 15.0 * P.microadjust_bias_pos_indel_fold * P.bias_prior_var_DP_mul > (aDP + aDPplus) * dp_coef
    is_in_indel_len = max(f.X("APDP", 1), f.X("APDP", 2)) * P.bias_prior_var_DP_mul > (aDP + aDPplus) * dp_coef
    is_in_indel_rtr = max(f.X("APDP", 3), f.X("APDP", 4)) * P.bias_prior_var_DP_mul > (aDP + aDPplus) * dp_coef
    is_in_rtr = max(rtr1_tracklen, rtr2_tracklen) > cround(P.indel_polymerase_size)
    is_in_dnv_read = (SEQUENCING_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) and (a_dnv_dp * 2 > a_snv_dp)
    is_outlier_del = False
    if is_in_indel_read or is_in_dnv_read or ((is_ins(symbol) or is_del(symbol)) and (f.X("APXM", 0) > f.X("APXM", 1) * P.microadjust_bias_pos_indel_misma_to_indel_ratio)):
        _aPpriorfreq -= P.bias_priorfreq_indel_in_read_div
        _aBpriorfreq -= P.bias_priorfreq_indel_in_read_div
    if LINK_M != symbol and LINK_NN != symbol:
        maxpf = 0
        if is_in_indel_len: maxpf = max(maxpf, P.bias_priorfreq_indel_in_var_div2)
        if is_in_indel_rtr: maxpf = max(maxpf, P.bias_priorfreq_indel_in_str_div2)
        if is_in_rtr: maxpf = max(maxpf, P.bias_priorfreq_var_in_str_div2)
        if is_outlier_del: maxpf = max(maxpf, 10)
        _aBpriorfreq -= maxpf
        _aPpriorfreq -= maxpf
    aPpriorfreq = _aPpriorfreq + allbias_allprior
    aBpriorfreq = _aBpriorfreq + allbias_allprior
    out["nPF"] = [cround(aPpriorfreq), cround(aBpriorfreq)]

    aIpriorfreq = (P.bias_priorfreq_ipos_snv if is_subst(symbol) else P.bias_priorfreq_ipos_indel) + allbias_allprior
    homopol_len = (rtr1_tracklen if 1 == rtr1_unitlen else 0) + (rtr2_tracklen if 1 == rtr2_unitlen else 0)
    if is_subst(symbol):
        dec = (min(5 * homopol_len, 20) if ((SEQUENCING_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) and (homopol_len > 0)
                                             and (is_subst(symbol) or LINK_D1 == symbol or LINK_I1 == symbol)) else 0)
        aSBpriorfreq = min(non_neg_minus(f.a("aBQ"), dec), f.a("bMQ")) + P.bias_priorfreq_strand_snv_base
    else:
        aSBpriorfreq = P.bias_priorfreq_strand_indel
    aSBpriorfreq += allbias_allprior

    dedup_A2C1_frac = min(1.0, float(max(CDP1, P.bias_reduction_by_high_sequencingDP_min_n_totDepth)) / float(max(ADP1, 1)))
    dedup_a2c1_frac = min(1.0, float(max(cDP1, P.bias_reduction_by_high_sequencingDP_min_n_altDepth)) / float(max(aDP1, 1)))
    dedup_frag_frac = max(dedup_A2C1_frac, dedup_a2c1_frac)
    pc_read = P.bias_FA_pseudocount_indel_in_read if is_in_indel_read else 0.5
    pl = P.powlaw_exponent

    def seg_pc(one, two, ALL2, length, ALLlength, prior):
        return dp4_to_pcFA(False, False, dedup_frag_frac, f.a(one), aDP, f.X(ALL2) + f.a(one) - f.a(two), ADP, pl, phred2nat(prior),
                           max(1, f.a(length)) / float(max(1, f.a("aBQ2"))), max(1, f.X(ALLlength)) / float(max(1, f.X("ABQ2"))), pc_read)
    aLPFA = seg_pc("aLP1", "aLP2", "ALP2", "aLPL", "ALPL", aPpriorfreq)[0]
    aRPFA = seg_pc("aRP1", "aRP2", "ARP2", "aRPL", "ARPL", aPpriorfreq)[0]
    aLBFA = seg_pc("aLB1", "aLB2", "ALB2", "aLBL", "ALBL", aBpriorfreq)[0]
    aRBFA = seg_pc("aRB1", "aRB2", "ARB2", "aRBL", "ARBL", aBpriorfreq)[0]
    is_tmore_amplicon = is_weak_amplicon if not tprov else is_strong_amplicon

    normCDP1 = f.sumpair("CDP12b") + 1
    normBDP = f.sumpair("BDPb") + 1
    c2DP = f.a("cDP2f") + f.a("cDP2r")
    try_tier2 = (c2DP >= 2) and (normBDP * P.fam_bias_overseq_perc >= normCDP1 * 100) and (a_umi_dp * 100 > a_dp * 50)
    tier2 = bool(int(d["tki_tier2"])) if is_rescued else try_tier2
    out["tier2"] = int(tier2)
    # (fmt.c2LP0[0] * 4: index 0 of the allele vector is the allele itself here)
    cFA2L = ((cdiv(f.a("c2LP0") ** 2 * 2, max(1, min(c2DP, f.a("c2LP0") * 4))) + c2altpc) / (f.X("C2LP0") + 1.0)) if tier2 else 1.0
    cFA2R = ((cdiv(f.a("c2RP0") ** 2 * 2, max(1, min(c2DP, f.a("c2RP0") * 4))) + c2altpc) / (f.X("C2RP0") + 1.0)) if tier2 else 1.0
    c2LPFA = c2RPFA = c2LBFA = c2RBFA = 1.0
    if tier2:
        C2DP = f.sumpair("CDP2b")
        prior_dec = 0
        c2Ppriorfreq = max(0, aPpriorfreq - prior_dec)
        c2Bpriorfreq = max(0, aBpriorfreq - prior_dec)

        def fam_pc(one, two, ALL2, length, ALLlength, prior):
            return dp4_to_pcFA(False, True, -1, f.a(one), c2DP, f.X(ALL2) + f.a(one) - f.a(two), C2DP, pl, phred2nat(prior),
                               max(1, f.a(length)) / float(max(1, f.a("c2BQ2"))), max(1, f.X(ALLlength)) / float(max(1, f.X("C2BQ2"))), c2altpc, 1.0)
        c2LPFA = fam_pc("c2LP1", "c2LP2", "C2LP2", "c2LPL", "C2LPL", c2Ppriorfreq)[0]
        c2RPFA = fam_pc("c2RP1", "c2RP2", "C2RP2", "c2RPL", "C2RPL", c2Ppriorfreq)[0]
        c2LBFA = fam_pc("c2LB1", "c2LB2", "C2LB2", "c2LBL", "C2LBL", c2Bpriorfreq)[0]
        c2RBFA = fam_pc("c2RB1", "c2RB2", "C2RB2", "c2RBL", "C2RBL", c2Bpriorfreq)[0]

    ALpd = (f.X("ALI2") + 0.5) / (f.X("ADPfr") + f.X("ADPrr") - f.X("ALI2") + 0.5)
    aLpd = (f.a("aLI1") + ALpd / (1.0 + ALpd)) / (f.a("aDPfr") + f.a("aDPrr") - f.a("aLI1") + 1.0 / (1.0 + ALpd))
    _aLIFAx2 = dp4_to_pcFA(False, False, dedup_frag_frac, f.a("aLI1"), f.a("aDPfr") + f.a("aDPrr"), f.X("ALI2") + f.a("aLI1") - f.a("aLI2"), f.X("ADPfr") + f.X("ADPrr"),
                           pl, phred2nat(aIpriorfreq), aLpd, ALpd, 0.25, 0.5)
    aLIFA = _aLIFAx2[0] * (dir_bias_div if is_tmore_amplicon else max(dir_bias_div, aDPFA / _aLIFAx2[1]))
    ARpd = (f.X("ARI2") + 0.5) / (f.X("ADPff") + f.X("ADPrf") - f.X("ARI2") + 0.5)
    aRpd = (f.a("aRI1") + ARpd / (1.0 + ARpd)) / (f.a("aDPff") + f.a("aDPrf") - f.a("aRI1") + 1.0 / (1.0 + ARpd))
    _aRIFAx2 = dp4_to_pcFA(False, False, dedup_frag_frac, f.a("aRI1"), f.a("aDPff") + f.a("aDPrf"), f.X("ARI2") + f.a("aRI1") - f.a("aRI2"), f.X("ADPff") + f.X("ADPrf"),
                           pl, phred2nat(aIpriorfreq), aRpd, ARpd, 0.25, 0.5)
    aRIFA = _aRIFAx2[0] * (dir_bias_div if is_tmore_amplicon else max(dir_bias_div, aDPFA / _aRIFAx2[1]))
    aSIFA = max((f.a("aLI1") + 0.5) / (f.X("ALI2") + f.a("aLI1") - f.a("aLI2") + 1.0), (f.a("aRI1") + 0.5) / (f.X("ARI2") + f.a("aRI1") - f.a("aRI2") + 1.0))

    if is_ins(symbol) or is_del(symbol):
        indel_multialleles_coef = max(1, f.a("bDPa")) / float(max(1, f.a("bDPf") + f.a("bDPr")))
        is_in_indel_major_reg = (max(f.X("APDP", 1), f.X("APDP", 3)) + max(f.X("APDP", 2), f.X("APDP", 4))) * 0.5 * (1.0 + FLT_EPSILON) < aDP * indel_multialleles_coef
        if ((min(gapSa_size, P.microadjust_nobias_pos_indel_maxlen) * aDPFA * indel_multialleles_coef >= P.nobias_pos_indel_lenfrac_thres)
                or (max(rtr1_tracklen, rtr2_tracklen) >= P.nobias_pos_indel_str_track_len and is_in_indel_major_reg and (not is_outlier_del)
                    and not (f.X("APXM", 0) > f.X("APXM", 1) * P.microadjust_nobias_pos_indel_misma_to_indel_ratio))):
            aLPFA += 2.0; aRPFA += 2.0; aLBFA += 2.0; aRBFA += 2.0
            if tier2:
                c2LPFA += 2.0; c2RPFA += 2.0; c2LBFA += 2.0; c2RBFA += 2.0
        if f.a("bMQ") >= P.microadjust_nobias_pos_indel_bMQ and f.a("a2XM2") * 100 >= aDP * 100 * P.microadjust_nobias_pos_indel_perc:
            aLIFA += 2.0; aRIFA += 2.0
    elif LINK_M == symbol or LINK_NN == symbol:
        pc = P.bias_FA_pseudocount_indel_in_read
        aLBFA = min(aLBFA, (pc + f.a("aLB1")) / float(pc * 2 + ADP))
        aRBFA = min(aRBFA, (pc + f.a("aRB1")) / float(pc * 2 + ADP))
    elif refsymbol == symbol:
        aLIFA = aRIFA = max(aLIFA, aRIFA)

    avg_sqr_indel_len = max(cdiv(f.X("APXM", 4), max(1, f.X("APDP", 1))), cdiv(f.X("APXM", 5), max(1, f.X("APDP", 2))))
    if ((not is_subst(symbol)) and (P.microadjust_nobias_pos_indel_maxlen ** 2 < avg_sqr_indel_len)
            and (LINK_M == symbol or LINK_NN == symbol or ((gapSa_size * 2) ** 2 < avg_sqr_indel_len))):
        pc = P.bias_FA_pseudocount_indel_in_read
        aLPFA_minA = (pc + f.a("aLP1")) / float(pc * 2 + f.X("ALP1"))
        aRPFA_minA = (pc + f.a("aRP1")) / float(pc * 2 + f.X("ALP1"))      # sic: ALP1
        aLPFA = min(aLPFA, aLPFA_minA)
        aRPFA = min(aRPFA, aRPFA_minA)
        if tier2:
            c2LPFA = min(c2LPFA, aLPFA_minA)
            c2RPFA = min(c2RPFA, aRPFA_minA)
    if tprov or (SEQUENCING_PLATFORM_IONTORRENT == P.inferred_sequencing_platform):
        aLIFA = aRIFA = max(aLIFA, aRIFA)

    aPFFA = (f.a("aPF1") + pfa * 100.0) / (f.X("APF2") + (f.a("aPF1") - f.a("aPF2")) + 100.0)
    aSSFAx2 = dp4_to_pcFA(True, False, dedup_frag_frac, f.a("aRIf"), f.a("aLIr"), f.X("ARIf"), f.X("ALIr"), pl, phred2nat(aSBpriorfreq))
    bias_priorfreq_orientation_base = (P.bias_priorfreq_orientation_snv_base if is_subst(symbol) else P.bias_priorfreq_orientation_indel_base) + allbias_allprior
    bias_priorfreq_orientation_all = math.log(max(aDPFA, P.bias_orientation_min_effective_allelefrac) ** 2) + phred2nat(bias_priorfreq_orientation_base)
    _cROFA1x2 = dp4_to_pcFA(True, False, dedup_frag_frac, f.a("cDP1f"), f.a("cDP1r"), f.X("CDP1b", 0), f.X("CDP1b", 1), pl, bias_priorfreq_orientation_all)
    if P.bias_is_orientation_artifact_mixed_with_sequencing_error:
        cROFA10x2 = dp4_to_pcFA(True, False, dedup_frag_frac, f.a("cDP1f"), f.a("cDP1r"), f.X("CDP1b", 0), f.X("CDP1b", 1), pl, bias_priorfreq_orientation_all)
        cROFA12x2 = dp4_to_pcFA(True, False, dedup_frag_frac, f.a("cDP12f"), f.a("cDP12r"), f.X("CDP12b", 0), f.X("CDP12b", 1), pl, bias_priorfreq_orientation_all)
        _cROFA1x2 = cROFA12x2 if ((f.X("ADPff") * 8 >= ADP) and (f.X("ADPfr") * 8 >= ADP) and (f.X("ADPrf") * 8 >= ADP) and (f.X("ADPrr") * 8 >= ADP)) else cROFA10x2
    cROFA1x2 = _cROFA1x2
    cROFA2x2 = dp4_to_pcFA(True, True, -1, f.a("cDP2f"), f.a("cDP2r"), f.X("CDP2b", 0), f.X("CDP2b", 1), pl, bias_priorfreq_orientation_all, -1, -1, c2altpc, 1.0)
    aSSFA = aSSFAx2[0] * dir_bias_div
    cROFA1 = cROFA1x2[0] * dir_bias_div
    cROFA2 = cROFA2x2[0] * dir_bias_div

    bAD, AD = f.a("bAD"), f.a("AD")
    if is_ins(symbol) or is_del(symbol):
        bAD = min(bAD, f.a("bDPa"))
        AD = min(AD, f.a("cDP0a"))
    out["bAD"], out["AD"] = bAD, AD

    bFA = (f.a("bDPa") + pfa) / (f.sumpair("BDPb") + 1.0)
    cFA0 = (f.a("cDP0a") + pfa * (P.lib_nonwgs_ad_pseudocount if does_fmt_imply_short_frag(d, P.lib_wgs_min_avg_fraglen) else 1.0)) / (f.sumpair("CDP1b") + 1.0)
    is_strand_r_weak = (f.X("ADPfr") + f.X("ADPrr")) * P.microadjust_nobias_strand_all_fold < (f.X("ADPff") + f.X("ADPrf")) * unbias_ratio
    is_strand_f_weak = (f.X("ADPff") + f.X("ADPrf")) * P.microadjust_nobias_strand_all_fold < (f.X("ADPfr") + f.X("ADPrr")) * unbias_ratio
    if is_strand_r_weak:
        aLIFA += 4.0; aSSFA += 4.0
    if is_strand_f_weak:
        aRIFA += 4.0; aSSFA += 4.0

    aLPFA2 = max(aDPFA * 0.01, aLPFA); aRPFA2 = max(aDPFA * 0.01, aRPFA); aLBFA2 = max(aDPFA * 0.01, aLBFA); aRBFA2 = max(aDPFA * 0.01, aRBFA)
    c2LPFA2 = max(cFA2 * 0.01, c2LPFA); c2RPFA2 = max(cFA2 * 0.01, c2RPFA); c2LBFA2 = max(cFA2 * 0.01, c2LBFA); c2RBFA2 = max(cFA2 * 0.01, c2RBFA)
    aLIFA2 = max(aDPFA * 0.01, aLIFA); aRIFA2 = max(aDPFA * 0.01, aRIFA); aSSFA2 = max(aDPFA * 0.05, aSSFA)
    cROFA1 = max(aDPFA * 1e-4, cROFA1)
    cROFA2 = max(aDPFA * 1e-4, cROFA2)

    fBTA = float(f.sumpair("BTAb") + 200)
    fBTB = float(f.sumpair("BTBb") + 6)
    fbTA = float(f.a("bTAf") + f.a("bTAr") + 100)
    fbTB = float(f.a("bTBf") + f.a("bTBr") + 3)
    frag_sidelen_frac = 1.0 - min(
        between(cdiv(f.a("aLIT"), max(1, f.a("aDPfr") + f.a("aDPrr"))) - P.microadjust_longfrag_sidelength_min, 0, P.microadjust_longfrag_sidelength_max),
        between(cdiv(f.a("aRIT"), max(1, f.a("aDPff") + f.a("aDPrf"))) - P.microadjust_longfrag_sidelength_min, 0, P.microadjust_longfrag_sidelength_max)) / P.microadjust_longfrag_sidelength_zeroMQpenalty
    _alt_frac_mut_affected_tpos = fbTB / fbTA
    alt_frac_mut_affected_tpos = (max(0, _alt_frac_mut_affected_tpos - 0.2) * 1.25) if is_nmore_amplicon else _alt_frac_mut_affected_tpos
    nonalt_frac_mut_affected_tpos = (fBTB + P.contam_any_mul_frac * fbTB - fbTB) / (fBTA + P.contam_any_mul_frac * fbTA - fbTA)
    frac_mut_affected_pos = max(P.syserr_MQ_NMR_expfrac,
                                P.syserr_MQ_NMR_altfrac_coef * alt_frac_mut_affected_tpos * frag_sidelen_frac - P.syserr_MQ_NMR_nonaltfrac_coef * nonalt_frac_mut_affected_tpos)
    bNMQ = cround(numstates2phred(math.pow(frac_mut_affected_pos / P.syserr_MQ_NMR_expfrac, P.syserr_MQ_NMR_pl_exponent)) * frac_mut_affected_pos)
    out["bNMa"] = cround(100 * alt_frac_mut_affected_tpos)
    out["bNMb"] = cround(100 * nonalt_frac_mut_affected_tpos)
    out["bNMQ"] = bNMQ

    is_tmore_amplicon_with_primerlen = is_tmore_amplicon or ((P.primerlen > 0) and not (0x4 & P.primer_flag))
    bFAa = bFA
    tier1_selfonly_aFA_min = min(cROFA1, aLPFA2, aRPFA2, aLBFA2, aRBFA2, cFA0,
                                 aDPFA * between(1.0 + aDPFA - alt_frac_mut_affected_tpos, 0.1, 1.0), aPFFA * aSSFA2 / max(aSSFA2, aSSFAx2[1]))
    tier1_selfplus_aFA_min = min(aSSFA2, aLIFA2, aRIFA2, max(aDPFA * 0.01, aSIFA), bFAa)
    cFA2a = (cFA2 * P.powlaw_amplicon_allele_fraction_coef) if (is_tmore_amplicon_with_primerlen and not is_rescued) else cFA2
    cFA3a = cFA3 if (normBDP * 100 > normCDP1 * (cdiv(P.fam_tier3DP_bias_overseq_perc - 100, (2 if is_rescued else 1)) + 100)) else 1.0
    c23FA = cFA2a
    tier2_selfonly_c2FA_min = min(cROFA2, c2LPFA2, c2RPFA2, c2LBFA2, c2RBFA2, cFA2a, cFA3a, cFA2L, cFA2R)

    out["nNFA"] = [-numstates2deciphred(v) for v in (counterbias_P_FA, counterbias_BQ_FA, aDPFA, bFA, cFA0, cFA2)]
    fts_bits, fts_pct, nAFA, nBCFA = 0, [], [], []
    bit = 0

    def push(vec, refFA, biasFA):
        nonlocal fts_bits, bit
        vec.append(-numstates2deciphred(biasFA))
        fired = biasFA < refFA * P.bias_thres_FTS_FA
        if fired:
            fts_bits |= (1 << bit)
        fts_pct.append(cround(100.0 * biasFA / refFA) if fired else 0)
        bit += 1
    for v in (aSSFA2, aPFFA, aSIFA, aLBFA2, aRBFA2, aLPFA2, aRPFA2, aLIFA2, aRIFA2):
        push(nAFA, aDPFA, v)
    push(nBCFA, bFA, cFA0)
    push(nBCFA, cFA0, bFA)
    push(nBCFA, cFA0, cROFA1)
    for v in (cROFA2, c2LPFA2, c2RPFA2, c2LBFA2, c2RBFA2, cFA2L, cFA2R):
        push(nBCFA, cFA2, v)
    out["nAFA"], out["nBCFA"], out["FTS"], out["FTSpct"] = nAFA, nBCFA, fts_bits, fts_pct

    aNCFA = (max((f.a("aNC") + 0.5) / (ADP + 1.0), between((f.a("cDP1f") + f.a("cDP1r")) / 300.0, 1.0 / 3.0, 2.0 / 3.0) * aDPFA)
             if ((not tprov) and does_fmt_imply_short_frag(d, P.lib_wgs_min_avg_fraglen) and (is_ins(symbol) or is_del(symbol)) and gapSa_size >= P.lib_nonwgs_clip_penal_min_indelsize)
             else 2.0)
    counterbias_normalgerm_FA = (1e-9 if ((not tprov) or not does_fmt_imply_short_frag(d, P.lib_wgs_min_avg_fraglen))
                                 else between(aPFFA * aPFFA * (1.0 / P.lib_nonwgs_normal_full_self_rescue_fa), aPFFA * P.lib_nonwgs_normal_min_self_rescue_fa_ratio, aPFFA))
    counterbias_FA = max(counterbias_P_FA, counterbias_BQ_FA, counterbias_normalgerm_FA)
    dedup_FA = min(bFA, cFA0) if not tprov else max(bFA, cFA0)
    frac_umi2seg = min(1.0, c23FA / aDPFA, aDPFA / c23FA)
    refbias = 0.0
    if (is_ins(symbol) or is_del(symbol)) and is_rescued:
        indel_noinfo_nbases = gapSa_size * (2 if is_ins(symbol) else 1) + max(gapSa_size, rtr1_tracklen, rtr2_anyTR_tracklen)
        refbias = float(indel_noinfo_nbases) / (float(min(f.X("ALPL"), f.X("ARPL")) * 2 + indel_noinfo_nbases) / float(f.X("ABQ2") + 0.5))
        refbias = min(refbias, P.microadjust_refbias_indel_max)
    sCDP1, sCDP2 = f.sumpair("CDP1b"), f.sumpair("CDP2b")
    min_abcFA_v = max(min(min(tier1_selfplus_aFA_min, tier1_selfonly_aFA_min), aNCFA), counterbias_FA)
    out["cDP1v"] = trunc(calc_normFA_from_rawFA_refbias(min_abcFA_v, refbias) * sCDP1 * 100)
    min_abcFA_w = max(min(aLPFA2, aRPFA2, aLBFA2, aRBFA2, bFA, aNCFA), counterbias_FA)
    out["cDP1w"] = trunc(calc_normFA_from_rawFA_refbias(min_abcFA_w, refbias) * sCDP1 * 100)
    min_abcFA_x = min(aPFFA, dedup_FA)
    if tprov:
        min_abcFA_x = max(min_abcFA_x, counterbias_FA)
    out["cDP1x"] = 1 + trunc(min_abcFA_x * sCDP1 * 100)
    c2XBFA2 = between(3.0 * c2LBFA2 * c2RBFA2 * aSSFA2 / (cFA2 ** 3), min(c2LBFA2, c2RBFA2) / 8.0, min(c2LBFA2, c2RBFA2))
    c2XPFA2 = between(3.0 * c2LPFA2 * c2RPFA2 * aSSFA2 / (cFA2 ** 3), min(c2LPFA2, c2RPFA2) / 8.0, min(c2LPFA2, c2RPFA2))
    c2XXFA2 = min(c2XBFA2, c2XPFA2)
    min_c23FA_v = max(min(min(tier1_selfplus_aFA_min, tier2_selfonly_c2FA_min, c2XXFA2), aNCFA), counterbias_FA * frac_umi2seg)
    out["cDP2v"] = trunc(calc_normFA_from_rawFA_refbias(min_c23FA_v, refbias) * sCDP2 * 100)
    min_c23FA_w = max(min(c2LPFA2, c2RPFA2, c2XXFA2, c2LBFA2, c2RBFA2, cFA2, aNCFA), counterbias_FA * frac_umi2seg)
    out["cDP2w"] = trunc(calc_normFA_from_rawFA_refbias(min_c23FA_w, refbias) * sCDP2 * 100)
    min_c23FA_x = min(aPFFA, c23FA)
    out["cDP2x"] = 1 + trunc(min_c23FA_x * sCDP2 * 100)
    return out


def sum_DPv(outs, symbols):
    """BcfFormat_symbol_sum_DPv, main.hpp:4888-4906, over the alleles of one (position, symbol type) group -> (sums [6], NN values [6])."""
    keys = ("cDP1v", "cDP1w", "cDP1x", "cDP2v", "cDP2w", "cDP2x")
    s1, s2 = [0] * 6, [0] * 6
    for o, sym in zip(outs, symbols):
        for i, k in enumerate(keys):
            s1[i] += o[k]
        if sym in (BASE_NN, LINK_NN):
            for i, k in enumerate(keys):
                s2[i] = o[k]
    return s1, s2


# ------------------------------------------------------------------------------------------------------------------------------------
# BcfFormat_symbol_calc_qual, main.hpp:4908-5343 (+ PhredMutationTable :213-262, calc_binom_10log10_likeratio main_conversion.hpp:222-237,
# logit2 :211-219, indel_phred main.hpp:794-801, indel_len_rusize_phred :757-790).  C++ types are followed where they change a result:
# `auto` and ternaries that mix int and double are double, a double assigned to an integer variable or pushed into a FORMAT vector is
# truncated, int64mul() truncates its operands first.
# ------------------------------------------------------------------------------------------------------------------------------------
BASE_A, BASE_C, BASE_G, BASE_T = 0, 1, 2, 3
INS_N_ANCHOR_BASES = 1              # main.hpp:155
TIN_CONTAM_MICRO_VQ_DELTA = 0       # main.hpp:157


def calc_binom_10log10_likeratio(prob, a, b, bidirectional=False, set_max_prob_to_one=False):
    if set_max_prob_to_one:
        prob = min(1.0, prob)
    prob = (prob + DBL_EPSILON) / (1.0 + (2.0 * DBL_EPSILON))
    a += DBL_EPSILON
    b += DBL_EPSILON
    A = prob * (a + b)
    B = (1.0 - prob) * (a + b)
    if bidirectional or a > A:
        return 10.0 / math.log(10.0) * (a * math.log(a / A) + b * math.log(b / B))
    return 0.0


def logit2(a, b):
    return math.log(prob2odds((a + DBL_EPSILON) / (a + b + 2.0 * DBL_EPSILON)))


def indel_phred(ampfact, repeatsize_at_max_repeatnum, max_repeatnum):
    region_size = repeatsize_at_max_repeatnum * max_repeatnum
    num_slips = ((float(region_size - 8)) if region_size > 64 else math.log1p(math.exp(float(region_size) - 8.0))) * ampfact / float(repeatsize_at_max_repeatnum * repeatsize_at_max_repeatnum)
    return int(math.floor(-10 * math.log((1.0 - DBL_EPSILON) / (num_slips + 1.0)) / math.log(10)))


N_UNITS_TO_PHRED = [0, 0, 3, 5, 6, 7, 8, 8, 9, 10, 10, 10, 11, 11, 11, 12, 12, 12, 13]


def indel_len_rusize_phred(indel_len, repeatunit_size):
    if 0 == (indel_len % repeatunit_size):
        return N_UNITS_TO_PHRED[min(indel_len // repeatunit_size, len(N_UNITS_TO_PHRED) - 1)]
    return N_UNITS_TO_PHRED[min(indel_len, len(N_UNITS_TO_PHRED) - 1)]


def sscs_phred_err_rate(P, con_symbol, alt_symbol):
    """PhredMutationTable::toPhredErrRate with the table calc_qual builds (main.hpp:4935-4942)."""
    if is_ins(con_symbol) or is_del(con_symbol):
        raw = P.fam_phred_sscs_indel_open
    elif con_symbol == LINK_M:
        if alt_symbol in (LINK_D1, LINK_I1): raw = P.fam_phred_sscs_indel_open
        elif alt_symbol in (LINK_D2, LINK_I2): raw = P.fam_phred_sscs_indel_open + P.fam_phred_sscs_indel_ext * 1
        else: raw = P.fam_phred_sscs_indel_open + P.fam_phred_sscs_indel_ext * 2
    elif (con_symbol == BASE_C and alt_symbol == BASE_T) or (con_symbol == BASE_G and alt_symbol == BASE_A): raw = P.fam_phred_sscs_transition_CG_TA
    elif (con_symbol == BASE_A and alt_symbol == BASE_G) or (con_symbol == BASE_T and alt_symbol == BASE_C): raw = P.fam_phred_sscs_transition_AT_GC
    elif (con_symbol == BASE_C and alt_symbol == BASE_A) or (con_symbol == BASE_G and alt_symbol == BASE_T): raw = P.fam_phred_sscs_transversion_CG_AT
    else: raw = P.fam_phred_sscs_transversion_other
    return raw + (3 if P.tumor_vcf_fname_nonempty else 0)


def calc_qual(d, dpv, cdpv, extra, P):
    """d: traced inputs of the record; dpv: calc_DPv's outputs for it; cdpv = (sums [6], NN [6]) of its group from sum_DPv;
    extra = (ins_cdepth, del_cdepth, ins1_cdepth, del1_cdepth, repeatunit.size(), repeatnum).  Returns the FORMAT values the function pushes."""
    f = _F(d)
    out = {}
    tprov = bool(P.tumor_vcf_is_provided)
    is_rescued = tprov                                        # main.cpp:979
    ins_cdepth, del_cdepth, ins1_cdepth, del1_cdepth, ru_size, repeatnum = [int(v) for v in extra]
    rtr1_tracklen, rtr1_unitlen, rtr2_tracklen, rtr2_unitlen = int(d["rtr1_tracklen"]), int(d["rtr1_unitlen"]), int(d["rtr2_tracklen"]), int(d["rtr2_unitlen"])
    tpfa = float(d["tpfa_qual"])
    refsymbol, symbol = int(d["refsymbol"]), f.a("symbol")
    indel_size = f.a("gapSa_len")
    cDP1v, cDP1w, cDP1x, cDP2v, cDP2w = dpv["cDP1v"], dpv["cDP1w"], dpv["cDP1x"], dpv["cDP2v"], dpv["cDP2w"]
    CDP1v0, CDP1x0 = cdpv[0][0], cdpv[0][2]
    tier2 = bool(dpv["tier2"])
    bNMQ = dpv["bNMQ"]
    sCDP1, sCDP2, sCDP12, sBDP = f.sumpair("CDP1b"), f.sumpair("CDP2b"), f.sumpair("CDP12b"), f.sumpair("BDPb")
    a_pcr_dp, a_snv_dp = f.X("APDP", 5), f.X("APDP", 6)

    cFA2 = (f.a("cDP2f") + f.a("cDP2r") + 0.5) / (sCDP2 + 1.0)
    powlaw_sscs_phrederr = sscs_phred_err_rate(P, refsymbol, symbol) + (0 if not tprov else 4)
    umi_cFA = (float(cDP2v) + 0.5) / float(sCDP2 * 100 + 1.0)
    umi_cFA_w = (float(cDP2w) + 0.5) / float(sCDP2 * 100 + 1.0)
    powlaw_sscs_inc1 = trunc(powlaw_sscs_phrederr - ((float(P.fam_phred_pow_sscs_transversion_AT_TA_origin) if ((BASE_A == refsymbol and BASE_T == symbol) or (BASE_T == refsymbol and BASE_A == symbol))
                                                       else P.fam_phred_pow_sscs_snv_origin) if is_subst(symbol) else P.fam_phred_pow_sscs_indel_origin))
    powlaw_sscs_inc4tn = trunc((max(P.fam_phred_sscs_transition_CG_TA, P.fam_phred_sscs_transition_AT_GC, P.fam_phred_sscs_transversion_CG_AT, P.fam_phred_sscs_transversion_other)
                                - P.fam_phred_pow_sscs_snv_origin) if is_subst(symbol) else float(powlaw_sscs_inc1))
    is_substitution_oxidation = (BASE_C == refsymbol and BASE_A == symbol) or (BASE_G == refsymbol and BASE_T == symbol)
    powlaw_sscs_inc4tn += P.tn_q_inc_max_sscs_CG_AT if is_substitution_oxidation else P.tn_q_inc_max_sscs_other
    t2n_contam_frac = (tpfa if tpfa > 0 else 0) * P.contam_t2n_mul_frac
    contamfrac = P.contam_any_mul_frac + (1.0 - P.contam_any_mul_frac) * t2n_contam_frac

    aDP = f.a("aDPff") + f.a("aDPfr") + f.a("aDPrf") + f.a("aDPrr")
    ADP = f.X("ADPff") + f.X("ADPrf") + f.X("ADPfr") + f.X("ADPrr")
    cDP0, CDP0 = f.a("cDP1f") + f.a("cDP1r"), sCDP1
    cDP2, CDP2 = f.a("cDP2f") + f.a("cDP2r"), sCDP2
    aavgMQ = cdiv(f.a("aMQs"), max(1, aDP))
    diffAaMQs = cdiv(f.X("AMQs") - f.a("aMQs"), max(1, ADP - aDP)) - aavgMQ
    tn_q_inc_max = P.tn_q_inc_max
    noUMI_bias_inc = min(P.bias_FA_powerlaw_noUMI_phred_inc_snv, cdiv(aDP, 2))
    pl_noUMI_phred_inc = P.powlaw_anyvar_base + (noUMI_bias_inc if is_subst(symbol) else P.bias_FA_powerlaw_noUMI_phred_inc_indel)          # double
    withUMI_bias_inc = min(P.bias_FA_powerlaw_withUMI_phred_inc_snv - P.bias_FA_powerlaw_noUMI_phred_inc_snv, cdiv(cDP2, 2)) + noUMI_bias_inc
    pl_withUMI_phred_inc = P.powlaw_anyvar_base + (withUMI_bias_inc if is_subst(symbol) else P.bias_FA_powerlaw_withUMI_phred_inc_indel)    # double
    prior_weight = 1.0 / (f.a("cDPmf") + f.a("cDPmr") + 1.0)
    fam_thres_highBQ = P.fam_thres_highBQ_snv if is_subst(symbol) else P.fam_thres_highBQ_indel
    cMmQ = cround(numstates2phred((f.a("cDPMf") + f.a("cDPmf") + f.a("cDPMr") + f.a("cDPmr") + math.pow(10, fam_thres_highBQ / 10.0) * prior_weight)
                                  / (f.a("cDPmf") + f.a("cDPmr") + prior_weight)))
    nbases_x100_1 = f.a("bIADb") * 100 + 1
    nbases_x100_2 = min(nbases_x100_1, cDP1v + 1)
    perbase_likeratio_q_x10_1 = cdiv(10 * f.a("bIAQb"), max(1, f.a("bIADb")))
    perbase_likeratio_q_x10_2 = perbase_likeratio_q_x10_1 + cround(10 * numstates2phred(float(nbases_x100_2) / float(nbases_x100_1)))
    duped_frag_binom_qual = cdiv((perbase_likeratio_q_x10_1 if (is_ins(symbol) or is_del(symbol)) else perbase_likeratio_q_x10_2) * nbases_x100_2, 10 * 100)
    contam_frag_withmin_qual = cround(calc_binom_10log10_likeratio(t2n_contam_frac, cDP0, CDP0 - cDP0)) + 9 - 3
    phred_het3al_chance_inc_snp = max(0, 2 * P.germ_phred_hetero_snp - P.germ_phred_het3al_snp - TIN_CONTAM_MICRO_VQ_DELTA)
    phred_het3al_chance_inc_indel = max(0, 2 * P.germ_phred_hetero_indel - P.germ_phred_het3al_indel - TIN_CONTAM_MICRO_VQ_DELTA)
    phred_het3al_chance_inc = phred_het3al_chance_inc_snp if is_subst(symbol) else phred_het3al_chance_inc_indel
    if is_ins(symbol) or is_del(symbol):
        phred_het3al_chance_inc = non_neg_minus(phred_het3al_chance_inc_indel + 1, indel_size)
    contam_syserr_phred_bypassed = phred_het3al_chance_inc
    normcDP1 = f.a("cDP12f") + f.a("cDP12r") + 1
    normCDP1 = sCDP12 + 1
    normBDP = sBDP + 1
    sscs_dec1_div = 2 if is_rescued else 1
    sscs_dec1a = 0 if ((cdiv(P.fam_min_n_copies, sscs_dec1_div) <= normCDP1) or (cdiv(P.fam_min_n_copies_DPxAD, sscs_dec1_div) <= normCDP1 * normcDP1)) else (powlaw_sscs_inc1 + 3)
    sscs_dec1b = 0 if ((cdiv(P.fam_min_overseq_perc - 100, sscs_dec1_div) + 100) * normCDP1 <= 100 * normBDP) else (powlaw_sscs_inc1 + 3)
    sscs_dec1 = max(sscs_dec1a, sscs_dec1b)
    sscs_dec2 = non_neg_minus(fam_thres_highBQ, cMmQ)
    cIADnormcnt = (f.a("cIADf") + f.a("cIADr")) * 100 + 1
    cIADmincnt = min(cIADnormcnt, cDP2v + 1)
    sscs_binom_qual_fw = f.a("cIAQf") + cdiv(f.a("cIAQr") * min(P.fam_phred_dscs_all - f.a("cIDQf"), f.a("cIDQr")), max(f.a("cIDQr"), 1))
    sscs_binom_qual_rv = f.a("cIAQr") + cdiv(f.a("cIAQf") * min(P.fam_phred_dscs_all - f.a("cIDQr"), f.a("cIDQf")), max(f.a("cIDQf"), 1))
    contam_sscs_withmin_qual = cround(calc_binom_10log10_likeratio(t2n_contam_frac, cDP2, CDP2 - cDP2)) + 9 - 3
    mx = max(sscs_binom_qual_fw, sscs_binom_qual_rv)
    sub = numstates2phred(cIADnormcnt / float(cIADmincnt)) * cIADnormcnt / 100.0
    nnm = (mx - sub) if mx > sub else 0                       # non_neg_minus(int64, double): a double (or the int 0)
    sscs_binom_qual = cdiv(trunc(nnm) * cIADmincnt, cIADnormcnt)
    if mx > P.microadjust_fam_binom_qual_halving_thres and is_subst(symbol):
        sscs_binom_qual = min(sscs_binom_qual, P.microadjust_fam_binom_qual_halving_thres + cdiv(mx - P.microadjust_fam_binom_qual_halving_thres, 2))
    sscs_binom_qual -= sscs_dec1 + sscs_dec2
    min_bcFA_v = (float(cDP1v) + 0.5) / float(sCDP1 * 100 + 1.0)
    dedup_frag_powlaw_qual_v = cround(P.powlaw_exponent * numstates2phred(min_bcFA_v) + pl_noUMI_phred_inc)
    min_bcFA_w = (float(cDP1w) + 0.5) / float(sCDP1 * 100 + 1.0)
    dedup_frag_powlaw_qual_w = cround(P.powlaw_exponent * numstates2phred(min_bcFA_w) + pl_noUMI_phred_inc + tn_q_inc_max)
    ds_vq_inc_powlaw = trunc(cround(10 / math.log(10) * min(math.log((f.a("cDP12f") + 0.5) / (f.X("CDP12b", 0) + 1.0)), math.log((f.a("cDP12r") + 0.5) / (f.X("CDP12b", 1) + 1.0)))) + powlaw_sscs_phrederr)
    ds_vq_inc_binom = 3 * min(f.a("cDP2f"), f.a("cDP2r"))
    powlaw_sscs_inc2 = max(0, min(sscs_binom_qual_fw, sscs_binom_qual_rv, ds_vq_inc_powlaw, ds_vq_inc_binom, 3)) * (1 if cFA2 > 0.002 else 0)
    sscs_dec3 = -3 if is_rescued else (0 if cFA2 >= 0.003 else 5)
    sscs_base_2 = trunc(pl_withUMI_phred_inc + powlaw_sscs_inc1 + powlaw_sscs_inc2 - sscs_dec1 - sscs_dec2 - sscs_dec3)
    sscs_base_2tn = trunc(pl_withUMI_phred_inc + powlaw_sscs_inc4tn + powlaw_sscs_inc2 - sscs_dec1 - sscs_dec2 - sscs_dec3)
    sscs_powlaw_qual_v = cround(P.powlaw_exponent * numstates2phred(umi_cFA) + sscs_base_2)
    sscs_powlaw_qual_w = cround(P.powlaw_exponent * numstates2phred(umi_cFA_w) + sscs_base_2tn)
    dFA = float(f.a("dDP2") + 0.5) / float(f.X("DDP1") + 1.0)
    dSNR = float(f.a("dDP2") + 0.5) / float(f.a("dDP1") + 1.0)
    dnormFA = dFA * math.pow(dSNR, 1.0 / P.powlaw_exponent)
    fam_phred_dscs_estimated = cround((P.fam_phred_dscs_max + powlaw_sscs_phrederr) / 2.0)
    dFA_vq_binom = cdiv((fam_phred_dscs_estimated - cround(numstates2phred(1.0 / dnormFA))) * f.a("dDP2") * cIADmincnt, cIADnormcnt)
    dFA_vq_powlaw = trunc(P.powlaw_anyvar_base + (fam_phred_dscs_estimated - P.fam_phred_pow_dscs_all_origin)
                          + cround(numstates2phred(dnormFA * min(1.0, float(cDP1v + 0.5) / float(sCDP1 * 100 + 1.0)))))
    out["cMmQ"] = cMmQ

    eps = FLT_EPSILON
    is_indel_penal_applied = (SEQUENCING_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) and (not tprov)
    indel_penal_base = (cround(P.indel_multiallele_samepos_penal / math.log(2) * math.log(float(max(aDP + eps, f.X("APDP", 1), f.X("APDP", 2))) / float(aDP + eps)))
                        if is_indel_penal_applied else 0)
    indel_penal4multialleles = 0
    indel_penal4multialleles_g = 0
    indel_penal4multialleles_soma = 0
    indel_UMI_penal = 0
    if indel_size > 0 and f.a("cDP0a") > 0:
        indel_pq = float(min(indel_phred(P.indel_polymerase_slip_rate, ru_size, repeatnum), 24)) + 2 - 10.0
        eff_tracklen1 = ru_size * max(1, repeatnum) - ru_size
        eff_tracklen2 = cdiv(max(rtr1_tracklen - rtr1_unitlen, rtr2_tracklen - rtr2_unitlen), 3)
        indel_ic = (numstates2phred(float(max(indel_size + (INS_N_ANCHOR_BASES if is_ins(symbol) else 0), 1)) / float(max(eff_tracklen1, eff_tracklen2) + 1))
                    + ((numstates2phred(P.indel_del_to_ins_err_ratio) * min(200, f.a("cDP0a")) / 200) if is_ins(symbol) else 0))
        indelcdepth = ins_cdepth if is_ins(symbol) else del_cdepth
        if LINK_D1 == symbol:
            indelcdepth += ins1_cdepth
        if LINK_I1 == symbol:
            indelcdepth = trunc(indelcdepth + del1_cdepth / P.indel_del_to_ins_err_ratio)     # int += double
        nearInDelDP = f.X("APDP", 1) if is_ins(symbol) else f.X("APDP", 2)
        indel_penal4multialleles1 = cround(P.indel_multiallele_samepos_penal / math.log(2.0) * math.log(float(indelcdepth + eps) / float(f.a("cDP0a") + eps)))
        if SEQUENCING_PLATFORM_IONTORRENT == P.inferred_sequencing_platform:
            indel_penal4multialleles1 = trunc(non_neg_minus(indel_penal4multialleles1, P.indel_multiallele_samepos_penal))
        indel_penal4multialleles2 = cround(P.indel_multiallele_diffpos_penal / math.log(2.0) * math.log(float(nearInDelDP + eps) / float(max(aDP, nearInDelDP) + eps)))
        indel_penal4multialleles_g = trunc(cround(P.indel_tetraallele_germline_penal_value / math.log(2.0) * math.log(float(ins_cdepth + del_cdepth + eps) / float(f.a("cDP0a") + eps)))
                                           - P.indel_tetraallele_germline_penal_thres)
        if is_ins(symbol):
            indel_penal4multialleles = cdiv(indel_penal4multialleles1 * P.indel_ins_penal_pseudocount, P.indel_ins_penal_pseudocount + indel_size)
            indel_penal4multialleles_soma = cdiv(indel_penal4multialleles1 * P.indel_ins_penal_pseudocount, P.indel_ins_penal_pseudocount + indel_size)
        else:
            indel_penal4multialleles = max(indel_penal4multialleles1, indel_penal4multialleles2)
            indel_penal4multialleles_soma = indel_penal4multialleles1
        dedup_frag_powlaw_qual_v = trunc(dedup_frag_powlaw_qual_v + cround(indel_ic))
        dedup_frag_powlaw_qual_w = trunc(dedup_frag_powlaw_qual_w + cround(indel_ic))
        duped_frag_binom_qual = trunc(duped_frag_binom_qual + cround(indel_pq))
        sscs_indel_ic = numstates2phred(float(max(indel_size, 1) ** 2) / float(max(eff_tracklen1, eff_tracklen2) + 1))
        sscs_ins_vs_del_inc = cround(P.powlaw_exponent * numstates2phred(P.indel_del_to_ins_err_ratio))
        x = sscs_indel_ic * (0 if is_ins(symbol) else max(eff_tracklen1, eff_tracklen2)) / cround(P.indel_polymerase_size)
        extra_reward = trunc(((sscs_ins_vs_del_inc - x) if sscs_ins_vs_del_inc > x else 0) - cdiv(sscs_ins_vs_del_inc, 2))
        sscs_powlaw_qual_v = trunc(sscs_powlaw_qual_v + cround(sscs_indel_ic) + extra_reward)
        sscs_powlaw_qual_w = trunc(sscs_powlaw_qual_w + cround(sscs_indel_ic) + extra_reward)
        sscs_binom_qual = trunc(sscs_binom_qual + cround(indel_pq) + extra_reward)
        if tier2:
            v1 = (sBDP + 1.0) / float(sCDP1 + 1.0) * P.fam_indel_nonUMI_phred_dec_per_fold_overseq
            v2 = (P.fam_thres_emperr_all_flat_indel + 1) * P.fam_indel_nonUMI_phred_dec_per_fold_overseq
            indel_UMI_penal = trunc((v1 - v2) if v1 > v2 else 0)
    if is_substitution_oxidation and tprov:
        sscs_binom_qual = max(sscs_binom_qual, min(aDP, 3))
    out["aAaMQ"] = diffAaMQs

    readlenMQcap = cdiv(f.X("APXM", 2), max(1, f.X("APDP", 0))) - 17
    diffMQ = max(0, diffAaMQs)
    is_aln_extra_accurate = P.inferred_maxMQ > 60
    _systematicMQVQadd = 0 if symbol == refsymbol else min(P.germ_phred_homalt_snp, ADP * 3)
    _systematicMQVQadd_somatic = 0 if symbol != refsymbol else min(P.germ_phred_homalt_snp, ADP * 3)
    is_MQ_unadjusted = is_aln_extra_accurate or (not is_subst(symbol)) or (aDP > cdiv(ADP * 3, 4))
    _systematicMQVQminus = ((0 if is_MQ_unadjusted else cdiv(non_neg_minus(60 - 30, aavgMQ) * 2, 5))
                            + (0 if (is_MQ_unadjusted or refsymbol != symbol) else non_neg_minus(min(15, diffMQ), aavgMQ)))
    diffMQ2 = diffMQ
    if f.a("bMQ") < 20 and not tprov:
        aDPxf = f.a("aDPff") + f.a("aDPrf") + 0.5
        aDPxr = f.a("aDPfr") + f.a("aDPrr") + 0.5
        ADPxf = f.X("ADPff") + f.X("ADPrf") + 1.0
        ADPxr = f.X("ADPfr") + f.X("ADPrr") + 1.0
        if ((aDPxr / ADPxr) * 2 < (aDPxf / ADPxf) or (aDPxf / ADPxf) * 2 < (aDPxr / ADPxr)
                or (f.a("aLI1") + 0.5) / (f.X("ALI2") + 1.0) * (2 * (1.0 + DBL_EPSILON)) < aDPxr / ADPxr
                or (f.a("aRI1") + 0.5) / (f.X("ARI2") + 1.0) * (2 * (1.0 + DBL_EPSILON)) < aDPxf / ADPxf):
            diffMQ2 = max(diffMQ2, 20 - min(f.a("bMQ"), 20))
    _systematicMQ_base = (f.a("bMQ") * (P.syserr_MQ_max - P.syserr_MQ_nonref_base) / P.syserr_MQ_max + P.syserr_MQ_nonref_base) - diffMQ2 - bNMQ    # double
    _systematicMQ = trunc(float(f.a("bMQ")) if ((refsymbol == symbol) and (ADP > aDP * 2)) else (_systematicMQ_base - trunc(numstates2phred((ADP + 1.0) / (aDP + 0.5)))))
    is_nonWGS = does_fmt_imply_short_frag(d, P.lib_wgs_min_avg_fraglen)
    normal_rescued_MQ = min(non_neg_minus(readlenMQcap, 60), (P.lib_nonwgs_normal_max_rescued_MQ if is_nonWGS else P.lib_wgs_normal_max_rescued_MQ))
    systematicMQVQ1 = min(max(_systematicMQ, P.syserr_MQ_min) + _systematicMQVQadd, readlenMQcap)
    systematicBQVQ = f.a("aBQQ") if ((SEQUENCING_PLATFORM_IONTORRENT != P.inferred_sequencing_platform) and is_subst(symbol)) else 200
    is_strong_amplicon = (a_pcr_dp * 100) > f.X("APDP", 0) * 50
    is_weak_amplicon = (a_pcr_dp * 100) > f.X("APDP", 0) * 30
    is_tmore_amplicon = is_weak_amplicon if not tprov else is_strong_amplicon
    go_per_dp = cdiv(f.X("APXM", 1), max(f.X("APDP", 0), 1))
    if is_tmore_amplicon and (is_ins(symbol) or is_del(symbol)) and systematicMQVQ1 > 70 and go_per_dp > 20:
        systematicMQVQ1 = 70 + cdiv((systematicMQVQ1 - 70) * 5, go_per_dp - 15)
    indel_penal_base_add = 0
    if not tprov:
        delAPDP = max(f.X("APDP", 2), f.X("APDP", 4))
        if ((f.X("APDP", 0) < 3 * delAPDP) and (f.X("APDP", 0) < 3 * a_snv_dp) and (aDP * 3 < delAPDP) and (aDP * 3 < a_snv_dp)
                and is_subst(symbol) and (rtr2_tracklen >= 8 * rtr2_unitlen)):
            indel_penal_base_add = P.microadjust_germline_mix_with_del_snv_penalty
        if is_tmore_amplicon and is_del(symbol):
            if aDP * 4 < f.X("APDP", 2):
                indel_penal_base_add = max(indel_penal_base_add, 5)
            elif f.a("cDP0a") * 3 < 2 * del_cdepth:
                indel_penal_base_add = max(indel_penal_base_add, 2)
    systematicMQVQ = max(0, systematicMQVQ1)
    indel_penal_base2 = indel_penal_base + indel_penal_base_add
    fmtADPfx, fmtADPrx = f.X("ADPff") + f.X("ADPfr"), f.X("ADPrf") + f.X("ADPrr")
    fmtADPxf, fmtADPxr = f.X("ADPff") + f.X("ADPrf"), f.X("ADPfr") + f.X("ADPrr")
    fold = P.microadjust_strand_orientation_absence_DP_fold
    is_fmtADPfrx_imba = max(fmtADPfx, fmtADPrx) > fold * (min(fmtADPfx, fmtADPrx) + 1)
    is_fmtADPxfr_imba = max(fmtADPxf, fmtADPxr) > fold * (min(fmtADPxf, fmtADPxr) + 1)
    dedup_frag_powlaw_qual_v_minus = (((P.microadjust_orientation_absence_snv_penalty if is_fmtADPfrx_imba else 0) + (P.microadjust_strand_absence_snv_penalty if is_fmtADPxfr_imba else 0))
                                      if is_subst(symbol) else (P.microadjust_dedup_absence_indel_penalty if is_tmore_amplicon else 0))
    tn_syserr_q = systematicMQVQ + P.tn_q_inc_max + normal_rescued_MQ
    out["bMQQ"] = systematicMQVQ
    bIAQ = duped_frag_binom_qual - indel_penal_base2
    cIAQ = sscs_binom_qual - indel_penal_base
    cPCQ1 = min(dedup_frag_powlaw_qual_w - indel_penal_base2, tn_syserr_q)
    cPLQ1 = dedup_frag_powlaw_qual_v - indel_penal_base2 - dedup_frag_powlaw_qual_v_minus
    cPCQ2 = min(sscs_powlaw_qual_w - indel_penal_base, tn_syserr_q)
    cPLQ2 = sscs_powlaw_qual_v - indel_penal_base
    bTINQ = contam_frag_withmin_qual + contam_syserr_phred_bypassed
    cTINQ = contam_sscs_withmin_qual + contam_syserr_phred_bypassed
    out.update(bIAQ=bIAQ, cIAQ=cIAQ, cPCQ1=cPCQ1, cPLQ1=cPLQ1, cPCQ2=cPCQ2, cPLQ2=cPLQ2, bTINQ=bTINQ, cTINQ=cTINQ)
    aDPpc = 1 if refsymbol == symbol else 0
    penal4BQerr = (5 + cdiv(P.penal4lowdep, max(1, aDP + aDPpc) ** 2)) if is_subst(symbol) else 0
    indel_q_inc = 0 if (((not is_ins(symbol)) and (not is_del(symbol))) or is_rescued) else indel_len_rusize_phred(indel_size, repeatnum)
    out["gVQ1"] = trunc(max(0, indel_q_inc + min(min(systematicBQVQ, non_neg_minus(systematicMQVQ, _systematicMQVQminus)), bIAQ - penal4BQerr, cPLQ1)
                            - 2 * max(0, indel_penal4multialleles - P.indel_multiallele_soma_penal_thres, indel_penal4multialleles_g)))
    systematicVQsomatic_minus = 0 if is_rescued else (15 - min(cdiv(ADP * 15, 100), aDP, 15))
    systematicVQsomatic = non_neg_minus(min(systematicBQVQ, systematicMQVQ + _systematicMQVQadd_somatic), systematicVQsomatic_minus)
    bcVQ1 = min(systematicVQsomatic, bIAQ - (0 if is_rescued else penal4BQerr), cPLQ1) - indel_penal4multialleles_soma
    out["cVQ1"] = max(0, min(bcVQ1, bTINQ) - indel_UMI_penal)
    mincVQ2 = 0
    if is_ins(symbol) or is_del(symbol):
        sscs_floor_qual_v = trunc(min(P.germ_phred_homalt_indel + numstates2phred(umi_cFA), cdiv(cDP2v * 3, 100)) + ((INS_N_ANCHOR_BASES if is_ins(symbol) else 0) - INS_N_ANCHOR_BASES) * 3)
        mincVQ2 = max(mincVQ2, sscs_floor_qual_v)
    dVQinc = min(min(dFA_vq_binom, dFA_vq_powlaw) - max(0, min(cIAQ, cPLQ2)), P.fam_phred_dscs_inc_max)
    out["dVQinc"] = dVQinc
    cVQ2 = min(systematicVQsomatic, cIAQ + max(0, dVQinc), cPLQ2 + max(0, dVQinc)) - indel_penal4multialleles
    out["cVQ2"] = max(mincVQ2, min(cVQ2, cTINQ))
    cDP1y = cDP1x if is_rescued else cDP1v
    CDP1y0 = CDP1x0 if is_rescued else CDP1v0
    binom_contam_LODQ = calc_binom_10log10_likeratio(contamfrac, cDP1y, CDP1y0)
    power_contam_LODQ = cround(10.0 / math.log(10.0) * P.powlaw_exponent * max(logit2((cDP1y + 1) / float(CDP1y0 + 1), contamfrac), 0.0))
    out["CONTQ"] = trunc(min(binom_contam_LODQ, power_contam_LODQ))
    return out
