"""The calling step of the oracle (main.cpp:990-1168, output_germline, the arithmetic of append_vcf_record) against a second,
pure-Python restatement written from the reference text: tumor-only records and the normal sample of a T/N pair.  The Python side
reads the oracle's own per-allele outputs of calc_qual (gVQ1, cVQ1, CONTQ, cDP1v, ...) plus a few planes, and recomputes everything
behind them.  No GPU."""
import math

import numpy as np
import pytest

from uvc_amd import _ffi, region, synth
from util import run_region

EPS = 2.220446049250313e-16
E = _ffi.ENUMS
NN_BASE, LINK_M, LINK_NN, END = 5, 6, 13, 14


def binom_llr(prob, a, b):                       # calc_binom_10log10_likeratio<false, false>, main_conversion.hpp:221-237
    prob = (prob + EPS) / (1.0 + 2.0 * EPS)
    a += EPS; b += EPS
    A, B = prob * (a + b), (1.0 - prob) * (a + b)
    return 10.0 / math.log(10.0) * (a * math.log(a / A) + b * math.log(b / B)) if a > A else 0.0


def logit2(a, b):                                # main_conversion.hpp:205-219
    p = (a + EPS) / (a + b + 2.0 * EPS)
    return math.log(p / (1.0 - p))


def cround(x):                                   # C round(): half away from zero
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def het_lodq(a1, a2, frac, ple):                 # hetLODQ, main.hpp:5457-5462
    return min(int(binom_llr(frac, a1, a2)), cround(10.0 / math.log(10.0) * ple * max(logit2((a1 + 0.5) * 0.5 / frac, (a2 + 0.5) * 0.5 / (1.0 - frac)), 0.0)))


N_UNITS = {7: -3, 8: -2, 9: -1, 10: 3, 11: 2, 12: 1}
is_ins = lambda s: s in (10, 11, 12)
is_del = lambda s: s in (7, 8, 9)


def germline(P, refsymbol, recs, tprov):
    """output_germline up to the returned tuple, main.hpp:5483-5612.  recs: dicts of one (position, symbol type) group in emission order."""
    v = [dict(r, rec=i) for i, r in enumerate(recs) if r["symbol"] != NN_BASE]
    while len(v) <= 4:
        v.append(dict(symbol=END, gVQ1=0, CONTQ=0, cDP0a=0, cDP1v=50, rec=-1))
    v.sort(key=lambda r: -r["gVQ1"])             # Python's sort is stable, like the insertion sort behind std::sort for these sizes
    isref = lambda r: r["symbol"] in (refsymbol, NN_BASE, LINK_NN)
    ref = next(r for r in v if isref(r))
    alts = [r for r in v if not isref(r)][:3]
    a0, a1, a2, a3 = ref["gVQ1"], alts[0]["gVQ1"], alts[1]["gVQ1"], alts[2]["gVQ1"]
    subst = refsymbol <= NN_BASE
    ad0, ad1, ad2 = ref["cDP1v"] / 100.0, alts[0]["cDP1v"] / 100.0, alts[1]["cDP1v"] / 100.0
    h = [het_lodq(ad0, ad1, 1.0 - P.germ_hetero_FA, P.powlaw_exponent), het_lodq(ad1, ad0, P.germ_hetero_FA, P.powlaw_exponent),
         het_lodq(ad1, ad2, 0.5, P.powlaw_exponent), het_lodq(ad2, ad1, 0.5, P.powlaw_exponent)]
    hetero, homalt, tri = ((P.germ_phred_hetero_snp, P.germ_phred_homalt_snp, P.germ_phred_het3al_snp) if subst
                           else (P.germ_phred_hetero_indel, P.germ_phred_homalt_indel, P.germ_phred_het3al_indel))
    if tprov:
        a0, a1, a2, a3 = min(a0, ref["CONTQ"]), min(a1, alts[0]["CONTQ"]), min(a2, alts[1]["CONTQ"]), min(a3, alts[2]["CONTQ"])
    else:
        a0 = min(a0, ref["CONTQ"])
    a2penal, a3penal = max(a2 - (tri - hetero), 0), max(a3 - hetero, 0)
    a01, a12, a03 = max(max(h[0], h[1]), 0), max(max(h[2], h[3]) - 3, 0), max(a0, a3)
    s1, s2 = alts[0]["symbol"], alts[1]["symbol"]
    penal = 0
    if is_ins(s1) and is_ins(s2):
        penal += 3
        if s1 == s2:
            penal += 3
            if s1 == 10: penal += 3
    n1, n2 = N_UNITS.get(s1, 0), N_UNITS.get(s2, 0)
    if n1 and n2: penal -= min(max(abs(n1 - n2) * 3 - 5, 0), 9)
    gl = [0 - a1 - a2penal - a3penal,
          -hetero - max(a01, a2) - max(min(a01, a2) - hetero, 0) - a3penal,
          -homalt - max(a0, a2) - max(min(a0, a2) - hetero, 0) - a3penal,
          -tri - max(a12, a03) - max(min(a12, a03) - hetero, 0) - max(min(a12, min(a0, a3)) - hetero, 0) - penal]
    return dict(ret=gl[0] - max(gl[1:]), GL4=gl, GST=[a0, a1, a2, a3] + h, alt1=alts[0], alt2=alts[1], ref=ref)


def normv_quals(tAD, tDP, tVQ, cap, nAD, nDP, nVQ, coef, prior, dec_xm, ple):   # main.hpp:5982-6009
    binom = int(binom_llr((tDP - tAD) / tDP, nDP - nAD, nAD))
    plus = nAD * min(max(nDP / tDP - 1.0, 0), 1)
    frac = ((tAD + 0.5) / (tDP + 1.0)) / ((nAD + 0.5 + plus) / (nDP + 1.0 + plus))
    powlaw = cround(ple * 10.0 / math.log(10.0) * math.log(frac))
    inc = max(-prior, (-int(nAD)) * 3, min(binom - prior, powlaw - prior))
    dec = max(0, nVQ - max(0, min(binom - prior, int((math.log(max(frac, 1.001)) / math.log(2)) ** 2 * coef))))
    dec = max(dec, min(nVQ + 9, dec_xm))
    return [binom, powlaw, dec, min(cap, tVQ + inc) - dec]


def record_call(P, r, g, ref_bDP, own_bDP, aBQ2_own, ABQ2_tot, tk, group, all_out, germ_any, refsymbol):
    """main.cpp:1081-1147 + append_vcf_record, tumor-only (tk None) or normal sample (tk = the tumor record dict)"""
    symbol = r["symbol"]
    germ_phred = P.germ_phred_hetero_snp if symbol <= NN_BASE else P.germ_phred_hetero_indel
    single = g["ret"] - 3 + germ_phred
    totBDP = r["bDP"]
    nlodv = END
    if tk is None:
        nlodq1 = single
        t = dict(BDP=totBDP, bDP=own_bDP, CDP1x=r["CDP1x0"], cDP1x=r["cDP1x"], cVQ1=r["cVQ1"], cPCQ1=r["cPCQ1"], CDP2x=r["CDP2x0"], cDP2x=r["cDP2x"], cVQ2=r["cVQ2"], cPCQ2=r["cPCQ2"], bNMQ=r["bNMQ"], tDP=0)
        n = dict(cDP1x=0, CDP1x=0, cDP2x=0, CDP2x=0, cVQ1=0, cVQ2=0, BDP=0, CDP1=0)
    else:
        inc = 999
        for fp in (g["alt1"], g["alt2"]):
            real = fp["rec"] >= 0
            tAD, tDP = (tk["cDP1x"] + 50) / 100.0, (tk["CDP1x"] + 100) / 100.0
            nAD, nDP = ((fp["cDP1x"] if real else 50) + 50) / 100.0, ((fp["CDP1x0"] if real else 0) + 100) / 100.0
            frac = (tAD / tDP) / (nAD / nDP)
            binom, powlaw = int(binom_llr((tDP - tAD) / tDP, nDP - nAD, nAD)), int(P.powlaw_exponent * 10 / math.log(10) * math.log(frac))
            tri = 0
            if fp["symbol"] != symbol:
                tri = (2 * P.germ_phred_hetero_snp - P.germ_phred_het3al_snp) if symbol <= NN_BASE else (2 * P.germ_phred_hetero_indel - P.germ_phred_het3al_indel)
            new = int(min(max(min(binom, powlaw), -3), P.powlaw_anyvar_base)) + tri
            if inc > new: inc, nlodv = new, fp["symbol"]
        n_norm_alts = (totBDP - ref_bDP) + own_bDP
        nlodq1 = max(max(g["ret"], germ_phred + inc), tk["vHGQ"] + min(3, totBDP - n_norm_alts * cround(0.5 / P.contam_any_mul_frac)))
        t = tk
        n = dict(cDP1x=r["cDP1x"], CDP1x=r["CDP1x0"], cDP2x=r["cDP2x"], CDP2x=r["CDP2x0"], cVQ1=r["cVQ1"], cVQ2=r["cVQ2"], BDP=totBDP, CDP1=r["DP"])
    inc_snp, inc_indel = max(0, 2 * P.germ_phred_hetero_snp - P.germ_phred_het3al_snp), max(0, 2 * P.germ_phred_hetero_indel - P.germ_phred_het3al_indel)
    het3 = inc_snp if symbol <= NN_BASE else inc_indel
    if is_ins(symbol) or is_del(symbol): het3 = max(inc_indel + 1 - r["gapSa_len"], 0)
    qmin, qmax = P.microadjust_syserr_MQ_NMR_tn_syserr_no_penal_qual_min, P.microadjust_syserr_MQ_NMR_tn_syserr_no_penal_qual_max
    dec_xm = min(max(min(r["bNMQ"], t["bNMQ"]), qmin), qmax) - qmin
    add1 = add2 = 0.0
    dec_both = 0
    if tk is not None:
        if r["short_frag"]: add1, add2 = P.lib_nonwgs_normal_add_mul_ad * n["cDP1x"] / 100.0, P.lib_nonwgs_normal_add_mul_ad * n["cDP2x"] / 100.0
        if t["tDP"] > 500 and r["DP"] > 500 and is_del(symbol) and r["APDP2"] * 3 > r["APDP0"]: dec_both = min(max(n["cVQ1"] - 31, 0), 9)
    prior = 11 if P.inferred_sequencing_platform == 2 else 3
    assert P.tn_syserr_norm_devqual >= 0
    b4 = normv_quals((t["cDP1x"] + 0.5) / 100.0, (t["CDP1x"] + 1.0) / 100.0, t["cVQ1"], t["cPCQ1"], (n["cDP1x"] + 0.5) / 100.0 + add1, (n["CDP1x"] + 1.0) / 100.0 + add1,
                     max(n["cVQ1"] - het3, 0), P.tn_syserr_norm_devqual, prior, dec_xm, P.powlaw_exponent)
    conv = n["cVQ1"] - (3 * (n["BDP"] + 1) // (n["CDP1"] + 1))
    c4 = normv_quals((t["cDP2x"] + 0.5) / 100.0, (t["CDP2x"] + 1.0) / 100.0, t["cVQ2"], t["cPCQ2"], (n["cDP2x"] + 0.5) / 100.0 + add2, (n["CDP2x"] + 1.0) / 100.0 + add2,
                     max(n["cVQ2"] - (max(het3, 3) - 3), 0), P.tn_syserr_norm_devqual, prior, max(dec_xm, min(max(n["cVQ2"], conv), 12)), P.powlaw_exponent)
    tl1 = max(b4[3], c4[3])
    deanim = (refsymbol, symbol) in ((1, 3), (2, 0))
    realphred = lambda p: -10 * math.log(p) / math.log(10)
    lowest = max(5 - realphred((t["bDP"] + 1e-3) / (t["BDP"] + 1)) / 10.0, 7 - realphred((t["cDP2x"] * 0.01 + 1e-5) / (t["CDP2x"] * 0.01 + 1) / (5 if deanim else 1)) / 10.0)
    tlodq = (tl1 if tl1 >= 10 else tl1 * 3 - 20) - dec_both
    nlodq = nlodq1 - dec_both
    sq = min(tlodq, nlodq)
    v = float(sq) if tk is not None else max(float(tlodq), float(np.float32(lowest)))
    if v < 10.0:
        base = np.float32(10.0 ** 0.1)
        v = float(np.log1p(np.power(base, np.float32(v), dtype=np.float32), dtype=np.float32) / np.log(base, dtype=np.float32))
    keep_var = ((v >= P.vqual) or (tk is None and ((aBQ2_own >= P.vad1 and ABQ2_tot >= P.vdp1 and ABQ2_tot * P.vfa1 <= aBQ2_own) or (t["bDP"] >= P.vad2 and t["BDP"] >= P.vdp2 and t["BDP"] * P.vfa2 <= t["bDP"])))) \
        and (symbol != refsymbol or all_out or germ_any)
    keep = keep_var and t["bDP"] >= (P.min_r_ad if symbol == refsymbol else P.min_a_ad)
    return dict(vHGQ=single, NLODQ=nlodq, NLODV=nlodv, TLODQ=tlodq, SomaticQ=sq, TNBQF=b4, TNCQF=c4, QUAL=v, FILTER=min(int(v // 10), 6), keep=int(keep))


def check(oracle_lib, reads, P, rec, R, tumor_keys=None):
    frag, seg, prep32, prep64 = R.fetch("FRAG"), R.fetch("SEG32"), R.fetch("PREP32"), R.fetch("PREP64")
    n = len(rec["refpos"])
    rows = [{k: int(rec[k][i]) for k in rec} for i in range(n)]
    i = 0
    n_checked = 0
    while i < n:
        j = i
        zpos = rows[i]["refpos"] + (1 if rows[i]["symbol"] <= NN_BASE else 0)
        groups = {}
        while j < n and rows[j]["refpos"] + (1 if rows[j]["symbol"] <= NN_BASE else 0) == zpos:
            groups.setdefault(0 if rows[j]["symbol"] <= NN_BASE else 1, []).append(rows[j]); j += 1
        calls = {st: germline(P, g[0]["refsymbol"], g, tumor_keys is not None) for st, g in groups.items()}
        for st, g in groups.items():
            c = calls[st]
            for r in g:
                assert r["vNLODQ"] == c["ret"] and [r["GL4_%d" % k] for k in range(4)] == c["GL4"] and [r["GST%d" % k] for k in range(8)] == c["GST"], (r["refpos"], r["symbol"])
                if not r["out"]: continue
                x = r["refpos"] - reads["beg"]
                sym, refsym = r["symbol"], r["refsymbol"]
                bd = lambda s: int(frag[0, E["UVC_FRAG_bDP"], s, x] + frag[1, E["UVC_FRAG_bDP"], s, x])
                syms = range(0, 6) if st == 0 else range(6, 14)
                r["short_frag"] = (int(prep64[E["UVC_P_a_LI"], x]) + int(prep64[E["UVC_P_a_RI"], x])) < (int(prep32[E["UVC_P_a_LIDP"], x]) + int(prep32[E["UVC_P_a_RIDP"], x])) * P.lib_wgs_min_avg_fraglen
                r["APDP0"], r["APDP2"] = int(prep32[E["UVC_P_a_dp"], x]), int(prep32[E["UVC_P_a_near_del_dp"], x])
                tk = None
                if tumor_keys is not None:
                    names = ("refpos", "symbol", "cDP1x", "CDP1x", "bDP", "BDP", "tier2", "indel_len", "cVQ1", "cPCQ1", "cDP2x", "CDP2x", "cVQ2", "cPCQ2", "bNMQ", "vHGQ", "tDP")
                    tk = dict(zip(names, tumor_keys[r["tkey"]]))
                want = record_call(P, r, c, bd(refsym), bd(sym), int(seg[E["UVC_S_aBQ2"], sym, x]), int(np.int32(sum(int(seg[E["UVC_S_aBQ2"], s, x]) for s in syms))), tk, g, False,
                                   any(gg[0]["germ_emit"] for gg in groups.values()), refsym)
                got = dict(vHGQ=r["vHGQ"], NLODQ=r["NLODQ"], NLODV=r["NLODV"], TLODQ=r["TLODQ"], SomaticQ=r["SomaticQ"], TNBQF=[r["TNBQF%d" % k] for k in range(4)], TNCQF=[r["TNCQF%d" % k] for k in range(4)],
                           FILTER=r["FILTER"], keep=r["keep"])
                q = float(np.array([r["QUAL"]], np.int32).view(np.float32)[0])
                assert abs(q - want.pop("QUAL")) <= 1e-4 * max(1.0, abs(q)), (r["refpos"], sym, q)
                assert got == want, (r["refpos"], sym, got, want)
                n_checked += 1
        i = j
    return n_checked


def test_tumor_only_calls(oracle_lib):
    reads = synth.generate_region(region_len=4000, depth=120, seed=21, snv_every=150, somatic_every=400, indel_every=250)
    P = region.default_params(oracle_lib)
    R = run_region(oracle_lib, reads, params=P)
    rec = R.score()
    assert check(oracle_lib, reads, P, rec, R) > 100
    assert rec["keep"].sum() > 10 and len(set(rec["germ_GT"].tolist())) >= 2


def test_normal_sample_calls(oracle_lib):
    reads = synth.generate_region(region_len=3000, depth=80, seed=22, snv_every=200, somatic_every=300, indel_every=300)
    base = run_region(oracle_lib, reads).score()
    keys = []
    for i in range(0, len(base["refpos"]), 2):
        sym = int(base["symbol"][i])
        keys.append((int(base["refpos"][i]), sym, int(base["cDP1x"][i]) * 2, int(base["CDP1x0"][i]) * 2, int(base["bAD"][i]) + 3, int(base["bDP"][i]) * 2, 0, 2 if 7 <= sym <= 12 else 0,
                     int(base["cVQ1"][i]) + 5, int(base["cPCQ1"][i]), int(base["cDP2x"][i]), int(base["CDP2x0"][i]), int(base["cVQ2"][i]), int(base["cPCQ2"][i]), int(base["bNMQ"][i]), int(base["vHGQ"][i]), int(base["DP"][i]) * 9))
    keys = sorted(set(keys), key=lambda k: (k[0], k[1]))
    P = region.default_params(oracle_lib); P.tumor_vcf_is_provided = 1
    R = run_region(oracle_lib, reads, params=P)
    rec = R.score(tumor_keys=keys)
    assert check(oracle_lib, reads, P, rec, R, tumor_keys=keys) == sum(1 for k in keys if k[1] != LINK_NN)   # LINK_NN records are not written (OUTVAR_LINK_NN)
    assert len(set(rec["NLODV"][rec["out"] == 1].tolist())) > 1
