#!/usr/bin/env python3
"""Generates tests/golden/chain_*.npz: fuzzed reads + the per-position planes of the accumulate path computed by the chain of INDEPENDENT
Python restatements (tests/rtr_cases.py -> prep_restatement.py -> p2_restatement.py / segbias_restatement.py -> p3_restatement.py ->
p45_restatement.py), each written from the reference text.  No library is loaded here: neither the oracle nor the HIP library produces a
number in these files, and the parameters come from tests/golden/params_default.json (the reference's own defaults, dumped by
oracle/ref_params_dump.cpp from CmdLineArgs.hpp as it lies) with CommandLineArgs::selfUpdateByPlatform applied in Python.

tests/test_chain_golden.py holds the oracle (CPU suite) and the HIP path (-m gpu) to these planes, bit for bit.
Run from the repo root:  python tests/golden/make_chain_golden.py     (about ten seconds)
"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from uvc_amd import region  # noqa: E402  (apply_platform: plain attribute arithmetic, no library)
from rtr_cases import python_tracks  # noqa: E402
from prep_restatement import PREP32, PREP64, THRES, prep_sets, thres_sets  # noqa: E402
from p2_restatement import update_by_aln  # noqa: E402
from segbias_restatement import SEG_FIELDS  # noqa: E402
from p3_restatement import fragment_pass  # noqa: E402
from p45_restatement import FAM, FI32, FI64, family_passes  # noqa: E402
from gather_restatement import Planes, gather  # noqa: E402
from score_restatement import calc_DPv, calc_qual, sum_DPv  # noqa: E402
from indel_alleles_restatement import DEL, INS, allele_rows, context, majority_alleles  # noqa: E402
from test_call_cpu import germline, record_call  # noqa: E402  (the calling step: output_germline, NLODQ / TLODQ / QUAL / FILTER, main.cpp:990-1168)
from test_gpu_fuzz import weird_region  # noqa: E402

CASES = {
    "chain_illumina_umi": dict(seed=61, n_frag=110, ref_len=410, umi=True, platform=1, normal=0),
    "chain_illumina_plain": dict(seed=62, n_frag=130, ref_len=450, umi=False, platform=1, normal=0),
    "chain_iontorrent_umi_normal": dict(seed=63, n_frag=100, ref_len=390, umi=True, platform=2, normal=1),
    "chain_illumina_umi_deep": dict(seed=64, n_frag=600, ref_len=520, umi=True, platform=1, normal=0),
    "chain_iontorrent_plain": dict(seed=65, n_frag=140, ref_len=430, umi=False, platform=2, normal=0),
}
READ_KEYS = ("pos", "mpos", "isize", "flag", "mapq", "nm", "l_qseq", "seq_off", "cigar_off", "n_cigar", "frag_id", "fam_id", "fam_strand", "fam_dflag", "bases", "quals", "cigars")
VQ_ORDER = ("a1BQf", "a1BQr", "a2BQf", "a2BQr", "bMQ", "bIAQb", "bIADb", "bIDQb", "cIAQf", "cIADf", "cIDQf", "cIAQr", "cIADr", "cIDQr")   # enum of include/uvcgpu.h


def params_for(platform, normal):
    P = types.SimpleNamespace(**json.load(open(os.path.join(ROOT, "tests", "golden", "params_default.json"))))
    region.apply_platform(P, platform, 150, 60)
    P.tumor_vcf_is_provided = normal
    return P


def i32(a):
    a = np.asarray(a).astype(np.int64)
    return (((a + 2 ** 31) % 2 ** 32) - 2 ** 31).astype(np.int32)


def chain_planes(reads, P, platform, normal):
    proton = (platform == 2)
    rtr, baq = python_tracks(reads["refseq"], smax=P.indel_str_repeatsize_max, vmax=P.indel_vntr_repeatsize_max, bq_max=P.indel_BQ_max,
                             slip_rate=P.indel_polymerase_slip_rate, del_to_ins=P.indel_del_to_ins_err_ratio, polymerase_size=P.indel_polymerase_size,
                             str_phred_per_region=P.indel_str_phred_per_region, nonstr_phred_per_base=P.indel_nonSTR_phred_per_base)
    codes = np.array([{"A": 0, "C": 1, "G": 2, "T": 3}.get(c.upper(), 4) for c in reads["refseq"]], dtype=np.int32)
    prep = prep_sets(reads, P, rtr, baq[0], np.append(codes, 4))
    thres, ip = thres_sets(prep, rtr[3], P, is_normal=bool(normal), iontorrent=proton)
    seg, bqsum = update_by_aln(reads, P, rtr, ip, baq[0], baq[1], codes, prep, thres, proton)
    alleles = {}
    frag, vq3 = fragment_pass(reads, P, rtr, ip, baq[0], codes, prep, thres, seg, bqsum, proton, alleles=alleles)
    famp, fi, dup, vq45 = family_passes(reads, P, rtr, ip, baq[0], baq[1], codes, prep, thres, proton=proton, alleles=alleles)
    alleles.setdefault("bq", ({}, {}))
    vq = dict(vq3); vq.update(vq45)
    for name in SEG_FIELDS[34:38]:
        vq[name] = seg[name]
    rtr_after = np.array(rtr, dtype=np.int64); rtr_after[3] = ip       # UVC_F_RTR: the tracks AFTER P1b edited indelphred
    return alleles, {
        "PREP32": np.stack([i32(prep[n]) for n in PREP32]), "PREP64": np.stack([np.asarray(prep[n], dtype=np.int64) for n in PREP64]),
        "THRES": np.stack([i32(thres[n]) for n in THRES]),
        "SEG32": np.stack([i32(seg[n]) for n in SEG_FIELDS[:30]]), "SEG64": np.stack([np.asarray(seg[n], dtype=np.int64) for n in SEG_FIELDS[30:34]]),
        "VQ": np.stack([i32(vq[n]) for n in VQ_ORDER]), "BQSUM": i32(bqsum),
        "FRAG": np.stack([np.stack([i32(frag[st][f]) for f in range(3)]) for st in range(2)]),
        "FAM": np.stack([np.stack([i32(famp[st][k]) for k in range(len(FAM))]) for st in range(2)]),
        "FAMINFO32": np.stack([i32(fi[n]) for n in FI32]), "FAMINFO64": np.stack([np.asarray(fi[n], dtype=np.int64) for n in FI64]),
        "DUPLEX": np.stack([i32(dup[k]) for k in range(2)]),
        "RTR": i32(rtr_after), "BAQ": np.stack([np.asarray(baq[0], dtype=np.int64), np.asarray(baq[1], dtype=np.int64)]),
    }


GATHERED = ("refsymbol", "DP", "bDP", "c2DP", "c2AD", "bDPa", "cDP0a", "gapSa_len", "a2BQf", "a2BQr", "aBQ", "aBQQ", "bMQ")
MARGIN = 3   # positions next to the region's ends are left out: which of them a library scores is its caller's business (main.cpp:608, 643)
LINK_SYMBOLS = (6, 12, 11, 10, 9, 8, 7, 13)   # SYMBOL_TYPE_TO_SYMBOLS[LINK_SYMBOL], main_conversion.hpp:399


def score_group(rows, pl, x, group, extra, P, all_out=True):
    """calc_DPv -> sum_DPv -> calc_qual -> output_germline -> the per-record call values for the records of one (position, symbol type) group,
    appended to `rows` (field -> list)."""
    def put(k, v): rows.setdefault(k, []).append(int(v))
    outs = [calc_DPv(d, P) for d in group]
    symbols = [int(d["symbol"]) for d in group]
    sums = sum_DPv(outs, symbols)
    quals = [calc_qual(d, o, sums, extra, P) for d, o in zip(group, outs)]
    refsym = int(group[0]["refsymbol"])
    crecs = [dict(symbol=int(d["symbol"]), gVQ1=q["gVQ1"], CONTQ=q["CONTQ"], cDP0a=d["cDP0a"], cDP1v=o["cDP1v"], cDP1x=o["cDP1x"], cDP2x=o["cDP2x"], CDP1x0=sums[0][2], CDP2x0=sums[0][5],
                  cVQ1=q["cVQ1"], cPCQ1=q["cPCQ1"], cVQ2=q["cVQ2"], cPCQ2=q["cPCQ2"], bNMQ=o["bNMQ"], bDP=d["bDP"], DP=d["DP"], gapSa_len=d["gapSa_len"])
             for d, o, q in zip(group, outs, quals)]
    tprov = bool(P.tumor_vcf_is_provided)
    g = germline(P, refsym, crecs, tprov)
    bd = lambda s_: pl.frag(0, "bDP", s_, x) + pl.frag(1, "bDP", s_, x)
    type_syms = range(6) if refsym <= 5 else range(6, 14)
    abq2_tot = int(np.int32(sum(pl.seg("aBQ2", s_, x) for s_ in type_syms)))
    for d, o, q, r in zip(group, outs, quals, crecs):
        put("refpos", d["refpos"]); put("symbol", d["symbol"])
        for k in GATHERED: put(k, d[k])
        put("nPF0", o["nPF"][0]); put("nPF1", o["nPF"][1])
        for k in ("AD", "bAD", "bNMa", "bNMb", "bNMQ", "FTS", "tier2", "cDP1v", "cDP1w", "cDP1x", "cDP2v", "cDP2w", "cDP2x"): put(k, o[k])
        for i, v in enumerate(o["nNFA"]): put("nNFA%d" % i, v)
        for i, v in enumerate(o["nAFA"]): put("nAFA%d" % i, v)
        for i, v in enumerate(o["nBCFA"]): put("nBCFA%d" % i, v)
        pct = [min(max(int(v), 0), 255) for v in o["FTSpct"]] + [0]          # 19 percentages, four per record field
        for w in range(5): put("FTSpct%d" % w, sum(pct[4 * w + b] << (8 * b) for b in range(4)))
        for t, k in enumerate(("CDP1v", "CDP1w", "CDP1x", "CDP2v", "CDP2w", "CDP2x")):
            put(k + "0", sums[0][t]); put(k + "1", sums[1][t])
        for k, v in q.items(): put(k, v)
        put("vNLODQ", g["ret"])
        for i, v in enumerate(g["GL4"]): put("GL4_%d" % i, v)
        for i, v in enumerate(g["GST"]): put("GST%d" % i, v)
        # per-record call values: meaningful where the record is written (the test compares them where the library says `out`)
        tk = d.get("tk")
        if tprov:     # what the T/N arm of main.cpp:1081-1147 reads besides the record (does_fmt_imply_short_frag, the near-deletion depth)
            r["short_frag"] = (pl.prep("a_LI", x) + pl.prep("a_RI", x)) < (pl.prep("a_LIDP", x) + pl.prep("a_RIDP", x)) * int(P.lib_wgs_min_avg_fraglen)
            r["APDP0"], r["APDP2"] = pl.prep("a_dp", x), pl.prep("a_near_del_dp", x)
        c = record_call(P, r, g, bd(refsym), bd(r["symbol"]), pl.seg("aBQ2", r["symbol"], x), abq2_tot, tk, crecs, all_out, False, refsym)
        put("has_key", 1 if tk is not None else 0)
        for k in ("vHGQ", "NLODQ", "NLODV", "TLODQ", "SomaticQ", "FILTER", "keep"): put("call__" + k, c[k])
        for i in range(4): put("call__TNBQF%d" % i, c["TNBQF"][i]); put("call__TNCQF%d" % i, c["TNCQF"][i])
        rows.setdefault("call__QUAL", []).append(float(c["QUAL"]))


def chain_records(planes, reads, P, rows_alleles, all_out=True):
    """The scored records of every symbol at every inner position (tumor-only): gather (BcfFormat_symboltype_init / _symbol_init /
    fill_symbol_VQ_fmts) -> calc_DPv -> sum_DPv over the records of the (position, symbol type) group -> calc_qual -> the calling step, all by
    the independent restatements, from the chain's own planes and allele-keyed maps.  A base symbol's, LINK_M's or LINK_NN's record takes
    bDPa / cDP0a from its own depths and has no InDel string (main.cpp:810-817, 897-903); an InDel symbol has one record per majority allele
    (fill_by_indel_info + indel_get_majority, main.cpp:853-895), "<L..>" with zero depths when it has none.
    all_out = False: the default gate of main.cpp:835-841 -- an ALT symbol needs min_altdp_thres fragments, the REF symbol as many non-REF
    fragments of its type at the position; the sums of sum_DPv then run over the records that passed."""
    pl = Planes(lambda g: planes[g])
    npos, beg, refseq = planes["RTR"].shape[1], int(reads["beg"]), reads["refseq"]
    codes = np.array([{"A": 0, "C": 1, "G": 2, "T": 3}.get(c.upper(), 4) for c in refseq], dtype=np.int32)
    rows = {}
    for x in range(MARGIN, npos - 1 - MARGIN):
        for stype, symbols in ((0, range(6)), (1, LINK_SYMBOLS)):
            group = []
            ins_c = del_c = ins1_c = del1_c = 0
            type_bdp = sum(pl.frag(st, "bDP", s_, x) for st in (0, 1) for s_ in symbols)          # bDPcDP[0] of BcfFormat_symboltype_init
            for sym in symbols:
                base = gather(pl, x, sym, codes, P, False, npos)
                refsym = int(base["refsymbol"])
                bdepth = base["bDPf"] + base["bDPr"]
                cdepth = max(base["cDP1f"], base["cDP12f"]) + max(base["cDP1r"], base["cDP12r"])
                if sym in INS:
                    ins_c += cdepth; ins1_c += cdepth if sym == 12 else 0
                if sym in DEL:
                    del_c += cdepth; del1_c += cdepth if sym == 9 else 0
                if not all_out:
                    ref_bdepth = pl.frag(0, "bDP", refsym, x) + pl.frag(1, "bDP", refsym, x)
                    if (refsym != sym and bdepth < int(P.min_altdp_thres)) or (refsym == sym and type_bdp - ref_bdepth < int(P.min_altdp_thres)):
                        continue
                if sym in INS or sym in DEL:
                    alls = majority_alleles(rows_alleles, beg + x, sym, (base["bDPf"], base["bDPr"]))
                else:
                    alls = [(bdepth, cdepth, "")]
                for b, c, text in alls:
                    d = dict(base)
                    d.update(bDPa=b, cDP0a=c, gapSa_len=len(text), refpos=beg + x, tki_tier2=0, tpfa_dpv=-1.0, tpfa_qual=-1.0)
                    group.append(d)
            if not group:
                continue
            # the InDel depths and the repeat context belong to the LINK position of the same loop iteration (main.cpp:608-640); base symbols do not read them
            extra = (ins_c, del_c, ins1_c, del1_c) + context(refseq, x, int(P.indel_str_repeatsize_max)) if stype == 1 else (0, 0, 0, 0, 0, 0)
            score_group(rows, pl, x, group, extra, P, all_out)
    return {k: np.array(v, dtype=(np.float64 if k == "call__QUAL" else np.int64)) for k, v in rows.items()}


KEY_FIELDS = ("refpos", "symbol", "cDP1x", "CDP1x", "bDP", "BDP", "tier2", "indel_len", "cVQ1", "cPCQ1", "cDP2x", "CDP2x", "cVQ2", "cPCQ2", "bNMQ", "vHGQ", "tDP")   # UvcTumorKey


def tumor_keys_from_chain(gated):
    """A tumor-sample channel made from the chain's own tumor-only records under the default gate: every second record becomes a key
    (one per (position, symbol); the InDel length is the record's)."""
    keys, seen = [], set()
    for i in range(0, len(gated["refpos"]), 2):
        pos, sym = int(gated["refpos"][i]), int(gated["symbol"][i])
        if (pos, sym) in seen:
            continue
        seen.add((pos, sym))
        g = lambda k: int(gated[k][i])
        keys.append((pos, sym, g("cDP1x"), g("CDP1x0"), g("bAD"), g("bDP"), g("tier2") | (i // 2 % 2), g("gapSa_len") if sym in INS or sym in DEL else 0,
                     g("cVQ1"), g("cPCQ1"), g("cDP2x"), g("CDP2x0"), g("cVQ2"), g("cPCQ2"), g("bNMQ"), g("call__vHGQ"), g("DP") * (1 + 3 * (i % 3 == 0))))
    return sorted(keys, key=lambda k: (k[0], k[1]))


def chain_records_normal(planes, reads, P, rows_alleles, keys):
    """The normal sample of a T/N pair (IS_PROVIDED(vcf_tumor_fname)): only positions that carry a tumor key are scored, every symbol of both
    symbol types there (main.cpp:529-538, 843-845); a (position, symbol) with a key has one record per key with the key's tier-2 flag, tpfa and --
    for an InDel -- the key's string length (main.cpp:849-903, 928-934, 985-986), the others are scored with tpfa = -1."""
    pl = Planes(lambda g: planes[g])
    npos, beg, refseq = planes["RTR"].shape[1], int(reads["beg"]), reads["refseq"]
    codes = np.array([{"A": 0, "C": 1, "G": 2, "T": 3}.get(c.upper(), 4) for c in refseq], dtype=np.int32)
    by = {}
    for k in keys:
        by[(k[0], k[1])] = dict(zip(KEY_FIELDS, k))
    rows = {}
    for pos in sorted({k[0] for k in keys}):
        x = pos - beg
        if not (MARGIN <= x < npos - 1 - MARGIN):
            continue
        for stype, symbols in ((0, range(6)), (1, LINK_SYMBOLS)):
            group = []
            ins_c = del_c = ins1_c = del1_c = 0
            for sym in symbols:
                base = gather(pl, x, sym, codes, P, False, npos)
                bdepth = base["bDPf"] + base["bDPr"]
                cdepth = max(base["cDP1f"], base["cDP12f"]) + max(base["cDP1r"], base["cDP12r"])
                if sym in INS:
                    ins_c += cdepth; ins1_c += cdepth if sym == 12 else 0
                if sym in DEL:
                    del_c += cdepth; del1_c += cdepth if sym == 9 else 0
                tk = by.get((pos, sym))
                if tk is not None:
                    alls = [(bdepth, cdepth, "x" * tk["indel_len"] if (sym in INS or sym in DEL) else "")]
                elif sym in INS or sym in DEL:
                    alls = majority_alleles(rows_alleles, pos, sym, (base["bDPf"], base["bDPr"]))
                else:
                    alls = [(bdepth, cdepth, "")]
                for b, c, text in alls:
                    d = dict(base)
                    d.update(bDPa=b, cDP0a=c, gapSa_len=len(text), refpos=pos, tki_tier2=(tk["tier2"] if tk else 0),
                             tpfa_dpv=((tk["cDP1x"] + 1.0) / (tk["CDP1x"] + 2.0) if tk else -1.0), tpfa_qual=((tk["bDP"] + 0.5) / (tk["BDP"] + 1.0) if tk else -1.0))
                    if tk is not None:
                        d["tk"] = tk
                    group.append(d)
            extra = (ins_c, del_c, ins1_c, del1_c) + context(refseq, x, int(P.indel_str_repeatsize_max)) if stype == 1 else (0, 0, 0, 0, 0, 0)
            score_group(rows, pl, x, group, extra, P, False)
    return {k: np.array(v, dtype=(np.float64 if k == "call__QUAL" else np.int64)) for k, v in rows.items()}


if __name__ == "__main__":
    for name, kw in CASES.items():
        reads = weird_region(kw["seed"], n_frag=kw["n_frag"], ref_len=kw["ref_len"], umi=kw["umi"])
        P = params_for(kw["platform"], kw["normal"])
        alleles, planes = chain_planes(reads, P, kw["platform"], kw["normal"])
        out = {"planes__" + g: v for g, v in planes.items()}
        arows = allele_rows(alleles, reads["refseq"], int(reads["beg"]))
        akeys = sorted(arows)
        out["alleles__rows"] = np.array([[k[0], k[1], k[2], len(k[3])] + list(arows[k]) for k in akeys], dtype=np.int64).reshape(len(akeys), 8)   # refpos symbol strand len bAD1 cAD1 c2AD c2dAD
        out["alleles__text"] = np.array(";".join(k[3] for k in akeys))
        recs = chain_records(planes, reads, P, arows) if not kw["normal"] else {"refpos": np.zeros(0, dtype=np.int64)}   # a normal sample only scores what its tumor's keys name
        out.update({"records__" + k: v for k, v in recs.items()})
        gated = chain_records(planes, reads, P, arows, all_out=False) if not kw["normal"] else {"refpos": np.zeros(0, dtype=np.int64)}
        out.update({"gated__" + k: v for k, v in gated.items()})
        if kw["normal"]:   # the tumor pass of the same reads (tumor parameters) gives the keys, the normal pass is scored on them
            Pt = params_for(kw["platform"], 0)
            al_t, planes_t = chain_planes(reads, Pt, kw["platform"], 0)
            keys = tumor_keys_from_chain(chain_records(planes_t, reads, Pt, allele_rows(al_t, reads["refseq"], int(reads["beg"])), all_out=False))
            nrecs = chain_records_normal(planes, reads, P, arows, keys)
            out["normal__keys"] = np.array(keys, dtype=np.int64)
            out.update({"normal__" + k: v for k, v in nrecs.items()})
            print("   normal sample:", len(keys), "tumor keys,", len(nrecs["refpos"]), "records")
        for k in READ_KEYS:
            out["reads__" + k] = np.asarray(reads[k])
        out["meta"] = np.array(json.dumps(dict(tid=int(reads["tid"]), beg=int(reads["beg"]), end=int(reads["end"]), refseq=reads["refseq"], n_reads=int(reads["n_reads"]),
                                               n_fams=int(reads["n_fams"]), platform=kw["platform"], normal=kw["normal"])))
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, "KiB", {g: (v.shape, int(np.abs(v.astype(np.float64)).sum())) for g, v in planes.items() if g in ("SEG32", "FAM", "DUPLEX")},
              len(recs["refpos"]), "records x", len(recs), "fields,", len(gated["refpos"]), "under the default gate,", len(akeys), "allele rows")
