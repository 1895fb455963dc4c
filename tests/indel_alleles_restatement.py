"""Independent restatement of the InDel allele side of the scoring, written from the reference text: the rows fill_by_indel_info reads
(instcode.hpp:8-95 through main.hpp:5351-5378), indel_get_majority (main.hpp:5406-5455), the record enumeration of one LINK group
(main.cpp:641-660, 804-904) and indelpos_to_context (main.hpp:723-755).  Input: the allele-keyed maps collected by the restated fragment and
family passes (tests/p3_restatement.py, tests/p45_restatement.py: `alleles`).  Test infrastructure; nothing here is shipped."""
from rtr_cases import _more_str

LINK_M, LINK_NN = 6, 13
INS, DEL = (10, 11, 12), (7, 8, 9)
DESC = {7: "<LD3P>", 8: "<LD2>", 9: "<LD1>", 10: "<LI3P>", 11: "<LI2>", 12: "<LI1>"}      # SYMBOL_TO_DESC_ARR, main_conversion.hpp:336-346


def allele_string(sym, key, refseq, x):
    """The inserted text, or the deleted reference bases refchars.substr(refpos - begin, length) (instcode.hpp:45-49)."""
    return key if sym in INS else refseq[x:x + key]


def allele_rows(alleles, refseq, beg):
    """Every row fill_by_indel_info would read: {(refpos, symbol, strand, string): (bAD1, cAD1, c2AD, c2dAD)}; rows exist where the fragment
    map has the allele (instcode.hpp:42-58) and its string is not empty."""
    rows = {}
    for strand in (0, 1):
        for (sym, pos), d in alleles["bq"][strand].items():
            for key, bq in d.items():
                text = allele_string(sym, key, refseq, pos - beg)
                if not text:
                    continue
                get = lambda name: alleles[name][strand].get((sym, pos), {}).get(key, 0)
                rows[(pos, sym, strand, text)] = (bq, get("fq"), get("c2"), get("c2d"))
    return rows


def majority_alleles(rows, refpos, sym, frag_bdp):
    """fill_by_indel_info on the strands with fragment depth (main.cpp:853-866) + indel_get_majority: [(bAD1, cAD1, string)] in record order --
    at least a quarter of the best fragment support, by bAD1^2 * length descending; ("<L..>", 0, 0) when there is no row at all."""
    merged = {}
    for strand in (0, 1):
        if frag_bdp[strand] <= 0:
            continue
        for (p, s, st, text), v in rows.items():
            if (p, s, st) == (refpos, sym, strand):
                b, c = merged.get(text, (0, 0))
                merged[text] = (b + v[0], c + v[1])
    if not merged:
        return [(0, 0, DESC[sym])]
    best = max(v[0] for v in merged.values())
    kept = [(v[0], v[1], text) for text, v in sorted(merged.items()) if v[0] >= (best + 3) // 4]
    # std::sort over reverse iterators with "x < y iff key(x) < key(y)": descending by key; equal keys stay unordered in the reference
    # (callers of this restatement compare such records as a set)
    kept.sort(key=lambda t: -(t[0] * t[0] * len(t[2])))
    return kept


def context(refseq, refidx, smax):
    """indelpos_to_context, main.hpp:723-755 -> (repeat unit length, repeat count)."""
    n = len(refseq)
    if refidx >= n:
        return 0, 0
    best_rs, best_rn = 0, 0
    for rs in range(1, smax + 1):
        q = refidx
        while q + rs < n and refseq[q] == refseq[q + rs]:
            q += 1
        rn = (q - refidx) // rs + 1
        if _more_str(rs, rn, best_rs, best_rn, smax):
            best_rs, best_rn = rs, rn
    return best_rs, best_rn
