"""BcfFormat_symbol_calc_DPv + BcfFormat_symbol_sum_DPv + BcfFormat_symbol_calc_qual (main.hpp:4253-5343): the oracle's restatement against an independent Python
restatement written from the reference text (tests/score_restatement.py), record by record, on the inputs the oracle itself gathered
(test hook uvc_oracle_score_trace) -- all-out scoring of fuzzed reads, UMI families, the IonTorrent arm and the normal sample of a T/N
pair (is_rescued arms).  VERDICT r2 "missing" #2."""
import ctypes as C

import numpy as np
import pytest

from uvc_amd import _ffi, region, synth
from score_restatement import calc_DPv, calc_qual, sum_DPv
from test_gpu_fuzz import weird_region
from util import run_region


def traced_score(lib, R, all_out=False, tumor_keys=None, is_amplicon=False):
    """(records dict, list of per-record input dicts) of one score call."""
    lib.dll.uvc_oracle_score_trace_names.restype = C.c_char_p
    names = lib.dll.uvc_oracle_score_trace_names().decode().split(";")[:-1]
    rec = R.score(all_out=all_out, tumor_keys=tumor_keys, is_amplicon=is_amplicon)
    n = len(rec["refpos"])
    req, _keep = R.make_request(all_out, -1, -1, is_amplicon, None, tumor_keys, False, False, 0, kept_only=False)
    buf = np.zeros(n * len(names), dtype=np.float64)
    buf2 = np.zeros(n * 6, dtype=np.float64)
    nv = C.c_int64()
    fn = lib.dll.uvc_oracle_score_trace
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
    assert fn(R.h, C.byref(req), buf.ctypes.data, buf.size, C.byref(nv), buf2.ctypes.data) == 0, lib.last_error()
    assert nv.value == buf.size, (nv.value, buf.size)
    rows = buf.reshape(n, len(names))
    return rec, [dict(zip(names, r)) for r in rows], buf2.reshape(n, 6)


def check(lib, R, P, **kw):
    rec, ins, extra = traced_score(lib, R, **kw)
    n = len(ins)
    assert n > 0
    outs = [calc_DPv(d, P) for d in ins]
    bad = []
    for i, o in enumerate(outs):
        got = {"nPF": [int(rec["nPF0"][i]), int(rec["nPF1"][i])], "bNMa": int(rec["bNMa"][i]), "bNMb": int(rec["bNMb"][i]), "bNMQ": int(rec["bNMQ"][i]),
               "nNFA": [int(rec["nNFA%d" % k][i]) for k in range(6)], "nAFA": [int(rec["nAFA%d" % k][i]) for k in range(9)],
               "nBCFA": [int(rec["nBCFA%d" % k][i]) for k in range(10)], "FTS": int(rec["FTS"][i]), "tier2": int(rec["tier2"][i]),
               "AD": int(rec["AD"][i]), "bAD": int(rec["bAD"][i])}
        for k in ("cDP1v", "cDP1w", "cDP1x", "cDP2v", "cDP2w", "cDP2x"):
            got[k] = int(rec[k][i])
        pct = []
        for b in range(19):
            pct.append((int(rec["FTSpct%d" % (b // 4)][i]) >> (8 * (b % 4))) & 0xFF)
        got["FTSpct"] = pct
        want = dict(o); want["FTSpct"] = [min(max(v, 0), 255) for v in o["FTSpct"]]
        if got != want:
            bad.append((i, int(rec["refpos"][i]), int(rec["symbol"][i]), {k: (got[k], want[k]) for k in got if got[k] != want[k]}))
    assert not bad, (len(bad), n, bad[:3])
    # sum_DPv per (zerobased_pos, symbol type) group: consecutive records with the same (refpos, type)
    keys = [(int(p), int(s) > 5) for p, s in zip(rec["refpos"], rec["symbol"])]
    i = 0
    while i < n:
        j = i
        while j < n and keys[j] == keys[i]:
            j += 1
        s1, s2 = sum_DPv(outs[i:j], [int(s) for s in rec["symbol"][i:j]])
        for q in range(i, j):
            for t, k in enumerate(("CDP1v", "CDP1w", "CDP1x", "CDP2v", "CDP2w", "CDP2x")):
                assert (int(rec[k + "0"][q]), int(rec[k + "1"][q])) == (s1[t], s2[t]), (q, k)
            # BcfFormat_symbol_calc_qual on top of the restated calc_DPv / sum_DPv results
            want = calc_qual(ins[q], outs[q], (s1, s2), extra[q], P)
            got = {k: int(rec[k][q]) for k in want}
            if got != want:
                bad.append((q, int(rec["refpos"][q]), int(rec["symbol"][q]), {k: (got[k], want[k]) for k in got if got[k] != want[k]}))
        i = j
    assert not bad, ("calc_qual", len(bad), n, bad[:4])
    return n


@pytest.mark.parametrize("case", ["fuzz_illumina", "fuzz_umi", "fuzz_iontorrent", "synth_umi_default_gate", "synth_tn_normal"])
def test_calc_DPv_against_the_independent_restatement(case, oracle_lib):
    lib = oracle_lib
    if case.startswith("fuzz"):
        n = 0
        for seed in ((0, 4, 8) if case == "fuzz_illumina" else (2, 5) if case == "fuzz_umi" else (3, 7)):
            reads = weird_region(seed, umi=(case == "fuzz_umi"))
            platform = 2 if case == "fuzz_iontorrent" else 1
            P = region.default_params(lib, platform=platform)
            R = run_region(lib, reads, params=P)
            n += check(lib, R, P, all_out=True)
        assert n > 15000
    elif case == "synth_umi_default_gate":
        reads = synth.generate_region(seed=11, region_len=2000, depth=400, umi=True)
        P = region.default_params(lib)
        R = run_region(lib, reads, params=P)
        assert check(lib, R, P, all_out=False) > 100
    else:
        from test_gpu_parity import tumor_keys_from
        reads = synth.generate_region(seed=12345, region_len=6000, depth=60)
        keys = tumor_keys_from(run_region(lib, reads).score(all_out=False))
        P = region.default_params(lib)
        P.tumor_vcf_is_provided = 1
        R = run_region(lib, reads, params=P)
        assert check(lib, R, P, all_out=False, tumor_keys=keys) > 50
