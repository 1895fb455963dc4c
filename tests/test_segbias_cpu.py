"""dealwith_segbias<isGap> (main.hpp:1360-1595), the function behind 34 counters of every (position, symbol): the oracle's restatement
against an independent Python restatement written from the reference text (tests/segbias_restatement.py), call by call on fuzzed
arguments -- both template arms, every orientation / pairing flag combination, thresholds and BAQ arrays at their decision points,
the amplicon / UMI / normal-filter parameter arms, short-read micro-adjustment.  VERDICT r2 "missing" #2."""
import ctypes as C

import numpy as np
import pytest

from uvc_amd import _ffi, region
from rtr_cases import fuzz_reference
from segbias_restatement import SEG_FIELDS, dealwith_segbias

THRES = "aLPxT aRPxT aLI1T aLI2T aRI1T aRI2T aLI1t aLI2t aRI1t aRI2t aLP1t aLP2t aRP1t aRP2t aLB1t aLB2t aRB1t aRB2t".split()   # UVC_T_* order


def one_case(rng, lib, R, baq, beg, n, P):
    isGap = int(rng.integers(0, 2))
    pos = int(beg + rng.integers(0, n - 160))
    span = int(rng.integers(1, 152))
    endpos = pos + span
    rpos = int(rng.integers(pos, endpos))
    if rng.random() < 0.3:
        rpos = pos if rng.random() < 0.5 else endpos - 1
    isize = int(rng.choice([0, 0, span, span + 30, 400, -400, 2500, -3000, int(rng.integers(-700, 700))]))
    mpos = int(pos + rng.integers(-500, 500)) if rng.random() < 0.8 else pos
    flag = 0
    for bit, p_ in ((0x1, 0.8), (0x8, 0.2), (0x10, 0.5), (0x20, 0.5), (0x40, 0.5), (0x80, 0.5)):
        if rng.random() < p_:
            flag |= bit
    bq = int(rng.choice([0, 2, 11, 19, 20, 21, 24, 25, 29, 30, 31, 37, 41, int(rng.integers(0, 60))]))
    args = [isGap, bq, rpos, int(rng.integers(0, 14)), pos, endpos, mpos, isize, flag, int(rng.integers(0, 61)),
            int(rng.choice([0, 5, 20, 21, 40, 99, int(rng.integers(0, 300))])), int(rng.choice([0, 5, 20, 21, 40, 99, int(rng.integers(0, 300))])),
            int(rng.choice([0, 1, 2])), int(rng.choice([0, 1, 2, 5, 17, 40])), int(rng.choice([0, 3, 4, 5, 6, 50, 99999])), int(rng.integers(0, 16)), int(rng.choice([0, 0, 1, 2]))]
    seg_l, seg_r = rpos - pos + 1, endpos - rpos
    def near(v):   # a threshold at, just below or just above the value it is compared with, or anywhere
        return int(rng.choice([v - 1, v, v + 1, int(rng.integers(0, 300))]))
    th = {k: near(int(rng.choice([seg_l, seg_r, 10, 100, 2000]))) for k in THRES}
    thres = (C.c_int32 * 18)(*[th[k] for k in THRES])
    out = (C.c_int64 * len(SEG_FIELDS))()
    a = (C.c_int32 * 17)(*args)
    rc = lib.dll.uvc_oracle_test_segbias(R.h, a, thres, out)
    assert rc == 0, lib.last_error()
    aln = dict(pos=pos, endpos=endpos, mpos=mpos, isize=isize, flag=flag, qual=args[9])
    want = dealwith_segbias(bool(isGap), bq, rpos, th, aln, args[10], args[11], baq[0], baq[1], beg, args[12], args[13], args[14], args[15], args[16], P)
    got = dict(zip(SEG_FIELDS, list(out)))
    assert got == want, (args, th, {k: (got[k], want[k]) for k in got if got[k] != want[k]})


@pytest.mark.parametrize("arm", ["default", "short_reads", "amplicon_primer", "tn_primer_filter", "iontorrent_like"])
def test_segbias_against_the_independent_restatement(arm, oracle_lib):
    lib = oracle_lib
    lib.dll.uvc_oracle_test_segbias.restype = C.c_int
    lib.dll.uvc_oracle_test_segbias.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    P = region.default_params(lib)
    if arm == "short_reads":
        P.central_readlen = 40                          # below microadjust_median_readlen_thres: the per-base BAQ floor arm
    elif arm == "amplicon_primer":
        P.primerlen = 20; P.primer_flag = 0; P.primerlen2 = 15
    elif arm == "tn_primer_filter":
        P.primerlen = 20; P.primer_flag = 1; P.tn_is_paired = 1
    elif arm == "iontorrent_like":
        P.bias_thres_PFBQ1 = 7; P.bias_thres_PFBQ2 = 12; P.bias_thres_highBQ = 7; P.bias_thres_interfering_indel = 50; P.microadjust_nobias_pos_indel_maxlen = 3
    n, beg = 3000, 500000
    ref = fuzz_reference(5, n)                           # repeat-rich: the BAQ prefix sums have both kinds of increments
    R = region.Region(lib, P, 0, beg, beg + n, ref)
    baq = R.fetch("BAQ")
    rng = np.random.default_rng(["default", "short_reads", "amplicon_primer", "tn_primer_filter", "iontorrent_like"].index(arm) + 77)
    for _ in range(4000):
        one_case(rng, lib, R, baq, beg, n, P)
    R.close()
