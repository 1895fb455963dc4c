"""Pins the oracle's math primitives against the reference's own known answers:
  * the compile-time static_asserts of main_conversion.hpp:205-209 and :251-254,
  * the generating formulas the reference quotes next to its lookup tables (main.hpp:762, main_conversion.hpp:926)."""
import ctypes as C
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def m(oracle_lib):
    d = oracle_lib.dll
    d.uvc_oracle_calc_binom_10log10_likeratio.restype = C.c_double
    d.uvc_oracle_calc_binom_10log10_likeratio.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
    d.uvc_oracle_prob2odds.restype = C.c_double; d.uvc_oracle_prob2odds.argtypes = [C.c_double]
    d.uvc_oracle_odds2prob.restype = C.c_double; d.uvc_oracle_odds2prob.argtypes = [C.c_double]
    d.uvc_oracle_indel_len_rusize_phred.restype = C.c_int32; d.uvc_oracle_indel_len_rusize_phred.argtypes = [C.c_int32, C.c_int32]
    d.uvc_oracle_indel_phred.restype = C.c_int32; d.uvc_oracle_indel_phred.argtypes = [C.c_double, C.c_int32, C.c_int32]
    d.uvc_oracle_dp4_to_pcFA.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int] + [C.c_double] * 11
    d.uvc_oracle_infer_max_qual.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32]
    return d


def test_reference_static_asserts_binom(m):
    f = m.uvc_oracle_calc_binom_10log10_likeratio
    assert abs(f(0.1, 10, 90, 0, 0)) < 1e-4                 # main_conversion.hpp:251
    assert 763 < f(0.1, 90, 10, 0, 0) < 764                 # :252-253
    assert abs(f(0.1, 1, 99, 0, 0)) < 1e-4                  # :254
    # the closed form quoted in the reference comment: 10/log(10) * (90*log(9)+10*log(1/9))
    assert f(0.1, 90, 10, 0, 0) == pytest.approx(10 / math.log(10) * (90 * math.log(9) + 10 * math.log(1 / 9)), rel=1e-9)


def test_reference_static_asserts_odds(m):
    assert 0.99 < m.uvc_oracle_prob2odds(m.uvc_oracle_odds2prob(1.0)) < 1.01     # main_conversion.hpp:205-206
    assert 0.65 < m.uvc_oracle_odds2prob(m.uvc_oracle_prob2odds(0.66)) < 0.67    # :208-209


def test_indel_len_table_matches_the_quoted_formula(m):
    # main.hpp:762: "derived from the python code: int(round(10.0/log(10.0)*log(i)))", i = 1..18, unit size 1
    for i in range(1, 19):
        assert m.uvc_oracle_indel_len_rusize_phred(i, 1) == int(round(10.0 / math.log(10.0) * math.log(i)))
    assert m.uvc_oracle_indel_len_rusize_phred(40, 1) == 13          # clamps at 18 units
    assert m.uvc_oracle_indel_len_rusize_phred(6, 3) == 3            # 2 units of size 3
    assert m.uvc_oracle_indel_len_rusize_phred(7, 3) == 8            # not a multiple: indexed by the length


def test_indel_phred_against_closed_form(m):
    # indel_phred (main.hpp:794-801) = floor(-10 log10((1-eps)/(slips+1))): the Phred-scaled DECREMENT caused by polymerase
    # slippage, so it grows with the tract length; recomputed here independently in python
    vals = [m.uvc_oracle_indel_phred(8.0, 1, n) for n in range(1, 30)]
    assert all(a <= b for a, b in zip(vals, vals[1:]))
    exp = [math.floor(-10 * math.log10((1 - 2.220446049250313e-16) / (math.log1p(math.exp(n - 8)) * 8.0 + 1))) for n in range(1, 30)]
    assert vals == exp
    assert m.uvc_oracle_indel_phred(8.0, 2, 40) == math.floor(-10 * math.log10((1 - 2.220446049250313e-16) / ((80 - 8) * 8.0 / 4 + 1)))   # region > 64: linear branch


def test_dp4_to_pcFA_basic_properties(m):
    out = (C.c_double * 2)()
    # no bias: pass and fail fractions equal -> returns the pass fraction, second value = pooled fraction
    m.uvc_oracle_dp4_to_pcFA(out, 0, 0, 1.0, 10, 10, 100, 100, 3.0, math.log(501), -1, -1, 0.5, 1.0)
    assert out[0] == pytest.approx(10.5 / 101) and out[1] == pytest.approx(21.0 / 202)
    # strong bias: the ALT is only in the "fail" class -> the pass-class fraction bounds the result from below
    m.uvc_oracle_dp4_to_pcFA(out, 0, 0, 1.0, 0, 50, 1000, 1000, 3.0, math.log(501), -1, -1, 0.5, 1.0)
    assert 0.5 / 1001 <= out[0] < 50.5 / 1001


def test_infer_max_qual(m):
    out = (C.c_int32 * 3)()
    distr = (C.c_int32 * 16)(*([30] + [0] * 15))
    m.uvc_oracle_infer_max_qual(out, 37, 1, distr, 30)
    assert list(out) == [30 * 37, 30, 37]                            # all fragments in the top bucket, AD == DP
    distr = (C.c_int32 * 16)(*([0, 0, 5] + [0] * 13))
    m.uvc_oracle_infer_max_qual(out, 40, 1, distr, 50)
    exp = int(5 * (38 - 10 / math.log(10) * math.log(50 / 5 + np.finfo(float).eps)))
    assert list(out) == [exp, 5, 38]
