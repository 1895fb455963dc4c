"""Row a3 on the CPU: the oracle's refstring2repeatvec / BAQ prefix sums (oracle_accumulate.cpp) against the independent Python
restatement of tests/rtr_cases.py (written from main.hpp:699-721, 794-874 and main.cpp:400-429) on fuzzed references.  The reference's
own hot-path headers need htslib and cannot be compiled here, so this is the second opinion the oracle gets."""
import numpy as np
import pytest

from uvc_amd import region
from rtr_cases import EDGE_REFERENCES, fuzz_reference, python_tracks


def oracle_tracks(lib, ref, params=None):
    p = params if params is not None else region.default_params(lib)
    R = region.Region(lib, p, 0, 100000, 100000 + len(ref), ref)
    rtr, baq = R.fetch("RTR"), R.fetch("BAQ")
    R.close()
    rtr[0] -= 0; return rtr, baq


def check(lib, ref, **kw):
    rtr, baq = oracle_tracks(lib, ref, kw.pop("params", None))
    prtr, pbaq = python_tracks(ref, **kw)
    for f, name in enumerate(("begpos", "tracklen", "unitlen", "indelphred", "anyTR_begpos", "anyTR_tracklen", "anyTR_unitlen")):
        bad = np.flatnonzero(rtr[f] != prtr[f])
        assert bad.size == 0, (name, len(ref), int(bad[0]), int(rtr[f][bad[0]]), int(prtr[f][bad[0]]), ref[max(0, bad[0] - 20): bad[0] + 20])
    assert np.array_equal(baq, pbaq)


@pytest.mark.parametrize("i", range(len(EDGE_REFERENCES)))
def test_edge_references(i, oracle_lib):
    check(oracle_lib, EDGE_REFERENCES[i])


@pytest.mark.parametrize("seed", range(12))
def test_fuzzed_references(seed, oracle_lib):
    n = [37, 300, 1024, 2048, 3073, 5000, 7000, 9000][seed % 8]
    check(oracle_lib, fuzz_reference(seed, n))


def test_other_parameters(oracle_lib):
    p = region.default_params(oracle_lib)
    p.indel_str_repeatsize_max = 4; p.indel_vntr_repeatsize_max = 20; p.indel_BQ_max = 30
    p.indel_polymerase_slip_rate = 3.0; p.indel_del_to_ins_err_ratio = 2.0; p.indel_polymerase_size = 5.0
    p.indel_str_phred_per_region = 17; p.indel_nonSTR_phred_per_base = 3
    check(oracle_lib, fuzz_reference(99, 4000), params=p, smax=4, vmax=20, bq_max=30, slip_rate=3.0, del_to_ins=2.0, polymerase_size=5.0,
          str_phred_per_region=17, nonstr_phred_per_base=3)
