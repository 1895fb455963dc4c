"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must reproduce them
(CPU), and the HIP path must reproduce them through the C ABI (-m gpu)."""
import os

import numpy as np
import pytest

from uvc_amd import _ffi, synth
from util import INT_GROUPS, run_region

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import importlib.util
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
_mg = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_mg)


def check(lib, name, exact_records):
    gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    reads = synth.generate_region(**_mg.CASES[name])
    R = run_region(lib, reads)
    for g in INT_GROUPS:
        assert np.array_equal(R.fetch(g), gold["planes__" + g]), g
    for all_out in (0, 1):
        rec = R.score(all_out=bool(all_out))
        got = np.stack([rec[f] for f in _ffi.SCORE_FIELDS]).astype(np.int64)
        exp = gold["records%d" % all_out].astype(np.int64)
        assert got.shape == exp.shape
        if exact_records:
            assert np.array_equal(got, exp)
        else:   # quality class: +-1 Phred, x100 depth-like fields +-1 %
            tol = np.maximum(1, np.abs(exp) // 100)
            assert (np.abs(got - exp) <= tol).all()


@pytest.mark.parametrize("name", list(_mg.CASES))
def test_oracle_reproduces_golden(name, oracle_lib):
    check(oracle_lib, name, exact_records=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(_mg.CASES))
def test_gpu_reproduces_golden(name, gpu_lib):
    check(gpu_lib, name, exact_records=False)


def test_oracle_normal_sample_structure(oracle_lib):
    """T/N normal-sample scoring, oracle only: exactly the positions with a tumor record are emitted, all 14 symbols of
    each, one record per tumor key of an (position, symbol), and a record without a key is scored with tpfa = -1 (its quality
    fields equal those of the same allele when the key list holds only *other* symbols of that position)."""
    import numpy as np
    from uvc_amd import region, synth
    from util import run_region
    reads = synth.generate_region(region_len=3000, depth=40, seed=17)
    base = run_region(oracle_lib, reads).score(all_out=False)
    alt = [i for i in range(len(base["refpos"])) if base["symbol"][i] != base["refsymbol"][i] and base["symbol"][i] < 5][:6]
    assert len(alt) >= 3
    keys = sorted({(int(base["refpos"][i]), int(base["symbol"][i]), int(base["cDP1x"][i]), int(base["CDP1x0"][i]), int(base["bAD"][i]), int(base["bDP"][i]), 0, 0) for i in alt})
    p = region.default_params(oracle_lib); p.tumor_vcf_is_provided = 1
    R = run_region(oracle_lib, reads, params=p)
    rec = R.score(tumor_keys=keys)
    pos = sorted(set(k[0] for k in keys))
    assert sorted(set(rec["refpos"].tolist())) == pos
    for q in pos:
        syms = sorted(rec["symbol"][rec["refpos"] == q].tolist())
        assert syms == list(range(14)), (q, syms)
    # dropping one key changes only the records of that (position, symbol) ... and, through the cross-allele sums, nothing else's inputs
    rec2 = R.score(tumor_keys=keys[1:])
    k0 = keys[0]
    if k0[0] in set(k[0] for k in keys[1:]):
        a = {(int(r), int(s)): i for i, (r, s) in enumerate(zip(rec["refpos"], rec["symbol"]))}
        b = {(int(r), int(s)): i for i, (r, s) in enumerate(zip(rec2["refpos"], rec2["symbol"]))}
        assert rec["cDP1x"][a[(k0[0], k0[1])]] != -12345    # present in both
        same = [key for key in a if key in b and key[0] != k0[0]]
        assert all(rec["cVQ1"][a[key]] == rec2["cVQ1"][b[key]] for key in same)
    with_none = R.score(tumor_keys=None)
    assert len(with_none["refpos"]) == 0                     # no tumor record, nothing to rescue
