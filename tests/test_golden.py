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
