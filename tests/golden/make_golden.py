#!/usr/bin/env python3
"""Generates tests/golden/oracle_*.npz: seeded synthetic inputs + the oracle's planes and scoring records.

These fixtures pin the ORACLE (and, through the -m gpu tests, the HIP path) against accidental change.  They are
NOT reference outputs: the reference's hot path cannot be built here (htslib absent, stand-ins not allowed), see DESIGN.md.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from uvc_amd import _ffi, synth  # noqa: E402
from util import INT_GROUPS, run_region  # noqa: E402

CASES = {
    "oracle_nonumi_1500bp_40x": dict(seed=101, region_len=1500, depth=40, snv_every=300, somatic_every=700, indel_every=400),
    "oracle_umi_800bp_300x": dict(seed=102, region_len=800, depth=300, umi=True, snv_every=250, somatic_every=0, indel_every=350),
}

if __name__ == "__main__":
    lib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")
    for name, kw in CASES.items():
        reads = synth.generate_region(**kw)
        R = run_region(lib, reads)
        out = {"planes__" + g: R.fetch(g) for g in INT_GROUPS}
        for all_out in (0, 1):
            rec = R.score(all_out=bool(all_out))
            out["records%d" % all_out] = np.stack([rec[f] for f in _ffi.SCORE_FIELDS])
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, "KiB", {k: int(np.abs(v).sum()) for k, v in list(out.items())[:3]})
