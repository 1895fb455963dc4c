#!/usr/bin/env python3
"""GPU-box helper (not a test of the suite): the family-assignment comparison of tests/test_group.py::test_gpu_against_oracle over many seeds,
sizes and parameter settings for a given number of seconds.    python3 scripts/gpu_soak_group.py SECONDS [FIRST_SEED]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from uvc_amd import _ffi, group, region  # noqa: E402
from test_group import canon, make_alignments  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
glib = region.gpu_lib(); assert glib.dll.uvcgpu_init(0) == 0
olib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")
t0, n_ok, fails = time.time(), 0, []
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    umi, amplicon = bool(rng.integers(0, 2)), bool(rng.integers(0, 3) == 0)
    n_pairs, length = int(rng.choice([30, 300, 1400, 6000, 30000])), int(rng.choice([400, 3000, 20000, 200000]))
    cols, qnames, tb, te = make_alignments(seed=seed, n_pairs=n_pairs, length=length, umi=umi, amplicon=amplicon)
    P = group.default_params(olib, tb, te)
    if rng.random() < 0.3:
        P.pair_end_merge, P.end2end, P.kept_aln_min_aln_len, P.kept_aln_min_mapqual = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.choice([0, 10, 60])), int(rng.choice([0, 20, 60]))
    if rng.random() < 0.3:
        P.inferred_sequencing_platform = 2
    try:
        ro, rg = group.group_families(olib, P, cols), group.group_families(glib, P, cols)
        for k in ("filter_reason", "isize_norm"):
            assert np.array_equal(ro[k], rg[k]), k
        for k in ("n_kept", "n_fams", "n_frags", "ext_beg", "ext_end", "n_amplicon", "n_visited_qnames"):
            assert ro[k] == rg[k], (k, ro[k], rg[k])
        assert canon(ro) == canon(rg), "families"
        n_ok += 1
    except (AssertionError, Exception) as e:   # noqa: BLE001
        fails.append(seed); print("FAIL seed", seed, dict(umi=umi, amplicon=amplicon, n_pairs=n_pairs, length=length), repr(e)[:300], flush=True)
    seed += 1
print("group soak: %d inputs equal, %d FAILED %s in %.0f s" % (n_ok, len(fails), fails[:20], time.time() - t0))
sys.exit(1 if fails else 0)
