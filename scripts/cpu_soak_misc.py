#!/usr/bin/env python3
"""CPU helper (not a test of the suite): two more rows of the oracle against their independent Python restatements over many seeds --
apply_bq_err_correction3 (tests/test_bq_correction.py) on fuzzed and synthetic reads, and the family assignment (tests/test_group.py: filter reasons,
families, fragments, amplicon / visited counts) on random alignment sets.   python3 scripts/cpu_soak_misc.py SECONDS [FIRST_SEED]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from uvc_amd import _ffi, group, synth  # noqa: E402
from test_bq_correction import corrected, expected_quals  # noqa: E402
from test_group import canon, make_alignments, py_group  # noqa: E402
from test_gpu_fuzz import weird_region  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
olib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")
t0, n_bq, n_grp, fails = time.time(), 0, 0, []
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    try:
        if rng.random() < 0.6: reads = weird_region(seed, n_frag=int(rng.choice([60, 260])), ref_len=int(rng.choice([300, 700])))
        else: reads = synth.generate_region(seed=seed, region_len=2000, depth=int(rng.choice([20, 60])), clip_frac=float(rng.choice([0.0, 0.3])))
        bq_max, bq_inc = int(rng.choice([30, 37, 41, 60])), int(rng.choice([0, 1, 4, 10]))
        R, q = corrected(olib, reads, bq_max, bq_inc)
        exp = expected_quals(reads, bq_max, bq_inc)
        assert np.array_equal(q, exp), ("bq", np.flatnonzero(q != exp)[:6].tolist())
        R.close(); n_bq += 1
        umi, amplicon = bool(rng.integers(0, 2)), bool(rng.integers(0, 3) == 0)
        cols, qnames, tb, te = make_alignments(seed=seed, n_pairs=int(rng.choice([30, 300, 1400])), length=int(rng.choice([400, 3000, 20000])), umi=umi, amplicon=amplicon)
        P = group.default_params(olib, tb, te)
        if rng.random() < 0.3:
            P.pair_end_merge, P.end2end, P.kept_aln_min_aln_len, P.kept_aln_min_mapqual = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.choice([0, 10, 60])), int(rng.choice([0, 20, 60]))
        res = group.group_families(olib, P, cols)
        reason, fams, n_amp, n_vis = py_group(P, cols)
        assert np.array_equal(res["filter_reason"], reason), "filter reasons"
        assert res["n_amplicon"] == n_amp and res["n_visited_qnames"] == n_vis and res["n_fams"] == len(fams), "counts"
        expf = sorted((tuple(sorted((s, i) for (s, _), idx in fr.items() for i in idx)), key[4], key[5]) for key, fr in fams.items())
        got, frags = canon(res)
        assert got == expf and frags == sorted(tuple(idx) for fr in fams.values() for idx in fr.values()), "families"
        n_grp += 1
    except AssertionError as e:
        fails.append(seed); print("FAIL seed", seed, repr(e)[:400], flush=True)
    seed += 1
print("misc soak: %d read sets through the BQ correction, %d alignment sets through the family assignment equal, %d FAILED %s in %.0f s" % (n_bq, n_grp, len(fails), fails[:20], time.time() - t0))
sys.exit(1 if fails else 0)
