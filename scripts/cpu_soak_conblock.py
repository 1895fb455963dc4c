#!/usr/bin/env python3
"""CPU helper (not a test of the suite): the consensus blocks (row a9; host code of libuvcgpu.so, no GPU needed) against the oracle and the independent
Python restatement of tests/test_conblock.py over many seeds and read sets, synthetic and fuzzed.   python3 scripts/cpu_soak_conblock.py SECONDS [FIRST_SEED]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from uvc_amd import _ffi, consensus, region, synth  # noqa: E402
from test_conblock import py_family_blocks, py_to_seq  # noqa: E402
from test_gpu_fuzz import weird_region  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
plib = _ffi.Lib(_ffi.gpu_library_path(), "uvcgpu_")
olib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")
t0, n_ok, n_blocks, fails = time.time(), 0, 0, []
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    umi = bool(rng.integers(0, 2))
    if rng.random() < 0.5:
        reads = weird_region(seed, n_frag=int(rng.choice([60, 260])), ref_len=int(rng.choice([300, 700])), umi=umi)
    else:
        reads = synth.generate_region(seed=seed, region_len=int(rng.choice([1500, 4000])), depth=int(rng.choice([40, 200])), umi=umi, indel_every=int(rng.choice([100, 400])), clip_frac=float(rng.choice([0.05, 0.3])))
    P = region.default_params(plib)
    if rng.random() < 0.3: P.primerlen = int(rng.integers(1, 30))
    mf = int(rng.choice([1, 2, 3]))
    try:
        mine = consensus.family_blocks(plib, P, reads, min_fragments=mf)
        theirs = consensus.family_blocks(olib, P, reads, min_fragments=mf)
        want = py_family_blocks(reads, P, mf)
        assert len(mine) == len(theirs) == len(want), ("count", len(mine), len(theirs), len(want))
        for a, b in zip(mine, theirs):
            key = (a["fam_id"], a["strand"], a["type"], a["refpos"])
            assert key == (b["fam_id"], b["strand"], b["type"], b["refpos"]) and a["n_fragments"] == b["n_fragments"] == want[key][0], ("head", key)
            assert a["rows"].tolist() == b["rows"].tolist() == want[key][1], ("rows", key)
            trim = [None, (20, 3), (60, 1), (150, 2)][int(rng.integers(0, 4))]
            r2l = (a["type"] == 2)
            s = consensus.block_to_seq(plib, a["rows"], r2l, trim)
            assert s == consensus.block_to_seq(olib, a["rows"], r2l, trim) == py_to_seq(a["rows"], r2l, trim), ("seq", key, trim)
        n_blocks += len(mine); n_ok += 1
    except AssertionError as e:
        fails.append(seed); print("FAIL seed", seed, dict(umi=umi, mf=mf, primerlen=P.primerlen), repr(e)[:400], flush=True)
    seed += 1
print("consensus-block soak: %d read sets (%d blocks) equal three ways, %d FAILED %s in %.0f s" % (n_ok, n_blocks, len(fails), fails[:20], time.time() - t0))
sys.exit(1 if fails else 0)
