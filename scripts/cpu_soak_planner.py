#!/usr/bin/env python3
"""CPU helper (not a test of the suite): uvcio_plan_regions (SamIter::iternext, grouping.cpp:225-312, with its memory model) against the Python
restatement of tests/test_io.py over many seeds, thread counts and memory budgets; one-shot and streamed (uvcio_planner_*, random piece sizes).   python3 scripts/cpu_soak_planner.py SECONDS [FIRST_SEED]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from uvc_amd import io as uio  # noqa: E402
from test_io import _plan_py  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0, n_ok, fails = time.time(), 0, []
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    tlen = [int(v) for v in rng.choice([300, 8000, 50000, 120000], size=int(rng.integers(1, 5)))]
    tid, pos = [], []
    gap_p, step_hi = float(rng.choice([0.0, 0.002, 0.02])), int(rng.choice([2, 3, 40]))
    for t, L in enumerate(tlen):
        at = int(rng.integers(0, 300))
        while at < L - 200:
            at += int(rng.integers(150, 2500)) if rng.random() < gap_p else int(rng.integers(0, step_hi))
            if at < L - 200:
                tid.append(t); pos.append(at)
    tid, pos = np.array(tid, dtype=np.int32), np.array(pos, dtype=np.int32)
    endpos = pos + rng.integers(1, int(rng.choice([2, 151, 3000])), len(pos)).astype(np.int32)
    flag = np.where(rng.random(len(pos)) < float(rng.choice([0.0, 0.01, 0.3])), 4, 0).astype(np.uint16)
    nthreads, mem = int(rng.choice([1, 2, 4, 8, 64])), int(rng.choice([1, 2, 64, 1536]))
    try:
        got = uio.plan_regions(tid, pos, endpos, flag, tlen, nthreads=nthreads, mem_per_thread_mb=mem)
        want = _plan_py(tid, pos, endpos, flag, tlen, nthreads, mem)
        assert got == want, (len(got), len(want), [(a, b) for a, b in zip(got, want) if a != b][:2])
        # the streaming form (what uvc1-mi355x plans with): the same columns fed in pieces, cuts taken as they become final
        piece = int(rng.choice([1, 7, 100, 1000, 100000]))
        got2 = uio.plan_regions_stream(tid, pos, endpos, flag, tlen, nthreads=nthreads, mem_per_thread_mb=mem, piece=piece)
        assert got2 == want, ("streamed, piece %d" % piece, len(got2), len(want), [(a, b) for a, b in zip(got2, want) if a != b][:2])
        n_ok += 1
    except AssertionError as e:
        fails.append(seed); print("FAIL seed", seed, dict(tlen=tlen, n=len(pos), nthreads=nthreads, mem=mem), repr(e)[:400], flush=True)
    seed += 1
print("planner soak: %d inputs equal, %d FAILED %s in %.0f s" % (n_ok, len(fails), fails[:20], time.time() - t0))
sys.exit(1 if fails else 0)
