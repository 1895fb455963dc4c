#!/usr/bin/env python3
"""A/B check of a scoring-path change on the GPU box: the library as built against a reference build of the same ABI
(default build_ab/libuvcgpu_r3.so, the end-of-round-3 library), same reads, every score field compared EXACTLY, at sizes
where the CPU checker takes minutes.  Prints one line per case and UVCGPU_TIMING-style wall times of the score call.

    python scripts/gpu_ab_score.py [--ref build_ab/libuvcgpu_r3.so] [--kb 200] [--depth 300] [--umi-kb 50]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uvc_amd import _ffi, region, synth   # noqa: E402


def run(lib, reads, all_out, kept_only=False, repeat=3):
    R = region.Region(lib, region.default_params(lib), reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    R.set_reads(reads)
    R.accumulate()
    out, best = None, 1e9
    for _ in range(repeat):
        t0 = time.perf_counter()
        out = R.score(all_out=all_out, kept_only=kept_only)
        best = min(best, time.perf_counter() - t0)
    alleles = R.indel_alleles()
    R.close()
    return out, best, alleles


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default=os.path.join(ROOT, "build_ab", "libuvcgpu_r3.so"))
    ap.add_argument("--kb", type=int, default=200)
    ap.add_argument("--depth", type=int, default=300)
    ap.add_argument("--umi-kb", type=int, default=50)
    a = ap.parse_args()
    new = region.gpu_lib()
    rc = new.dll.uvcgpu_init(0)
    assert rc == 0, new.last_error()
    old = _ffi.Lib(a.ref, "uvcgpu_")
    old.dll.uvcgpu_init.restype, old.dll.uvcgpu_init.argtypes = C.c_int, [C.c_int]
    assert old.dll.uvcgpu_init(0) == 0
    cases = [("nonumi %d kb x %d" % (a.kb, a.depth), dict(seed=7, region_len=a.kb * 1000, depth=a.depth)),
             ("umi %d kb x 2000" % a.umi_kb, dict(seed=11, region_len=a.umi_kb * 1000, depth=2000, umi=True)),
             ("small 3 kb x 60 dense variants", dict(seed=3, region_len=3000, depth=60, snv_every=50, indel_every=100, somatic_every=200))]
    bad = 0
    for name, kw in cases:
        reads = synth.generate_region(**kw)
        for all_out, kept in ((False, False), (True, False), (False, True)):
            o_new, t_new, al_new = run(new, reads, all_out, kept)
            o_old, t_old, al_old = run(old, reads, all_out, kept)
            n = len(o_new["refpos"])
            diffs = []
            if len(o_old["refpos"]) != n:
                diffs.append("n_records %d vs %d" % (n, len(o_old["refpos"])))
            else:
                for k in o_new:
                    if not np.array_equal(o_new[k], o_old[k]):
                        w = np.nonzero(o_new[k] != o_old[k])[0]
                        diffs.append("%s: %d differ, first at %d (%d vs %d, refpos %d symbol %d)" % (k, len(w), w[0], o_new[k][w[0]], o_old[k][w[0]], o_new["refpos"][w[0]], o_new["symbol"][w[0]]))
            print("%-34s all_out=%d kept_only=%d  %8d records  score call %.2f ms (ref build %.2f ms)  %s" % (name, all_out, kept, n, 1e3 * t_new, 1e3 * t_old, "IDENTICAL" if not diffs else "DIFFERENT"), flush=True)
            for d in diffs[:12]:
                print("     " + d)
            bad += bool(diffs)
    print("cases that differ: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
