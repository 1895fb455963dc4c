#!/usr/bin/env python3
"""One tile through accumulate once, then `--n` score calls: the program to put behind `rocprofv3 --kernel-trace --stats` (or a --pmc pass)
when only the scoring kernels are of interest.   python scripts/gpu_score_profile.py [--kb 1000] [--depth 300] [--all-out] [--umi] [--n 10]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uvc_amd import region, synth   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kb", type=int, default=1000)
ap.add_argument("--depth", type=int, default=300)
ap.add_argument("--all-out", action="store_true")
ap.add_argument("--umi", action="store_true")
ap.add_argument("--n", type=int, default=10)
a = ap.parse_args()
lib = region.gpu_lib()
assert lib.dll.uvcgpu_init(0) == 0, lib.last_error()
reads = synth.generate_region(seed=12345, region_len=a.kb * 1000, depth=a.depth, umi=a.umi)
R = region.Region(lib, region.default_params(lib), reads["tid"], reads["beg"], reads["end"], reads["refseq"])
R.set_reads(reads)
R.accumulate()
ts = []
for _ in range(a.n):
    t0 = time.perf_counter()
    out = R.score(all_out=a.all_out, kept_only=not a.all_out, copy=False)
    ts.append(time.perf_counter() - t0)
print("%d kb x %d%s %s: %d records returned, score call min %.3f ms median %.3f ms" % (a.kb, a.depth, " umi" if a.umi else "", "all-out" if a.all_out else "default gate, kept_only",
                                                                                      len(out["refpos"]), 1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2]))
R.close()
