"""Debug helper: accumulate twice on one region and name the planes that differ."""
import sys, numpy as np
sys.path.insert(0, ".")
from uvc_amd import _ffi, region, synth
E = _ffi.ENUMS
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
import os
if os.environ.get("DBG_LIB"):
    _ffi.gpu_library_path = lambda: os.environ["DBG_LIB"]
lib = region.gpu_lib()
assert lib.dll.uvcgpu_init(0) == 0
reads = synth.generate_region(seed=12345, region_len=kb * 1000, depth=300)
R = region.Region(lib, region.default_params(lib), reads["tid"], reads["beg"], reads["end"], reads["refseq"])
R.set_reads(reads)
R.accumulate()
groups = ("VQ", "FRAG", "FAM")
a = {g: R.fetch(g).copy() for g in groups}
R2 = region.Region(lib, region.default_params(lib), reads["tid"], reads["beg"], reads["end"], reads["refseq"])
R2.set_reads(reads)
R2.accumulate()
for g in groups: print("fresh handle", g, "differing cells", int((R2.fetch(g) != a[g]).sum()))
import zlib
print("crc of first VQ", zlib.crc32(a["VQ"].tobytes()))
R.score(capacity=400_000)
for it in range(2):
    R.accumulate()
    b = {g: R.fetch(g).copy() for g in groups}
    for g in groups:
        d = a[g] != b[g]
        print("pass", it, g, a[g].shape, "differing cells", int(d.sum()))
        if d.any():
            idx = np.argwhere(d)
            names = {v: k for k, v in E.items() if k.startswith("UVC_VQ_")}
            import collections
            c = collections.Counter((int(i[0]), int(i[1])) for i in idx[:200000])
            for (f, s), n in sorted(c.items())[:40]:
                print("   field", f, names.get(f), "sym", s, "cells", n)
            for i in idx[:8]:
                print("   ", tuple(int(v) for v in i), int(a[g][tuple(i)]), int(b[g][tuple(i)]))
    R.score(capacity=400_000)
