import sys, os
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tests")
import numpy as np
from uvc_amd import _ffi, region, synth
from util import diff_groups
glib = region.gpu_lib(); assert glib.dll.uvcgpu_init(0) == 0
olib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")
def run(lib, reads, P=None):
    P = P or region.default_params(lib)
    R = region.Region(lib, P, reads["tid"], reads["beg"], reads["end"], reads["refseq"]); R.set_reads(reads); R.accumulate(); return R
for depth, L, seed in ((2000, 20000, 1), (2000, 10000, 1), (1000, 20000, 1), (2000, 14000, 2), (2000, 17000, 3), (3000, 12000, 4)):
    reads = synth.generate_region(seed=seed, region_len=L, depth=depth)
    o, g = run(olib, reads), run(glib, reads)
    bad = diff_groups(o, g)
    print(depth, L, seed, "n_reads", reads["n_reads"], {k: v[0] for k, v in bad.items()}, flush=True)
    if "FRAG" in bad:
        a, b = o.fetch("FRAG"), g.fetch("FRAG")
        d = np.argwhere(a != b)
        print("  strands", np.unique(d[:,0]), "fields", np.unique(d[:,1]), "syms", np.unique(d[:,2]), "pos range", d[:,3].min(), d[:,3].max(), "n distinct pos", len(np.unique(d[:,3])))
        for i in d[:5]: print("   ", tuple(i), int(a[tuple(i)]), int(b[tuple(i)]))
        pos = np.unique(d[:,3]); print("  first positions", pos[:20], "gaps", np.unique(np.diff(pos))[:10])
    o.close(); g.close()
