"""Runs each InDel read of one fuzz seed alone on both libraries and prints the reads whose planes differ: python scripts/fuzz_isolate.py SEED (GPU box)."""
import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import importlib.util, numpy as np
from uvc_amd import _ffi, region
from util import diff_groups
spec = importlib.util.spec_from_file_location("fz", "/root/repo/tests/test_gpu_fuzz.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
ol = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_"); gl = region.gpu_lib()
seed = int(sys.argv[1])
reads = fz.weird_region(seed, umi=(seed % 3 == 2))
OPS = "MIDNSH"
def subset(i):
    r = dict(reads)
    for k in ("pos", "mpos", "isize", "flag", "mapq", "nm", "l_qseq", "seq_off", "cigar_off", "n_cigar", "frag_id", "fam_id", "fam_strand"):
        r[k] = reads[k][i:i + 1].copy()
    r["n_reads"] = 1; r["frag_id"][:] = 0; r["fam_id"][:] = 0; r["n_fams"] = 1; r["fam_dflag"] = reads["fam_dflag"][int(reads["fam_id"][i]):int(reads["fam_id"][i]) + 1]
    return r
nbad = 0
for i in range(int(reads["n_reads"])):
    if reads["n_cigar"][i] == 1: continue
    r = subset(i)
    try:
        o = fz.run(ol, r); g = fz.run(gl, r)
    except region.UvcError as e:
        print("read", i, "refused", e); continue
    bad = diff_groups(o, g)
    if bad:
        cg = reads["cigars"][int(reads["cigar_off"][i]):int(reads["cigar_off"][i]) + int(reads["n_cigar"][i])]
        so = int(reads["seq_off"][i]); lq = int(reads["l_qseq"][i])
        print("BAD read", i, "pos", int(reads["pos"][i]) - reads["beg"], "".join("%d%s" % (int(c >> 4), OPS[int(c) & 0xF]) for c in cg), "flag", hex(int(reads["flag"][i])), {k: (v[0], v[1][:3]) for k, v in bad.items()})
        print("   quals", reads["quals"][so:so + lq].tolist())
        nbad += 1
        if nbad >= 6: break
print("done, bad =", nbad)
