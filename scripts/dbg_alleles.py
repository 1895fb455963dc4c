"""Prints the InDel allele rows that differ between the oracle and the HIP path: python scripts/dbg_alleles.py SEED UMI N_FAM (GPU box)."""
import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import importlib.util
from uvc_amd import _ffi, region
spec = importlib.util.spec_from_file_location("ta", "/root/repo/tests/test_gpu_indel_alleles.py"); ta = importlib.util.module_from_spec(spec); spec.loader.exec_module(ta)
ol = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_"); gl = region.gpu_lib()
seed, umi, n = int(sys.argv[1]), int(sys.argv[2]) != 0, int(sys.argv[3])
reads = ta.indel_sites_region(seed, n_fam=n, umi=umi)
o, g = ta.run(ol, reads), ta.run(gl, reads)
key = lambda r: (r["refpos"], r["symbol"], r["strand"], r["len"], r["seq"])
ro = {key(r): r for r in o.indel_alleles()}; rg = {key(r): r for r in g.indel_alleles()}
for k in sorted(set(ro) | set(rg), key=lambda k: (k[0], k[1], k[2], k[3], k[4] or "")):
    a, b = ro.get(k), rg.get(k)
    va = a and (a["bAD1"], a["cAD1"], a["c2AD"], a["c2dAD"]); vb = b and (b["bAD1"], b["cAD1"], b["c2AD"], b["c2dAD"])
    if va != vb: print(k, "oracle", va, "gpu", vb)
# isolate: each family alone
import numpy as np
OPS = "MIDNSH"
shown = 0
for fam in range(int(reads["n_fams"])):
    idx = np.nonzero(reads["fam_id"] == fam)[0]
    r = dict(reads)
    for k in ("pos", "mpos", "isize", "flag", "mapq", "nm", "l_qseq", "seq_off", "cigar_off", "n_cigar", "frag_id", "fam_id", "fam_strand"):
        r[k] = reads[k][idx].copy()
    r["n_reads"] = len(idx); r["fam_id"][:] = 0; r["n_fams"] = 1; r["fam_dflag"] = reads["fam_dflag"][fam:fam + 1]
    o1, g1 = ta.run(ol, r), ta.run(gl, r)
    a = {key(x): (x["bAD1"], x["cAD1"], x["c2AD"], x["c2dAD"]) for x in o1.indel_alleles()}; b = {key(x): (x["bAD1"], x["cAD1"], x["c2AD"], x["c2dAD"]) for x in g1.indel_alleles()}
    if a != b:
        print("FAMILY", fam, "dflag", int(reads["fam_dflag"][fam]))
        for k in sorted(set(a) | set(b), key=lambda k: (k[0], k[1], k[2], k[3], k[4] or "")):
            if a.get(k) != b.get(k): print("   ", k, "oracle", a.get(k), "gpu", b.get(k))
        for i in idx:
            cg = reads["cigars"][int(reads["cigar_off"][i]):int(reads["cigar_off"][i]) + int(reads["n_cigar"][i])]
            so = int(reads["seq_off"][i]); qp = 0; ins = []
            for c in cg:
                op, ln = int(c) & 0xF, int(c) >> 4
                if op == 1: ins.append(("".join("ACGTN"[b] for b in reads["bases"][so + qp:so + qp + ln]), reads["quals"][so + qp - 1:so + qp + ln + 1].tolist()))
                if op in (0, 1, 4): qp += ln
            print("    read", i, "frag", int(reads["frag_id"][i]), "strand", int(reads["fam_strand"][i]), "pos", int(reads["pos"][i]) - reads["beg"], "".join("%d%s" % (int(c) >> 4, OPS[int(c) & 0xF]) for c in cg), "flag", hex(int(reads["flag"][i])), "nm", int(reads["nm"][i]), ins)
        shown += 1
        if shown >= 2: break
