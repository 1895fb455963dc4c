#!/usr/bin/env python3
"""GPU-box helper (not a test of the suite): the two-pass T/N flow of bin/uvcTN.sh on random file sets -- uvc1-mi355x tumor pass (--tn-is-paired 1
--bed-out-fname), normal pass (--bed-in-fname --tumor-vcf) -- against the Python chain on the HIP libraries (same text) and on the oracle
libraries fed with the keys libuvcio reads from the tumor VCF (records in the tolerance classes).   python3 scripts/gpu_soak_tn_cli.py SECONDS [FIRST_SEED]"""
import gzip
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from uvc_amd import _ffi, io as uio, pipeline, region, synth  # noqa: E402
import bamwriter  # noqa: E402
from test_gpu_parity import compare_records  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
exe = os.path.join(ROOT, "uvc_amd", "csrc", "uvc1-mi355x")
glib = region.gpu_lib(); assert glib.dll.uvcgpu_init(0) == 0
olib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")


def cli(args):
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-400:]


t0, n_ok, n_rec, fails = time.time(), 0, 0, []
with tempfile.TemporaryDirectory() as d:
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        L, tile = int(rng.choice([3000, 5000, 9000])), int(rng.choice([2000, 3500]))
        kw = dict(seed=seed, region_len=L, beg=int(rng.choice([4000, 40000])), snv_every=int(rng.choice([200, 400])), somatic_every=int(rng.choice([300, 700])), indel_every=int(rng.choice([300, 900])),
                  umi=bool(rng.integers(0, 2)))
        tb, nb, fa = [os.path.join(d, n) for n in ("tumor.bam", "normal.bam", "tn.fa")]
        tv, nv, bed = [os.path.join(d, n) for n in ("T.vcf.gz", "N.vcf.gz", "T.bed")]
        for name, depth, path in (("tumor", int(rng.choice([80, 150])), tb), ("normal", int(rng.choice([30, 60])), nb)):
            reads = synth.generate_region(depth=depth, **kw)
            umis = None
            if kw["umi"]:
                r2 = np.random.default_rng(seed * 7 + depth)
                umis = ["".join("ACGT"[i] for i in r2.integers(0, 4, 6)) + "+" + "".join("ACGT"[i] for i in r2.integers(0, 4, 6)) for _ in range(int(reads["n_fams"]))]
            chrom_len = reads["end"] + 4000
            seq = "".join("ACGT"[i] for i in np.random.default_rng(6).integers(0, 4, chrom_len))
            seq = seq[:reads["beg"]] + reads["refseq"] + seq[reads["end"]:]
            bamwriter.write_bam(path, [("chrT", chrom_len)], bamwriter.records_from_reads(reads, tid=0, umis=umis))
            bamwriter.write_fasta(fa, [("chrT", seq)])
        b0 = kw["beg"]
        try:
            cli([tb, "-f", fa, "-o", tv, "-s", "TUM", "--targets", "chrT:%d-%d" % (b0 + 1, b0 + L), "--tile", str(tile), "--tn-is-paired", "1", "--bed-out-fname", bed, "-t", str(int(rng.choice([1, 3])))])
            cli([nb, "-f", fa, "-o", nv, "-s", "NOR", "--tn-is-paired", "1", "--bed-in-fname", bed, "--tumor-vcf", tv, "-t", str(int(rng.choice([1, 3])))])
            n_lines = [l for l in gzip.open(nv, "rt").read().splitlines() if not l.startswith("##")]
            T = uio.TumorVcf(tv, ["chrT"])
            bam, fasta = uio.Bam(nb), uio.Fasta(fa)
            got = None
            for lib in (glib, olib):
                p = region.default_params(lib)
                p.tumor_vcf_is_provided, p.tn_is_paired = 1, 1
                res = list(pipeline.call_contig(lib, bam, fasta, "chrT", b0, b0 + L, tile=tile, params=p, tumor_vcf=T, vcf=(lib is glib)))
                if lib is glib:
                    got = res
                else:
                    assert len(res) == len(got), "tiles"
                    for a, b in zip(res, got):
                        assert a["score_range"] == b["score_range"], "score range"
                        compare_records(a["records"], b["records"])
                        n_rec += len(a["records"]["refpos"])
            assert "".join(t["vcf"] for t in got).splitlines() == n_lines[1:], "command line against the Python chain"
            T.close()
            n_ok += 1
        except (AssertionError, region.UvcError) as e:
            fails.append(seed); print("FAIL seed", seed, dict(L=L, tile=tile, **{k: v for k, v in kw.items() if k != "seed"}), repr(e)[:500], flush=True)
        seed += 1
print("T/N command-line soak: %d file sets (%d normal-sample records) equal, %d FAILED %s in %.0f s" % (n_ok, n_rec, len(fails), fails[:20], time.time() - t0))
sys.exit(1 if fails else 0)
