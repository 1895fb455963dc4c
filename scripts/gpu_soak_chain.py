#!/usr/bin/env python3
"""GPU-box helper (not a test of the suite): BAM + FASTA -> records through the whole chain (reader, family assignment, region, scoring) with
the HIP libraries against the oracle libraries, tests/test_pipeline.py::test_chain_gpu_equals_oracle over many seeds, read sets and
sub-ranges.  (Two tilings of one contig are not compared here: the BAQ prefix sums of a region start at its first base and are divided by ten
afterwards, so a quality next to a cut may differ by 1 between tilings and a record at the --vqual threshold may be kept in one only -- DESIGN.md 4c,
tests/test_tiles.py.)   python3 scripts/gpu_soak_chain.py SECONDS [FIRST_SEED]"""
import os
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
glib = region.gpu_lib(); assert glib.dll.uvcgpu_init(0) == 0
olib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")
t0, n_ok, fails = time.time(), 0, []
with tempfile.TemporaryDirectory() as d:
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        umi = bool(rng.integers(0, 2))
        L, depth = int(rng.choice([3000, 6000, 12000])), int(rng.choice([30, 60, 150]))
        reads = synth.generate_region(seed=seed, region_len=L, depth=depth, beg=int(rng.choice([2000, 30000])), umi=umi, snv_every=int(rng.choice([150, 300])), somatic_every=900,
                                      indel_every=int(rng.choice([200, 500])), err_rate=float(rng.choice([1e-3, 1e-2])), clip_frac=float(rng.choice([0.01, 0.2])))
        umis = None
        if umi:
            umis = ["".join("ACGT"[i] for i in rng.integers(0, 4, 6)) + "+" + "".join("ACGT"[i] for i in rng.integers(0, 4, 6)) for _ in range(int(reads["n_fams"]))]
        recs = bamwriter.records_from_reads(reads, tid=0, umis=umis)
        chrom_len = reads["end"] + int(rng.choice([300, 5000]))
        seq = "".join("ACGT"[i] for i in rng.integers(0, 4, chrom_len))
        seq = seq[:reads["beg"]] + reads["refseq"] + seq[reads["end"]:]
        bamwriter.write_bam(os.path.join(d, "s.bam"), [("chrT", chrom_len)], recs)
        bamwriter.write_fasta(os.path.join(d, "s.fa"), [("chrT", seq)])
        bam, fa = uio.Bam(os.path.join(d, "s.bam")), uio.Fasta(os.path.join(d, "s.fa"))
        a, b = reads["beg"] + int(rng.integers(0, L // 3)), reads["beg"] + L - int(rng.integers(0, L // 3))
        try:
            ro = pipeline.call_region(olib, bam, fa, "chrT", a, b, molecule_tag=0)
            rg = pipeline.call_region(glib, bam, fa, "chrT", a, b, molecule_tag=0)
            assert (ro is None) == (rg is None)
            if ro is not None:
                assert (ro["n_reads"], ro["n_fams"], ro["rpos"], ro["ext"]) == (rg["n_reads"], rg["n_fams"], rg["rpos"], rg["ext"]), "region"
                assert ro["alleles"] == rg["alleles"], "alleles"
                compare_records(ro["records"], rg["records"])
            n_ok += 1
        except (AssertionError, region.UvcError) as e:
            fails.append(seed); print("FAIL seed", seed, dict(umi=umi, L=L, depth=depth, a=a, b=b), repr(e)[:400], flush=True)
        seed += 1
print("chain soak: %d file sets equal, %d FAILED %s in %.0f s" % (n_ok, len(fails), fails[:20], time.time() - t0))
sys.exit(1 if fails else 0)
