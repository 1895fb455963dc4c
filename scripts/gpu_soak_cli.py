#!/usr/bin/env python3
"""GPU-box helper (not a test of the suite): uvc1-mi355x (the chain in C++, tiles in flight on threads) against uvc_amd/pipeline.py on random
file sets -- thread counts, tile lengths, device inflate, two --shard processes + --concat -- the VCF text must be the same.
    python3 scripts/gpu_soak_cli.py SECONDS [FIRST_SEED]"""
import gzip
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from uvc_amd import pipeline, region, synth  # noqa: E402
import bamwriter  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
exe = os.path.join(ROOT, "uvc_amd", "csrc", "uvc1-mi355x")
glib = region.gpu_lib(); assert glib.dll.uvcgpu_init(0) == 0


def text(path):
    return [l for l in gzip.open(path, "rt").read().splitlines() if not l.startswith(("##fileDate=", "##variantCallerCommand="))]


t0, n_ok, fails = time.time(), 0, []
with tempfile.TemporaryDirectory() as d:
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        umi = bool(rng.integers(0, 2))
        L, depth = int(rng.choice([4000, 9000, 20000])), int(rng.choice([30, 80, 200]))
        reads = synth.generate_region(seed=seed, region_len=L, depth=depth, beg=int(rng.choice([3000, 40000])), umi=umi, snv_every=int(rng.choice([150, 400])), somatic_every=900,
                                      indel_every=int(rng.choice([200, 600])), err_rate=float(rng.choice([1e-3, 1e-2])), clip_frac=float(rng.choice([0.01, 0.2])))
        umis = None
        if umi:
            umis = ["".join("ACGT"[i] for i in rng.integers(0, 4, 6)) + "+" + "".join("ACGT"[i] for i in rng.integers(0, 4, 6)) for _ in range(int(reads["n_fams"]))]
        chrom_len = reads["end"] + int(rng.choice([300, 5000]))
        seq = "".join("ACGT"[i] for i in rng.integers(0, 4, chrom_len))
        seq = seq[:reads["beg"]] + reads["refseq"] + seq[reads["end"]:]
        bam, fa = os.path.join(d, "s.bam"), os.path.join(d, "s.fa")
        bamwriter.write_bam(bam, [("chrT", chrom_len)], bamwriter.records_from_reads(reads, tid=0, umis=umis))
        bamwriter.write_fasta(fa, [("chrT", seq)])
        b0 = reads["beg"]
        tile = int(rng.choice([1500, 2500, 7000]))
        target = "chrT:%d-%d" % (b0 + 1, b0 + L)
        threads = int(rng.choice([1, 2, 3, 5]))
        env = dict(os.environ)
        if rng.random() < 0.3: env.update(UVC1_DEVICE_INFLATE="1", UVC1_DEVICE_INFLATE_MIN="1")
        if rng.random() < 0.2: env.update(UVC1_PINNED="1")
        out_c, out_py, p0, p1, out_s = [os.path.join(d, n) for n in ("c.vcf.gz", "py.vcf.gz", "p0.vcf.gz", "p1.vcf.gz", "s.vcf.gz")]
        try:
            base = [exe, bam, "-f", fa, "-s", "T1", "--targets", target, "--tile", str(tile)]
            r = subprocess.run(base + ["-o", out_c, "-t", str(threads)], capture_output=True, text=True, timeout=300, env=env)
            assert r.returncode == 0, r.stderr[-300:]
            pipeline.write_vcf(glib, bam, fa, "chrT", b0, b0 + L, out_py, sample="T1", tile=tile)
            a, b = text(out_c), text(out_py)
            assert [l for l in a if not l.startswith("##")] == [l for l in b if not l.startswith("##")], "command line against the Python chain"
            if rng.random() < 0.5:   # one process per shard + --concat
                for i, p in enumerate((p0, p1)):
                    r = subprocess.run(base + ["-o", p, "-t", "2", "--shard", "%d/2" % i], capture_output=True, text=True, timeout=300, env=env)
                    assert r.returncode == 0, r.stderr[-300:]
                r = subprocess.run([exe, "--concat", out_s, p0, p1], capture_output=True, text=True, timeout=120)
                assert r.returncode == 0, r.stderr[-300:]
                assert text(out_s) == a, "shards"
            n_ok += 1
        except AssertionError as e:
            fails.append(seed); print("FAIL seed", seed, dict(umi=umi, L=L, depth=depth, tile=tile, threads=threads), repr(e)[:400], flush=True)
        seed += 1
print("cli soak: %d file sets equal, %d FAILED %s in %.0f s" % (n_ok, len(fails), fails[:20], time.time() - t0))
sys.exit(1 if fails else 0)
