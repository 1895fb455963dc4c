"""files -> VCF with the native command line on a synthetic 1 Mb x 300x BAM, zlib against the library's own inflate:
python scripts/bench_cli.py  (GPU box).  The BAM / FASTA are written first (tests/bamwriter.py, slow Python, not timed)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from uvc_amd import synth
import bamwriter
tile_kb, depth = 1000, 300
d = tempfile.mkdtemp()
t0 = time.perf_counter()
reads = synth.generate_region(seed=3, region_len=tile_kb * 1000, depth=depth, beg=50000)
recs = bamwriter.records_from_reads(reads)
chrom_len = reads["end"] + 50000
rng = np.random.default_rng(1)
seq = "".join("ACGT"[i] for i in rng.integers(0, 4, chrom_len))
seq = seq[:reads["beg"]] + reads["refseq"] + seq[reads["end"]:]
bamwriter.write_bam(os.path.join(d, "t.bam"), [("chrT", chrom_len)], recs)
bamwriter.write_fasta(os.path.join(d, "t.fa"), [("chrT", seq)])
print("files written in %.1f s: %d reads, BAM %.1f MB" % (time.perf_counter() - t0, len(recs), os.path.getsize(os.path.join(d, "t.bam")) / 1e6), flush=True)
exe = "/root/repo/uvc_amd/csrc/uvc1-mi355x"
beg, end = reads["beg"], reads["end"]
for label, env in (("zlib", {"UVCIO_ZLIB": "1"}), ("own inflate, heap columns", {}), ("own inflate, page-locked columns", {"UVC1_PINNED": "1"})):
    for threads in (4, 8, 14):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([exe, os.path.join(d, "t.bam"), "-f", os.path.join(d, "t.fa"), "-o", os.path.join(d, "o_%d.vcf.gz" % len(label)), "--targets", "chrT:%d-%d" % (beg + 1, end),
                            "--tile", "1000000", "-t", str(threads), "--timing", "--repeat", str(2 * threads)], capture_output=True, text=True, env=e)
        print("%s, -t %d: %s" % (label, threads, " | ".join(l.strip() for l in r.stderr.splitlines() if "positions/s" in l or "thread-seconds" in l)), flush=True)
a = open(os.path.join(d, "o_4.vcf.gz"), "rb").read(); b = open(os.path.join(d, "o_32.vcf.gz"), "rb").read()
print("outputs identical:", a == b, len(a))
