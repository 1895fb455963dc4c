"""Times the BAM -> records chain (uvc_amd/pipeline.py) stage by stage on a synthetic tile: python scripts/bench_pipeline.py [tile_kb] [depth]  (GPU box).
The BAM / FASTA files are written first (tests/bamwriter.py, slow Python, not timed)."""
import os, sys, tempfile, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from uvc_amd import group, io as uio, pipeline, region, synth
import bamwriter

tile_kb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 300
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
lib = region.gpu_lib()
assert lib.dll.uvcgpu_init(0) == 0
bam, fa = uio.Bam(os.path.join(d, "t.bam")), uio.Fasta(os.path.join(d, "t.fa"))
beg, end = reads["beg"], reads["end"]
for rep in range(3):
    t = [time.perf_counter()]
    cols = bam.fetch(0, max(0, beg - 2000), end + 2000); t.append(time.perf_counter())
    kind, h = group._digest_batch(lib, cols["qnames"], 0, 0); t.append(time.perf_counter())
    gp = group.default_params(lib, beg, end)
    g = group.group_families(lib, gp, dict(tid=cols["tid"], pos=cols["pos"], endpos=cols["endpos"], mtid=cols["mtid"], mpos=cols["mpos"], isize=cols["isize"], flag=cols["flag"], mapq=cols["mapq"],
                                          qname_hash31=h[0], qname_hash17=h[1], umi_hash31=h[2], umi_hash17=h[3], umi_kind=kind)); t.append(time.perf_counter())
    res = pipeline.call_region(lib, bam, fa, "chrT", beg, end); t.append(time.perf_counter())
    print("rep %d: fetch+decode %.3f s, digests %.3f s, grouping %.3f s | whole call_region %.3f s = %.2f M positions/s, %d reads kept, %d records kept" % (
        rep, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], (end - beg) / (t[4] - t[3]) / 1e6, res["n_reads"], int(res["records"]["keep"].sum())), flush=True)

# whole "contig" in 100 kb tiles, serial and with several tiles in flight
if tile_kb >= 200:
    for workers in (1, 2, 4, 8):
        t = time.perf_counter()
        n_rec = 0; n_pos = 0
        for res in pipeline.call_contig(lib, os.path.join(d, "t.bam"), os.path.join(d, "t.fa"), "chrT", beg, end, tile=100_000, workers=workers):
            n_rec += int(res["records"]["keep"].sum()); n_pos += res["rpos"][1] - res["rpos"][0]
        dt = time.perf_counter() - t
        print("tiles of 100 kb, %d in flight: %.3f s for %d positions = %.2f M positions/s (files -> records), %d records kept" % (workers, dt, n_pos, n_pos / dt / 1e6, n_rec), flush=True)

# the native command line (uvc_amd/csrc/uvc1-mi355x): files -> block-gzipped VCF
import subprocess
exe = "/root/repo/uvc_amd/csrc/uvc1-mi355x"
if os.path.exists(exe):
    for tile, threads in ((100_000, 1), (100_000, 4), (100_000, 8), (100_000, 16), (250_000, 4), (250_000, 8), (500_000, 2), (500_000, 4), (1_000_000, 1), (1_000_000, 2), (1_000_000, 4), (1_000_000, 6), (1_000_000, 8), (500_000, 8)):
        if tile > tile_kb * 1000:
            continue
        r = subprocess.run([exe, os.path.join(d, "t.bam"), "-f", os.path.join(d, "t.fa"), "-o", os.path.join(d, "o.vcf.gz"), "--targets", "chrT:%d-%d" % (beg + 1, end),
                            "--tile", str(tile), "-t", str(threads), "--timing", "--repeat", str(max(4, 2 * threads))], capture_output=True, text=True)
        print("uvc1-mi355x tile %d, -t %d: %s" % (tile, threads, " | ".join(l.strip() for l in r.stderr.splitlines() if "positions/s" in l or "thread-seconds" in l)), flush=True)
