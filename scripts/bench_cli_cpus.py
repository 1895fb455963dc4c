"""files -> VCF with the native command line: how many host cores does the chain really have, and does it use them?
python scripts/bench_cli_cpus.py  (GPU box).  Prints the box's CPU limits, then runs uvc1-mi355x with the thread pools sized by the
readers' shared pool (uvc_io.cpp) at its default size (the quota-aware core count, uvc_cpus.h), at 8 and 32 threads and with zlib,
with wall clock and user / system CPU seconds of the child."""
import os, resource, subprocess, sys, tempfile, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from uvc_amd import synth
import bamwriter
def rd(p):
    try: return open(p).read().strip()
    except OSError: return None
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "cpu.max", rd("/sys/fs/cgroup/cpu.max"),
      "cfs_quota_us", rd("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), "cfs_period_us", rd("/sys/fs/cgroup/cpu/cpu.cfs_period_us"), flush=True)
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
print("files written in %.1f s" % (time.perf_counter() - t0), flush=True)
exe = "/root/repo/uvc_amd/csrc/uvc1-mi355x"
beg, end = reads["beg"], reads["end"]
outs = []
for label, env in (("host inflate", {}), ("device inflate", {"UVC1_DEVICE_INFLATE": "1"}), ("device inflate, page-locked columns", {"UVC1_DEVICE_INFLATE": "1", "UVC1_PINNED": "1"})):
    for threads in (4, 8, 12):
        e = dict(os.environ); e.update(env)
        out = os.path.join(d, "o_%d_%d.vcf.gz" % (len(outs), threads)); outs.append((threads, out))
        r0 = resource.getrusage(resource.RUSAGE_CHILDREN); w0 = time.perf_counter()
        r = subprocess.run([exe, os.path.join(d, "t.bam"), "-f", os.path.join(d, "t.fa"), "-o", out, "--targets", "chrT:%d-%d" % (beg + 1, end),
                            "--tile", "1000000", "-t", str(threads), "--timing", "--repeat", str(2 * threads)], capture_output=True, text=True, env=e)
        w = time.perf_counter() - w0; r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
        print("%s, -t %d: wall %.2f s, user %.1f s, sys %.1f s (%.1f cores busy) | %s" % (label, threads, w, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime,
              (r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime) / w, " | ".join(l.strip() for l in r.stderr.splitlines() if "positions/s" in l or "thread-seconds" in l)), flush=True)
        if r.returncode: print(r.stderr[-2000:]); sys.exit(1)
by_t = {}
for th, o in outs: by_t.setdefault(th, []).append(open(o, "rb").read())
print("outputs identical per thread count:", all(all(x == v[0] for x in v) for v in by_t.values()))
