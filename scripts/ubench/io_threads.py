import sys, os, time, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from uvc_amd import io as uio, synth
import bamwriter
d = tempfile.mkdtemp()
reads = synth.generate_region(seed=3, region_len=50000, depth=300, beg=50000)
recs = bamwriter.records_from_reads(reads)
bamwriter.write_bam(os.path.join(d, "t.bam"), [("chrT", 200000)], recs)
print(len(recs), os.path.getsize(os.path.join(d, "t.bam")))
b = uio.Bam(os.path.join(d, "t.bam"))
for rep in range(3):
    t = time.perf_counter(); c = b.fetch(0, 40000, 110000); print("fetch %.3f s" % (time.perf_counter() - t), c["n_alns"])
