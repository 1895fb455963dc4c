"""GPU box: where uvcgpu_region_vcf_records spends its time on the 1 Mb x 300x tile (UVCGPU_TIMING=1 prints the laps)."""
import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["UVCGPU_TIMING"] = "1"
from uvc_amd import region, synth
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reads = synth.generate_region(seed=12345, region_len=kb * 1000, depth=300)
lib = region.gpu_lib()
R = region.Region(lib, region.default_params(lib), reads["tid"], reads["beg"], reads["end"], reads["refseq"])
for rep in range(3):
    R.set_reads(reads); R.accumulate()
    t0 = time.perf_counter(); rec = R.score(kept_only=True); t1 = time.perf_counter()
    txt = R.vcf_records("chr20", rec); t2 = time.perf_counter()
    print("rep %d: score %.1f ms (%d records returned), vcf_records %.1f ms (%d lines, %d bytes)" % (rep, 1e3 * (t1 - t0), len(rec["refpos"]), 1e3 * (t2 - t1), txt.count("\n"), len(txt)), file=sys.stderr, flush=True)
