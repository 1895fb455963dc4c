import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from uvc_amd import region, synth
from util import run_region
lib = region.gpu_lib()
for umi, depth, L in ((False, 100, 3000), (True, 400, 2000), (True, 2000, 20000)):
    reads = synth.generate_region(region_len=L, depth=depth, seed=3, umi=umi)
    R = run_region(lib, reads)
    for it in range(3):
        try:
            r = R.score(release_state=True)
            print(umi, depth, it, "score ok", len(r["refpos"]))
        except Exception as e:
            print(umi, depth, it, "score FAILED", e)
        try:
            R.accumulate()
        except Exception as e:
            print(umi, depth, it, "accumulate FAILED", e)
