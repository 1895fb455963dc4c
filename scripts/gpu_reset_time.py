"""GPU-box helper: wall time of uvcgpu_region_reset (side arrays on the device, uvc_rtr.hip) for a 1 Mb tile, synchronised."""
import ctypes as C, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from uvc_amd import region
from rtr_cases import fuzz_reference
lib = region.gpu_lib(); assert lib.dll.uvcgpu_init(0) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
import numpy as np
from uvc_amd import synth
# the bench's own kind of reference (synth.make_reference: random + planted STRs every few kb), then a repeat-dense fuzz reference
refs = [np.frombuffer(b"ACGT", dtype=np.uint8)[synth.make_reference(np.random.default_rng(s), n)].tobytes() for s in range(2)] + [fuzz_reference(7, n, kinds="plain").encode()]
R = region.Region(lib, region.default_params(lib), 0, 1000000, 1000000 + n, refs[0])
lib.dll.uvcgpu_region_sync.argtypes = [C.c_void_p]
for rep in range(3):
    for ref in refs:
        lib.dll.uvcgpu_region_sync(R.h); t0 = time.perf_counter()
        R.reset(0, 1000000, 1000000 + n, ref); t1 = time.perf_counter()
        lib.dll.uvcgpu_region_sync(R.h); t2 = time.perf_counter()
        print("reset %d bp: host call %.3f ms, until the stream is idle %.3f ms" % (n, 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
