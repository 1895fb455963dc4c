"""One-off sweep of the normal-sample (T/N) scoring arm over the fuzz generator: python scripts/tn_sweep.py FIRST LAST  (GPU box).
Tumor keys come from the oracle's tumor-only records of the same reads; oracle and GPU are then scored as the normal sample (both arms of
the somatic quality), with every record group returned and with kept_only."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from uvc_amd import _ffi, region
from test_gpu_fuzz import weird_region
from test_gpu_parity import compare_records, tumor_keys_from
from util import run_region
ol = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_"); gl = region.gpu_lib()
bad = n = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    reads = weird_region(seed, n_frag=150 + (seed * 37) % 400, ref_len=300 + (seed * 91) % 900, umi=(seed % 3 == 2))
    try:
        keys = tumor_keys_from(run_region(ol, reads).score(all_out=(seed % 2 == 0)))
    except region.UvcError:
        continue
    if len(keys) < 2:
        continue
    out = []
    try:
        for lib in (ol, gl):
            p = region.default_params(lib); p.tumor_vcf_is_provided = 1
            if seed % 4 >= 2:
                p.tn_syserr_norm_devqual = -1.0; p.outvar_flag = 63
            R = run_region(lib, reads, params=p)
            out.append((R.score(tumor_keys=keys), R.score(tumor_keys=keys, kept_only=True) if lib is gl else None, R))
    except region.UvcError as e:
        print("seed", seed, "refused", e, flush=True); continue
    n += 1
    try:
        compare_records(out[0][0], out[1][0])
        full, kept, Rg = out[1]
        assert Rg.vcf_records("chrF", kept, tumor_keys=keys) == Rg.vcf_records("chrF", full, tumor_keys=keys)
    except AssertionError as e:
        bad += 1; print("seed", seed, str(e)[:400], flush=True)
    for o in out:
        o[2].close()
print("swept", sys.argv[1:], "regions", n, "bad", bad)
