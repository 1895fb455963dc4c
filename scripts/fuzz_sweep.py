"""One-off wide sweep of tests/test_gpu_fuzz.py's generator: python scripts/fuzz_sweep.py FIRST LAST  (GPU box)."""
import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import importlib.util
from uvc_amd import _ffi, region
from util import diff_groups
spec = importlib.util.spec_from_file_location("fz", "/root/repo/tests/test_gpu_fuzz.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from test_gpu_parity import compare_records
ol = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_"); gl = region.gpu_lib()
nbad = nref = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    reads = fz.weird_region(seed, n_frag=120 + (seed * 37) % 400, ref_len=300 + (seed * 91) % 900, umi=(seed % 3 == 2))
    platform = 2 if seed % 4 == 3 else 1
    out = []
    for lib in (ol, gl):
        try: out.append(fz.run(lib, reads, platform=platform, correct_bq=(seed % 2 == 1)))
        except region.UvcError as e: out.append(e.code)
    o, g = out
    if isinstance(o, int) or isinstance(g, int):
        nref += 1
        if not ((g == -3 and not isinstance(o, int)) or o == g): print("seed", seed, "REFUSAL MISMATCH", o, g, flush=True); nbad += 1
        continue
    bad = diff_groups(o, g)
    if bad:
        nbad += 1
        print("seed", seed, "PLANES", {k: (v[0], v[1][:2]) for k, v in bad.items()}, flush=True); continue
    if o.indel_alleles() != g.indel_alleles(): nbad += 1; print("seed", seed, "ALLELE ROWS differ", flush=True); continue
    try: compare_records(o.score(all_out=True), g.score(all_out=True))
    except AssertionError as e:
        nbad += 1; print("seed", seed, "RECORDS", str(e)[:400], flush=True)
print("swept", sys.argv[1:], "bad", nbad, "refused", nref)
