"""One-off sweep of the record-line comparison of tests/test_vcf_text.py over the fuzz generator: python scripts/vcf_sweep.py FIRST LAST  (GPU box)."""
import ctypes as C, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from uvc_amd import _ffi, region
import test_vcf_text as T
from test_gpu_fuzz import run, weird_region
ol = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_"); gl = region.gpu_lib()
L = C.CDLL(T.REF_SO)
for n in ("uvc_ref_format_string", "uvc_ref_format_id", "uvc_ref_format_line", "uvc_ref_filter_id", "uvc_ref_filter_line"): getattr(L, n).restype = C.c_char_p
L.uvc_ref_stream_format.restype = C.c_int64; L.uvc_ref_stream_format.argtypes = [C.c_char_p, C.c_char_p, C.c_int64]
bad = nlines = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    reads = weird_region(seed, n_frag=120 + (seed * 37) % 400, ref_len=300 + (seed * 91) % 900, umi=(seed % 3 == 2))
    platform = 2 if seed % 4 == 3 else 1
    try:
        Ro, Rg = run(ol, reads, platform=platform), run(gl, reads, platform=platform)
    except region.UvcError:
        continue
    all_out = (seed % 2 == 0)
    mine = Rg.vcf_records("chrF", Rg.score(all_out=all_out)).splitlines()
    want = T._oracle_lines(ol, L, Ro, "chrF", all_out=all_out)
    try:
        T.compare_lines(mine, want); nlines += len(want)
    except AssertionError as e:
        bad += 1; print("seed", seed, str(e)[:500], flush=True)
print("swept", sys.argv[1:], "lines", nlines, "bad", bad)
