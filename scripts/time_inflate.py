"""GPU box: throughput of uvcgpu_bgzf_inflate on BAM-record bytes cut into BGZF-sized (0xff00) blocks, against zlib on one core."""
import ctypes, os, sys, time, zlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from uvc_amd import region, synth
import bamwriter
kb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reads = synth.generate_region(seed=3, region_len=kb * 1000, depth=300, beg=50000)
recs = bamwriter.records_from_reads(reads)
path = "/tmp/t_inflate.bam"
bamwriter.write_bam(path, [("chrT", int(reads["end"]) + 1000)], recs)
import gzip
raw = gzip.open(path, "rb").read()      # BGZF is a multi-member gzip file
blocks = [raw[i:i + 0xff00] for i in range(0, len(raw), 0xff00)]
t = time.perf_counter()
comps = []
for b in blocks:
    c = zlib.compressobj(6, zlib.DEFLATED, -15); comps.append(c.compress(b) + c.flush())
print("%d blocks, %.1f MB inflated, %.1f MB compressed (%.1f s to compress)" % (len(blocks), len(raw) / 1e6, sum(map(len, comps)) / 1e6, time.perf_counter() - t), flush=True)
t = time.perf_counter()
for c in comps: zlib.decompress(c, -15)
dt_z = time.perf_counter() - t
print("zlib, one core: %.3f s = %.0f MB/s" % (dt_z, len(raw) / dt_z / 1e6), flush=True)
lib = region.gpu_lib()
fn = lib.dll.uvcgpu_bgzf_inflate
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
comp = np.frombuffer(b"".join(comps) + b"\0" * 8, np.uint8).copy()
in_len = np.array([len(c) for c in comps], np.int32); in_off = np.concatenate([[0], np.cumsum(in_len[:-1])]).astype(np.int64)
out_len = np.array([len(b) for b in blocks], np.int32); out_off = np.concatenate([[0], np.cumsum(out_len[:-1])]).astype(np.int64)
out = np.zeros(len(raw) + 8, np.uint8)
os.environ["UVCGPU_TIMING"] = "1"
for rep in range(4):
    t = time.perf_counter()
    rc = fn(None, comp.ctypes.data, len(comp), in_off.ctypes.data, in_len.ctypes.data, out_off.ctypes.data, out_len.ctypes.data, len(comps), out.ctypes.data, len(out))
    dt = time.perf_counter() - t
    assert rc == 0, lib.last_error()
    print("device (H2D + kernel + D2H, pageable host buffers): %.1f ms = %.2f GB/s of output" % (1e3 * dt, len(raw) / dt / 1e9), flush=True)
assert out[:len(raw)].tobytes() == raw
print("bytes identical")
