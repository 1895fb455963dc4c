#!/usr/bin/env python3
"""CPU helper (not a test of the suite): the library's own DEFLATE decoder (uvc_inflate_fast.h behind uvcio_inflate_raw_fast) on damaged streams --
bit flips, truncations, spliced and random bytes, wrong output sizes -- under AddressSanitizer (scripts/cpu_sanitize.sh builds the library): it must
decline or decode without touching memory outside its buffers, and whatever it accepts for an intact stream must be zlib's bytes.
    python3 scripts/cpu_fuzz_inflate.py SECONDS [SEED]"""
import ctypes
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from uvc_amd import io as uio  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = uio.dll().uvcio_inflate_raw_fast
f.restype, f.argtypes = ctypes.c_int, [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64]
rng = np.random.default_rng(seed)


def make(kind, n):
    if kind == 0: return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 1: return rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes()
    if kind == 2: return (b"read_name_0123456789:" * (n // 21 + 1))[:n]
    if kind == 3: return (rng.integers(0, 41, n, dtype=np.uint8) + 33).tobytes()
    return b"\x00" * n


t0, n_intact, n_damaged, n_accepted = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    n = int(rng.choice([0, 1, 7, 100, 1000, 5000, 20000, 65280]))
    data = make(int(rng.integers(0, 5)), n)
    co = zlib.compressobj(int(rng.choice([0, 1, 6, 9])), zlib.DEFLATED, -15, 9, int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])))
    comp = co.compress(data) + co.flush()
    out = ctypes.create_string_buffer(max(n, 1))
    if f(comp, len(comp), out, n):
        assert out.raw[:n] == data
    n_intact += 1
    for _ in range(8):
        bad = bytearray(comp)
        kind = int(rng.integers(0, 5))
        if kind == 0 and bad:
            for _ in range(int(rng.integers(1, 6))): bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1 and bad: bad = bad[:int(rng.integers(0, len(bad)))]
        elif kind == 2: bad = bytearray(rng.integers(0, 256, int(rng.integers(0, 300)), dtype=np.uint8).tobytes())
        elif kind == 3 and len(bad) > 4:
            a = int(rng.integers(0, len(bad) - 2)); bad[a:a + int(rng.integers(1, 40))] = rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8).tobytes()
        m = n if kind != 4 else max(0, n + int(rng.integers(-50, 50)))        # a wrong output size for an intact stream
        out = ctypes.create_string_buffer(max(m, 1))
        n_accepted += int(f(bytes(bad), len(bad), out, m) != 0)
        n_damaged += 1
print("inflate fuzz: %d intact streams equal zlib, %d damaged streams handled (%d accepted: a flipped literal decodes), no sanitizer report, %.0f s" % (n_intact, n_damaged, n_accepted, time.time() - t0))
