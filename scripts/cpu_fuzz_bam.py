#!/usr/bin/env python3
"""CPU helper (not a test of the suite): the BAM reader of libuvcio.so on DAMAGED files under AddressSanitizer (scripts/cpu_sanitize.sh builds the
library): the inflated record stream of a good file gets bit flips / spliced bytes / truncations and is compressed again (valid BGZF, valid CRC), so
that the damage reaches the record walk and the field decode.  The reader must refuse the file or return something -- never touch memory outside its
buffers.    python3 scripts/cpu_fuzz_bam.py SECONDS [SEED]"""
import os
import struct
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from uvc_amd import io as uio, synth  # noqa: E402
import bamwriter  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)


def blocks(raw):
    out, at = [], 0
    while at < len(raw):
        bsize = struct.unpack_from("<H", raw, at + 16)[0] + 1
        out.append(zlib.decompress(raw[at + 18:at + bsize - 8], -15))
        at += bsize
    return out


t0, n_files, n_refused, n_read = time.time(), 0, 0, 0
with tempfile.TemporaryDirectory() as d:
    reads = synth.generate_region(seed=3, region_len=6000, depth=30, beg=20000, indel_every=500, clip_frac=0.1)
    recs = bamwriter.records_from_reads(reads, tid=0)
    good = os.path.join(d, "g.bam")
    bamwriter.write_bam(good, [("chrT", 40000)], recs, block_bytes=4000, with_index=False)
    payload = b"".join(blocks(open(good, "rb").read()))
    while time.time() - t0 < budget:
        bad = bytearray(payload)
        kind = int(rng.integers(0, 4))
        lo = 0 if rng.random() < 0.2 else 40          # mostly behind the header text, sometimes inside the header itself
        if kind == 0:
            for _ in range(int(rng.integers(1, 8))): bad[int(rng.integers(lo, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            a = int(rng.integers(lo, len(bad) - 8)); bad[a:a + 4] = struct.pack("<i", int(rng.choice([-1, 0, 1, 2 ** 31 - 1, -2 ** 31, 70000, 10 ** 9])))   # a length / count field, perhaps
        elif kind == 2: bad = bad[:int(rng.integers(lo, len(bad)))]
        else:
            a = int(rng.integers(lo, len(bad) - 2)); bad[a:a + int(rng.integers(1, 60))] = rng.integers(0, 256, int(rng.integers(0, 60)), dtype=np.uint8).tobytes()
        path = os.path.join(d, "b.bam")
        with open(path, "wb") as fh:
            bs = int(rng.choice([700, 4000, 60000]))
            for at in range(0, len(bad), bs): fh.write(bamwriter.bgzf_block(bytes(bad[at:at + bs])))
            fh.write(bamwriter.bgzf_block(b""))
        os.environ["UVCIO_SERIAL_WALK"] = str(rng.choice(["", "1"])); os.environ["UVCIO_THREADS"] = str(rng.choice(["1", "4"]))
        if not os.environ["UVCIO_SERIAL_WALK"]: os.environ.pop("UVCIO_SERIAL_WALK")
        try:
            b = uio.Bam(path)
            for tid, qb, qe in ((0, 0, 40000), (0, 21000, 21500)):
                got = b.fetch(tid, qb, qe); n_read += int(got["n_alns"])
            b.close()
        except (uio.UvcIoError if hasattr(uio, "UvcIoError") else Exception):   # refused: fine
            n_refused += 1
        n_files += 1
print("BAM damage fuzz: %d damaged files, %d refused, %d alignments returned from the others, no sanitizer report, %.0f s" % (n_files, n_refused, n_read, time.time() - t0))
