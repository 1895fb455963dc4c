#!/usr/bin/env python3
"""CPU helper (not a test of the suite): the FASTA / .fai reader of libuvcio.so under AddressSanitizer -- intact files against Python slicing on random
ranges and line widths, then damaged .fai entries (lengths, offsets, line widths made huge / negative / zero / non-numeric, fields dropped) and
truncated sequence files: refuse or read, never outside the buffers.    python3 scripts/cpu_fuzz_fasta.py SECONDS [SEED]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from uvc_amd import io as uio  # noqa: E402
import bamwriter  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0, n_ok, n_q, n_bad, n_refused = time.time(), 0, 0, 0, 0
with tempfile.TemporaryDirectory() as d:
    while time.time() - t0 < budget:
        seqs = [("c%d" % i, "".join("ACGTNacgtn"[j] for j in rng.integers(0, 10, int(rng.choice([1, 59, 60, 61, 1000, 30000]))))) for i in range(int(rng.integers(1, 4)))]
        path = os.path.join(d, "r.fa")
        bamwriter.write_fasta(path, seqs, width=int(rng.choice([1, 7, 60, 70, 100000])))
        F = uio.Fasta(path)
        for name, s in seqs:
            assert F.seq_len(name) == len(s)
            for _ in range(5):
                a = int(rng.integers(0, len(s))); b = min(len(s), a + int(rng.choice([1, 10, 200, 100000])))
                assert F.fetch(name, a, b) == s[a:b].upper(), (name, a, b); n_q += 1     # the reader hands the bases over in upper case
        F.close(); n_ok += 1
        # damage
        fai = open(path + ".fai").read().splitlines()
        i = int(rng.integers(0, len(fai))); c = fai[i].split("\t")
        kind = int(rng.integers(0, 4))
        if kind == 0: c[int(rng.integers(1, 5))] = str(rng.choice(["-1", "0", "99999999999999", "4294967296", "x", ""]))
        elif kind == 1: c.pop(int(rng.integers(0, len(c))))
        elif kind == 2: open(path, "r+").truncate(int(rng.integers(0, os.path.getsize(path))))
        else: c[2] = str(int(c[2]) + int(rng.integers(-50, 5000)))
        fai[i] = "\t".join(c)
        open(path + ".fai", "w").write("\n".join(fai) + "\n")
        try:
            F = uio.Fasta(path)
            for name, s in seqs:
                L = F.seq_len(name)
                if L > 0 and L < 10 ** 8:
                    a = int(rng.integers(0, L)); F.fetch(name, a, min(L, a + 500))
            F.close()
        except (IOError, UnicodeError, ValueError):
            n_refused += 1
        n_bad += 1
print("FASTA fuzz: %d intact files (%d ranges equal), %d damaged (%d refused), no sanitizer report, %.0f s" % (n_ok, n_q, n_bad, n_refused, time.time() - t0))
