#!/usr/bin/env python3
"""CPU helper (not a test of the suite): the tumor-VCF reader of libuvcio.so (rescue_variants_from_vcf, main.cpp:183-398, on text) on damaged text
under AddressSanitizer: fields dropped, swapped, emptied, made huge or non-numeric, lines cut, bytes flipped.  It must refuse the file or read it.
    python3 scripts/cpu_fuzz_tumor_vcf.py SECONDS [SEED]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from uvc_amd import io as uio  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
fmt = "GT:VTI:BDPb:bDPf:bDPr:CDP1x:cDP1x:cVQ1:cPCQ1:CDP2x:cDP2x:cVQ2:cPCQ2:bNMQ:vHGQ:CDP1b:cDP1f:cDP1r:CDP2b"


def smp(vti, k):
    return "./1:%s:%d,%d:9,%d:8,%d:%d:100,%d:50,%d:60,%d:%d:10,%d:40,%d:45,%d:30,%d:%d:70,%d:20,%d:21,%d:5,%d" % (
        vti, 100 + k, 90 + k, 3 + k, 4 + k, 9000 + k, 300 + k, 31 + k, 32 + k, 800 + k, 30 + k, 41 + k, 42 + k, 17 + k, 55 + k, 60 + k, 6 + k, 7 + k, 1 + k)


def good_lines(n):
    out = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tTUMOR1"]
    pos = 1
    for k in range(n):
        pos += int(rng.integers(0, 50))
        kind = int(rng.integers(0, 6))
        if kind == 0: out.append("chrA\t%d\t.\tC\tT\t50\tPASS\tANY_VAR\t%s\t%s" % (pos, fmt, smp("1,3", k)))
        elif kind == 1: out.append("chrA\t%d\t.\tGAC\tG\t50\tPASS\tANY_VAR\t%s:_C2XP\t%s:x" % (pos, fmt, smp("6,8", k)))
        elif kind == 2: out.append("chrA\t%d\t.\tG\tGTTT\t50\tPASS\tANY_VAR\t%s\t%s" % (pos, fmt, smp("6,10", k)))
        elif kind == 3: out.append("chrA\t%d\t.\tT\t<NON_REF>\t.\t.\tMGVCF_BLOCK\tGT:VTI:POS_VT_BDP_CDP_HomRefQ\t.:3,15:1000,2,.,5,5,5,30,.,2001" % pos)
        elif kind == 4: out.append("chrA\t%d\t.\tT\t<ADDITIONAL_INDEL_CANDIDATE>\t.\t.\tADDITIONAL_INDEL_CANDIDATE;RU=A;RC=9\tGT:VTI:clipDP\t.:3,16:40,12" % pos)
        else: out.append("chrB\t%d\t.\tA\tG\t50\tPASS\tANY_VAR\t%s\t%s" % (pos, fmt, smp("0,2", k)))
    return out


t0, n_files, n_refused, n_keys = time.time(), 0, 0, 0
with tempfile.TemporaryDirectory() as d:
    while time.time() - t0 < budget:
        lines = good_lines(int(rng.integers(1, 40)))
        for _ in range(int(rng.integers(1, 6))):
            i = int(rng.integers(0, len(lines)))
            c = lines[i].split("\t")
            kind = int(rng.integers(0, 8))
            if kind == 0 and len(c) > 1: c.pop(int(rng.integers(0, len(c))))
            elif kind == 1: c[int(rng.integers(0, len(c)))] = ""
            elif kind == 2: c[int(rng.integers(0, len(c)))] = str(rng.choice(["99999999999999999999", "-1", "1e9", "nan", ":", ",,,", "0x10", "\x00", "." * 300]))
            elif kind == 3 and len(c) >= 10:
                f = c[9].split(":"); f[int(rng.integers(0, len(f)))] = str(rng.choice(["", ",", "a,b", "1," * 50, "-", "4294967296,1"])); c[9] = ":".join(f)
            elif kind == 4 and len(c) >= 9:
                f = c[8].split(":"); rng.shuffle(f); c[8] = ":".join(f)
            elif kind == 5: lines[i] = lines[i][:int(rng.integers(0, len(lines[i]) + 1))]; continue
            elif kind == 6:
                b = bytearray(lines[i].encode("latin1"))
                if b: b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
                lines[i] = b.decode("latin1"); continue
            else: lines.insert(i, lines[i])
            lines[i] = "\t".join(c)
        path = os.path.join(d, "t.vcf.gz" if rng.random() < 0.5 else "t.vcf")
        text = ("\n".join(lines) + ("\n" if rng.random() < 0.8 else "")).encode("latin1")
        if path.endswith(".gz"):
            w = uio.BgzfWriter(path); w.write(text.decode("latin1")); w.close()
        else:
            open(path, "wb").write(text)
        try:
            T = uio.TumorVcf(path, ["chrA", "chrB"], is_tumor_format_retrieved=bool(rng.integers(0, 2)))
            for tid in (0, 1):
                keys, cols = T.fetch(tid, 0, 10 ** 9)
                n_keys += len(keys) if keys is not None else 0
            T.close()
        except (IOError, UnicodeError, ValueError):
            n_refused += 1
        n_files += 1
print("tumor VCF fuzz: %d damaged files, %d refused, %d keys read from the others, no sanitizer report, %.0f s" % (n_files, n_refused, n_keys, time.time() - t0))
