#!/usr/bin/env python3
"""CPU helper (not a test of the suite): the BAM reader of libuvcio.so against files written by tests/bamwriter.py -- random records, block sizes,
packed / block-aligned layouts, with and without index, batch sizes, both record walks, random region queries: every column of every overlapping
alignment, in file order (the check of tests/test_io.py::test_fetch_equals_the_overlap_definition).   python3 scripts/cpu_soak_bam.py SECONDS [FIRST_SEED]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from uvc_amd import io as uio, synth  # noqa: E402
import bamwriter  # noqa: E402
from test_io import expected  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0, n_ok, n_q, fails = time.time(), 0, 0, []
with tempfile.TemporaryDirectory() as d:
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        L, beg = int(rng.choice([3000, 20000, 60000])), int(rng.choice([0, 16000, 100000]))
        reads = synth.generate_region(seed=seed, region_len=L, depth=int(rng.choice([5, 25, 80])), beg=beg + 100, indel_every=int(rng.choice([300, 700])), clip_frac=float(rng.choice([0.0, 0.05, 0.3])),
                                      umi=bool(rng.integers(0, 2)))
        recs = bamwriter.records_from_reads(reads, tid=1)
        extra = [dict(tid=0, pos=50 + 10 * k, qname="first%d" % k, flag=0, mapq=30, cigar=[(0, 20)], bases=[k % 4] * 20, quals=[30] * 20, nm=1) for k in range(int(rng.integers(0, 60)))]
        if rng.random() < 0.5:
            p = beg + 100 + int(rng.integers(0, L))
            recs.append(dict(tid=1, pos=p, qname="unmapped_mate", flag=0x4 | 0x1 | 0x80, mapq=0, cigar=[], bases=[0, 1, 2, 3, 4], quals=[2] * 5, mtid=1, mpos=p, tlen=0))
            recs.append(dict(tid=1, pos=p + 1, qname="q" * int(rng.integers(1, 250)), flag=0, mapq=60, cigar=[(4, 3), (0, 30), (1, 2), (0, 10), (2, 4), (0, 5), (5, 7)], bases=list(np.arange(50) % 5),
                             quals=list(np.arange(50) % 42), nm=300, aux=b"XAZhello\0XBBc" + (3).to_bytes(4, "little") + b"\x01\x02\x03XFf" + bytes(4) + b"XSs" + (-5).to_bytes(2, "little", signed=True)))
        recs = extra + sorted(recs, key=lambda r: r["pos"])
        refs = [("chrA", 5000), ("chr20", beg + L + 5000)]
        packed, with_index = bool(rng.integers(0, 2)), bool(rng.integers(0, 4) > 0)
        path = os.path.join(d, "s.bam")
        for stale in (path + ".bai", path[:-4] + ".bai"):
            if os.path.exists(stale): os.remove(stale)
        bamwriter.write_bam(path, refs, recs, block_bytes=int(rng.choice([700, 3000, 20000, 60000])), with_index=with_index, packed=packed)
        for k, v in (("UVCIO_BATCH_BYTES", rng.choice(["", "65536", "70001", "300000"])), ("UVCIO_SERIAL_WALK", rng.choice(["", "1"])), ("UVCIO_ZLIB", rng.choice(["", "1"])), ("UVCIO_THREADS", rng.choice(["1", "4"]))):
            if v: os.environ[k] = str(v)
            else: os.environ.pop(k, None)
        try:
            b = uio.Bam(path)
            assert b.refs == refs and b.has_index == with_index
            for _ in range(6):
                tid = int(rng.integers(0, 2))
                qb = int(rng.integers(0, refs[tid][1])); qe = min(refs[tid][1], qb + int(rng.choice([1, 100, 5000, 100000])))
                got = b.fetch(tid, qb, qe)
                want = expected(recs, tid, qb, qe)
                assert got["n_alns"] == len(want), ("count", tid, qb, qe, got["n_alns"], len(want))
                for i, (r, e) in enumerate(want):
                    assert (got["pos"][i], got["endpos"][i], got["flag"][i], got["mapq"][i], got["qnames"][i]) == (r["pos"], e, r["flag"], r["mapq"], r["qname"]), ("fields", i)
                    assert (got["mtid"][i], got["mpos"][i], got["isize"][i]) == (r.get("mtid", -1), r.get("mpos", -1), r.get("tlen", 0)), ("mate", i)
                    nm = r.get("nm")
                    assert got["nm"][i] == (nm if nm is not None and nm >= 0 else -1), ("nm", i)
                    so, lq = int(got["seq_off"][i]), int(got["l_qseq"][i])
                    assert lq == len(r["bases"]) and list(got["bases"][so:so + lq]) == [int(x) for x in r["bases"]] and list(got["quals"][so:so + lq]) == [int(x) for x in r["quals"]], ("bases", i)
                    co, nc = int(got["cigar_off"][i]), int(got["n_cigar"][i])
                    assert [(int(c) & 0xF, int(c) >> 4) for c in got["cigars"][co:co + nc]] == list(r["cigar"]), ("cigar", i)
                n_q += 1
            b.close()
            n_ok += 1
        except (AssertionError, Exception) as e:   # noqa: BLE001
            fails.append(seed); print("FAIL seed", seed, dict(L=L, beg=beg, packed=packed, with_index=with_index, env={k: os.environ.get(k) for k in ("UVCIO_BATCH_BYTES", "UVCIO_SERIAL_WALK", "UVCIO_ZLIB", "UVCIO_THREADS")}), repr(e)[:400], flush=True)
        seed += 1
print("BAM reader soak: %d files (%d queries) equal, %d FAILED %s in %.0f s" % (n_ok, n_q, len(fails), fails[:20], time.time() - t0))
sys.exit(1 if fails else 0)
