"""libuvcio.so (BGZF / BAM / BAI / FASTA readers written from the SAM specification) against files produced by the independent
Python writer of tests/bamwriter.py: every column round-trips, region queries return exactly the overlapping alignments in
file order (the sam_itr_queryi contract), with and without an index.  No GPU."""
import numpy as np
import pytest

from uvc_amd import io as uio, synth
import bamwriter


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("io")
    reads = synth.generate_region(seed=31, region_len=40000, depth=25, beg=100000, indel_every=700, clip_frac=0.05)
    recs = bamwriter.records_from_reads(reads, tid=1)
    # a second reference in front (tid 0) with a few reads, an unmapped mate placed on tid 1, a read with many aux fields, a long name
    extra = [dict(tid=0, pos=50 + 10 * k, qname="first%d" % k, flag=0, mapq=30, cigar=[(0, 20)], bases=[k % 4] * 20, quals=[30] * 20, nm=1) for k in range(40)]
    recs.append(dict(tid=1, pos=120000, qname="unmapped_mate", flag=0x4 | 0x1 | 0x80, mapq=0, cigar=[], bases=[0, 1, 2, 3, 4], quals=[2] * 5, mtid=1, mpos=120000, tlen=0))
    recs.append(dict(tid=1, pos=120001, qname="q" * 200, flag=0, mapq=60, cigar=[(4, 3), (0, 30), (1, 2), (0, 10), (2, 4), (0, 5), (5, 7)], bases=list(np.arange(50) % 5), quals=list(np.arange(50) % 42),
                     nm=300, aux=b"XAZhello\0XBBc" + (3).to_bytes(4, "little") + b"\x01\x02\x03XFf" + bytes(4) + b"XSs" + (-5).to_bytes(2, "little", signed=True)))
    recs = extra + sorted(recs, key=lambda r: r["pos"])
    refs = [("chrA", 5000), ("chr20", 200000)]
    bamwriter.write_bam(str(d / "t.bam"), refs, recs, block_bytes=20000)
    bamwriter.write_bam(str(d / "noidx.bam"), refs, recs, block_bytes=50000, with_index=False)
    bamwriter.write_bam(str(d / "packed.bam"), refs, recs, block_bytes=3000, packed=True)   # records straddle the BGZF blocks (not what htslib writes)
    rng = np.random.default_rng(1)
    seqs = [("chrA", "".join("acgtN"[i] for i in rng.integers(0, 5, 5000))), ("chr20", "".join("ACGT"[i] for i in rng.integers(0, 4, 200000)))]
    bamwriter.write_fasta(str(d / "ref.fa"), seqs, width=70)
    return d, refs, recs, seqs


def expected(recs, tid, beg, end):
    out = []
    for r in recs:
        e = r["pos"] + sum(l for o, l in r["cigar"] if o in (0, 2, 3, 7, 8))
        if e == r["pos"]: e += 1
        if r["tid"] == tid and r["pos"] < end and e > beg: out.append((r, e))
    return out


@pytest.mark.parametrize("batch", [None, "65536", "70001"])
@pytest.mark.parametrize("walk", ["blocks", "serial"])
@pytest.mark.parametrize("name", ["t.bam", "noidx.bam", "packed.bam"])
def test_fetch_equals_the_overlap_definition(files, name, batch, walk, monkeypatch):
    """t.bam / noidx.bam: every BGZF block begins with a record (htslib's layout): the reader walks the blocks in parallel; packed.bam: records
    straddle the blocks, the parallel walk notices and the sequential one takes over; UVCIO_SERIAL_WALK forces the sequential one."""
    d, refs, recs, _ = files
    if batch: monkeypatch.setenv("UVCIO_BATCH_BYTES", batch)     # several batches per query: records and blocks straddle them
    else: monkeypatch.delenv("UVCIO_BATCH_BYTES", raising=False)
    if walk == "serial": monkeypatch.setenv("UVCIO_SERIAL_WALK", "1")
    else: monkeypatch.delenv("UVCIO_SERIAL_WALK", raising=False)
    monkeypatch.setenv("UVCIO_THREADS", "4")
    b = uio.Bam(str(d / name))
    assert b.refs == refs and b.has_index == (name != "noidx.bam")
    for tid, beg, end in [(1, 100000, 140000), (1, 118000, 121000), (1, 0, 100001), (1, 139990, 200000), (0, 0, 5000), (0, 100, 101), (1, 16384 * 7, 16384 * 7 + 1), (1, 150000, 160000)]:
        got = b.fetch(tid, beg, end)
        want = expected(recs, tid, beg, end)
        assert got["n_alns"] == len(want), (tid, beg, end)
        for i, (r, e) in enumerate(want):
            assert (got["pos"][i], got["endpos"][i], got["flag"][i], got["mapq"][i], got["qnames"][i]) == (r["pos"], e, r["flag"], r["mapq"], r["qname"])
            assert (got["mtid"][i], got["mpos"][i], got["isize"][i]) == (r.get("mtid", -1), r.get("mpos", -1), r.get("tlen", 0))
            nm = r.get("nm")
            assert got["nm"][i] == (nm if nm is not None and nm >= 0 else -1)
            so, lq = int(got["seq_off"][i]), int(got["l_qseq"][i])
            assert lq == len(r["bases"]) and list(got["bases"][so:so + lq]) == [int(x) for x in r["bases"]] and list(got["quals"][so:so + lq]) == [int(x) for x in r["quals"]]
            co, nc = int(got["cigar_off"][i]), int(got["n_cigar"][i])
            assert [(int(c) & 0xF, int(c) >> 4) for c in got["cigars"][co:co + nc]] == list(r["cigar"])
    b.close()


def test_readers_on_several_threads_share_the_pool(files, monkeypatch):
    """uvc1-mi355x keeps several tiles in flight, one reader each; their inflate / walk / decode slices go through one process-wide pool
    and every caller works on the queue (its own slices or another reader's) while it waits.  Same columns as a reader that is alone."""
    import threading
    d, refs, recs, _ = files
    monkeypatch.setenv("UVCIO_BATCH_BYTES", "65536")
    queries = [(1, 100000, 140000), (1, 118000, 121000), (0, 0, 5000), (1, 139990, 200000)]
    keys = ("pos", "endpos", "flag", "mapq", "isize", "seq_off", "l_qseq", "bases", "quals", "cigars")
    solo = uio.Bam(str(d / "t.bam"))
    want = [{k: np.array(g[k]).copy() for k in keys} for g in (solo.fetch(*q) for q in queries)]
    solo.close()
    bad = []
    def work(seed):
        b = uio.Bam(str(d / ("t.bam" if seed % 2 else "packed.bam")))
        for rep in range(6):
            qi = (seed + rep) % len(queries)
            g = b.fetch(*queries[qi])
            for k in keys:
                if not np.array_equal(np.array(g[k]), want[qi][k]): bad.append((seed, rep, k))
        b.close()
    th = [threading.Thread(target=work, args=(s,)) for s in range(6)]
    for x in th: x.start()
    for x in th: x.join()
    assert not bad, bad[:5]


def test_fasta_fetch(files):
    d, _, _, seqs = files
    f = uio.Fasta(str(d / "ref.fa"))
    assert f.seq_len("chr20") == 200000 and f.seq_len("nope") == -1
    for name, s in seqs:
        for beg, end in [(0, 1), (0, 70), (69, 71), (1234, 4321), (len(s) - 5, len(s)), (0, len(s))]:
            assert f.fetch(name, beg, end) == s[beg:end].upper()
    with pytest.raises(IOError):
        f.fetch("chrA", 10, 6000)
    f.close()


def test_bad_files_are_refused(tmp_path):
    p = tmp_path / "x.bam"
    p.write_bytes(b"not a bam file at all")
    with pytest.raises(IOError):
        uio.Bam(str(p))
    with pytest.raises(IOError):
        uio.Fasta(str(tmp_path / "missing.fa"))


# ---- BGZF writer and the region planner ----
def test_bgzf_writer_blocks_and_eof_marker(tmp_path):
    import gzip
    import struct
    from uvc_amd import io as uio
    rng = np.random.default_rng(5)
    payload = b"".join(b"chr20\t%d\t.\tA\tC\t%d\n" % (i, rng.integers(0, 99)) for i in range(40000)) + bytes(rng.integers(0, 256, 200000, dtype=np.uint8))
    path = str(tmp_path / "out.vcf.gz")
    with uio.BgzfWriter(path) as w:
        for at in range(0, len(payload), 77777):     # writes that do not line up with block boundaries
            w.write(payload[at:at + 77777])
    raw = open(path, "rb").read()
    assert gzip.decompress(raw) == payload          # a BGZF file is a multi-member gzip file
    assert raw[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")   # SAMv1 4.1.2 end-of-file marker
    at, sizes = 0, []
    while at < len(raw):                              # every member carries the BC extra field with its own size
        assert raw[at:at + 4] == b"\x1f\x8b\x08\x04" and raw[at + 12:at + 16] == b"BC\x02\x00"
        bsize = struct.unpack("<H", raw[at + 16:at + 18])[0] + 1
        isize = struct.unpack("<I", raw[at + bsize - 4:at + bsize])[0]
        assert isize <= 0xff00
        sizes.append(isize)
        at += bsize
    assert at == len(raw) and sum(sizes) == len(payload) and sizes[-1] == 0 and all(s == 0xff00 for s in sizes[:-2])


def _plan_py(tid, pos, endpos, flag, tlen, nthreads, mem):
    """An independent restatement of SamIter::iternext (no BED), written as the reference's control flow: one generator turn per iternext call."""
    B_POS, B_READ, STR, UNITS = 8192, 512, 100, 8
    cuts, state = [], dict(tid=-1, beg=-1, end=-1)
    i, n, batch = 0, len(tid), 0
    done = (n == 0)
    while not done:
        tot = [0, 0, 0, 0]
        rr = rp = rpa = 0
        cur, early = -1, False
        while True:
            ret = 0 if i < n else -1
            if ret >= 0:
                cur = i; i += 1
            if cur < 0:
                break
            if not (flag[cur] & 4):
                ct, cb, ce = int(tid[cur]), int(pos[cur]), int(endpos[cur])
                memfree = (1024 * 1024 // UNITS) * mem
                used = rr * B_READ + (rp + rpa) * (B_POS + 1024)
                ovl = min(max(max(state["end"], 0) - cb, 0), 150)
                sub = used > memfree + memfree * ovl // 150
                changed = ct != state["tid"]
                far = (not changed) and state["end"] + 2 * STR < cb
                rflag = 16 * changed + 8 * far + 4 * sub + 2 * (ret == -1)
                if rflag:
                    first = state["tid"] == -1
                    norm_end = min(state["end"], (2 ** 31 - 1) if first else int(tlen[state["tid"]]))
                    if not first and state["beg"] < norm_end:
                        cuts.append(dict(tid=state["tid"], beg=state["beg"], end=norm_end, flag=rflag, batch=batch, n_reads=rr))
                        s = rp + rpa
                        tot = [tot[0] + rr, tot[1] + rr * rr, tot[2] + s, tot[3] + s * s]
                        rr = rp = rpa = 0
                    state["tid"] = ct
                    state["beg"] = cb if changed else max(max(state["beg"], cb), norm_end)
                    by_reads = min(tot[1] // max(1, tot[0]) * nthreads, tot[0]) * B_READ
                    by_pos = (min(tot[3] // max(1, tot[2]) * nthreads, tot[2]) + 2 * STR * nthreads) * B_POS
                    if by_reads + by_pos + tot[2] * 1024 > 1024 * 1024 * mem * nthreads:
                        state["end"] = max(state["beg"], norm_end)
                        early = True
                        break
                if changed:
                    state["beg"], state["end"] = cb, ce
                    rpa += rp
                else:
                    state["end"] = max(state["end"], ce)
                rr += 1
                rp = state["end"] - state["beg"]
            if ret < 0:
                break
        done = not early
        batch += 1
    return cuts


@pytest.mark.parametrize("seed,mem,nthreads", [(1, 1536, 1), (2, 1, 1), (3, 1, 4), (4, 2, 2), (5, 1536, 8)])
def test_region_planner_follows_samiter(seed, mem, nthreads):
    from uvc_amd import io as uio
    rng = np.random.default_rng(seed)
    tlen = [50000, 8000, 120000]
    tid, pos = [], []
    for t, L in enumerate(tlen):
        at = int(rng.integers(0, 300))
        while at < L - 200:
            if rng.random() < 0.002:
                at += int(rng.integers(150, 2500))          # coverage gaps on both sides of the 200 bp threshold
            else:
                at += int(rng.integers(0, 3))
            if at < L - 200:
                tid.append(t); pos.append(at)
    tid, pos = np.array(tid, dtype=np.int32), np.array(pos, dtype=np.int32)
    endpos = pos + rng.integers(30, 151, len(pos)).astype(np.int32)
    flag = np.where(rng.random(len(pos)) < 0.01, 4, 0).astype(np.uint16)   # a few unmapped mates placed on the contig
    got = uio.plan_regions(tid, pos, endpos, flag, tlen, nthreads=nthreads, mem_per_thread_mb=mem)
    want = _plan_py(tid, pos, endpos, flag, tlen, nthreads, mem)
    assert got == want
    # the streaming form (what uvc1-mi355x feeds window by window): the same cuts whatever the piece size, also when a piece ends on the record that
    # closes a batch or on an unmapped one
    for piece in (1, 7, 1000, len(tid) + 5):
        assert uio.plan_regions_stream(tid, pos, endpos, flag, tlen, nthreads=nthreads, mem_per_thread_mb=mem, piece=piece) == want, piece
    for cutoff in (len(tid) - 1, len(tid) // 2, 3):   # files that end elsewhere (the end-of-file step runs on the record the last call holds)
        w2 = _plan_py(tid[:cutoff], pos[:cutoff], endpos[:cutoff], flag[:cutoff], tlen, nthreads, mem)
        assert uio.plan_regions_stream(tid[:cutoff], pos[:cutoff], endpos[:cutoff], flag[:cutoff], tlen, nthreads=nthreads, mem_per_thread_mb=mem, piece=11) == w2, cutoff
    assert len(got) >= 3 and {16, 8} <= {c["flag"] & 24 for c in got} | {c["flag"] & 16 for c in got} | {c["flag"] & 8 for c in got}
    if mem == 1:
        assert any(c["flag"] & 4 for c in got) and max(c["batch"] for c in got) > 0   # the memory model cuts blocks and closes batches
    for a, b in zip(got, got[1:]):      # blocks are disjoint and ordered
        assert (a["tid"], a["end"]) <= (b["tid"], b["beg"]) or a["tid"] < b["tid"]
    mapped = (flag & 4) == 0
    assert sum(c["n_reads"] for c in got) <= int(mapped.sum())


def test_tumor_vcf_reader(tmp_path):
    """uvcio_tumor_vcf_*: rescue_variants_from_vcf (main.cpp:183-398) on text.  Keys: symbolpos = POS - 1 for substitutions and the two
    position-level line types, POS for InDels (:279); symbolic ALTs other than <NON_REF> / <ADDITIONAL_INDEL_CANDIDATE> are skipped (:265-272),
    lines without FORMAT/VTI too (:275); the integers are the ones :294-372 read."""
    fmt = "GT:VTI:BDPb:bDPf:bDPr:CDP1x:cDP1x:cVQ1:cPCQ1:CDP2x:cDP2x:cVQ2:cPCQ2:bNMQ:vHGQ:CDP1b:cDP1f:cDP1r:CDP2b"
    def smp(vti, k):
        return "./1:%s:%d,%d:9,%d:8,%d:%d:100,%d:50,%d:60,%d:%d:10,%d:40,%d:45,%d:30,%d:%d:70,%d:20,%d:21,%d:5,%d" % (
            vti, 100 + k, 90 + k, 3 + k, 4 + k, 9000 + k, 300 + k, 31 + k, 32 + k, 800 + k, 30 + k, 41 + k, 42 + k, 17 + k, 55 + k, 60 + k, 6 + k, 7 + k, 1 + k)
    lines = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tTUMOR1",
             "chrB\t101\t.\tA\tG\t50\tPASS\tANY_VAR\t" + fmt + "\t" + smp("0,2", 0),
             "chrA\t501\t.\tC\tT\t50\tPASS\tANY_VAR\t" + fmt + ":_C2XP\t" + smp("1,3", 1) + ":x",
             "chrA\t700\t.\tGAC\tG\t50\tPASS\tANY_VAR\t" + fmt + "\t" + smp("6,8", 2),          # deletion of 2: symbolpos = POS
             "chrA\t700\t.\tG\tGTTT\t50\tPASS\tANY_VAR\t" + fmt + "\t" + smp("6,10", 3),        # insertion of 3 at the same place
             "chrA\t800\t.\tG\t<LD3P>\t50\tPASS\tANY_VAR\t" + fmt + "\t" + smp("6,7", 4),       # symbolic: skipped
             "chrA\t1001\t.\tT\t<NON_REF>\t.\t.\tMGVCF_BLOCK\tGT:VTI:POS_VT_BDP_CDP_HomRefQ\t.:3,15:1000,2,.,5,5,5,30,.,2001",
             "chrA\t1200\t.\tT\t<ADDITIONAL_INDEL_CANDIDATE>\t.\t.\tADDITIONAL_INDEL_CANDIDATE;RU=A;RC=9\tGT:VTI:clipDP\t.:3,16:40,12",
             "chrA\t1300\t.\tT\tA\t50\tPASS\tANY_VAR\tGT:DP\t./1:5",                           # no VTI: skipped
             "chrZ\t5\t.\tT\tA\t50\tPASS\tANY_VAR\t" + fmt + "\t" + smp("3,0", 5)]              # contig the BAM does not have
    path = str(tmp_path / "t.vcf.gz")
    w = uio.BgzfWriter(path); w.write("\n".join(lines) + "\n"); w.close()
    T = uio.TumorVcf(path, ["chrA", "chrB"])
    assert T.sample == "TUMOR1" and T.n_records == 6
    keys, cols = T.fetch(0, 0, 10 ** 9)
    got = [(k.refpos, k.symbol) for k in keys]
    assert got == [(500, 3), (700, 8), (700, 10), (1000, 15), (1199, 16)]
    k = keys[0]
    assert (k.BDP, k.bDP, k.CDP1x, k.cDP1x, k.cVQ1, k.cPCQ1, k.CDP2x, k.cDP2x, k.cVQ2, k.cPCQ2, k.bNMQ, k.vHGQ, k.tDP, k.tAD0, k.tAD1, k.t2DP, k.tier2, k.indel_len) == \
        (101 + 91, 4 + 5, 9001, 301, 32, 33, 801, 31, 42, 43, 18, 56, 61 + 70, 20 + 21, 7 + 8, 5 + 2, 1, 0)
    assert (keys[1].indel_len, keys[2].indel_len, keys[1].tier2) == (2, 3, 0) and cols[1] == smp("6,8", 2) and cols[3].startswith(".:3,15:")
    assert [(k.refpos, k.symbol) for k in T.fetch(0, 700, 1000)[0]] == [(700, 8), (700, 10), (1000, 15)]
    assert T.fetch(0, 701, 999) == (None, []) and [(k.refpos, k.symbol) for k in T.fetch(1, 0, 200)[0]] == [(100, 2)]
    T2 = uio.TumorVcf(path, ["chrA", "chrB"], is_tumor_format_retrieved=False)   # then the two position-level line types are dropped as well
    assert T2.n_records == 4
    T.close(); T2.close()
    bad = str(tmp_path / "bad.vcf")
    open(bad, "w").write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS\nchrA\t5\t.\tA\tC\t1\t.\t.\tGT:VTI:BDPb\t./1:0,1:7\n")
    with pytest.raises(IOError):
        uio.TumorVcf(bad, ["chrA"])


def test_crc32_equals_zlib():
    """uvcio_crc32 (carry-less-multiplication folding where the CPU has it, zlib's table walk otherwise) against zlib.crc32 at every
    length around the 16- and 64-byte steps of the folding loop and at BGZF block sizes."""
    import ctypes
    import zlib
    dll = uio.dll()
    dll.uvcio_crc32.restype = ctypes.c_uint32
    dll.uvcio_crc32.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    rng = np.random.default_rng(0)
    for n in list(range(1, 200)) + [255, 256, 1000, 4095, 4096, 0xff00, 0xff01, 100003]:
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert dll.uvcio_crc32(b, n) == zlib.crc32(b), n


def test_fast_inflate_equals_zlib_on_every_block_type():
    """The library's own DEFLATE decoder (uvc_inflate_fast.h: what the BGZF reader tries before zlib) against zlib: stored, fixed and dynamic
    blocks, every strategy and level, literal-heavy and match-heavy data, empty and one-byte streams, BGZF-sized blocks.  A stream it
    declines is fine (the reader falls back); a stream it decodes must be byte-identical.  Corrupted streams must be declined or decoded
    without touching memory outside the output."""
    import ctypes
    import zlib
    dll = uio.dll()
    f = dll.uvcio_inflate_raw_fast
    f.restype, f.argtypes = ctypes.c_int, [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64]
    rng = np.random.default_rng(5)

    def make(kind, n):
        if kind == 0: return rng.integers(0, 256, n, dtype=np.uint8).tobytes()                       # incompressible
        if kind == 1: return rng.choice(np.frombuffer(b"ACGT", np.uint8), n).tobytes()               # 2 bits per byte
        if kind == 2: return (b"read_name_0123456789:" * (n // 21 + 1))[:n]                          # long matches
        if kind == 3: return (rng.integers(0, 41, n, dtype=np.uint8) + 33).tobytes()                 # quality strings: literals
        return b"\x00" * n                                                                           # one run
    decoded = total = 0
    for trial in range(120):
        n = int(rng.choice([0, 1, 2, 7, 100, 1000, 5000, 20000, 65280]))
        data = make(trial % 5, n)
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
                comp = co.compress(data) + co.flush()
                out = ctypes.create_string_buffer(max(n, 1) + 16)
                out.raw = b"\xAA" * (max(n, 1) + 16)
                total += 1
                if f(comp, len(comp), out, n):
                    decoded += 1
                    assert out.raw[:n] == data, (trial, n, level, strategy)
                    assert out.raw[n:] == b"\xAA" * (len(out.raw) - n)                               # nothing behind the block
    assert decoded == total                                                                          # zlib's own output: nothing to decline
    for trial in range(400):                                                                         # bit flips: no crash, no overrun
        data = make(trial % 5, int(rng.choice([100, 5000])))
        comp = bytearray(zlib.compress(data, 6)[2:-4])
        for _ in range(int(rng.integers(1, 5))):
            comp[int(rng.integers(0, len(comp)))] ^= 1 << int(rng.integers(0, 8))
        out = ctypes.create_string_buffer(len(data) + 16)
        out.raw = b"\xAA" * (len(data) + 16)
        f(bytes(comp), len(comp), out, len(data))
        assert out.raw[len(data):] == b"\xAA" * 16
