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
@pytest.mark.parametrize("name", ["t.bam", "noidx.bam"])
def test_fetch_equals_the_overlap_definition(files, name, batch, monkeypatch):
    d, refs, recs, _ = files
    if batch: monkeypatch.setenv("UVCIO_BATCH_BYTES", batch)     # several batches per query: records and blocks straddle them
    else: monkeypatch.delenv("UVCIO_BATCH_BYTES", raising=False)
    b = uio.Bam(str(d / name))
    assert b.refs == refs and b.has_index == (name == "t.bam")
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
