"""Region boundaries: who scores a zerobased_pos that two adjacent regions share, regions that start at contig position 0, and the
MGVCF block that opens at the begin of a BED line.

Reference behaviour (main.cpp:608-656): zerobased_pos runs over [rpos_inclu_beg, rpos_exclu_end] inclusive, the BASE sub-position of
the first is skipped, an MGVCF block opens where refpos % 1000 == 0 or refpos == incluBegPosition.  Two adjacent regions therefore both
write the LINK records of their common end point.  The tilers of this repository give every zerobased_pos one owner instead
(UvcScoreRequest::base_at_pos_beg): the -m "not gpu" tests pin that on the oracle, the -m gpu tests compare the HIP library with it.
"""
import numpy as np
import pytest

from uvc_amd import io as uio, pipeline, region, synth
from util import run_region

IDENT = ("refpos", "symbol", "refsymbol", "DP", "AD", "bDP", "bAD", "bDPa", "cDP0a", "gapSa_len", "keep", "out", "FILTER", "cVQ1", "cVQ2", "TLODQ", "NLODQ", "QUAL", "vAC0", "vAC1", "germ_GT")


def _reads(seed=3, n=4000, depth=50):
    return synth.generate_region(seed=seed, region_len=n, depth=depth, snv_every=250, somatic_every=700, indel_every=400)


def _concat(parts):
    return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}


def _split_equals_uncut(lib, all_out):
    reads = _reads()
    R = run_region(lib, reads)
    lo, hi = reads["beg"] + 150, reads["end"] - 150
    whole = R.score(all_out=all_out, pos_beg=lo, pos_end=hi)
    for cut in (lo + 1, lo + 1234, hi - 1):
        a = R.score(all_out=all_out, pos_beg=lo, pos_end=cut)
        b = R.score(all_out=all_out, pos_beg=cut, pos_end=hi, base_at_pos_beg=True)
        both = _concat([a, b])
        for k in whole:
            if k in ("germ_ref", "germ_alt1", "germ_alt2"):     # record indices: relative to the call that made them
                continue
            assert np.array_equal(both[k], whole[k]), (cut, k)
        # without the flag the second call drops the BASE records of refpos cut - 1, as process_batch does at the start of a region
        c = R.score(all_out=all_out, pos_beg=cut, pos_end=hi)
        assert len(c["refpos"]) == len(b["refpos"]) - int(((b["refpos"] == cut - 1) & (b["symbol"] <= 5)).sum())
    R.close()
    return len(whole["refpos"])


def test_split_requests_equal_the_uncut_request_on_the_oracle(oracle_lib):
    assert _split_equals_uncut(oracle_lib, False) > 20
    assert _split_equals_uncut(oracle_lib, True) > 20000


@pytest.mark.gpu
def test_split_requests_equal_the_uncut_request(gpu_lib):
    assert _split_equals_uncut(gpu_lib, False) > 20
    assert _split_equals_uncut(gpu_lib, True) > 20000


def _region_at_contig_start(seed=9):
    """A region whose reads begin at contig position 0 (chrM, small or viral references, decoys: the first tile of such a contig)."""
    r = synth.generate_region(seed=seed, region_len=3000, depth=40, beg=0, snv_every=200, indel_every=350)
    h = synth.HALO
    out = dict(r)
    out["refseq"] = r["refseq"][h:]
    out["beg"], out["end"] = 0, r["end"] - h
    out["pos"], out["mpos"] = r["pos"] - h, r["mpos"] - h
    assert out["pos"].min() == 0
    return out


def test_position_zero_on_the_oracle(oracle_lib):
    reads = _region_at_contig_start()
    R = run_region(oracle_lib, reads)
    rec = R.score(pos_beg=0, pos_end=reads["end"] - 100)
    assert len(rec["refpos"]) > 10 and rec["refpos"].min() >= 0
    allo = R.score(all_out=True, pos_beg=0, pos_end=50)
    assert (allo["refpos"][allo["symbol"] >= 6] == 0).sum() == 8 and not ((allo["refpos"] == -1).any())   # LINK at 0 is scored, BASE of refpos -1 is not
    R.close()


@pytest.mark.gpu
def test_position_zero(gpu_lib, oracle_lib):
    """uvcgpu_region_score refused pos_beg == region begin, which every first tile of a contig with a read at 0 asks for."""
    from test_gpu_parity import compare_records
    reads = _region_at_contig_start()
    Ro, Rg = run_region(oracle_lib, reads), run_region(gpu_lib, reads)
    for all_out, pe in ((False, reads["end"] - 100), (True, 300)):
        compare_records(Ro.score(all_out=all_out, pos_beg=0, pos_end=pe), Rg.score(all_out=all_out, pos_beg=0, pos_end=pe))
    text = Rg.vcf_records("chrM", Rg.score(all_out=True, pos_beg=0, pos_end=40), pos_beg=0, pos_end=40)
    cols = [l.split("\t") for l in text.splitlines()]
    assert cols and all(int(c[1]) >= 0 for c in cols)
    assert all(c[3] == "n" for c in cols if c[1] == "0")         # append_vcf_record: POS 0, REF "n" for an InDel in front of the first base (main.hpp:6066-6090)
    with pytest.raises(region.UvcError):
        Rg.score(pos_beg=0, pos_end=40, base_at_pos_beg=True)   # refpos -1 does not exist
    Ro.close(); Rg.close()


def _block_positions(text):
    return [int(l.split("\t")[1]) for l in text.splitlines() if l.split("\t")[4] == "<NON_REF>"]


@pytest.mark.gpu
def test_mgvcf_block_opens_at_the_region_begin(gpu_lib, oracle_lib):
    """`refpos == incluBegPosition` (main.cpp:655-656) is the begin of the BED line, not the begin of the state."""
    from test_vcf_text import _oracle_lines, ref_vcf as _rv   # noqa: F401
    reads = _reads(seed=5, n=5000)
    Rg = run_region(gpu_lib, reads)
    rb = reads["beg"] + 437                                      # a target that does not start on a multiple of 1000
    kw = dict(pos_beg=rb, pos_end=reads["end"] - 150, region_beg=rb)
    rec = Rg.score(**kw)
    with_rb = _block_positions(Rg.vcf_records("c", rec, **kw))
    without = _block_positions(Rg.vcf_records("c", rec, pos_beg=kw["pos_beg"], pos_end=kw["pos_end"]))
    assert with_rb[0] == rb + 1 and with_rb[1:] == without and all(p % 1000 == 1 for p in without)
    Rg.close()


def test_mgvcf_block_opens_at_the_region_begin_on_the_oracle(oracle_lib):
    import test_vcf_text as tv
    reads = _reads(seed=5, n=5000)
    Ro = run_region(oracle_lib, reads)
    rb = reads["beg"] + 437
    lines = tv._oracle_lines(oracle_lib, tv._load_ref_vcf(), Ro, "c", pos_beg=rb, pos_end=reads["end"] - 150, region_beg=rb)
    blocks = _block_positions("\n".join(lines))
    assert blocks[0] == rb + 1 and all(p % 1000 == 1 for p in blocks[1:]) and len(blocks) >= 4
    Ro.close()


def _tile_sets(lib, bam, fa, tile, all_out):
    out = {}
    dup = 0
    for t in pipeline.call_contig(lib, bam, fa, "chrT", tile=tile, all_out=all_out):
        r = t["records"]
        for i in range(len(r["refpos"])):
            k = (int(r["refpos"][i]), int(r["symbol"][i]), int(r["gapSa_len"][i]), int(r["bDPa"][i]))
            dup += k in out
            out[k] = {f: int(r[f][i]) for f in ("DP", "AD", "bDP", "bAD", "cDP0a", "TLODQ", "keep")}
    return out, dup


def test_tiles_of_a_stretch_give_the_records_of_the_uncut_region(tmp_path, oracle_lib):
    """Every (position, symbol, allele) of the uncut region appears in exactly one tile, with the same depths.  (Phred-like fields may move
    by a unit next to nothing: the reference's BAQ prefix sums are divided by 10 after summation, main.cpp:400-429, so their differences
    depend on where the region starts -- a property of the algorithm, not of the tiling.)"""
    import test_pipeline as tp
    tp.make_files(tmp_path, 0)
    bam, fa = uio.Bam(str(tmp_path / "u0.bam")), uio.Fasta(str(tmp_path / "u0.fa"))
    for all_out in (False, True):
        whole, d0 = _tile_sets(oracle_lib, bam, fa, 10 ** 7, all_out)
        assert d0 == 0 and len(whole) > (50000 if all_out else 100)
        for tile in (2500, 1000):
            tiles, dup = _tile_sets(oracle_lib, bam, fa, tile, all_out)
            assert dup == 0 and set(tiles) == set(whole), (tile, all_out, dup, len(set(tiles) ^ set(whole)))
            for f in ("DP", "AD", "bDP", "bAD", "cDP0a"):
                assert all(tiles[k][f] == whole[k][f] for k in whole), (tile, f)
            assert sum(abs(tiles[k]["TLODQ"] - whole[k]["TLODQ"]) > 1 for k in whole) == 0
            assert sum(tiles[k]["keep"] != whole[k]["keep"] for k in whole) <= 1


def test_shard_plan_and_bgzf_concat(tmp_path):
    """uvcio_plan_shards: contiguous, monotone, balanced; uvcio_bgzf_concat: bcftools concat -n of the shard outputs."""
    import ctypes as C
    import gzip
    from uvc_amd import shard
    rng = np.random.default_rng(1)
    for n, k in ((1, 1), (5, 2), (64, 8), (3, 8), (1000, 7)):
        cost = rng.integers(0, 1000, n)
        s = shard.plan_contiguous(cost, k)
        assert len(s) == n and (np.diff(s) >= 0).all() and s.min() >= 0 and s.max() < k
        if n >= 8 * k:
            loads = np.bincount(s, weights=cost, minlength=k)
            assert loads.max() <= cost.sum() / k + cost.max()
    assert list(shard.plan_contiguous([0, 0, 0, 0], 2)) == [0, 0, 1, 1]
    parts = []
    for i, text in enumerate(["##h\n#CHROM\n1\t5\n", "1\t9\n", "", "2\t1\n"]):
        p = str(tmp_path / ("s%d.vcf.gz" % i))
        w = uio.BgzfWriter(p); w.write(text); w.close()
        parts.append(p)
    out = str(tmp_path / "all.vcf.gz")
    shard.concat_bgzf(out, parts)
    assert gzip.open(out, "rt").read() == "##h\n#CHROM\n1\t5\n1\t9\n2\t1\n"
    raw = open(out, "rb").read()
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    assert raw.endswith(eof) and raw.count(eof) == 1
