"""BASELINE.json's full bench tile (1 Mb at 300x, 2 M reads) through size-independent properties, and the edge cases of the
boundary.  The oracle would need minutes here, so nothing below calls it: the expected values are recomputed with numpy
directly from the read arrays, or are invariants (repeatability, additivity of independent regions)."""
import zlib

import numpy as np
import pytest

from uvc_amd import _ffi, region, synth

pytestmark = pytest.mark.gpu
E = _ffi.ENUMS


def ref_consuming_coverage(reads):
    """depth of reference-consuming CIGAR ops (M/=/X/D/N) per position, and of M/=/X only"""
    n = reads["end"] - reads["beg"] + 1
    d_all = np.zeros(n + 1, np.int64)
    d_m = np.zeros(n + 1, np.int64)
    op = (reads["cigars"] & 0xF).astype(np.int64)
    ln = (reads["cigars"] >> 4).astype(np.int64)
    owner = np.repeat(np.arange(reads["n_reads"]), reads["n_cigar"])
    consumes = np.isin(op, (0, 2, 3, 7, 8))
    step = np.where(consumes, ln, 0)
    # start offset of every op inside its read = exclusive prefix sum of the ref-consuming lengths, restarted per read
    cs = np.cumsum(step) - step
    first = np.zeros(reads["n_reads"], np.int64)
    first_idx = reads["cigar_off"].astype(np.int64)
    first = cs[first_idx]
    start = reads["pos"].astype(np.int64)[owner] - reads["beg"] + (cs - first[owner])
    for mask, d in ((consumes, d_all), (np.isin(op, (0, 7, 8)), d_m)):
        np.add.at(d, start[mask], 1)
        np.add.at(d, start[mask] + ln[mask], -1)
    return np.cumsum(d_all)[:n], np.cumsum(d_m)[:n]


@pytest.fixture(scope="module")
def full_tile(gpu_lib):
    reads = synth.generate_region(seed=12345, region_len=1_000_000, depth=300)
    R = region.Region(gpu_lib, region.default_params(gpu_lib), reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    R.set_reads(reads)
    R.accumulate()
    return reads, R


def checksum(R, groups=("PREP32", "SEG32", "SEG64", "VQ", "BQSUM", "FRAG", "FAM", "FAMINFO32", "DUPLEX")):
    return {g: zlib.crc32(R.fetch(g).tobytes()) for g in groups}


def test_depth_planes_equal_a_numpy_pileup(full_tile):
    reads, R = full_tile
    cov_all, cov_m = ref_consuming_coverage(reads)
    prep = R.fetch("PREP32")
    assert np.array_equal(prep[E["UVC_P_a_dp"]], cov_all)                   # one count per reference position a read spans (M and D ops)
    seg = R.fetch("SEG32")
    ad = seg[E["UVC_S_aDPff"]] + seg[E["UVC_S_aDPfr"]] + seg[E["UVC_S_aDPrf"]] + seg[E["UVC_S_aDPrr"]]
    base_depth = ad[:6].sum(axis=0)
    # every aligned base adds one BASE symbol; a deleted base adds BASE_NN when the deletion is far enough from the read ends
    assert (base_depth >= cov_m).all() and (base_depth - cov_m).sum() <= (seg[E["UVC_S_aDPff"]][5] + seg[E["UVC_S_aDPfr"]][5] + seg[E["UVC_S_aDPrf"]][5] + seg[E["UVC_S_aDPrr"]][5]).sum()
    frag = R.fetch("FRAG")                                                   # [2][NFRAG][14][npos]
    bdp = frag[:, E["UVC_FRAG_bDP"]].sum(axis=0)
    assert (bdp[:6].sum(axis=0) <= base_depth).all()                         # a fragment counts once where both mates overlap
    assert (bdp[:6].sum(axis=0) * 2 >= base_depth).all()


def test_second_accumulate_reproduces_every_plane(full_tile):
    _, R = full_tile
    c1 = checksum(R)
    rec1 = R.score(capacity=400_000)
    R.accumulate()
    assert checksum(R) == c1
    rec2 = R.score(capacity=400_000)
    assert all(np.array_equal(rec1[k], rec2[k]) for k in rec1)
    assert len(rec1["refpos"]) > 10_000 and np.all(np.diff(rec1["refpos"]) >= 0)   # emission order: by position


def test_allele_rows_and_call_fields_are_consistent_with_the_planes(full_tile):
    """Size-independent properties of the InDel allele tables and of the calling step on the full tile:
    every fragment that votes for an InDel symbol carries exactly one allele of it (rows.bAD1 sums to FRAG_bDP per strand), the
    per-allele records of a site split that support, a record's NLODQ / QUAL / FILTER / keep follow from its own fields."""
    reads, R = full_tile
    rows = R.indel_alleles()
    assert len(rows) > 200
    frag = R.fetch("FRAG")
    fam = R.fetch("FAM")
    by_site = {}
    for r in rows:
        k = (r["strand"], r["symbol"], r["refpos"] - reads["beg"])
        by_site.setdefault(k, [0, 0]); by_site[k][0] += r["bAD1"]; by_site[k][1] += r["cAD1"]
    for (s, sym, x), (b, c) in by_site.items():
        assert b == frag[s, E["UVC_FRAG_bDP"], sym, x], (s, sym, x)
        assert c == fam[s, E["UVC_FAM_cDP12"], sym, x], (s, sym, x)                 # non-UMI data: every family votes after filtering
    indel = np.isin(frag[:, E["UVC_FRAG_bDP"], 7:13].sum(axis=0) > 0, True)
    assert int((frag[:, E["UVC_FRAG_bDP"], 7:13] > 0).sum()) == len(by_site)        # no (strand, symbol, position) without rows
    rec = R.score(capacity=400_000)
    ind = rec["gapSa_len"] > 0
    assert ind.sum() > 100 and (rec["gapSa"][ind] >= 0).all()
    for i in np.nonzero(ind)[0][:500]:
        row = rows[rec["gapSa"][i]]
        assert (row["refpos"], row["symbol"], row["len"]) == (rec["refpos"][i], rec["symbol"][i], rec["gapSa_len"][i])
    # tumor-only arithmetic of main.cpp:1081-1147 / append_vcf_record on the record's own fields
    out = rec["out"] == 1
    assert (out == (rec["symbol"] != 13)).all()                                      # everything but LINK_NN reaches append_vcf_record by default
    germ = np.where(rec["symbol"] <= 5, 31, 40)
    assert np.array_equal(rec["vHGQ"][out], (rec["vNLODQ"] - 3 + germ)[out]) and np.array_equal(rec["NLODQ"][out], rec["vHGQ"][out])
    assert np.array_equal(rec["SomaticQ"][out], np.minimum(rec["TLODQ"], rec["NLODQ"])[out])
    tl1 = np.maximum(rec["TNBQF3"], rec["TNCQF3"])
    assert np.array_equal(rec["TLODQ"][out], np.where(tl1 >= 10, tl1, tl1 * 3 - 20)[out])
    q = rec["QUAL"].view(np.float32)
    assert (q[out] >= np.maximum(rec["TLODQ"][out], 0) - 1e-3).all() and (q[out] > 0).all()
    assert np.array_equal(rec["FILTER"][out], np.minimum(np.floor(q[out] / 10), 6).astype(np.int32))
    alt = out & (rec["symbol"] != rec["refsymbol"])
    assert (rec["keep"][alt & (q >= 15)] == 1).all() and (rec["keep"][out & ~alt] == 0).all()   # REF alleles are only written with -A or a GERMLINE line
    assert rec["keep"].sum() > 500
    assert (rec["germ_GT"] >= 0).all() and (rec["germ_GT"] <= 3).all() and (rec["vNLODQ"] == rec["GL4_0"] - np.maximum(np.maximum(rec["GL4_1"], rec["GL4_2"]), rec["GL4_3"])).all()


def test_two_half_tiles_add_up(gpu_lib):
    """Regions are independent: the planes of two tiles laid side by side equal the planes of each tile alone."""
    a = synth.generate_region(seed=5, region_len=20_000, depth=300, beg=2_000_000)
    Ra = region.Region(gpu_lib, region.default_params(gpu_lib), a["tid"], a["beg"], a["end"], a["refseq"])
    Ra.set_reads(a); Ra.accumulate()
    before = checksum(Ra)
    b = synth.generate_region(seed=6, region_len=20_000, depth=300, beg=3_000_000)
    Rb = region.Region(gpu_lib, region.default_params(gpu_lib), b["tid"], b["beg"], b["end"], b["refseq"])
    Rb.set_reads(b); Rb.accumulate()
    assert checksum(Ra) == before        # another handle's work does not leak into this one


def test_no_reads_and_bad_inputs(gpu_lib):
    reads = synth.generate_region(seed=9, region_len=500, depth=5)
    p = region.default_params(gpu_lib)
    R = region.Region(gpu_lib, p, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    with pytest.raises(region.UvcError) as e:
        R.accumulate()
    assert e.value.code == E["UVCGPU_ENOREADS"]                              # process_batch returns -1, main.cpp:520-523
    empty = {k: (v[:0] if isinstance(v, np.ndarray) and k not in ("fam_dflag",) else v) for k, v in reads.items()}
    empty.update(n_reads=0, n_fams=0, fam_dflag=np.zeros(0, np.uint8))
    R.set_reads(empty)
    with pytest.raises(region.UvcError):
        R.accumulate()
    outside = dict(reads); outside["pos"] = reads["pos"].copy(); outside["pos"][0] = reads["beg"] - 10
    with pytest.raises(region.UvcError) as e:
        R.set_reads(outside)
    assert e.value.code == E["UVCGPU_EINVAL"]
    ragged = dict(reads); ragged["l_qseq"] = reads["l_qseq"].copy(); ragged["l_qseq"][1] += 1   # CIGAR no longer spans the read
    with pytest.raises(region.UvcError):
        R.set_reads(ragged)
    R.set_reads(reads)                                                       # the handle is still usable
    R.accumulate()
    assert R.fetch("PREP32")[E["UVC_P_a_dp"]].sum() == ref_consuming_coverage(reads)[0].sum()


# ---- BASELINE config 4 at full size: a 200 kb duplex-UMI panel tile at 2000x (2.7 M reads, ~95 M (unit, position) cells) ----
FAM_GROUPS = ("FRAG", "FAM", "FAMINFO32", "DUPLEX", "VQ", "SEG32", "PREP32")


def _umi_tile_planes(gpu_lib, reads, fam_path, monkeypatch):
    if fam_path:
        monkeypatch.setenv("UVCGPU_FAM_PATH", fam_path)      # read by set_reads: generic = one thread per (unit, position), window = LDS window kernels without the digest
    else:
        monkeypatch.delenv("UVCGPU_FAM_PATH", raising=False)
    R = region.Region(gpu_lib, region.default_params(gpu_lib), reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    R.set_reads(reads)
    R.accumulate()
    sums = checksum(R, FAM_GROUPS)
    rec = R.score(capacity=600_000)
    keep = {k: rec[k].copy() for k in ("refpos", "symbol", "cDP1x", "cDP2x", "cVQ1", "cVQ2", "QUAL", "keep")}
    planes = {g: R.fetch(g) for g in ("FRAG", "FAM", "DUPLEX")} if not fam_path else None
    R.close()
    return sums, keep, planes


def test_config4_tile_three_forms_of_the_family_kernels_agree(gpu_lib, monkeypatch):
    """The digest kernels (k_fam_p4d / k_fam_win<5, true> / k_duplex_d, what a deep tile runs by default), the window kernels without the
    digest and the one-thread-per-(unit, position) kernels are three implementations of P4 / P5 / duplex: at BASELINE config 4's full
    tile size they must leave bit-identical planes and identical records.  Plus relations the consensus counters obey by construction."""
    reads = synth.generate_region(seed=4242, region_len=200_000, depth=2000, umi=True)
    assert reads["n_reads"] > 2_000_000 and reads["n_fams"] > 100_000
    base_sums, base_rec, planes = _umi_tile_planes(gpu_lib, reads, None, monkeypatch)
    for path in ("window", "generic"):
        sums, rec, _ = _umi_tile_planes(gpu_lib, reads, path, monkeypatch)
        assert sums == base_sums, path
        assert all(np.array_equal(rec[k], base_rec[k]) for k in rec), path
    frag, fam, dup = planes["FRAG"], planes["FAM"], planes["DUPLEX"]
    bdp = frag[:, E["UVC_FRAG_bDP"]]                                          # [strand][symbol][position]
    c12, c1, c2, c3 = (fam[:, E[k]] for k in ("UVC_FAM_cDP12", "UVC_FAM_cDP1", "UVC_FAM_cDP2", "UVC_FAM_cDP3"))
    assert (c12.sum(axis=1) <= bdp.sum(axis=1)).all()                          # a family-strand unit votes once where its fragments vote at all
    assert (c12[:, :6].sum(axis=1) * 20 >= bdp[:, :6].sum(axis=1)).all()       # ... and no unit of this data set has more than 20 fragments on one strand at a position
    assert (c2 <= c1).all() and c3.sum() < c2.sum()                           # the stricter family-size / agreement threshold of cDP2 (main.hpp:3140-3220)
    assert (fam[:, E["UVC_FAM_cDP21"]] <= c12).all()
    d1, d2 = dup[E["UVC_DUPLEX_dDP1"]], dup[E["UVC_DUPLEX_dDP2"]]
    assert (d2 <= d1).all() and d2.sum() > 10_000                              # duplex families whose two strands agree are a subset of the duplex families
    assert (d1.sum(axis=0) <= np.minimum(c12[0].sum(axis=0), c12[1].sum(axis=0))).all()   # a duplex family needs a unit on each strand
    assert base_rec["keep"].sum() > 50 and len(base_rec["refpos"]) > 100_000


def test_k_frag_with_32_bit_buckets_equals_the_packed_form(gpu_lib, monkeypatch):
    """k_frag keeps its LDS bucket histogram packed two per word while fewer than 65 536 fragments cover a position (k_frag16) and in 32-bit
    words otherwise (k_frag: a depth no test tile has).  UVCGPU_FRAG32 forces the second form: same planes, same records, on a tile with
    InDel fragments (the LINK_M arm outside the interval sums) and rare symbols (the event queue)."""
    reads = synth.generate_region(seed=777, region_len=120_000, depth=300)
    out = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("UVCGPU_FRAG32", "1")     # read by set_reads
        else:
            monkeypatch.delenv("UVCGPU_FRAG32", raising=False)
        R = region.Region(gpu_lib, region.default_params(gpu_lib), reads["tid"], reads["beg"], reads["end"], reads["refseq"])
        R.set_reads(reads)
        R.accumulate()
        sums = checksum(R)
        rec = R.score(capacity=200_000)
        out.append((sums, {k: rec[k].copy() for k in ("refpos", "symbol", "QUAL", "keep")}))
        R.close()
    assert out[0][0] == out[1][0]
    assert all(np.array_equal(out[0][1][k], out[1][1][k]) for k in out[0][1])
    assert len(out[0][1]["refpos"]) > 1_000


def test_split_window_kernels_equal_the_one_wave_per_window_kernels(gpu_lib, monkeypatch):
    """Deep, short regions run k_prep_fast / k_p2_fast / k_frag16 with a block per window (four waves share the window's reads, LDS reduction);
    everything else one wave per window.  UVCGPU_SPLIT forces either form: same planes, same records, on a UMI tile with InDel fragments."""
    reads = synth.generate_region(seed=99, region_len=30_000, depth=600, umi=True)
    out = []
    for force in ("0", "1"):
        monkeypatch.setenv("UVCGPU_SPLIT", force)     # read by every accumulate
        R = region.Region(gpu_lib, region.default_params(gpu_lib), reads["tid"], reads["beg"], reads["end"], reads["refseq"])
        R.set_reads(reads)
        R.accumulate()
        sums = checksum(R)
        rec = R.score(capacity=200_000)
        out.append((sums, {k: rec[k].copy() for k in ("refpos", "symbol", "QUAL", "keep")}))
        R.close()
    assert out[0][0] == out[1][0]
    assert all(np.array_equal(out[0][1][k], out[1][1][k]) for k in out[0][1])
    assert len(out[0][1]["refpos"]) > 1_000
