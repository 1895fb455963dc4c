"""the insertion / soft-clip consensus blocks (ConsensusBlockSet, main_consensus.hpp:52-225; filled at main.hpp:2009,
2100-2116, 2259-2279; folded per family at main.hpp:1722, 2875-2911).  Host code of libuvcgpu.so (no GPU needed) against
  * a hand-built family whose blocks can be written down by eye,
  * an independent pure-Python restatement (dictionaries, written from the reference text),
  * the oracle's restatement in the reference's map-of-vectors form."""
import numpy as np
import pytest

from uvc_amd import _ffi, consensus, region, synth


@pytest.fixture(scope="module")
def product_lib():
    return _ffi.Lib(_ffi.gpu_library_path(), "uvcgpu_")        # loading and the host-only entry points need no GPU


def make_reads(specs, tid=0):
    """specs: list of (fam_id, strand, frag_id, pos, cigar string, bases string, quals list, flag, mpos, isize)"""
    ops = "MIDNSHP=X"
    pos, mpos, isize, flag, lq, so, co, nc, frag, fam, fs, bases, quals, cig = ([] for _ in range(14))
    for (f, s, g, p, c, b, q, fl, mp, isz) in specs:
        so.append(len(bases)); co.append(len(cig))
        n = 0; num = ""
        for ch in c:
            if ch.isdigit():
                num += ch
            else:
                cig.append(int(num) << 4 | ops.index(ch)); num = ""; n += 1
        nc.append(n)
        assert len(b) == len(q)
        bases += ["ACGTN".index(x) if x in "ACGT" else 4 for x in b]; quals += list(q)
        pos.append(p); mpos.append(mp); isize.append(isz); flag.append(fl); lq.append(len(b)); frag.append(g); fam.append(f); fs.append(s)
    n = len(specs)
    return dict(tid=tid, n_reads=n, pos=np.array(pos, np.int32), mpos=np.array(mpos, np.int32), isize=np.array(isize, np.int32), flag=np.array(flag, np.uint16),
                mapq=np.full(n, 60, np.uint8), nm=np.zeros(n, np.int32), l_qseq=np.array(lq, np.int32), seq_off=np.array(so, np.int64), cigar_off=np.array(co, np.int64),
                n_cigar=np.array(nc, np.int32), frag_id=np.array(frag, np.int32), fam_id=np.array(fam, np.int32), fam_strand=np.array(fs, np.uint8),
                bases=np.array(bases, np.uint8), quals=np.array(quals, np.uint8), cigars=np.array(cig, np.uint32), n_fams=max(fam) + 1, fam_dflag=np.zeros(max(fam) + 1, np.uint8))


def test_a_family_written_down_by_hand(product_lib):
    p = region.default_params(product_lib)
    q30 = [30] * 10
    reads = make_reads([
        # family 0, strand 0: three fragments with an insertion at 105 (after 5 M): two say ACG, one says AT (shorter)
        (0, 0, 0, 100, "5M3I2M", "AAAAAACGTT", [30, 30, 30, 30, 30, 20, 25, 35, 30, 30], 0x0, 100, 0),
        (0, 0, 1, 100, "5M3I2M", "AAAAAACGTT", [30, 30, 30, 30, 30, 40, 10, 15, 30, 30], 0x0, 100, 0),
        (0, 0, 2, 100, "5M2I3M", "AAAAAATTTT", [30, 30, 30, 30, 30, 22, 33, 30, 30, 30], 0x0, 100, 0),
        # family 1, strand 1: one fragment, two reads that both clip at 200 on the left (stored reversed) and one right clip at 207
        (1, 1, 0, 200, "3S7M", "GCATTTTTTT", [11, 12, 13] + [30] * 7, 0x10, 200, 0),
        (1, 1, 0, 200, "2S5M3S", "TATTTTTACG", [21, 9] + [30] * 5 + [5, 6, 7], 0x10, 200, 0),
    ])
    blocks = consensus.family_blocks(product_lib, p, reads, min_fragments=1)
    key = {(b["fam_id"], b["strand"], b["type"], b["refpos"]): b for b in blocks}
    assert sorted(key) == [(0, 0, 1, 105), (1, 1, 0, 205), (1, 1, 2, 200)]
    ins = key[(0, 0, 1, 105)]
    assert ins["n_fragments"] == 3
    #                          A  C  G  T  N  NN  sum of (2 * major - total)   fragments
    assert ins["rows"].tolist() == [[3, 0, 0, 0, 0, 0, 20 + 40 + 22, 3],       # A A A
                                    [0, 2, 0, 1, 0, 0, 25 + 10 + 33, 3],       # C C T
                                    [0, 0, 2, 0, 0, 0, 35 + 15, 2]]            # G G (the third fragment's insertion has two bases)
    assert consensus.block_to_seq(product_lib, ins["rows"]) == [("A", 27, 3, 1), ("C", 22, 3, 0), ("G", 25, 2, 1)]
    left = key[(1, 1, 2, 200)]          # clips of both reads of ONE fragment: per base the maximum quality per symbol, then one vote
    # read 1 reversed: A(13) C(12) G(11); read 2 reversed: A(9) T(21)  ->  row 0: A = 13; row 1: C = 12, T = 21; row 2: G = 11
    assert left["rows"].tolist() == [[1, 0, 0, 0, 0, 0, 13, 1], [0, 0, 0, 1, 0, 0, 21 * 2 - 33, 1], [0, 0, 1, 0, 0, 0, 11, 1]]
    assert [b for b, _, _, _ in consensus.block_to_seq(product_lib, left["rows"], right_to_left=True)] == ["G", "T", "A"]
    right = key[(1, 1, 0, 205)]
    assert right["rows"][:, :5].argmax(axis=1).tolist() == [0, 1, 2] and right["rows"][:, 6].tolist() == [5, 6, 7]
    frag = consensus.fragment_blocks(product_lib, p, reads, 3, 2)
    fl = [b for b in frag if b["type"] == 2][0]
    assert fl["rows"].tolist() == [[13, 0, 0, 0, 0, 0, 13, 1], [0, 12, 0, 21, 0, 0, 21, 1], [0, 0, 11, 0, 0, 0, 11, 1]]
    # a family below the fragment threshold has no family-level blocks; one whose span lies in the previous region neither
    assert {b["fam_id"] for b in consensus.family_blocks(product_lib, p, reads, min_fragments=2)} == {0}
    assert consensus.family_blocks(product_lib, p, reads, min_fragments=1, curr=(0, 150)) and {b["fam_id"] for b in consensus.family_blocks(product_lib, p, reads, curr=(0, 150))} == {0}
    assert {b["fam_id"] for b in consensus.family_blocks(product_lib, p, reads, curr=(0, 1000), prev=(0, 90, 101))} == {1}


# ---- independent restatement: dictionaries keyed like the reference's maps ----
def py_events(reads, i, P):
    out = []
    cig = reads["cigars"][reads["cigar_off"][i]:reads["cigar_off"][i] + reads["n_cigar"][i]]
    pos, flag, isize, mpos = int(reads["pos"][i]), int(reads["flag"][i]), int(reads["isize"][i]), int(reads["mpos"][i])
    rend = pos + sum(int(c >> 4) for c in cig if int(c & 15) in (0, 2, 3, 7, 8))
    amplicon = bool(reads["fam_dflag"][reads["fam_id"][i]] & 4) or (P.primerlen > 0 and not (P.primer_flag & 2))
    normal = bool(P.tn_is_paired and (P.primer_flag & 1))
    single_rc = bool(flag & 0x10) and not (flag & 1)
    ibeg = (min(pos, mpos) + P.primerlen) if isize else (0 if single_rc else pos + P.primerlen)
    iend = max(min(pos, mpos) + abs(isize) - P.primerlen, 0) if isize else (max(rend - P.primerlen, 0) if single_rc else 2**31 - 1)
    q, r = int(reads["seq_off"][i]), pos
    for k, c in enumerate(cig):
        op, ln = int(c & 15), int(c >> 4)
        if op == 1 and (normal or not amplicon or ibeg <= r < iend):
            out.append((1, r, [(min(int(reads["bases"][q + j]), 4), int(np.int8(reads["quals"][q + j]))) for j in range(ln)]))
        if op == 4:
            seq = [(min(int(reads["bases"][q + j]), 4), int(np.int8(reads["quals"][q + j]))) for j in range(ln)]
            out.append((2, r, seq[::-1]) if k == 0 else (0, r, seq))
        if op in (0, 1, 4, 7, 8):
            q += ln
        if op in (0, 2, 3, 7, 8):
            r += ln
    return out


def py_family_blocks(reads, P, min_fragments):
    res = {}
    n = reads["n_reads"]
    unit_key = list(zip(reads["fam_id"].tolist(), reads["fam_strand"].tolist()))
    i = 0
    while i < n:
        j = i
        while j < n and unit_key[j] == unit_key[i]:
            j += 1
        frags = {}
        for k in range(i, j):
            frags.setdefault(int(reads["frag_id"][k]), []).append(k)
        if len(frags) >= min_fragments:
            fam = {}
            for g, members in frags.items():
                fb = {}
                for k in members:
                    for (t, r, seq) in py_events(reads, k, P):
                        rows = fb.setdefault((t, r), [])
                        while len(rows) < len(seq):
                            rows.append([0] * 8)
                        for x, (b, q) in enumerate(seq):
                            rows[x][b] = max(rows[x][b], q); rows[x][6] = max(rows[x][6], q); rows[x][7] = 1
                for key, rows in fb.items():
                    dst = fam.setdefault(key, [])
                    while len(dst) < len(rows):
                        dst.append([0] * 8)
                    for x, row in enumerate(rows):
                        con, cc, tot = 5, 0, 0
                        for b in range(5):
                            if row[b] > cc:
                                con, cc = b, row[b]
                            tot += row[b]
                        dst[x][con] += 1; dst[x][6] += max(2 * cc - tot, 0); dst[x][7] += 1
            for (t, r), rows in fam.items():
                res[(unit_key[i][0], unit_key[i][1], t, r)] = (len(frags), rows)
        i = j
    return res


def py_to_seq(rows, right_to_left, trim):
    rows = [list(map(int, r)) for r in rows]
    if trim is not None:
        perc, consec = trim
        mx = max([sum(r[:5]) for r in rows] + [0])
        kept, low = [], 0
        for r in rows:
            if sum(r[:5]) * 100 < mx * perc:
                low += 1
                if low >= consec:
                    kept = kept[:len(kept) - (low - 1)]
                    break
            kept.append(r)
        else:
            kept = kept
        rows = kept
    out = []
    for r in (rows[::-1] if right_to_left else rows):
        con, cc, tot = 5, 0, 0
        for b in range(5):
            if r[b] > cc:
                con, cc = b, r[b]
            tot += r[b]
        qual = int(np.int8(np.int32(int(r[6] / max(r[7], 1)))))
        out.append(("ACGTN*"[con], qual, tot, int(cc / max(tot, 1))))
    return out


@pytest.mark.parametrize("case", [dict(seed=31, region_len=4000, depth=200, umi=True, indel_every=100, clip_frac=0.2),
                                  dict(seed=32, region_len=3000, depth=100, indel_every=100, clip_frac=0.3, min_fragments=1, set=dict(primerlen=20))])
def test_product_equals_the_oracle_and_an_independent_restatement(case, product_lib, oracle_lib):
    case = dict(case); overrides = case.pop("set", {}); mf = case.pop("min_fragments", 2)
    reads = synth.generate_region(**case)
    P = region.default_params(product_lib)
    for k, v in overrides.items():
        setattr(P, k, v)
    mine = consensus.family_blocks(product_lib, P, reads, min_fragments=mf)
    theirs = consensus.family_blocks(oracle_lib, P, reads, min_fragments=mf)
    want = py_family_blocks(reads, P, mf)
    assert len(mine) > 20 and {b["type"] for b in mine} == {0, 1, 2} and max(len(b["rows"]) for b in mine) >= 3
    assert len(mine) == len(theirs) == len(want)
    for a, b in zip(mine, theirs):
        key = (a["fam_id"], a["strand"], a["type"], a["refpos"])
        assert key == (b["fam_id"], b["strand"], b["type"], b["refpos"]) and a["n_fragments"] == b["n_fragments"] == want[key][0]
        assert a["rows"].tolist() == b["rows"].tolist() == want[key][1], key
        for trim in (None, (20, 3), (60, 1), (150, 2)):
            r2l = (a["type"] == 2)
            s = consensus.block_to_seq(product_lib, a["rows"], r2l, trim)
            assert s == consensus.block_to_seq(oracle_lib, a["rows"], r2l, trim) == py_to_seq(a["rows"], r2l, trim), (key, trim)
    # fragment level, first unit with two reads in one fragment
    fr = consensus.fragment_blocks(product_lib, P, reads, 0, 2) if reads["frag_id"][0] == reads["frag_id"][1] else consensus.fragment_blocks(product_lib, P, reads, 0, 1)
    fo = consensus.fragment_blocks(oracle_lib, P, reads, 0, 2) if reads["frag_id"][0] == reads["frag_id"][1] else consensus.fragment_blocks(oracle_lib, P, reads, 0, 1)
    assert [(b["type"], b["refpos"], b["rows"].tolist()) for b in fr] == [(b["type"], b["refpos"], b["rows"].tolist()) for b in fo]


def test_bad_arguments(product_lib):
    p = region.default_params(product_lib)
    reads = make_reads([(0, 0, 0, 10, "2S3M", "ACGTA", [30] * 5, 0, 10, 0)])
    with pytest.raises(region.UvcError):
        consensus.fragment_blocks(product_lib, p, reads, 0, 5)
    assert consensus.block_to_seq(product_lib, np.zeros((0, 8), np.int32)) == []
    # an all-zero row (a base of quality 0) votes for BASE_NN: '*' with family size 0
    assert consensus.block_to_seq(product_lib, [[0, 0, 0, 0, 0, 1, 0, 1]]) == [("*", 0, 0, 0)]


@pytest.mark.gpu
def test_on_the_gpu_box_too(product_lib, oracle_lib):
    """Row a9 is host code inside libuvcgpu.so: the same comparison once more under `-m gpu`, so that the driver's GPU run loads the library
    that hosts it and exercises it there as well (the CPU suite above runs everywhere)."""
    test_a_family_written_down_by_hand(product_lib)
    test_product_equals_the_oracle_and_an_independent_restatement(dict(seed=33, region_len=3000, depth=150, umi=True, indel_every=120, clip_frac=0.25), product_lib, oracle_lib)
