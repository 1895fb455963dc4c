"""The oracle's InDel allele tables against a hand count (no GPU): every read is its own fragment and family, carries one
insertion at the same site, so the per-strand rows are plain counts of the distinct inserted sequences and the scored alleles
follow indel_get_majority (main.hpp:5406-5455): at least a quarter of the best support, ordered by bAD1^2 * length.  Outside a repeat
every insertion shorter than 6 bases counts as ONE unit (ref_to_phredvalue, main.hpp:876-922), so all of them are LINK_I1 alleles."""
import collections

import numpy as np

from uvc_amd import region

M, I = 0, 1


def test_rows_and_alleles_of_one_insertion_site(oracle_lib):
    rng = np.random.default_rng(9)
    ref_len, beg, site = 300, 5_000_000, 150
    ref = rng.integers(0, 4, ref_len)
    ref[site - 1], ref[site] = 0, 1                                      # no repeat context around the site
    alleles = [([2, 3], 20), ([3], 9), ([2, 2, 0, 1], 6), ([4], 2), ([1, 1], 1)]   # (inserted codes, number of reads)
    cols = collections.defaultdict(list)
    bases, quals, cigars = [], [], []
    fid = 0
    expect = collections.Counter()
    for seq, n in alleles:
        for k in range(n):
            strand = k % 2
            start = 60 + (fid % 17)
            left, right = site - start, 80
            q = list(ref[start:site]) + seq + list(ref[site:site + right])
            cols["pos"].append(beg + start); cols["flag"].append(0x10 if strand else 0); cols["mapq"].append(60); cols["mpos"].append(-1); cols["isize"].append(0)
            cols["nm"].append(len(seq)); cols["l_qseq"].append(len(q)); cols["seq_off"].append(len(bases)); cols["cigar_off"].append(len(cigars)); cols["n_cigar"].append(3)
            cols["frag_id"].append(fid); cols["fam_id"].append(fid); cols["fam_strand"].append(strand)
            bases += [int(b) for b in q]; quals += [37] * len(q); cigars += [(left << 4) | M, (len(seq) << 4) | I, (right << 4) | M]
            expect[(strand, "".join("ACGTN"[b] for b in seq))] += 1
            fid += 1
    dt = dict(pos=np.int32, mpos=np.int32, isize=np.int32, flag=np.uint16, mapq=np.uint8, nm=np.int32, l_qseq=np.int32, seq_off=np.int64, cigar_off=np.int64,
              n_cigar=np.int32, frag_id=np.int32, fam_id=np.int32, fam_strand=np.uint8)
    reads = {k: np.array(v, dt[k]) for k, v in cols.items()}
    reads.update(n_reads=fid, n_fams=fid, fam_dflag=np.zeros(fid, np.uint8), bases=np.array(bases, np.uint8), quals=np.array(quals, np.uint8), cigars=np.array(cigars, np.uint32))
    R = region.Region(oracle_lib, region.default_params(oracle_lib), 2, beg, beg + ref_len, "".join("ACGT"[b] for b in ref))
    R.set_reads(reads); R.accumulate()
    rows = [r for r in R.indel_alleles() if r["refpos"] == beg + site]
    assert {(r["strand"], r["seq"]): (r["bAD1"], r["cAD1"], r["c2AD"], r["c2dAD"]) for r in rows} == {k: (v, v, 0, 0) for k, v in expect.items()}
    for strand in (0, 1):                                                # pushed in descending (cAD1, bAD1, ..., text) order, instcode.hpp:62
        mine = [r for r in rows if r["strand"] == strand]
        assert mine == sorted(mine, key=lambda r: (r["cAD1"], r["bAD1"], r["seq"]), reverse=True)
    rec = R.score()
    m = (rec["refpos"] == beg + site) & (rec["gapSa_len"] > 0)
    total = collections.Counter()
    for (strand, s), v in expect.items(): total[s] += v
    keep = {s: v for s, v in total.items() if v >= (max(total.values()) + 3) // 4}
    want = sorted(keep.items(), key=lambda kv: -(kv[1] ** 2) * len(kv[0]))
    all_rows = R.indel_alleles()
    got = [(all_rows[i]["seq"], int(b)) for i, b in zip(rec["gapSa"][m], rec["bDPa"][m])]
    assert got == want and rec["cDP0a"][m].tolist() == [v for _, v in want] and rec["gapSa_len"][m].tolist() == [len(s) for s, _ in want]
