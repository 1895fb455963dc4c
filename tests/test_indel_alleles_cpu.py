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


def test_allele_rows_against_the_independent_restatements(oracle_lib):
    """Every row fill_by_indel_info reads (fragment, family, cDP2 and duplex support of every allele, per strand) on fuzzed reads: the oracle's
    tables against the allele-keyed maps of the restated fragment and family passes (tests/p3_restatement.py, tests/p45_restatement.py,
    tests/indel_alleles_restatement.py), chained behind the other restatements -- no oracle value enters."""
    import importlib.util, os
    from rtr_cases import python_tracks
    from prep_restatement import prep_sets, thres_sets
    from p2_restatement import update_by_aln
    from p3_restatement import fragment_pass
    from p45_restatement import family_passes
    from indel_alleles_restatement import allele_rows
    from util import run_region
    spec = importlib.util.spec_from_file_location("fz_alleles", os.path.join(os.path.dirname(__file__), "test_gpu_fuzz.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    n_rows = 0
    for seed, umi, platform in ((71, True, 1), (72, False, 1), (73, True, 2)):
        reads = fz.weird_region(seed, n_frag=150, ref_len=420, umi=umi)
        P = region.default_params(oracle_lib, platform=platform)
        R = run_region(oracle_lib, reads, params=P)
        rtr, baq = python_tracks(reads["refseq"], smax=P.indel_str_repeatsize_max, vmax=P.indel_vntr_repeatsize_max, bq_max=P.indel_BQ_max,
                                 slip_rate=P.indel_polymerase_slip_rate, del_to_ins=P.indel_del_to_ins_err_ratio, polymerase_size=P.indel_polymerase_size,
                                 str_phred_per_region=P.indel_str_phred_per_region, nonstr_phred_per_base=P.indel_nonSTR_phred_per_base)
        codes = np.array([{"A": 0, "C": 1, "G": 2, "T": 3}.get(c.upper(), 4) for c in reads["refseq"]], dtype=np.int32)
        prep = prep_sets(reads, P, rtr, baq[0], np.append(codes, 4))
        proton = (platform == 2)
        thres, ip = thres_sets(prep, rtr[3], P, is_normal=False, iontorrent=proton)
        seg, bqsum = update_by_aln(reads, P, rtr, ip, baq[0], baq[1], codes, prep, thres, proton)
        al = {}
        fragment_pass(reads, P, rtr, ip, baq[0], codes, prep, thres, seg, bqsum, proton, alleles=al)
        family_passes(reads, P, rtr, ip, baq[0], baq[1], codes, prep, thres, proton=proton, alleles=al)
        want = allele_rows(al, reads["refseq"], reads["beg"])
        got = {}
        for r in R.indel_alleles():
            x = r["refpos"] - reads["beg"]
            text = r["seq"] if r["seq"] is not None else reads["refseq"][x:x + r["len"]]
            got[(r["refpos"], r["symbol"], r["strand"], text)] = (r["bAD1"], r["cAD1"], r["c2AD"], r["c2dAD"])
        assert got == want, sorted(set(got.items()) ^ set(want.items()))[:6]
        n_rows += len(want)
        R.close()
    assert n_rows > 1000
