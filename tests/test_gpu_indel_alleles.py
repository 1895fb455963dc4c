"""InDel allele tables (fill_by_indel_info / indel_get_majority, main.hpp:5350-5455): the rows the HIP path derives from its
allele-keyed counters and the per-allele score records must equal the oracle's -- multi-allelic insertion sites, deletions of
several lengths at one position, mates of a fragment that disagree, UMI families with a minority allele, duplex families, long
insertions (hashed allele codes), insertions that differ only in an N."""
import numpy as np
import pytest

from uvc_amd import region
from test_gpu_parity import compare_records
from util import diff_groups

pytestmark = pytest.mark.gpu

M, I, D, S = 0, 1, 2, 4
INS_ALLELES = [[3], [2], [3, 3], [0, 1, 2], [0, 4, 2], [3] * 15 + [0, 1, 2], [3] * 15 + [0, 1, 1], [1, 1, 1, 1], [4]]
DEL_ALLELES = [1, 2, 3, 5, 9]


def indel_sites_region(seed, n_fam=90, ref_len=420, beg=3_000_000, umi=True, depth_like=1, max_frags=7):
    rng = np.random.default_rng(seed)
    ref = rng.integers(0, 4, ref_len)
    ref[200:212] = 3                                       # a homopolymer: n_units != length for InDels inside it
    refseq = "".join("ACGT"[b] for b in ref)
    sites = [(60, I), (95, D), (130, I), (131, D), (205, I), (206, D), (260, I), (300, D), (301, I)]
    prefer = {s: int(rng.integers(0, 9)) for s in sites}   # the locally dominant allele
    cols = dict(pos=[], mpos=[], isize=[], flag=[], mapq=[], nm=[], l_qseq=[], seq_off=[], cigar_off=[], n_cigar=[], frag_id=[], fam_id=[], fam_strand=[])
    bases, quals, cigars, fam_dflag = [], [], [], []
    frag = 0
    for fam in range(n_fam):
        duplex = umi and rng.random() < 0.5
        fam_dflag.append((0x3 if duplex else 0x1) if umi else 0)
        fam_start = int(rng.integers(5, ref_len - 200))
        fam_allele = {s: (prefer[s] if rng.random() < 0.7 else int(rng.integers(0, 9))) for s in sites}
        fam_has = {s: rng.random() < 0.6 for s in sites}
        for strand in ((0, 1) if duplex else (int(rng.integers(0, 2)),)):
            for _ in range(int(rng.integers(1 if max_frags <= 7 else max_frags // 2, max_frags)) if umi else 1):
                for mate in range(int(rng.choice([1, 2, 2]))):
                    start = fam_start + (0 if mate == 0 else int(rng.integers(0, 60)))
                    length = int(rng.integers(90, 150))
                    ops, q, qq = [], [], []
                    rp = start
                    end = min(start + length, ref_len - 3)

                    def add_m(n):
                        nonlocal rp
                        if n <= 0: return
                        seg = ref[rp:rp + n].copy(); mis = rng.random(n) < 0.01; seg[mis] = rng.integers(0, 4, mis.sum())
                        ops.append((M, n)); q.extend(int(b) for b in seg); qq.extend(int(v) for v in rng.choice([30, 37, 41], n)); rp += n
                    for (sp, kind) in sites:
                        if not (rp + 6 < sp < end - 12) or not fam_has[(sp, kind)]: continue
                        al = fam_allele[(sp, kind)] if rng.random() < 0.85 else int(rng.integers(0, 9))   # within-family / between-mate disagreement
                        add_m(sp - rp)
                        if kind == I:
                            seq = INS_ALLELES[al % len(INS_ALLELES)]
                            ops.append((I, len(seq))); q.extend(seq); qq.extend(int(v) for v in rng.choice([8, 22, 30, 37, 41], len(seq)))
                        else:
                            dl = DEL_ALLELES[al % len(DEL_ALLELES)]
                            ops.append((D, dl)); rp += dl
                    add_m(end - rp)
                    if rng.random() < 0.3:
                        n = int(rng.integers(2, 12)); ops.append((S, n)); q.extend(int(b) for b in rng.integers(0, 4, n)); qq.extend([20] * n)
                    merged = []
                    for o, l in ops:
                        if merged and merged[-1][0] == o: merged[-1] = (o, merged[-1][1] + l)
                        else: merged.append((o, l))
                    fl = (0x1 | (0x40 if mate == 0 else 0x80) | (0x10 if (mate == 1) != (strand == 1) else 0x20))
                    cols["pos"].append(beg + start); cols["flag"].append(fl); cols["mapq"].append(60)
                    cols["mpos"].append(beg + fam_start); cols["isize"].append(int(rng.choice([180, -180, 260])))
                    cols["nm"].append(int(rng.choice([-1, 2, 6]))); cols["l_qseq"].append(len(q)); cols["seq_off"].append(len(bases)); cols["cigar_off"].append(len(cigars))
                    cols["n_cigar"].append(len(merged)); cols["frag_id"].append(frag); cols["fam_id"].append(fam); cols["fam_strand"].append(strand)
                    bases += q; quals += qq; cigars += [(l << 4) | o for o, l in merged]
                frag += 1
    dt = dict(pos=np.int32, mpos=np.int32, isize=np.int32, flag=np.uint16, mapq=np.uint8, nm=np.int32, l_qseq=np.int32, seq_off=np.int64, cigar_off=np.int64,
              n_cigar=np.int32, frag_id=np.int32, fam_id=np.int32, fam_strand=np.uint8)
    r = {k: np.array(v, dt[k]) for k, v in cols.items()}
    r.update(n_reads=len(cols["pos"]), tid=5, beg=beg, end=beg + ref_len, refseq=refseq, n_fams=n_fam, fam_dflag=np.array(fam_dflag, np.uint8),
             bases=np.array(bases, np.uint8), quals=np.array(quals, np.uint8), cigars=np.array(cigars, np.uint32))
    return r


def run(lib, reads, platform=1):
    R = region.Region(lib, region.default_params(lib, platform=platform), reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    R.set_reads(reads); R.accumulate()
    return R


@pytest.mark.parametrize("seed,umi,n_fam", [(1, True, 90), (2, True, 160), (3, False, 400), (4, False, 60), (5, True, 30), (6, True, 220)])
def test_allele_rows_and_records(seed, umi, n_fam, oracle_lib, gpu_lib):
    reads = indel_sites_region(seed, n_fam=n_fam, umi=umi)
    o, g = run(oracle_lib, reads), run(gpu_lib, reads)
    assert not diff_groups(o, g)
    ro, rg = o.indel_alleles(), g.indel_alleles()
    assert len(ro) > 20
    assert sum(r["c2AD"] for r in ro) > 0 or not umi
    assert len({(r["refpos"], r["symbol"], r["strand"]) for r in ro}) < len(ro)        # multi-allelic sites exist
    assert ro == rg, next((a, b) for a, b in zip(ro + [None], rg + [None]) if a != b)
    so, sg = o.score(all_out=False), g.score(all_out=False)
    compare_records(so, sg)
    ind = so["gapSa_len"] > 0
    assert ind.sum() > 10 and (so["gapSa"][ind] >= 0).all()
    keys = list(zip(so["refpos"][ind].tolist(), so["symbol"][ind].tolist()))
    assert len(set(keys)) < len(keys)                                                  # several records of one (refpos, symbol): one per allele
    for i in np.nonzero(ind)[0][:200]:                                                 # a record's depths are those of its allele rows
        row = ro[so["gapSa"][i]]
        same = [r for r in ro if (r["refpos"], r["symbol"], r["len"], r["seq"]) == (row["refpos"], row["symbol"], row["len"], row["seq"])]
        assert so["bDPa"][i] == sum(r["bAD1"] for r in same) and so["cDP0a"][i] == sum(r["cAD1"] for r in same)
    so, sg = o.score(all_out=True), g.score(all_out=True)
    compare_records(so, sg)


@pytest.mark.parametrize("seed,n_fam,max_frags", [(11, 8, 30), (12, 5, 90), (13, 12, 45)])
def test_deep_families(seed, n_fam, max_frags, oracle_lib, gpu_lib):
    """Families of a panel's depth: (family, position) runs of tens to hundreds of InDel events.  k_gap_alleles reads runs inside its block's
    128-event window from LDS and the others (longer runs, runs with an insertion of more than 13 bases) from global memory; k_gap_rows sums
    runs of hundreds of increments with one atomic per increment.  Rows and records must equal the oracle's either way."""
    reads = indel_sites_region(seed, n_fam=n_fam, umi=True, max_frags=max_frags)
    o, g = run(oracle_lib, reads), run(gpu_lib, reads)
    assert not diff_groups(o, g)
    ro, rg = o.indel_alleles(), g.indel_alleles()
    assert len(ro) > 10 and max(r["bAD1"] for r in ro) >= max_frags // 2
    assert ro == rg, next((a, b) for a, b in zip(ro + [None], rg + [None]) if a != b)
    compare_records(o.score(all_out=False), g.score(all_out=False))
    compare_records(o.score(all_out=True), g.score(all_out=True))


def test_caller_alleles_override_the_tables(oracle_lib, gpu_lib):
    reads = indel_sites_region(7, n_fam=80)
    o, g = run(oracle_lib, reads), run(gpu_lib, reads)
    base = o.score()
    ind = np.nonzero(base["gapSa_len"] > 0)[0]
    p, s = int(base["refpos"][ind[0]]), int(base["symbol"][ind[0]])
    override = [(p, s, 7, 5, 2), (p, s, 3, 1, 9)]
    so, sg = o.score(indel_alleles=override), g.score(indel_alleles=override)
    compare_records(so, sg)
    m = (so["refpos"] == p) & (so["symbol"] == s)
    assert so["bDPa"][m].tolist() == [7, 3] and so["gapSa"][m].tolist() == [-1, -1] and so["gapSa_len"][m].tolist() == [2, 9]
    assert (so["gapSa"][(so["gapSa_len"] > 0) & ~m] >= 0).all()


def _colliding_insertions(length=16, n=600_000, seed=123):
    """Two different insertions of `length` bases whose 34-bit allele codes (k_gap's FNV hash of insertions longer than 13 bases) agree:
    a birthday search over random sequences, with the hash restated here."""
    rng = np.random.default_rng(seed)
    seqs = rng.integers(0, 4, (n, length), dtype=np.uint8)
    h = np.full(n, np.uint64(1469598103934665603) ^ np.uint64(length), dtype=np.uint64)
    with np.errstate(over="ignore"):
        for i in range(length):
            h = (h ^ seqs[:, i].astype(np.uint64)) * np.uint64(1099511628211)
    code = h >> np.uint64(30)
    order = np.argsort(code, kind="stable")
    same = np.nonzero(code[order][1:] == code[order][:-1])[0]
    for k in same:
        a, b = seqs[order[k]], seqs[order[k + 1]]
        if not np.array_equal(a, b):
            return a.tolist(), b.tolist()
    raise AssertionError("no collision found: enlarge n")


def test_long_insertions_with_one_hash_stay_two_alleles(oracle_lib, gpu_lib, monkeypatch):
    """insertions longer than 13 bases are keyed by a 34-bit hash; two alleles of one site that collide must not be
    merged -- k_gap_rows splits a run of equal keys by comparing the sequences."""
    import test_gpu_indel_alleles as me
    a, b = _colliding_insertions()
    assert a != b and len(a) == len(b) == 16
    monkeypatch.setattr(me, "INS_ALLELES", [a, b, a, b, a, b, [2], a, b])
    for seed, umi in ((11, True), (12, False)):
        reads = indel_sites_region(seed, n_fam=140 if umi else 300, umi=umi)
        o, g = run(oracle_lib, reads), run(gpu_lib, reads)
        assert not diff_groups(o, g)
        ro, rg = o.indel_alleles(), g.indel_alleles()
        ta, tb = "".join("ACGT"[x] for x in a), "".join("ACGT"[x] for x in b)
        sites_a = {(r["refpos"], r["symbol"], r["strand"]) for r in ro if r["seq"] == ta}
        sites_b = {(r["refpos"], r["symbol"], r["strand"]) for r in ro if r["seq"] == tb}
        assert len(sites_a & sites_b) >= 2                       # both alleles on one (site, strand): the colliding keys meet
        assert ro == rg, next((x, y) for x, y in zip(ro + [None], rg + [None]) if x != y)
        compare_records(o.score(), g.score())


def big_family_region(two_alleles, n_frag=24, beg=3_000_000, ref_len=400):
    """One UMI family of `n_frag` single-read fragments that all carry a 3-base insertion right behind their first few bases; with
    `two_alleles` 14 of them insert ACG and the others ACT (one insertion symbol, two sequences)."""
    rng = np.random.default_rng(4)
    ref = rng.integers(0, 4, ref_len)
    refseq = "".join("ACGT"[b] for b in ref)
    cols = dict(pos=[], mpos=[], isize=[], flag=[], mapq=[], nm=[], l_qseq=[], seq_off=[], cigar_off=[], n_cigar=[], frag_id=[], fam_id=[], fam_strand=[])
    bases, quals, cigars = [], [], []
    frag = 0
    for fam, site_off in enumerate(range(4, 40, 3)):           # insertion 4 .. 37 bases behind the start of the family's reads
        start = 20 + 3 * fam
        for k in range(n_frag):
            ins = [0, 1, 2] if (not two_alleles or k < 14) else [0, 1, 3]
            left, right = site_off, 100
            q = [int(x) for x in ref[start:start + left]] + ins + [int(x) for x in ref[start + left:start + left + right]]
            cols["pos"].append(beg + start); cols["flag"].append(0); cols["mapq"].append(60); cols["mpos"].append(beg + start); cols["isize"].append(0)
            cols["nm"].append(3); cols["l_qseq"].append(len(q)); cols["seq_off"].append(len(bases)); cols["cigar_off"].append(len(cigars)); cols["n_cigar"].append(3)
            cols["frag_id"].append(frag); cols["fam_id"].append(fam); cols["fam_strand"].append(0)
            bases += q; quals += [37] * len(q); cigars += [(left << 4) | M, (3 << 4) | I, (right << 4) | M]
            frag += 1
    n_fam = fam + 1
    dt = dict(pos=np.int32, mpos=np.int32, isize=np.int32, flag=np.uint16, mapq=np.uint8, nm=np.int32, l_qseq=np.int32, seq_off=np.int64, cigar_off=np.int64,
              n_cigar=np.int32, frag_id=np.int32, fam_id=np.int32, fam_strand=np.uint8)
    r = {k: np.array(v, dt[k]) for k, v in cols.items()}
    r.update(n_reads=len(cols["pos"]), tid=5, beg=beg, end=beg + ref_len, refseq=refseq, n_fams=n_fam, fam_dflag=np.full(n_fam, 0x1, np.uint8),
             bases=np.array(bases, np.uint8), quals=np.array(quals, np.uint8), cigars=np.array(cigars, np.uint32))
    return r


def test_family_position_bias_counts_the_majority_sequence(oracle_lib, gpu_lib):
    """`indel_len` of the FAM2 position-bias test is the number of fragments that carry the MAJORITY inserted
    sequence (read_family_con_ampl_getMajority_ins, main.hpp:188-198, 3239-3246), not the number of votes for the insertion symbol.  Families
    of 24 fragments, 14 with one inserted sequence and 10 with another, at sites 4 .. 37 bases behind the read start: the two counts (14 and
    24) fall on different sides of microadjust_nobias_pos_indel_maxlen + the distance threshold at some of the sites."""
    mixed, pure = big_family_region(True), big_family_region(False)
    om, op = run(oracle_lib, mixed), run(oracle_lib, pure)
    fm, fp = om.fetch("FAMINFO32"), op.fetch("FAMINFO32")
    ins = slice(10, 13)                                          # LINK_I3P .. LINK_I1
    assert not np.array_equal(fm[:, ins, :], fp[:, ins, :])      # the oracle is sensitive to the count at these sites: the case does provoke the difference
    gm = run(gpu_lib, mixed)
    assert not diff_groups(om, gm)
    gp = run(gpu_lib, pure)
    assert not diff_groups(op, gp)
    compare_records(om.score(), gm.score())
