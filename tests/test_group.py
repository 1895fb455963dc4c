"""Family assignment (grouping.cpp:608-997, MolecularID.hpp, Hash.hpp).
CPU: the oracle against an independent pure-Python restatement (dict / set based, written from the reference text).
GPU: uvcgpu_group_families against the oracle: per-alignment outputs bit-exact, the family / fragment structure equal as a
partition (the order of families is the one thing the two are allowed to differ in, see include/uvcgroup.h)."""
import numpy as np
import pytest

from uvc_amd import group

M64 = (1 << 64) - 1
MAXINS, MARGIN, OUTER, INNER = 2000, 2000, 10, 3


def py_strnhash(s, base, n=None):
    r = 0
    for ch in s.encode()[:n]:
        if ch == 0:
            break
        r = (r * base + ch) & M64
    return r


def py_digest(qname, molecule_tag=0, disable_duplex=0):
    q31, q17 = py_strnhash(qname, 31), py_strnhash(qname, 17)
    i = qname.find("#")
    umi_beg = i + 1 if i >= 0 else len(qname)
    j = qname.find("#", umi_beg)
    umi_end = j if j >= 0 else len(qname)
    if not (umi_beg + 1 < umi_end and molecule_tag != 1):
        return 0, q31, q17, 0, 0
    umi = qname[umi_beg:umi_end]
    half = (len(umi) - 1) // 2
    duplex = (len(umi) % 2 == 1) and umi[half] == "+" and not disable_duplex
    return 1 | (2 if duplex else 0), q31, q17, py_strnhash(umi, 31), py_strnhash(umi, 17)


def make_alignments(seed, n_pairs=400, beg=100_000, length=3000, umi=False, amplicon=False):
    rng = np.random.default_rng(seed)
    rows = []
    for k in range(n_pairs):
        ins = int(rng.integers(180, 420))
        if amplicon and k % 2 == 0:
            start, ins = beg + 700, 300                       # a pile of identical inserts
        else:
            start = int(rng.integers(beg - 300, beg + length + 100))
        if k % 37 == 0:
            ins = 2500                                        # beyond MAX_INSERT_SIZE: isize normalised to 0
        name = "q%05d" % k
        if umi:
            a, b = "".join(rng.choice(list("ACGT"), 4)), "".join(rng.choice(list("ACGT"), 4))
            name += "#" + (a + "+" + b if k % 3 else a + b) + ("#x" if k % 5 == 0 else "")
        rl = 100
        top = bool(rng.integers(0, 2))
        jit = 1 if (amplicon and k % 2 == 0 and k % 8) else 3       # most amplicon inserts share their ends exactly
        p1, p2 = start + int(rng.integers(-jit + 1, jit)), start + ins - rl + int(rng.integers(-jit + 1, jit))
        f1 = 0x1 | 0x2 | (0x40 if top else 0x80) | 0x20
        f2 = 0x1 | 0x2 | (0x80 if top else 0x40) | 0x10
        mq = int(rng.integers(0, 61))
        rows.append((0, p1, p1 + rl, 0, p2, p2 + rl - p1, f1, mq, name))
        if k % 11:
            rows.append((0, p2, p2 + rl, 0, p1, -(p2 + rl - p1), f2, mq, name))
        if k % 23 == 0:
            rows.append((0, p1, p1 + 50, 0, p2, 0, f1 | 0x900, mq, name))          # secondary + supplementary
        if k % 29 == 0:
            rows.append((0, p1, p1 + 1, -1, -1, 0, 0x4, 0, "u%05d" % k))           # unmapped
        if k % 31 == 0:
            rows.append((0, start, start + 80, -1, -1, 0, 0x0, mq, "s%05d" % k))   # single-end
    rows.sort(key=lambda r: r[1])
    cols = dict(tid=[], pos=[], endpos=[], mtid=[], mpos=[], isize=[], flag=[], mapq=[], qname=[])
    for r in rows:
        for k, v in zip(cols, r):
            cols[k].append(v)
    d = [py_digest(q) for q in cols["qname"]]
    out = {k: np.array(v) for k, v in cols.items() if k != "qname"}
    out.update(umi_kind=np.array([x[0] for x in d], np.uint8), qname_hash31=np.array([x[1] for x in d], np.uint64), qname_hash17=np.array([x[2] for x in d], np.uint64),
               umi_hash31=np.array([x[3] for x in d], np.uint64), umi_hash17=np.array([x[4] for x in d], np.uint64))
    return out, cols["qname"], beg, beg + length


def py_prefilter(P, flag, mapq, pos, endpos, mpos, isize_raw):
    isize = 0 if abs(isize_raw) >= MAXINS else isize_raw
    min_mapqual, min_aln_len = P.kept_aln_min_aln_len, P.kept_aln_min_mapqual        # swapped at the call site, grouping.cpp:672-673
    merge = P.pair_end_merge == 0
    if flag & 0x4: return 1, 0, 0, 0, 0, isize
    if flag & 0x900: return 2, 0, 0, 0, 0, isize
    if mapq < min_mapqual: return 3, 0, 0, 0, 0, isize
    if endpos - pos < min_aln_len: return 4, 0, 0, 0, 0, isize
    if isize == 0:
        if P.kept_aln_is_zero_isize_discarded: return 7, 0, 0, 0, 0, isize
    else:
        if abs(isize) < P.kept_aln_min_isize: return 5, 0, 0, 0, 0, isize
        if abs(isize) > P.kept_aln_max_isize: return 6, 0, 0, 0, 0, isize
    isrc = int(bool(flag & 0x10))
    isr2 = int(bool(flag & 0x80) and bool(flag & 0x1)) if merge else 0
    b, e = pos, endpos - 1
    if (not merge) or not (flag & 1) or (flag & 8) or isize == 0 or abs(isize) >= MARGIN:
        tB, tE = (e, b) if isrc else (b, e)
    else:
        l = min(b, mpos); r = l + abs(isize) - 1
        strand = bool(flag & 0x20) if (flag & 0x81) == 0x81 else bool(flag & 0x10)
        tB, tE = (r, l) if strand else (l, r)
    oB, oE = min(tB, tE), max(tB, tE)
    if oB + (MARGIN - OUTER) <= P.fetch_tbeg or P.fetch_tend - 1 + (MARGIN - OUTER) <= oE: return 8, isrc, isr2, tB, tE, isize
    if P.end2end and not (oB <= P.fetch_tbeg and oE >= P.fetch_tend): return 9, isrc, isr2, tB, tE, isize
    return 0, isrc, isr2, tB, tE, isize


def py_group(P, c):
    n = len(c["pos"])
    size = P.fetch_tend - P.fetch_tbeg + (MARGIN + OUTER) * 2
    begc = np.zeros((4, size), np.int64); endc = np.zeros((4, size), np.int64)
    pre, visited = [], set()
    for i in range(n):
        r = py_prefilter(P, int(c["flag"][i]), int(c["mapq"][i]), int(c["pos"][i]), int(c["endpos"][i]), int(c["mpos"][i]), int(c["isize"][i]))
        pre.append(r)
        if r[0]: continue
        cl = r[1] * 2 + r[2]
        bi, ei = r[3] + MARGIN - P.fetch_tbeg, r[4] + MARGIN - P.fetch_tbeg
        if 0 <= bi < size: begc[cl, bi] += 1
        if 0 <= ei < size: endc[cl, ei] += 1
        if not (max(r[3], r[4]) + 2 <= P.fetch_tbeg or P.fetch_tend <= min(r[3], r[4])):
            visited.add((int(c["qname_hash31"][i]), int(c["qname_hash17"][i])))
    border = np.concatenate([np.zeros((4, 1), np.int64), np.cumsum(begc + endc, axis=1)], axis=1)

    def centers(cnt):
        cen = np.zeros(size, np.int64)
        for lo in range(INNER, size - INNER):
            cen[lo] = lo; mx = cnt[lo]
            for hi in range(lo - INNER, lo + INNER + 1):
                if cnt[hi] > mx and (cnt[hi] + 1) > (cnt[lo] + 1) * P.dedup_center_mult ** abs(lo - hi):
                    cen[lo] = hi; mx = cnt[hi]
        return cen
    # only the bins that are looked up matter; computing all of them is what the reference does
    b2c = [centers(begc[k]) for k in range(4)]; e2c = [centers(endc[k]) for k in range(4)]
    fams, reason, n_amp = {}, [], 0
    for i in range(n):
        r = pre[i]; why = r[0]
        if c["pos"][i] < max(P.fetch_tbeg - (MAXINS + 1), 0) or c["endpos"][i] > P.fetch_tend + MAXINS + 1: reason.append(100); continue
        if (int(c["qname_hash31"][i]), int(c["qname_hash17"][i])) not in visited: reason.append(101); continue
        reason.append(why)
        if why: continue
        flag, isize = int(c["flag"][i]), r[5]
        umi, dup = int(c["umi_kind"][i]) & 1, (int(c["umi_kind"][i]) >> 1) & 1
        cl = r[1] * 2 + r[2]
        beg2, end2 = int(b2c[cl][r[3] + MARGIN - P.fetch_tbeg]), int(e2c[cl][r[4] + MARGIN - P.fetch_tbeg])
        bc, ec = int(begc[cl, beg2]), int(endc[cl, end2])
        iL, iR = min(beg2 + 6, end2), max(beg2, max(end2 - 6, 0))
        tot = int(border[cl, iR] - border[cl, iL])
        br = (bc * (iR - iL) + 1) / (tot + (iR - iL) + 1); er = (ec * (iR - iL) + 1) / (tot + (iR - iL) + 1)
        ba = br > P.dedup_amplicon_border_to_insert_cov_weak_avgDP_ratio and bc >= P.dedup_amplicon_border_weak_minDP and bc >= tot * P.dedup_amplicon_border_to_insert_cov_weak_totDP_ratio
        ea = er > P.dedup_amplicon_border_to_insert_cov_weak_avgDP_ratio and ec >= P.dedup_amplicon_border_weak_minDP and ec >= tot * P.dedup_amplicon_border_to_insert_cov_weak_totDP_ratio
        bs = br > P.dedup_amplicon_border_to_insert_cov_strong_avgDP_ratio and bc >= P.dedup_amplicon_border_strong_minDP and bc >= tot * P.dedup_amplicon_border_to_insert_cov_strong_totDP_ratio
        es = er > P.dedup_amplicon_border_to_insert_cov_strong_avgDP_ratio and ec >= P.dedup_amplicon_border_strong_minDP and ec >= tot * P.dedup_amplicon_border_to_insert_cov_strong_totDP_ratio
        amp = bs or es or (ba and ea); n_amp += int(amp)
        if P.dedup_flag: idf = P.dedup_flag
        elif P.inferred_sequencing_platform == 2: idf = 0x9 if umi else (0x7 if amp else 0x3)
        elif umi:
            idf = 0x9 if (bs and ea and bc > ec * P.dedup_amplicon_end2end_ratio) else (0xA if (es and ba and ec > bc * P.dedup_amplicon_end2end_ratio) else 0xB)
        else: idf = 0x7 if amp else 0x3
        pres = bool(flag & 1) and not (flag & 4) and not (flag & 8) and (abs(isize) >= MAXINS * 3 // 4 or isize == 0)
        bp = (int(c["tid"][i]) if not (flag & 4) else 2**31 - 2, int(c["pos"][i]) if pres else beg2 - MARGIN + P.fetch_tbeg)
        ep = (int(c["mtid"][i]) if (flag & 1) and not (flag & 8) else 2**31 - 2, int(c["mpos"][i]) if pres else end2 - MARGIN + P.fetch_tbeg)
        strand = int(bool(flag & 0x20) if (flag & 0x81) == 0x81 else bool(flag & 0x10))
        dfl = umi + 2 * dup + (4 if amp else 0) + (8 if pres else 0)
        kb, ke = (-1, -1), (-1, -1)
        if idf & 3 == 3: kb, ke = min(bp, ep), max(bp, ep)
        elif idf & 1: kb = bp
        elif idf & 2: ke = ep
        key = (kb, ke, (int(c["qname_hash31"][i]), int(c["qname_hash17"][i])) if idf & 4 else 0, (int(c["umi_hash31"][i]), int(c["umi_hash17"][i])) if (idf & 8 and umi) else 0, dfl, idf)
        fams.setdefault(key, {}).setdefault((strand, int(c["qname_hash17"][i])), []).append(i)
    return np.array(reason), fams, n_amp, len(visited)


def canon(res):
    """family / fragment structure as order-free sets"""
    fam, frag = {}, {}
    for k, i in enumerate(res["order"]):
        fam.setdefault(int(res["fam_id"][k]), []).append((int(res["fam_strand"][k]), int(i)))
        frag.setdefault(int(res["frag_id"][k]), []).append(int(i))
    fams = sorted((tuple(sorted(v)), int(res["fam_dflag"][f]), int(res["fam_idflag"][f])) for f, v in fam.items())
    return fams, sorted(tuple(v) for v in frag.values())          # fragments keep file order inside


@pytest.mark.parametrize("case", ["plain", "umi", "amplicon", "end2end_nomerge"])
def test_oracle_against_python(oracle_lib, case):
    cols, qnames, tb, te = make_alignments(seed=hash(case) % 1000, umi=(case == "umi"), amplicon=(case == "amplicon"), n_pairs=1400 if case == "amplicon" else 300)
    P = group.default_params(oracle_lib, tb, te)
    if case == "end2end_nomerge":
        P.pair_end_merge, P.end2end, P.kept_aln_min_aln_len, P.kept_aln_min_mapqual = 1, 0, 10, 60
    res = group.group_families(oracle_lib, P, cols)
    reason, fams, n_amp, n_vis = py_group(P, cols)
    assert np.array_equal(res["filter_reason"], reason)
    assert res["n_amplicon"] == n_amp and res["n_visited_qnames"] == n_vis and res["n_fams"] == len(fams)
    exp = sorted((tuple(sorted((s, i) for (s, _), idx in fr.items() for i in idx)), key[4], key[5]) for key, fr in fams.items())
    got, frags = canon(res)
    assert got == exp
    assert frags == sorted(tuple(idx) for fr in fams.values() for idx in fr.values())
    assert (np.diff(res["fam_id"]) >= 0).all() and (np.diff(res["frag_id"]) >= 0).all()
    if case == "amplicon":
        assert n_amp > 100 and any(k[5] == 0x7 for k in fams)
    if case == "umi":
        assert any(k[5] == 0xB for k in fams) and any(k[4] & 2 for k in fams)


def test_hashes_and_digest(oracle_lib):
    for s in ["", "a", "read/1#ACGT+TTGA", "x#AC#tail", "noumi", "q#A", "#", "longer_name_with_many_chars_0123456789#ACGTAC+GTTGCA#z"]:
        for base in (31, 17):
            assert group.strnhash(oracle_lib, s, base) == py_strnhash(s, base)
        assert group.strnhash(oracle_lib, s, 31, n=3) == py_strnhash(s, 31, n=3)
        for mt, dd in ((0, 0), (1, 0), (3, 1)):
            assert group.qname_digest(oracle_lib, s, mt, dd) == py_digest(s, mt, dd)
    assert group.hash2hash(oracle_lib, 12345678901234567, 98765) == (12345678901234567 * ((1 << 31) - 1) + 98765) & M64


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["plain", "umi", "amplicon", "end2end_nomerge", "big"])
def test_gpu_against_oracle(oracle_lib, gpu_lib, case):
    if case == "big":
        cols, qnames, tb, te = make_alignments(seed=77, n_pairs=60_000, length=200_000, umi=True)
    else:
        cols, qnames, tb, te = make_alignments(seed=hash(case) % 1000, umi=(case == "umi"), amplicon=(case == "amplicon"), n_pairs=1400 if case == "amplicon" else 300)
    P = group.default_params(oracle_lib, tb, te)
    if case == "end2end_nomerge":
        P.pair_end_merge, P.end2end, P.kept_aln_min_aln_len, P.kept_aln_min_mapqual = 1, 0, 10, 60
    ro = group.group_families(oracle_lib, P, cols)
    rg = group.group_families(gpu_lib, P, cols)
    for k in ("filter_reason", "isize_norm"):
        assert np.array_equal(ro[k], rg[k]), k
    for k in ("n_kept", "n_fams", "n_frags", "ext_beg", "ext_end", "n_amplicon", "n_visited_qnames"):
        assert ro[k] == rg[k], (k, ro[k], rg[k])
    assert canon(ro) == canon(rg)
    assert (np.diff(rg["fam_id"]) >= 0).all() and (np.diff(rg["frag_id"]) >= 0).all()
    for s in ["q#ACGT+TTGA", "plain"]:
        assert group.qname_digest(gpu_lib, s) == group.qname_digest(oracle_lib, s)


def py_bam2umihash(pattern, bases, flag, kind0):
    """grouping.cpp:569-606, 787-792, restated independently on base codes 0..4."""
    nt16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    pat = [nt16.get(c.upper(), 15) for c in pattern]
    code16 = [1, 2, 4, 8, 15]
    rc = {1: 8, 2: 4, 4: 2, 8: 1}
    if (kind0 & 1) or (flag & 1) or not pat:
        return kind0, 0
    for is_rc in (False, True):
        for i in range(5):
            patpos, h = 0, 0
            for j in range(i, len(bases)):
                if patpos >= len(pat):
                    break
                b = code16[bases[len(bases) - 1 - j]] if is_rc else code16[bases[j]]
                if is_rc:
                    b = rc.get(b, b)
                if pat[patpos] == b or pat[patpos] == 15:
                    if pat[patpos] == 15:
                        h = (h * 16 + b) & M64
                    patpos += 1
                else:
                    break
            if patpos == len(pat):
                return kind0 | 1, h
    return kind0, 0


def test_in_read_umi_pattern(oracle_lib):
    """a11: bam2umihash.  Product (a host function of libuvcgpu: no GPU needed) and oracle against the restatement above."""
    import ctypes as C
    from uvc_amd import _ffi
    product = _ffi.Lib.__new__(_ffi.Lib); product.dll = C.CDLL(_ffi.gpu_library_path()); product.prefix = "uvcgpu_"
    rng = np.random.default_rng(3)
    pattern = "NNNACTNNNTGA"
    reads, flags, kinds = [], [], []
    def planted(offset, revcomp):
        umi = rng.integers(0, 4, 12)
        for k, ch in enumerate(pattern):
            if ch != "N":
                umi[k] = "ACGT".index(ch)
        body = rng.integers(0, 4, 60)
        seq = np.concatenate([rng.integers(0, 4, offset), umi, body])
        if revcomp:
            seq = (3 - seq)[::-1]
        return seq.astype(np.uint8)
    for off in range(0, 7):
        for revcomp in (False, True):
            reads.append(planted(off, revcomp)); flags.append(0); kinds.append(0)
    reads.append(planted(0, False)); flags.append(0x1); kinds.append(0)      # paired: not searched
    reads.append(planted(0, False)); flags.append(0); kinds.append(3)        # a UMI in the name wins
    for _ in range(40):
        reads.append(rng.integers(0, 5, int(rng.integers(8, 90))).astype(np.uint8)); flags.append(0); kinds.append(0)
    reads.append(np.array([0, 1, 2], np.uint8)); flags.append(0); kinds.append(0)   # shorter than the pattern
    cols = dict(bases=np.concatenate(reads), seq_off=np.cumsum([0] + [len(r) for r in reads[:-1]]), l_qseq=np.array([len(r) for r in reads]), flag=np.array(flags))
    want = [py_bam2umihash(pattern, [int(x) for x in r], f, k) for r, f, k in zip(reads, flags, kinds)]
    assert sum(k & 1 for k, _ in want) >= 11 and want[10][0] == 0 and want[11][0] == 0 and want[12][0] == 0 and want[13][0] == 0   # offsets 5 and 6 are out of reach
    for lib in (oracle_lib, product):
        kind = np.array(kinds, np.uint8)
        h = group.umi_in_read_batch(lib, pattern, cols, kind)
        assert [(int(k), int(x)) for k, x in zip(kind, h)] == want
        kind2 = np.array(kinds, np.uint8)
        group.umi_in_read_batch(lib, "", cols, kind2)
        assert kind2.tolist() == kinds
