"""Synthetic pileup regions (SURVEY.md section 8d "Synthetic inputs").

The benchmark and the parity tests run on in-memory regions laid out as the `alns3`-equivalent SoA of
include/uvcgpu.h (`UvcReadSoA`); the same dicts become BAM files through tests/bamwriter.py for the
file-level tests (libuvcio reads BAM / BAI / FASTA itself, htslib is not available here).  The generator follows the SURVEY recipe: uniform random reference with planted
homopolymer and (AC)n tracks, 150-bp paired-end reads, insert ~N(350, 50) clipped to [200, 600],
both orientations 50/50, MAPQ 60 (5 % at 20-40), base quality 30+U{0..7} with the last 15
sequenced bases degraded, 1e-3 sequencing errors, germline-like SNVs every 1 kb (AF 0.5),
somatic-like SNVs every 10 kb (AF 0.02-0.1), 1-bp / 3-bp InDels every 5 kb, 1 % soft-clipped
reads, NM tag present.  Family assignment mirrors the reference's non-UMI capture key
(dedup_idflag 0x3 = begin + end position, grouping.cpp:838-840); the UMI mode builds duplex
families (duplexflag 0x3, grouping.cpp:931).

Everything here is host-side input preparation; it is not on the measured path.
"""
import numpy as np

READ_LEN = 150
HALO = 100  # MAX_STR_N_BASES, common.hpp:63

BAM_CMATCH, BAM_CINS, BAM_CDEL, BAM_CSOFT_CLIP = 0, 1, 2, 4


def make_reference(rng, n):
    ref = rng.integers(0, 4, size=n, dtype=np.uint8)
    # homopolymer runs 8-20 bp every ~2 kb, (AC)n every ~5 kb
    for p in range(1000, n - 100, 2000):
        q = p + int(rng.integers(0, 500))
        ln = int(rng.integers(8, 21))
        ref[q:q + ln] = rng.integers(0, 4)
    for p in range(2500, n - 100, 5000):
        q = p + int(rng.integers(0, 500))
        ln = int(rng.integers(5, 13))
        ref[q:q + 2 * ln:2] = 0
        ref[q + 1:q + 2 * ln:2] = 1
    return ref


def _pack(op, ln):
    return (np.uint32(ln) << np.uint32(4)) | np.uint32(op)


def generate_region(seed=12345, region_len=10000, depth=30, tid=19, beg=1000000, umi=False,
                    fam_mean=4.0, duplex_frac=0.6, snv_every=1000, somatic_every=10000, indel_every=5000,
                    err_rate=1e-3, clip_frac=0.01, dedup_by_position=True):
    """Returns a dict: refseq (str), tid, beg, end and the UvcReadSoA arrays (numpy)."""
    rng = np.random.default_rng(seed)
    n = int(region_len) + 2 * HALO
    ref = make_reference(rng, n)
    L = READ_LEN
    lo, hi = HALO, n - HALO          # reads live inside [lo, hi) (relative coordinates)
    # ---- molecules (fragments before PCR) ----
    mean_ins = 350
    n_pairs_target = int(round(depth * (hi - lo) / (2.0 * L)))
    if umi:
        n_mol = max(1, int(round(n_pairs_target / ((fam_mean + 1.0) * (1.0 + duplex_frac)))))
    else:
        n_mol = max(1, n_pairs_target)
    ins = np.clip(np.rint(rng.normal(mean_ins, 50, n_mol)), 200, 600).astype(np.int64)
    start = rng.integers(lo - 100, hi - 200 + 100, n_mol)
    start = np.clip(start, lo, hi - ins)   # keep the whole insert inside the region
    order = np.argsort(start, kind="stable")
    start, ins = start[order], ins[order]
    mol_top = rng.random(n_mol) < 0.5      # orientation of the (first) strand family
    mol_mapq = np.where(rng.random(n_mol) < 0.05, rng.integers(20, 41, n_mol), 60).astype(np.uint8)

    # ---- variants (relative coordinates) ----
    variants = []  # (pos, kind, arg, af)
    for p in (range(lo + 500, hi - 500, snv_every) if snv_every else ()):
        q = p + int(rng.integers(0, 200))
        variants.append((q, "snv", int((ref[q] + 1 + rng.integers(0, 3)) % 4), 0.5))
    for p in (range(lo + 700, hi - 500, somatic_every) if somatic_every else ()):
        q = p + int(rng.integers(0, 200))
        variants.append((q, "snv", int((ref[q] + 1 + rng.integers(0, 3)) % 4), float(rng.uniform(0.02, 0.1))))
    k = 0
    for p in (range(lo + 900, hi - 500, indel_every) if indel_every else ()):
        q = p + int(rng.integers(0, 200))
        kind = ("del", "ins")[k % 2]
        ln = (1, 3)[(k // 2) % 2]
        arg = ln if kind == "del" else rng.integers(0, 4, ln).astype(np.uint8)
        variants.append((q, kind, arg, 0.5 if k % 3 else 0.1))
        k += 1
    variants.sort(key=lambda v: v[0])

    # ---- expand molecules into sequenced fragments (families) ----
    if umi:
        fam_sizes_a = rng.poisson(fam_mean, n_mol) + 1
        has_b = rng.random(n_mol) < duplex_frac
        fam_sizes_b = np.where(has_b, rng.poisson(fam_mean, n_mol) + 1, 0)
    else:
        fam_sizes_a = np.ones(n_mol, dtype=np.int64)
        fam_sizes_b = np.zeros(n_mol, dtype=np.int64)
    # molecule-level allele choice (so that all PCR copies of a molecule agree)
    mol_alt = [rng.random(n_mol) < v[3] for v in variants]

    frag_mol = np.concatenate([np.repeat(np.arange(n_mol), fam_sizes_a), np.repeat(np.arange(n_mol), fam_sizes_b)])
    frag_isb = np.concatenate([np.zeros(int(fam_sizes_a.sum()), bool), np.ones(int(fam_sizes_b.sum()), bool)])
    o2 = np.lexsort((frag_isb, frag_mol))
    frag_mol, frag_isb = frag_mol[o2], frag_isb[o2]
    n_frag = frag_mol.size
    frag_top = np.where(frag_isb, ~mol_top[frag_mol], mol_top[frag_mol])   # strand slot 0 = R1 forward (flags 99/147)

    # ---- reads: two per fragment ----
    n_reads = 2 * n_frag
    r_frag = np.repeat(np.arange(n_frag), 2)
    r_is2 = np.tile(np.array([False, True]), n_frag)
    f_start, f_ins = start[frag_mol], ins[frag_mol]
    left_pos = f_start
    right_pos = f_start + f_ins - L
    # slot-0 fragments: R1 = left/forward, R2 = right/reverse; slot-1 fragments: R1 = right/reverse, R2 = left/forward
    r_top = frag_top[r_frag]
    r_isleft = np.where(r_top, ~r_is2, r_is2)
    r_pos = np.where(r_isleft, left_pos[r_frag], right_pos[r_frag]).astype(np.int64)
    r_rev = ~r_isleft
    r_mpos = np.where(r_isleft, right_pos[r_frag], left_pos[r_frag]).astype(np.int64)
    r_isize = np.where(r_isleft, f_ins[r_frag], -f_ins[r_frag]).astype(np.int64)
    flag = np.full(n_reads, 0x1 | 0x2, dtype=np.int64)
    flag |= np.where(r_rev, 0x10, 0x20)
    flag |= np.where(r_is2, 0x80, 0x40)

    idx = r_pos[:, None] + np.arange(L)[None, :]
    bases = ref[idx].copy()                                   # [n_reads, L]
    quals = rng.integers(30, 38, size=(n_reads, L), dtype=np.uint8)
    degr = rng.integers(0, 16, size=(n_reads, 15), dtype=np.uint8)
    fw = ~r_rev
    quals[fw, L - 15:] -= degr[fw]
    quals[r_rev, :15] -= degr[r_rev]

    # SNVs (molecule-level genotype), applied where the read covers the site
    cig_special = {}  # read index -> (bases, quals, cigar list, pos)
    r_mol = frag_mol[r_frag]
    by_pos = np.argsort(r_pos, kind="stable")
    sorted_pos = r_pos[by_pos]
    for vi, (vp, kind, arg, af) in enumerate(variants):
        cand = by_pos[np.searchsorted(sorted_pos, vp - L + 1, side="left"):np.searchsorted(sorted_pos, vp, side="right")]
        cover = np.sort(cand[mol_alt[vi][r_mol[cand]]])
        if kind == "snv":
            bases[cover, vp - r_pos[cover]] = arg
        else:
            for ri in cover:
                if ri in cig_special:
                    continue
                off = int(vp - r_pos[ri])
                ln = int(arg) if kind == "del" else len(arg)
                if off < 8 or off > L - 8 - ln:
                    continue
                p0 = int(r_pos[ri])
                if kind == "del":      # anchor base at vp, bases vp+1 .. vp+ln deleted
                    src = np.concatenate([np.arange(p0, vp + 1), np.arange(vp + 1 + ln, vp + 1 + ln + (L - off - 1))])
                    if src[-1] >= hi:
                        continue
                    b = ref[src].copy()
                    cig = [(BAM_CMATCH, off + 1), (BAM_CDEL, ln), (BAM_CMATCH, L - off - 1)]
                else:                  # inserted bases after the anchor at vp
                    left = ref[p0:vp + 1]
                    rest = L - (off + 1) - ln
                    b = np.concatenate([left, arg, ref[vp + 1:vp + 1 + rest]])
                    cig = [(BAM_CMATCH, off + 1), (BAM_CINS, ln), (BAM_CMATCH, rest)]
                cig_special[int(ri)] = [b.astype(np.uint8), cig]
    for ri, (b, cig) in cig_special.items():
        bases[ri] = b

    # sequencing errors
    flat = bases.reshape(-1)
    nerr = int(rng.binomial(flat.size, err_rate))
    eidx = rng.integers(0, flat.size, nerr)
    flat[eidx] = (flat[eidx] + rng.integers(1, 4, nerr).astype(np.uint8)) % 4
    # occasional N
    nidx = rng.integers(0, flat.size, int(rng.binomial(flat.size, 2e-5)))
    flat[nidx] = 4

    # soft clips on ~1 % of the plain reads
    clip_reads = np.nonzero(rng.random(n_reads) < clip_frac)[0]
    clips = {}
    for ri in clip_reads:
        if int(ri) in cig_special:
            continue
        c = int(rng.integers(10, 31))
        side_left = bool(rng.random() < 0.5)
        clips[int(ri)] = (c, side_left)
        if side_left:
            bases[ri, :c] = rng.integers(0, 4, c)
        else:
            bases[ri, L - c:] = rng.integers(0, 4, c)

    # ---- CIGARs, end positions, NM ----
    n_cigar = np.ones(n_reads, dtype=np.int32)
    pos_out = r_pos.copy()
    nm = np.count_nonzero(bases != ref[idx], axis=1).astype(np.int32)
    special = {}
    for ri, (b, cig) in cig_special.items():
        q = 0; r = int(r_pos[ri]); mm = 0      # NM = mismatches in aligned bases + indel length
        for op, ln in cig:
            if op == BAM_CMATCH:
                mm += int((bases[ri, q:q + ln] != ref[r:r + ln]).sum()); q += ln; r += ln
            elif op == BAM_CINS:
                mm += ln; q += ln
            else:
                mm += ln; r += ln
        nm[ri] = mm
        special[ri] = cig
    for ri, (c, side_left) in clips.items():
        if side_left:
            special[ri] = [(BAM_CSOFT_CLIP, c), (BAM_CMATCH, L - c)]
            pos_out[ri] = r_pos[ri] + c
            nm[ri] = int((bases[ri, c:] != ref[r_pos[ri] + c:r_pos[ri] + L]).sum())
        else:
            special[ri] = [(BAM_CMATCH, L - c), (BAM_CSOFT_CLIP, c)]
            nm[ri] = int((bases[ri, :L - c] != ref[r_pos[ri]:r_pos[ri] + L - c]).sum())
    for ri, cig in special.items():
        n_cigar[ri] = len(cig)
    cigar_off = np.zeros(n_reads, dtype=np.int64)
    cigar_off[1:] = np.cumsum(n_cigar[:-1])
    cigars = np.full(int(n_cigar.sum()), _pack(BAM_CMATCH, L), dtype=np.uint32)
    for ri, cig in special.items():
        for j, (op, ln) in enumerate(cig):
            cigars[cigar_off[ri] + j] = _pack(op, ln)

    # mate positions refer to the (possibly clip-shifted) mate start
    mate = np.arange(n_reads) ^ 1
    r_mpos = pos_out[mate]

    # ---- families ----
    if umi:
        # one family per molecule; the two strand families of a duplex share the family (alpha+beta UMI, grouping.cpp:778-783)
        fam_of_frag = frag_mol.copy()
        fam_dflag_val = 0x1 | 0x2
    elif dedup_by_position:
        # dedup_idflag 0x3: key = (begin, end) of the insert; both orientations share the family
        key = f_start * 4096 + f_ins
        _, fam_of_frag = np.unique(key, return_inverse=True)
        fam_dflag_val = 0x0
    else:
        fam_of_frag = np.arange(n_frag)
        fam_dflag_val = 0x0
    # contiguous nesting: sort reads by (family, strand slot, fragment)
    r_fam = fam_of_frag[r_frag]
    r_strand = np.where(frag_top[r_frag], 0, 1).astype(np.uint8)
    ordr = np.lexsort((r_is2, r_frag, r_strand, r_fam))
    uniq_f, fam_id = np.unique(r_fam[ordr], return_inverse=True)
    n_fams = int(uniq_f.size)

    def take(a):
        return np.ascontiguousarray(a[ordr])

    seq_off = np.arange(n_reads, dtype=np.int64) * L
    out = dict(
        tid=tid, beg=beg, end=beg + n, refseq="".join("ACGT"[b] for b in ref),
        n_reads=n_reads,
        pos=take(pos_out + beg).astype(np.int32), mpos=take(r_mpos + beg).astype(np.int32), isize=take(r_isize).astype(np.int32),
        flag=take(flag).astype(np.uint16), mapq=take(mol_mapq[r_mol]).astype(np.uint8), nm=take(nm).astype(np.int32),
        l_qseq=np.full(n_reads, L, dtype=np.int32), seq_off=seq_off,
        n_cigar=take(n_cigar).astype(np.int32), frag_id=take(r_frag).astype(np.int32),
        fam_id=fam_id.astype(np.int32), fam_strand=take(r_strand),
        bases=np.ascontiguousarray(bases[ordr].reshape(-1)).astype(np.uint8),
        quals=np.ascontiguousarray(quals[ordr].reshape(-1)).astype(np.uint8),
        n_fams=n_fams, fam_dflag=np.full(n_fams, fam_dflag_val, dtype=np.uint8),
        variants=[(int(v[0] + beg), v[1], (v[2].tolist() if hasattr(v[2], "tolist") else int(v[2])), float(v[3])) for v in variants],
    )
    # re-pack CIGARs in the new read order
    nc = out["n_cigar"]
    new_off = np.zeros(n_reads, dtype=np.int64)
    new_off[1:] = np.cumsum(nc[:-1])
    old_off = cigar_off[ordr]
    newc = np.empty(int(nc.sum()), dtype=np.uint32)
    single = nc == 1
    newc[new_off[single]] = cigars[old_off[single]]
    for i in np.nonzero(~single)[0]:
        newc[new_off[i]:new_off[i] + nc[i]] = cigars[old_off[i]:old_off[i] + nc[i]]
    out["cigars"] = newc
    out["cigar_off"] = new_off
    return out
