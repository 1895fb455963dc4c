"""apply_bq_err_correction3 (grouping.cpp:459-543): the oracle against an independent pure-Python restatement written from
the reference text, on hand-made reads (long soft clip, 3' homopolymer, poly-G) and on a seeded synthetic region; and
the HIP kernel against the oracle (qualities bit-exact, then every plane after accumulate)."""
import numpy as np
import pytest

from uvc_amd import region, synth
from util import diff_groups

S_CLIP = 4


def py_correct(bases, quals, flag, cigar, bq_max, bq_inc):
    """Line-by-line restatement on BAM 4-bit codes (A=1, C=2, G=4, T=8, N=15)."""
    l = len(bases)
    q = [min(int(v) + bq_inc, bq_max) for v in quals]
    code = [1 << int(b) if b < 4 else 15 for b in bases]
    if l == 0 or (flag & 0x4):
        return list(quals)
    isrc = 1 if (flag & 0x10) else 0
    inclu = [0, l - 1]
    exclu = [l, -1]
    end_clip = 0
    if cigar:
        op, ln = cigar[0]
        if op == S_CLIP:
            if isrc == 0:
                inclu[0] += ln
            else:
                exclu[1] += ln
                end_clip = ln
        op, ln = cigar[-1]
        if op == S_CLIP:
            if isrc == 1:
                inclu[1] -= ln
            else:
                exclu[0] -= ln
                end_clip = ln
    inc = -1 if isrc else 1
    prev_b, distinct = 0, 0
    start = exclu[isrc] - inc
    termpos = start
    while termpos != inclu[isrc] - inc:
        if code[termpos] != prev_b and q[termpos] >= 20:
            prev_b = code[termpos]
            distinct += 1
            if distinct == 2:
                break
        termpos -= inc
    track = abs(termpos - start)
    penal = (1 if end_clip >= 20 else 0) + (2 if track >= 15 else (1 if track >= 10 else 0))
    if penal > 0:
        pos = start
        while pos != inclu[isrc] - inc and pos != termpos:
            q[pos] = max(q[pos], penal + 1) - penal
            pos -= inc
    hl, prev_b = 0, 0
    pos = inclu[isrc]
    while pos != exclu[isrc]:
        if code[pos] == prev_b:
            hl += 1
            if hl >= 4 and code[pos] == 4:
                q[pos] = max(q[pos], 2) - 1
        else:
            prev_b, hl = code[pos], 1
        pos += inc
    return q


def make_reads(specs, beg=1_000_000, ref_len=400):
    """specs: list of (pos_offset, flag, cigar[(op,len)], bases, quals); one read per fragment / family."""
    rng = np.random.default_rng(0)
    refseq = "".join("ACGT"[i] for i in rng.integers(0, 4, ref_len))
    n = len(specs)
    r = dict(n_reads=n, tid=19, beg=beg, end=beg + ref_len, refseq=refseq, n_fams=n, fam_dflag=np.zeros(n, np.uint8))
    pos, flag, lq, so, co, nc, bases, quals, cig = [], [], [], [], [], [], [], [], []
    for off, fl, cg, b, q in specs:
        pos.append(beg + off); flag.append(fl); lq.append(len(b)); so.append(len(bases)); co.append(len(cig)); nc.append(len(cg))
        bases += list(b); quals += list(q); cig += [(ln << 4) | op for op, ln in cg]
    r.update(pos=np.array(pos, np.int32), mpos=np.array(pos, np.int32), isize=np.zeros(n, np.int32), flag=np.array(flag, np.uint16),
             mapq=np.full(n, 60, np.uint8), nm=np.full(n, -1, np.int32), l_qseq=np.array(lq, np.int32), seq_off=np.array(so, np.int64),
             cigar_off=np.array(co, np.int64), n_cigar=np.array(nc, np.int32), frag_id=np.arange(n, dtype=np.int32),
             fam_id=np.arange(n, dtype=np.int32), fam_strand=np.zeros(n, np.uint8), bases=np.array(bases, np.uint8),
             quals=np.array(quals, np.uint8), cigars=np.array(cig, np.uint32))
    return r


def hand_made():
    A, C_, G, T = 0, 1, 2, 3
    specs = []
    # forward read, 25-base soft clip at the 3' end, 12-base poly-T tail before it
    b = [A, C_, G, T] * 10 + [T] * 12 + [C_] * 25
    specs.append((10, 0x0, [(0, 52), (S_CLIP, 25)], b, [30] * len(b)))
    # reverse read: its 3' end is the left end; 22-base clip there, 16-base poly-A after it
    b = [G] * 22 + [A] * 16 + [A, C_, G, T] * 10
    specs.append((20, 0x10, [(S_CLIP, 22), (0, 56)], b, [25] * len(b)))
    # poly-G in the middle and low qualities inside the tail scan
    b = [A, C_] * 5 + [G] * 9 + [T, A] * 8 + [C_] * 11
    q = [35] * 10 + [30] * 9 + [12, 35] * 8 + [19] * 5 + [33] * 6
    specs.append((30, 0x0, [(0, len(b))], b, q))
    # qualities of 0 / 1 / 2 next to the floors "MAX(q, penal + 1) - penal"
    b = [C_] * 20 + [G] * 6 + [T] * 14
    q = [0, 1, 2, 3] * 10
    specs.append((40, 0x10, [(0, 40)], b, q))
    # N bases
    b = [4] * 5 + [A, G, G, G, G, G, 4, 4] + [T] * 12
    specs.append((50, 0x0, [(0, len(b))], b, [22] * len(b)))
    return make_reads(specs)


def expected_quals(reads, bq_max, bq_inc):
    out = reads["quals"].copy()
    for i in range(int(reads["n_reads"])):
        s, l = int(reads["seq_off"][i]), int(reads["l_qseq"][i])
        cg = [(int(c) & 0xF, int(c) >> 4) for c in reads["cigars"][int(reads["cigar_off"][i]):int(reads["cigar_off"][i]) + int(reads["n_cigar"][i])]]
        out[s:s + l] = py_correct(reads["bases"][s:s + l], reads["quals"][s:s + l], int(reads["flag"][i]), cg, bq_max, bq_inc)
    return out


def corrected(lib, reads, bq_max, bq_inc, accumulate=False):
    p = region.default_params(lib)
    p.assay_sequencing_BQ_max, p.assay_sequencing_BQ_inc = bq_max, bq_inc
    R = region.Region(lib, p, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    R.set_reads(reads)
    R.correct_bq()
    q = R.read_quals(len(reads["quals"]))
    if accumulate:
        R.accumulate()
    return R, q


@pytest.mark.parametrize("bq_max,bq_inc", [(37, 0), (41, 4), (30, 2)])
def test_oracle_hand_made(oracle_lib, bq_max, bq_inc):
    reads = hand_made()
    _, q = corrected(oracle_lib, reads, bq_max, bq_inc)
    exp = expected_quals(reads, bq_max, bq_inc)
    assert np.array_equal(q, exp), np.flatnonzero(q != exp)[:10]
    assert (q != np.minimum(reads["quals"].astype(int) + bq_inc, bq_max)).any()   # the penalties did fire


def test_oracle_synthetic(oracle_lib):
    reads = synth.generate_region(seed=31, region_len=3000, depth=40, clip_frac=0.3)
    _, q = corrected(oracle_lib, reads, 37, 1)
    assert np.array_equal(q, expected_quals(reads, 37, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("bq_max,bq_inc", [(37, 0), (41, 4)])
def test_gpu_hand_made(oracle_lib, gpu_lib, bq_max, bq_inc):
    reads = hand_made()
    _, qo = corrected(oracle_lib, reads, bq_max, bq_inc)
    _, qg = corrected(gpu_lib, reads, bq_max, bq_inc)
    assert np.array_equal(qo, qg), np.flatnonzero(qo != qg)[:10]


@pytest.mark.gpu
def test_gpu_synthetic_then_accumulate(oracle_lib, gpu_lib):
    """Corrected qualities feed every later pass: planes must still match bit for bit (this also moves reads across the
    low-quality-InDel eligibility test of the P2 work list)."""
    reads = synth.generate_region(seed=33, region_len=4000, depth=80, clip_frac=0.3, indel_every=400)
    Ro, qo = corrected(oracle_lib, reads, 37, 2, accumulate=True)
    Rg, qg = corrected(gpu_lib, reads, 37, 2, accumulate=True)
    assert np.array_equal(qo, qg)
    bad = diff_groups(Ro, Rg)
    assert not bad, "\n".join("%s: %d cells differ, e.g. %s" % (g, v[0], v[1]) for g, v in bad.items())
