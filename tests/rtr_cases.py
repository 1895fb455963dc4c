"""Fuzzed reference strings for the region side arrays (refstring2repeatvec main.hpp:803-874, the BAQ prefix sums
main.cpp:400-429) and an independent pure-Python restatement of both, written from the reference text -- not from oracle/ and not from
uvc_rtr.hip -- that the CPU suite holds the oracle against (tests/test_rtr_cpu.py); the GPU suite holds the kernels against the oracle on
the same strings (tests/test_gpu_rtr.py)."""
import math

import numpy as np


def fuzz_reference(seed, n, kinds="all"):
    """Random ACGT with planted repeat structure: homopolymers, di-/tri-/hexanucleotide STRs, VNTRs with 7..35-base units of 200..3000 bp
    (longer than the kernels' 1024-base push window and their 2048-base chunks), imperfect repeats, runs of N, soft-masked (lower-case)
    stretches, and repeats that touch either end of the region."""
    rng = np.random.default_rng(seed)
    s = rng.choice(list("ACGT"), size=n).tolist()

    def plant(at, text):
        at = max(0, min(at, n - 1))
        text = text[: n - at]
        s[at:at + len(text)] = list(text)

    def unit(k):
        return "".join(rng.choice(list("ACGT"), size=k))

    n_feat = max(3, n // 400)
    for _ in range(n_feat):
        kind = rng.integers(0, 8)
        at = int(rng.integers(0, n))
        if kind == 0:
            plant(at, unit(1) * int(rng.integers(4, 40)))
        elif kind == 1:
            plant(at, unit(int(rng.integers(2, 7))) * int(rng.integers(2, 30)))
        elif kind == 2:   # VNTR: unit 7..35, 200..3000 bases
            u = unit(int(rng.integers(7, 36)))
            plant(at, u * (int(rng.integers(200, 3000)) // len(u) + 1))
        elif kind == 3:   # imperfect repeat: an STR with a few substitutions
            u = unit(int(rng.integers(1, 7)))
            t = list(u * int(rng.integers(10, 60)))
            for _ in range(int(rng.integers(1, 4))):
                t[int(rng.integers(0, len(t)))] = rng.choice(list("ACGT"))
            plant(at, "".join(t))
        elif kind == 4:
            plant(at, "N" * int(rng.integers(1, 50)))
        elif kind == 5:   # soft-masked stretch: the scan compares characters, 'a' != 'A'
            ln = int(rng.integers(5, 200))
            plant(at, "".join(s[at:at + ln]).lower())
        elif kind == 6:   # a unit longer than indel_vntr_repeatsize_max: must NOT be found as a tandem repeat
            u = unit(int(rng.integers(36, 60)))
            plant(at, u * int(rng.integers(2, 8)))
        else:            # nested: a homopolymer inside a dinucleotide repeat
            u = unit(2)
            plant(at, u * 12 + u[0] * int(rng.integers(5, 25)) + u * 12)
    if kinds == "all":
        if n > 6000 and seed % 3 == 0:   # one run of N longer than a kernel window (3072) and a chunk
            plant(int(rng.integers(0, n - 5000)), "N" * int(rng.integers(3100, 5000)))
        if seed % 2 == 0:
            plant(0, unit(int(rng.integers(1, 7))) * 20)                       # repeat at the region start
        if seed % 4 < 2:
            u = unit(int(rng.integers(1, 7)))
            plant(n - 15 * len(u), u * 15)                                     # repeat clipped by the region end
    return "".join(s)


EDGE_REFERENCES = [
    "A", "AC", "AAAAA", "ACGTACGTACGT", "N" * 40, "ACGTT", "a" * 30 + "A" * 30,
    "AC" * 600,                       # one repeat over the whole region, longer than 1024
    "A" * 2047, "A" * 2048, "A" * 2049, "ACG" * 1024, "ACGT" * 768 + "A",   # lengths around the kernels' chunk (2048) and window (3072)
    "ACGTTGCA" * 200 + "T" * 1100 + "GATTACA" * 300,
    "G" * 1024 + "C", "G" * 1023 + "C" + "G" * 1024,
]


def _more_str(rulen1, rc1, rulen2, rc2, umax):
    """is_indel_context_more_STR, main.hpp:699-721."""
    if rulen2 * rc2 == 0:
        return True
    if rulen1 > umax or rulen2 > umax:
        return rulen1 < rulen2 or (rulen1 == rulen2 and rc1 > rc2)
    rank1 = (-rc1 * rulen1) if rc1 <= 1 else (rc1 - 1) * rulen1
    rank2 = (-rc2 * rulen1) if rc2 <= 1 else (rc2 - 1) * rulen2     # sic: rulen1
    if rc1 == 0 or rulen1 == 0:
        rank1 = -100
    if rc2 == 0 or rulen2 == 0:
        rank2 = -100
    return rank1 > rank2


def _indel_phred(ampfact, rs, rn):
    """indel_phred, main.hpp:794-801 + prob2phred, main_conversion.hpp:890-893."""
    region_size = rs * rn
    num_slips = ((region_size - 8.0) if region_size > 64 else math.log1p(math.exp(region_size - 8.0))) * ampfact / float(rs * rs)
    return math.floor(-10 * math.log((1.0 - 2.220446049250313e-16) / (num_slips + 1.0)) / math.log(10))


def python_tracks(ref, smax=6, vmax=35, bq_max=42, slip_rate=8.0, del_to_ins=5.0, polymerase_size=8.0, str_phred_per_region=10, nonstr_phred_per_base=5):
    """refstring2repeatvec (main.hpp:803-874) and region_repeatvec_to_baq_offsetarr<false / true> (main.cpp:400-429) as the reference
    runs them: the sequential walk over start positions, the run loop per unit length, strict-greater overwrites.
    Returns (rtr int32 [7][n + 1] in UVC_RTR_* order, baq int64 [2][n + 1]).  Runs are found with a precomputed run-length table
    (same values as the reference's while loop)."""
    n = len(ref)
    b = np.frombuffer(ref.encode(), dtype=np.uint8)
    # run[u][q] = number of consecutive q' >= q with q' + u < n and ref[q'] == ref[q' + u] (what the reference's while loop counts)
    run = np.zeros((vmax + 1, n + 1), dtype=np.int64)
    for u in range(1, vmax + 1):
        if u < n:
            ne = np.ones(n, dtype=bool)
            ne[: n - u] = b[: n - u] != b[u:]
            fails = np.flatnonzero(ne)                      # never empty: q >= n - u always fails
            q = np.arange(n)
            run[u][:n] = fails[np.searchsorted(fails, q)] - q
    rtr = np.zeros((7, n + 1), dtype=np.int64)
    rtr[3, :] = bq_max
    refpos = 0
    while refpos < n:
        rs_max, max_rn, rep_end = 0, 0, refpos
        a_rs, a_rn, a_end = 0, 0, refpos
        for rs in range(1, vmax + 1):
            qidx = refpos + int(run[rs][refpos])
            rn = (qidx - refpos) // rs + 1
            if rs <= smax and _more_str(rs, rn, rs_max, max_rn, smax):
                rs_max, max_rn, rep_end = rs, rn, qidx + rs
            if _more_str(rs, rn, a_rs, a_rn, vmax):
                a_rs, a_rn, a_end = rs, rn, qidx + rs
        stop = min(rep_end, n)
        tl = stop - refpos
        dec = _indel_phred(slip_rate * del_to_ins, rs_max, tl // rs_max)
        sel = np.arange(refpos, stop)
        sel = sel[tl > rtr[1, sel]]
        rtr[0, sel] = refpos; rtr[1, sel] = tl; rtr[2, sel] = rs_max; rtr[3, sel] = bq_max - min(bq_max - 1, dec)
        astop = min(a_end, n)
        atl = astop - refpos
        sel = np.arange(refpos, astop)
        sel = sel[atl > rtr[5, sel]]
        rtr[4, sel] = refpos; rtr[5, sel] = atl; rtr[6, sel] = a_rs
        nbases_to_next = smax + rs_max
        refpos += max(rs_max * max_rn, nbases_to_next + 1) - nbases_to_next
    rtr[:, n] = rtr[:, n - 1]
    baq = np.zeros((2, n + 1), dtype=np.int64)
    psize = int(round(polymerase_size))
    for any_tr in (0, 1):
        tl2 = rtr[5] if any_tr else rtr[1]
        reps = tl2 // rtr[2]
        is_str = (reps >= 3) | ((reps >= 2) & (tl2 >= psize))
        inc = np.where(is_str, (str_phred_per_region * 10) // np.maximum(tl2, 1) + 1, nonstr_phred_per_base * 10)
        baq[any_tr] = np.cumsum(inc) // 10
    return rtr.astype(np.int32), baq
