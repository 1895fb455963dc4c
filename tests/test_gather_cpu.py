"""The gather in front of the scoring: BcfFormat_symboltype_init, BcfFormat_symbol_init and fill_symbol_VQ_fmts
(main.hpp:3744-4251) plus the minABQ / RTR arguments of their caller (main.cpp:524-525, 904-940).  The inputs the oracle hands to
calc_DPv / calc_qual (test hook uvc_oracle_score_trace) against an independent Python restatement (tests/gather_restatement.py) that reads
only the fetched plane groups -- with tests/test_score_cpu.py behind it, every step from the planes to the scored record has a second
restatement."""
import numpy as np
import pytest

from uvc_amd import region, synth
from gather_restatement import Planes, gather
from test_gpu_fuzz import weird_region
from test_score_cpu import traced_score
from util import run_region

# what the restatement does not derive from the planes: the InDel allele table (bDPa / cDP0a / gapSa, tests/test_gpu_indel_alleles.py)
# and the tumor key of a T/N pair
NOT_GATHERED = {"bDPa", "cDP0a", "gapSa_len", "refpos", "tki_tier2", "tpfa_dpv", "tpfa_qual"}


def check(lib, reads, R, P, is_amplicon=False, **kw):
    rec, ins, _ = traced_score(lib, R, is_amplicon=is_amplicon, **kw)
    assert len(ins) > 0
    pl = Planes(R.fetch)
    codes = np.array([{"A": 0, "C": 1, "G": 2, "T": 3}.get(c.upper(), 4) for c in reads["refseq"]], dtype=np.int32)
    bad, seen = [], set()
    for i, d in enumerate(ins):
        x, sym = int(d["refpos"]) - R.beg, int(d["symbol"])
        want = gather(pl, x, sym, codes, P, is_amplicon, R.npos)
        seen |= set(want)
        diff = {k: (d[k], v) for k, v in want.items() if float(v) != d[k]}
        if diff:
            bad.append((i, int(d["refpos"]), sym, dict(list(diff.items())[:6])))
    assert not bad, (len(bad), len(ins), bad[:3])
    assert set(ins[0]) - seen == NOT_GATHERED, (set(ins[0]) - seen) ^ NOT_GATHERED
    return len(ins)


@pytest.mark.parametrize("case", ["fuzz_illumina", "fuzz_umi", "fuzz_iontorrent", "synth_umi_default_gate", "synth_homopolymers"])
def test_gather_against_the_independent_restatement(case, oracle_lib):
    lib = oracle_lib
    if case.startswith("fuzz"):
        n = 0
        for seed in ((0, 8) if case == "fuzz_illumina" else (2, 5) if case == "fuzz_umi" else (3,)):
            reads = weird_region(seed, umi=(case == "fuzz_umi"))
            P = region.default_params(lib, platform=2 if case == "fuzz_iontorrent" else 1)
            R = run_region(lib, reads, params=P)
            n += check(lib, reads, R, P, all_out=True)
            R.close()
        assert n > 3000
    elif case == "synth_umi_default_gate":
        reads = synth.generate_region(seed=11, region_len=2000, depth=400, umi=True)
        P = region.default_params(lib)
        R = run_region(lib, reads, params=P)
        assert check(lib, reads, R, P, all_out=False) > 100
        R.close()
    else:
        # the homopolymer arms of minABQ (main.cpp:904-928): a reference made of 1..5-base runs
        reads = synth.generate_region(seed=77, region_len=1200, depth=50)
        rng = np.random.default_rng(5)
        ref, runs = [], 0
        while len(ref) < len(reads["refseq"]):
            ref += ["ACGT"[rng.integers(4)]] * int(rng.integers(1, 6)); runs += 1
        reads["refseq"] = "".join(ref[:len(reads["refseq"])])
        P = region.default_params(lib)
        P.syserr_minABQ_pcr_snv, P.syserr_minABQ_pcr_indel = 150, 15          # the amplicon arm of main.cpp:524-525, told apart from the capture one
        R = run_region(lib, reads, params=P)
        assert check(lib, reads, R, P, all_out=True) > 5000
        assert check(lib, reads, R, P, is_amplicon=True, all_out=True) > 5000
        R.close()
