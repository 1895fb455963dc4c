"""P1 and P1b: update_seg_format_prep_sets_by_aln (main.hpp:924-1204) and update_seg_format_thres_from_prep_sets
(main.hpp:1206-1301).  The oracle's planes against an independent Python restatement written from the reference text
(tests/prep_restatement.py) on the fuzz generator's reads -- several InDels per read, an insertion next to a deletion, clips, reference
skips, amplicon-flagged families -- with the repeat tracks and BAQ sums of the independent restatement of tests/rtr_cases.py as inputs.
Every SegFormatPrepSet counter, every threshold and the edited indelphred of every position must agree."""
import importlib.util
import os

import numpy as np
import pytest

from uvc_amd import region
from rtr_cases import python_tracks
from prep_restatement import PREP32, PREP64, THRES, prep_sets, thres_sets
from util import run_region

_spec = importlib.util.spec_from_file_location("fz_prep", os.path.join(os.path.dirname(__file__), "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(fz)


@pytest.mark.parametrize("seed,umi,platform,normal", [(11, False, 1, 0), (12, True, 1, 0), (13, False, 2, 0), (14, False, 1, 1), (15, True, 2, 1), (16, False, 1, 0), (17, True, 1, 0), (18, False, 2, 0)])
def test_prep_and_thresholds_against_the_independent_restatement(seed, umi, platform, normal, oracle_lib):
    reads = fz.weird_region(seed, n_frag=150 + 40 * (seed % 4), ref_len=500 + 40 * seed, umi=umi)
    P = region.default_params(oracle_lib, platform=platform)
    P.tumor_vcf_is_provided = normal                 # the normal sample of a T/N pair takes the ...N... threshold percentages
    R = run_region(oracle_lib, reads, params=P)
    npos = reads["end"] - reads["beg"] + 1
    rtr, baq = python_tracks(reads["refseq"], smax=P.indel_str_repeatsize_max, vmax=P.indel_vntr_repeatsize_max, bq_max=P.indel_BQ_max,
                             slip_rate=P.indel_polymerase_slip_rate, del_to_ins=P.indel_del_to_ins_err_ratio, polymerase_size=P.indel_polymerase_size,
                             str_phred_per_region=P.indel_str_phred_per_region, nonstr_phred_per_base=P.indel_nonSTR_phred_per_base)
    assert rtr.shape[1] == npos
    codes = np.array([{"A": 0, "C": 1, "G": 2, "T": 3}.get(c.upper(), 4) for c in reads["refseq"]] + [4], dtype=np.int32)   # CHAR_TO_SYMBOL, main_conversion.hpp:473-486
    want = prep_sets(reads, P, rtr, baq[0], codes)
    got32, got64 = R.fetch("PREP32"), R.fetch("PREP64")
    bad = {}
    for k, name in enumerate(PREP32):
        w = want[name].astype(np.int64); w = ((w + 2 ** 31) % 2 ** 32) - 2 ** 31     # an int32 field
        if not np.array_equal(got32[k].astype(np.int64), w): bad[name] = np.nonzero(got32[k] != w)[0][:5].tolist()
    for k, name in enumerate(PREP64):
        if not np.array_equal(got64[k].astype(np.int64), want[name]): bad[name] = np.nonzero(got64[k] != want[name])[0][:5].tolist()
    assert not bad, bad
    assert got32[PREP32.index("a_near_ins_dp")].sum() > 0 and got32[PREP32.index("a_near_del_dp")].sum() > 0 and got32[PREP32.index("a_near_long_clip_dp")].sum() > 0
    t, ip = thres_sets(want, rtr[3], P, is_normal=bool(P.tumor_vcf_is_provided), iontorrent=(platform == 2))
    gt = R.fetch("THRES")
    bad = {name: np.nonzero(gt[k] != t[name])[0][:5].tolist() for k, name in enumerate(THRES) if not np.array_equal(gt[k].astype(np.int64), t[name])}
    assert not bad, bad
    assert np.array_equal(R.fetch("RTR")[3].astype(np.int64), ip), np.nonzero(R.fetch("RTR")[3] != ip)[0][:10]
    assert (ip != rtr[3]).any()                      # the edit happened somewhere
    R.close()
