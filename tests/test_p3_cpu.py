"""P3: the fragment pass of updateByAlns3UsingBQ (main.hpp:2620-2830).  The oracle's FRAG planes (bDP, bTA, bTB per strand)
and the VQ slots bMQ / bIAQb / bIADb / bIDQb against an independent Python restatement (tests/p3_restatement.py) chained behind the
restatements of the tracks, P1, P1b and P2 -- no oracle value enters the chain."""
import importlib.util
import os

import numpy as np
import pytest

from uvc_amd import region
from rtr_cases import python_tracks
from prep_restatement import prep_sets, thres_sets
from p2_restatement import update_by_aln
from p3_restatement import fragment_pass
from util import run_region

_spec = importlib.util.spec_from_file_location("fz_p3", os.path.join(os.path.dirname(__file__), "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(fz)


@pytest.mark.parametrize("seed,umi,platform,fam_flag", [(31, False, 1, 0), (32, True, 1, 0), (33, False, 2, 0), (34, True, 1, 1), (35, False, 1, 0)])
def test_fragment_pass_against_the_independent_restatements(seed, umi, platform, fam_flag, oracle_lib):
    reads = fz.weird_region(seed, n_frag=110 + 30 * (seed % 3), ref_len=400 + 30 * (seed % 7), umi=umi)
    P = region.default_params(oracle_lib, platform=platform)
    P.fam_flag = fam_flag
    R = run_region(oracle_lib, reads, params=P)
    rtr, baq = python_tracks(reads["refseq"], smax=P.indel_str_repeatsize_max, vmax=P.indel_vntr_repeatsize_max, bq_max=P.indel_BQ_max,
                             slip_rate=P.indel_polymerase_slip_rate, del_to_ins=P.indel_del_to_ins_err_ratio, polymerase_size=P.indel_polymerase_size,
                             str_phred_per_region=P.indel_str_phred_per_region, nonstr_phred_per_base=P.indel_nonSTR_phred_per_base)
    codes = np.array([{"A": 0, "C": 1, "G": 2, "T": 3}.get(c.upper(), 4) for c in reads["refseq"]], dtype=np.int32)
    prep = prep_sets(reads, P, rtr, baq[0], np.append(codes, 4))
    thres, ip = thres_sets(prep, rtr[3], P, is_normal=False, iontorrent=(platform == 2))
    proton = (platform == 2)
    seg, bqsum = update_by_aln(reads, P, rtr, ip, baq[0], baq[1], codes, prep, thres, proton)
    frag, vq = fragment_pass(reads, P, rtr, ip, baq[0], codes, prep, thres, seg, bqsum, proton)
    of, ov = R.fetch("FRAG"), R.fetch("VQ")
    bad = {}
    for st in range(2):
        for f, name in enumerate(("bDP", "bTA", "bTB")):
            if not np.array_equal(of[st][f].astype(np.int64), frag[st][f]): bad["%s[%d]" % (name, st)] = np.argwhere(of[st][f] != frag[st][f])[:4].tolist()
    for k, name in ((4, "bMQ"), (5, "bIAQb"), (6, "bIADb"), (7, "bIDQb")):       # UVC_VQ_* order
        if not np.array_equal(ov[k].astype(np.int64), vq[name]): bad[name] = np.argwhere(ov[k] != vq[name])[:4].tolist()
    assert not bad, bad
    assert of[:, 0, 7:13].sum() > 0 and of[:, 2].sum() > 0
    R.close()
