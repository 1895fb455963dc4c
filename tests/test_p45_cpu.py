"""P4 / P5: the family passes of updateByAlns3UsingFQ (main.hpp:2836-3590).  The oracle's FAM / FAMINFO32 / FAMINFO64 /
DUPLEX planes and the VQ slots cIAQf / cIADf / cIDQf / cIAQr / cIADr / cIDQr against an independent Python restatement
(tests/p45_restatement.py) chained behind the restatements of the tracks, P1, P1b and the per-read walk of P2 -- no oracle value enters."""
import importlib.util
import os

import numpy as np
import pytest

from uvc_amd import region
from rtr_cases import python_tracks
from prep_restatement import prep_sets, thres_sets
from p45_restatement import FAM, FI32, FI64, family_passes
from util import run_region

_spec = importlib.util.spec_from_file_location("fz_p45", os.path.join(os.path.dirname(__file__), "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(fz)


@pytest.mark.parametrize("seed,umi,platform,normal", [(41, True, 1, 0), (42, False, 1, 0), (43, True, 2, 0), (44, True, 1, 1), (45, True, 1, 0)])
def test_family_passes_against_the_independent_restatements(seed, umi, platform, normal, oracle_lib):
    reads = fz.weird_region(seed, n_frag=100 + 25 * (seed % 3), ref_len=380 + 30 * (seed % 5), umi=umi)
    P = region.default_params(oracle_lib, platform=platform)
    P.tumor_vcf_is_provided = normal
    R = run_region(oracle_lib, reads, params=P)
    rtr, baq = python_tracks(reads["refseq"], smax=P.indel_str_repeatsize_max, vmax=P.indel_vntr_repeatsize_max, bq_max=P.indel_BQ_max,
                             slip_rate=P.indel_polymerase_slip_rate, del_to_ins=P.indel_del_to_ins_err_ratio, polymerase_size=P.indel_polymerase_size,
                             str_phred_per_region=P.indel_str_phred_per_region, nonstr_phred_per_base=P.indel_nonSTR_phred_per_base)
    codes = np.array([{"A": 0, "C": 1, "G": 2, "T": 3}.get(c.upper(), 4) for c in reads["refseq"]], dtype=np.int32)
    prep = prep_sets(reads, P, rtr, baq[0], np.append(codes, 4))
    thres, ip = thres_sets(prep, rtr[3], P, is_normal=bool(normal), iontorrent=(platform == 2))
    famp, fi, dup, vq = family_passes(reads, P, rtr, ip, baq[0], baq[1], codes, prep, thres, proton=(platform == 2))
    of, o32, o64, od, ov = R.fetch("FAM"), R.fetch("FAMINFO32"), R.fetch("FAMINFO64"), R.fetch("DUPLEX"), R.fetch("VQ")
    bad = {}
    for st in range(2):
        for k, name in enumerate(FAM):
            if not np.array_equal(of[st][k].astype(np.int64), famp[st][k]): bad["%s[%d]" % (name, st)] = np.argwhere(of[st][k] != famp[st][k])[:4].tolist()
    for k, name in enumerate(FI32):
        if not np.array_equal(o32[k].astype(np.int64), fi[name]): bad[name] = np.argwhere(o32[k] != fi[name])[:4].tolist()
    for k, name in enumerate(FI64):
        if not np.array_equal(o64[k].astype(np.int64), fi[name]): bad[name] = np.argwhere(o64[k] != fi[name])[:4].tolist()
    for k in range(2):
        if not np.array_equal(od[k].astype(np.int64), dup[k]): bad["dDP%d" % (k + 1)] = np.argwhere(od[k] != dup[k])[:4].tolist()
    for k, name in enumerate("cIAQf cIADf cIDQf cIAQr cIADr cIDQr".split()):            # UVC_VQ_cIAQf = 8 ...
        if not np.array_equal(ov[8 + k].astype(np.int64), vq[name]): bad[name] = np.argwhere(ov[8 + k] != vq[name])[:4].tolist()
    assert not bad, bad
    assert of[:, 0].sum() > 0 and (not umi or (of[:, 2].sum() > 0 and od.sum() > 0 and o32[12].sum() > 0))
    R.close()
