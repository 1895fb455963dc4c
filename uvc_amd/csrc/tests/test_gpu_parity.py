"""GPU parity: every per-position plane of the HIP path equals the oracle bit for bit
(integer class of SURVEY section 8d; the bucket->quality VQ slots bIAQb/cIAQ* are the outputs of a
truncated fp64 log and are also required to match exactly here -- a flip would need a product
within 1e-13 of an integer)."""
import numpy as np
import pytest

from uvc_amd import synth
from util import diff_groups, run_region

pytestmark = pytest.mark.gpu

CASES = {
    "config1_10kb_30x": dict(region_len=10000, depth=30, seed=12345),
    "config2shape_5kb_300x": dict(region_len=5000, depth=300, seed=7),
    "nodedup_3kb_60x": dict(region_len=3000, depth=60, seed=3, dedup_by_position=False),
    "umi_duplex_2kb_400x": dict(region_len=2000, depth=400, seed=11, umi=True),
    "tiny_600bp_5x": dict(region_len=600, depth=5, seed=5),
}


@pytest.mark.parametrize("name", list(CASES))
def test_planes_match_oracle(name, oracle_lib, gpu_lib):
    reads = synth.generate_region(**CASES[name])
    Ro = run_region(oracle_lib, reads)
    Rg = run_region(gpu_lib, reads)
    bad = diff_groups(Ro, Rg)
    assert not bad, "\n".join("%s: %d cells differ, e.g. %s" % (g, v[0], v[1]) for g, v in bad.items())
