"""Oracle comparison at a size where the kernels run in the form the bench runs them: a 150 kb x 300x non-UMI region
(2 344 windows of 64 positions: the non-split kernel forms, real occupancy, radix sorts over 300 k reads, k_win_index over thousands of
windows, 147 carry blocks of k_prep_sums / k_frag_sums, a mismatch queue of hundreds of thousands of entries) and a 50 kb x 2000x
duplex-UMI region (the digest / window forms of the family kernels), and a 1 Mb x 20x region (positions >> reads): every plane bit-exact, every record inside the tolerance classes
of tests/test_gpu_parity.py.  The oracle runs these on all host cores of the box in well under a minute (it is single-threaded per
region: the regions run on a thread each)."""
import threading

import numpy as np
import pytest

from uvc_amd import synth
from util import diff_groups, run_region
from test_gpu_parity import compare_records

pytestmark = pytest.mark.gpu

LARGE = {
    "nonumi_150kb_300x": dict(region_len=150_000, depth=300, seed=4242),
    "duplex_50kb_2000x": dict(region_len=50_000, depth=2000, seed=4343, umi=True),
    # far more positions than reads, at least 65 536 fragments: the fragment-depth scan of the read preparation runs over npos + 1 elements
    # with the scratch sized for it (sized for the reads only it failed from ~4 positions per read on)
    "wgs_1mb_20x": dict(region_len=1_000_000, depth=20, seed=4444),
    # 3 % base errors at 2000x: more than 65 535 fragments leave the fast fragment statistics for the sweep list whose length only the device
    # knows (k_fragstat_overflow strides over it; found by scripts/gpu_soak.py, where a grid of 65 535 blocks left the rest with n_cov = 0)
    "noisy_20kb_2000x": dict(region_len=20_000, depth=2000, seed=30105, err_rate=0.03, snv_every=150, somatic_every=400, indel_every=800, clip_frac=0.1, dedup_by_position=False),
}


@pytest.fixture(scope="module")
def oracle_runs(oracle_lib):
    """Both oracle regions at once, one thread each (ctypes releases the GIL inside the library)."""
    out, errs = {}, []

    def work(name):
        try:
            reads = synth.generate_region(**LARGE[name])
            R = run_region(oracle_lib, reads)
            out[name] = (reads, R, R.score(all_out=False))
        except Exception as e:   # noqa: BLE001
            errs.append((name, e))
    th = [threading.Thread(target=work, args=(n,)) for n in LARGE]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    return out


@pytest.mark.parametrize("name", list(LARGE))
def test_large_region_matches_oracle(name, oracle_runs, gpu_lib):
    reads, Ro, ro = oracle_runs[name]
    Rg = run_region(gpu_lib, reads)
    bad = diff_groups(Ro, Rg)
    assert not bad, "\n".join("%s: %d cells differ, e.g. %s" % (g, v[0], v[1]) for g, v in bad.items())
    rg = Rg.score(all_out=False)
    assert len(ro["refpos"]) > 1000
    if name == "wgs_1mb_20x": assert reads["n_reads"] >= 2 * 65536 and Rg.npos > 4 * reads["n_reads"]
    worst = compare_records(ro, rg)
    print(name, reads["n_reads"], "reads,", len(ro["refpos"]), "records, worst differences", {k: v for k, v in worst.items() if v})
    # the InDel allele rows behind the records
    ao, ag = Ro.indel_alleles(), Rg.indel_alleles()
    assert ao == ag
    Rg.close()


@pytest.mark.skipif(not __import__("os").environ.get("UVC_FULL_ORACLE"), reason="minutes of oracle time and ~16 GB of host memory: UVC_FULL_ORACLE=1 (the soak runs set it)")
def test_the_bench_tile_itself_matches_oracle(oracle_lib, gpu_lib):
    """The 1 Mb x 300x tile bench.py times (its first tile: seed 12345, the generator's defaults) through the oracle: every plane bit-exact, the
    default-gate records inside the tolerance classes.  The largest comparison of the default suite is 150 kb; this one is behind a switch
    because the single-threaded oracle takes a few minutes on it."""
    reads = synth.generate_region(seed=12345, region_len=1_000_000, depth=300, beg=1000000)
    Ro = run_region(oracle_lib, reads)
    ro = Ro.score(all_out=False)
    Rg = run_region(gpu_lib, reads)
    bad = diff_groups(Ro, Rg)
    assert not bad, "\n".join("%s: %d cells differ, e.g. %s" % (g, v[0], v[1]) for g, v in bad.items())
    rg = Rg.score(all_out=False)
    assert len(ro["refpos"]) > 40000
    worst = compare_records(ro, rg)
    print("bench tile:", reads["n_reads"], "reads,", len(ro["refpos"]), "records, worst differences", {k: v for k, v in worst.items() if v})
    Ro.close(); Rg.close()
