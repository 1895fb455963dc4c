"""Row a3 on the device (uvc_rtr.hip): RegionalTandemRepeat tracks and the two BAQ prefix-sum arrays of a region, bit-exact against the
oracle on fuzzed references -- VNTRs of 200..3000 bases, runs of N longer than a kernel window, soft-masked stretches, repeats clipped by
either region end, lengths around the kernels' chunk sizes -- after create and after reset of a used handle."""
import numpy as np
import pytest

from uvc_amd import region
from rtr_cases import EDGE_REFERENCES, fuzz_reference, python_tracks

pytestmark = pytest.mark.gpu
NAMES = ("begpos", "tracklen", "unitlen", "indelphred", "anyTR_begpos", "anyTR_tracklen", "anyTR_unitlen")


def tracks(lib, ref, params=None, handle=None):
    p = params if params is not None else region.default_params(lib)
    if handle is None:
        R = region.Region(lib, p, 0, 100000, 100000 + len(ref), ref)
    else:
        R = handle
        R.reset(0, 100000, 100000 + len(ref), ref)
    out = R.fetch("RTR"), R.fetch("BAQ")
    if handle is None:
        R.close()
    return out


def same(a, b, ref):
    for f, name in enumerate(NAMES):
        bad = np.flatnonzero(a[0][f] != b[0][f])
        assert bad.size == 0, (name, len(ref), bad.size, int(bad[0]), int(a[0][f][bad[0]]), int(b[0][f][bad[0]]), ref[max(0, bad[0] - 30): bad[0] + 30])
    bad = np.argwhere(a[1] != b[1])
    assert bad.size == 0, ("BAQ", len(ref), [tuple(int(v) for v in x) for x in bad[:4]])


@pytest.mark.parametrize("i", range(len(EDGE_REFERENCES)))
def test_edge_references(i, oracle_lib, gpu_lib):
    ref = EDGE_REFERENCES[i]
    same(tracks(oracle_lib, ref), tracks(gpu_lib, ref), ref)


@pytest.mark.parametrize("seed", range(16))
def test_fuzzed_references(seed, oracle_lib, gpu_lib):
    n = [37, 300, 1023, 2048, 3073, 6143, 9000, 20000][seed % 8] + seed // 8
    ref = fuzz_reference(seed, n)
    same(tracks(oracle_lib, ref), tracks(gpu_lib, ref), ref)


def test_reset_of_one_handle_over_many_regions(oracle_lib, gpu_lib):
    """uvcgpu_region_reset re-runs the side-array kernels on the handle's scratch: longer, shorter, longer again."""
    R = region.Region(gpu_lib, region.default_params(gpu_lib), 0, 100000, 100000 + 50, "ACGTA" * 10)
    for seed, n in ((1, 5000), (2, 700), (3, 12000), (4, 2048), (5, 12001)):
        ref = fuzz_reference(100 + seed, n)
        same(tracks(oracle_lib, ref), tracks(gpu_lib, ref, handle=R), ref)
    R.close()


def test_other_parameters(oracle_lib, gpu_lib):
    ref = fuzz_reference(99, 6000)
    out = []
    for lib in (oracle_lib, gpu_lib):
        p = region.default_params(lib)
        p.indel_str_repeatsize_max = 4; p.indel_vntr_repeatsize_max = 20; p.indel_BQ_max = 30
        p.indel_polymerase_slip_rate = 3.0; p.indel_del_to_ins_err_ratio = 2.0; p.indel_polymerase_size = 5.0
        p.indel_str_phred_per_region = 17; p.indel_nonSTR_phred_per_base = 3
        out.append(tracks(lib, ref, p))
    same(out[0], out[1], ref)


def test_long_runs_do_not_cost_a_loop_per_position(gpu_lib):
    """1 Mb of N, then 30 kb of one 7-base unit and 150 kb of one base: the reference's walk jumps over them; the kernels must not loop over a run per position.
    Checked against the Python restatement (the oracle's run loop would need minutes)."""
    import time
    for ref in ("ACGT" * 50 + "N" * 1000000 + "TTGCA" * 40, "GATTACA" * 4286 + "ACGT" * 30 + "C" * 150000):
        t0 = time.perf_counter()
        got = tracks(gpu_lib, ref)
        dt = time.perf_counter() - t0
        want = python_tracks(ref)
        same(want, got, ref[:100])
        assert dt < 5.0, dt
