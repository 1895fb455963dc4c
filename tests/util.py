import numpy as np

from uvc_amd import _ffi, region, synth

INT_GROUPS = ["PREP32", "PREP64", "THRES", "SEG32", "SEG64", "BQSUM", "FRAG", "FAM", "FAMINFO32", "FAMINFO64", "DUPLEX", "RTR", "BAQ", "VQ"]


def run_region(lib, reads, params=None, platform=1):
    p = params if params is not None else region.default_params(lib, platform=platform)
    R = region.Region(lib, p, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    R.set_reads(reads)
    R.accumulate()
    return R


def diff_groups(Ra, Rb, groups=INT_GROUPS):
    """Returns {group: (n_mismatching_cells, first few mismatches)} for groups that differ."""
    bad = {}
    for g in groups:
        a, b = Ra.fetch(g), Rb.fetch(g)
        if not np.array_equal(a, b):
            idx = np.argwhere(a != b)
            bad[g] = (len(idx), [(tuple(int(v) for v in i), int(a[tuple(i)]), int(b[tuple(i)])) for i in idx[:8]])
    return bad
