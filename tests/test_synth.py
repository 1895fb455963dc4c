import numpy as np

from uvc_amd import synth


def test_deterministic_and_well_formed():
    a = synth.generate_region(seed=5, region_len=3000, depth=50)
    b = synth.generate_region(seed=5, region_len=3000, depth=50)
    for k in ("pos", "bases", "quals", "cigars", "fam_id", "flag"):
        assert np.array_equal(a[k], b[k]), k
    assert a["end"] - a["beg"] == len(a["refseq"])
    end = a["pos"].astype(np.int64).copy()
    for i in range(a["n_reads"]):
        for c in a["cigars"][a["cigar_off"][i]:a["cigar_off"][i] + a["n_cigar"][i]]:
            if (c & 0xF) in (0, 2, 3, 7, 8):
                end[i] += c >> 4
    assert a["pos"].min() >= a["beg"] + 100 and end.max() <= a["end"] - 100 + 3
    # grouping is contiguous: (fam, strand, frag) runs never repeat
    key = list(zip(a["fam_id"].tolist(), a["fam_strand"].tolist(), a["frag_id"].tolist()))
    seen, prev = set(), None
    for k in key:
        if k != prev:
            assert k not in seen
            seen.add(k)
        prev = k
    assert a["n_cigar"].max() >= 2          # InDel / clipped reads are present
    depth = a["n_reads"] * 150 / 3000.0
    assert 40 < depth < 60


def test_umi_mode_builds_duplex_families():
    a = synth.generate_region(seed=9, region_len=1500, depth=400, umi=True)
    assert (a["fam_dflag"] == 3).all()
    fam_strands = {}
    for f, s in zip(a["fam_id"].tolist(), a["fam_strand"].tolist()):
        fam_strands.setdefault(f, set()).add(s)
    both = sum(1 for v in fam_strands.values() if len(v) == 2)
    assert 0.3 < both / len(fam_strands) < 0.9
    frags_per_fam = a["n_reads"] / 2 / len(fam_strands)
    assert frags_per_fam > 3
