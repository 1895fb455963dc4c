#!/usr/bin/env python3
"""GPU-box helper (not a test of the suite): the fuzz comparison of tests/test_gpu_fuzz.py::test_weird_reads over many more seeds and shapes
for a given number of seconds -- planes bit for bit, InDel allele rows, all-out records in the tolerance classes, handles reused across
regions of different length.  Prints one line per failure and a summary; exit code 1 if anything differed.
    python3 scripts/gpu_soak.py SECONDS [FIRST_SEED] [fuzz|synth]
Half of the regions run under one of the parameter variants of tests/test_gpu_parity.py (primer gating, short reads, SSCS table, germline lines ...).
tn: synth + the normal-sample pass of a T/N pair with tumor keys made from the tumor-only records.
long: fuzzed reads of up to 3 000 bases.  big: synth at 10 .. 90 kb and 100 .. 2000x.  deep: 1 .. 3 kb at 5 000 .. 70 000x.
synth: regions of the synthetic generator (2 .. 8 kb at 20 .. 1500x, UMI / duplex, error / InDel / clip rates up to 30 times the defaults)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from uvc_amd import _ffi, region, synth  # noqa: E402
from util import diff_groups  # noqa: E402
from test_gpu_fuzz import weird_region  # noqa: E402
from test_gpu_parity import VARIANTS, compare_records, tumor_keys_from  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1000
mode = sys.argv[3] if len(sys.argv) > 3 else "fuzz"
import torch  # noqa: E402  (torch.cuda before libuvcgpu.so touches the device: uvc_amd/region.py device_reads)
torch.cuda.init(); dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
glib = region.gpu_lib(); assert glib.dll.uvcgpu_init(0) == 0
olib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")
handles = {}
t0, n_ok, n_refused, fails = time.time(), 0, 0, []
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    umi, platform, normal = bool(rng.integers(0, 2)), int(rng.choice([1, 1, 2])), 0
    n_frag, ref_len = int(rng.choice([40, 150, 260, 600, 1500])), int(rng.choice([200, 450, 700, 1300, 4100]))
    variant = dict(VARIANTS[sorted(VARIANTS)[int(rng.integers(0, len(VARIANTS)))]]) if rng.random() < 0.5 else {}
    if "platform" in variant: platform = variant["platform"]
    # a few thresholds moved inside their plausible ranges (both libraries get the same values): arms that the defaults never take
    TWEAK = dict(fam_thres_dup1add=(1, 4), fam_thres_dup1perc=(50, 101), fam_thres_dup2add=(2, 5), fam_thres_highBQ_snv=(0, 41), fam_thres_highBQ_indel=(0, 41),
                 bias_thres_highBQ=(0, 41), bias_thres_highBAQ=(0, 40), bias_thres_interfering_indel=(3, 40), bias_thres_interfering_indel_BQ=(0, 41), bias_thres_BAQ1=(10, 60), bias_thres_BAQ2=(20, 80),
                 syserr_mut_region_n_bases=(1, 60), min_altdp_thres=(0, 5), fam_flag=(0, 4), primerlen=(0, 31), primerlen2=(0, 41), indel_adj_tracklen_dist=(0, 12),
                 indel_adj_indellen_perc=(100, 301), bq_phred_added_misma=(0, 12), bq_phred_added_indel=(0, 12), microadjust_padded_deletion_flag=(0, 4), central_readlen=(50, 300),
                 bias_thres_PFBQ1=(10, 50), bias_thres_PFBQ2=(10, 60), fam_thres_emperr_all_flat_snv=(1, 6), fam_thres_emperr_con_perc_snv=(50, 101), bias_thres_strict_c2LRP0=(0, 20),
                 bias_thres_aLPxT_add=(0, 12), microadjust_nobias_pos_indel_maxlen=(0, 30), fam_thres_qseqlen=(0, 120), bias_thres_aLRP1t_minus=(0, 20), syserr_minABQ_cap_snv=(0, 300))
    tweaks = {}
    if mode == "params" or rng.random() < 0.3:
        for k in rng.choice(sorted(TWEAK), size=int(rng.integers(1, 6)), replace=False):
            tweaks[str(k)] = int(rng.integers(*TWEAK[str(k)]))
    if mode in ("synth", "tn", "big", "deep"):
        depth = int(rng.choice([20, 60, 150, 300, 600, 1500])); ref_len = int(rng.choice([2000, 3000, 5000, 8000])) if depth <= 300 else int(rng.choice([1000, 2000]))
        if mode == "big":   # tens of kb: the non-split kernel forms, hundreds of windows, carries across the interval-sum blocks
            depth = int(rng.choice([100, 300, 1000, 2000])); ref_len = int(rng.choice([20000, 50000, 90000])) if depth <= 300 else int(rng.choice([10000, 20000]))
        if mode == "deep":  # amplicon-like piles: up to 70 000 fragments on one position (the 16-bit bucket counters of k_frag16 end at 65 535)
            depth = int(rng.choice([5000, 20000, 70000])); ref_len = int(rng.choice([1000, 1500])) if depth > 5000 else int(rng.choice([1000, 3000]))
        n_frag = depth
        reads = synth.generate_region(seed=seed, region_len=ref_len, depth=depth, umi=umi, fam_mean=float(rng.choice([1.5, 4.0, 8.0])), duplex_frac=float(rng.choice([0.0, 0.6, 0.9])),
                                      snv_every=int(rng.choice([150, 1000])), somatic_every=int(rng.choice([400, 10000])), indel_every=int(rng.choice([200, 800, 5000])),
                                      err_rate=float(rng.choice([1e-3, 1e-2, 3e-2])), clip_frac=float(rng.choice([0.01, 0.1, 0.3])), dedup_by_position=bool(rng.integers(0, 2)))
    elif mode == "long":   # reads of up to 3 000 reference bases with the fuzz generator's CIGAR shapes (M runs stay <= 40: dozens of InDels per read)
        ref_len = int(rng.choice([1300, 4100, 9000])); n_frag = int(rng.choice([40, 150, 400]))
        reads = weird_region(seed, n_frag=n_frag, ref_len=ref_len, umi=umi, lengths=(1, 5, 60, 150, 300, 700, 1500, 3000))
    else:
        reads = weird_region(seed, n_frag=n_frag, ref_len=ref_len, umi=umi)
    out = []
    for name, lib in (("oracle", olib), ("gpu", glib)):
        P = region.default_params(lib, platform=platform)
        P.fam_flag = int(rng.integers(0, 2)) if name == "oracle" else out_fam_flag
        out_fam_flag = P.fam_flag
        for k, v in variant.get("set", {}).items():
            setattr(P, k, v)
        for k, v in tweaks.items():
            setattr(P, k, v)
        try:
            key = (name, platform, P.fam_flag, tuple(sorted(variant.get("set", {}).items())), tuple(sorted(tweaks.items())))
            if name == "gpu" and tweaks and len(handles) > 64: handles.pop(next(iter(handles))).close()
            R = handles.get(key)
            if R is None or name == "oracle":
                R = region.Region(lib, P, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
                if name == "gpu": handles[key] = R
            else:
                R.reset(reads["tid"], reads["beg"], reads["end"], reads["refseq"].encode())      # a handle that has held other regions
            if name == "gpu" and seed % 3 == 0:   # the form bench.py hands the columns over in: already in HBM, 4-bit bases, no offset columns
                R.set_reads_device(region.device_reads(region.compact_form(reads), dev))
            else:
                R.set_reads(reads)
            if seed % 2: R.correct_bq()
            R.accumulate(); R.fetch("PREP32")
            out.append(R)
        except region.UvcError as e:
            out.append(e.code)
    o, g = out
    try:
        if isinstance(o, int) or isinstance(g, int):
            assert o == g, ("refusal", o, g)
            n_refused += 1
        else:
            bad = diff_groups(o, g)
            assert not bad, {k: (v[0], v[1][:6]) for k, v in bad.items()}
            assert o.indel_alleles() == g.indel_alleles(), "allele rows"
            compare_records(o.score(all_out=True), g.score(all_out=True))
            ro = o.score(all_out=False)
            compare_records(ro, g.score(all_out=False))
            if mode == "tn" and len(ro["refpos"]) >= 4 and P.inferred_is_vcf_generated:
                keys = tumor_keys_from(ro, every=int(rng.integers(1, 4)))
                recs = []
                for lib in (olib, glib):
                    Pn = region.default_params(lib, platform=platform); Pn.fam_flag = out_fam_flag; Pn.tumor_vcf_is_provided = 1
                    for k, v in variant.get("set", {}).items():
                        setattr(Pn, k, v)
                    for k, v in tweaks.items():
                        setattr(Pn, k, v)
                    Rn = region.Region(lib, Pn, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
                    Rn.set_reads(reads)
                    if seed % 2: Rn.correct_bq()
                    Rn.accumulate()
                    recs.append(Rn.score(tumor_keys=keys)); Rn.close()
                compare_records(recs[0], recs[1])
            # the way bench.py and the command line end a region: only the record groups the writer reads, and the planes released (they are zeroed on
            # the side stream under the D2H; the next region on this handle starts from them)
            rfull = g.score(all_out=False)
            rk = g.score(all_out=False, release_state=True, kept_only=True)
            at = {}
            for i in range(len(rfull["refpos"])):
                at.setdefault((int(rfull["refpos"][i]), int(rfull["symbol"][i]), int(rfull["gapSa"][i])), i)
            for j in range(len(rk["refpos"])):
                i = at[(int(rk["refpos"][j]), int(rk["symbol"][j]), int(rk["gapSa"][j]))]
                for k in ("DP", "AD", "bDP", "cVQ1", "cVQ2", "gVQ1", "QUAL", "FILTER", "keep", "out", "TLODQ", "NLODQ", "cDP1v"):
                    assert rk[k][j] == rfull[k][i], ("kept_only record differs from the full one", k)
            assert int((rfull["keep"] & rfull["out"]).sum()) == int((rk["keep"] & rk["out"]).sum()), "kept_only lost a written record"
            n_ok += 1
    except AssertionError as e:
        fails.append(seed); print("FAIL seed", seed, dict(umi=umi, platform=platform, n_frag=n_frag, ref_len=ref_len, variant=variant, tweaks=tweaks), str(e)[:900], flush=True)
    if not isinstance(o, int): o.close()
    seed += 1
    if (n_ok + n_refused + len(fails)) % (50 if mode == "fuzz" else 10) == 0:
        print("... %d regions, %.0f s" % (n_ok + n_refused + len(fails), time.time() - t0), flush=True)
print("soak: %d regions equal, %d refused by both, %d FAILED %s in %.0f s" % (n_ok, n_refused, len(fails), fails[:20], time.time() - t0))
sys.exit(1 if fails else 0)
