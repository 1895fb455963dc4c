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
    "config4shape_1kb_2000x_duplex": dict(region_len=1000, depth=2000, seed=13, umi=True),   # BASELINE config 4 shape: deep duplex-UMI panel
    "deep_nonumi_800bp_3000x": dict(region_len=800, depth=3000, seed=14),                    # LDS queues / histograms far beyond one chunk
}


@pytest.mark.parametrize("name", list(CASES))
def test_planes_match_oracle(name, oracle_lib, gpu_lib):
    reads = synth.generate_region(**CASES[name])
    Ro = run_region(oracle_lib, reads)
    Rg = run_region(gpu_lib, reads)
    bad = diff_groups(Ro, Rg)
    assert not bad, "\n".join("%s: %d cells differ, e.g. %s" % (g, v[0], v[1]) for g, v in bad.items())


# SURVEY section 8(d) tolerance classes for the scoring records
EXACT_FIELDS = ["refpos", "symbol", "refsymbol", "DP", "AD", "bDP", "bAD", "c2DP", "c2AD", "bDPa", "cDP0a", "tier2", "FTS", "gapSa", "gapSa_len", "tkey",
                "cVQAM0", "cVQAM1", "cVQSM0", "cVQSM1", "vAC0", "vAC1", "germ_GT", "germ_emit", "germ_ref", "germ_alt1", "germ_alt2", "out", "NLODV", "FILTER", "keep"]
PCT_FIELDS = ["cDP1v", "cDP1w", "cDP1x", "cDP2v", "cDP2w", "cDP2x", "CDP1v0", "CDP1v1", "CDP1w0", "CDP1w1", "CDP1x0", "CDP1x1",
              "CDP2v0", "CDP2v1", "CDP2w0", "CDP2w1", "CDP2x0", "CDP2x1"]


def compare_records(ro, rg):
    """oracle vs GPU records: exact class bit-exact, depth-like x100 fields within 1 %, everything else within 1 Phred."""
    assert len(ro["refpos"]) == len(rg["refpos"]), (len(ro["refpos"]), len(rg["refpos"]))
    worst = {}
    for name in ro:
        if name == "QUAL":   # the bit pattern of a float (vcfqual): compare as numbers, device logf / powf are not correctly rounded
            fa, fb = ro[name].view(np.float32).astype(np.float64), rg[name].view(np.float32).astype(np.float64)
            assert (np.abs(fa - fb) <= 1e-3 * np.maximum(1.0, np.abs(fa))).all(), (name, int(np.argmax(np.abs(fa - fb))))
            worst[name] = 0
            continue
        a, b = ro[name].astype(np.int64), rg[name].astype(np.int64)
        if name.startswith("FTSpct"):   # four 8-bit percentages per field: each within 1
            d = np.max([np.abs(((a >> s) & 0xFF) - ((b >> s) & 0xFF)) for s in (0, 8, 16, 24)], axis=0)
            assert d.max(initial=0) <= 1, (name, int(np.argmax(d)))
            worst[name] = int(d.max(initial=0))
            continue
        d = np.abs(a - b)
        if name in EXACT_FIELDS:
            assert d.max(initial=0) == 0, (name, int(np.argmax(d)), int(a[np.argmax(d)]), int(b[np.argmax(d)]))
        elif name in PCT_FIELDS:
            tol = np.maximum(1, np.abs(a) // 100)
            assert (d <= tol).all(), (name, int(np.argmax(d - tol)))
        else:
            assert d.max(initial=0) <= 1, (name, int(np.argmax(d)), int(a[np.argmax(d)]), int(b[np.argmax(d)]))
        worst[name] = int(d.max(initial=0))
    return worst


@pytest.mark.parametrize("name,all_out", [("config1_10kb_30x", False), ("config1_10kb_30x", True), ("config2shape_5kb_300x", False), ("umi_duplex_2kb_400x", True),
                                          ("config4shape_1kb_2000x_duplex", False), ("deep_nonumi_800bp_3000x", True)])
def test_score_records_match_oracle(name, all_out, oracle_lib, gpu_lib):
    reads = synth.generate_region(**CASES[name])
    Ro = run_region(oracle_lib, reads)
    Rg = run_region(gpu_lib, reads)
    ro, rg = Ro.score(all_out=all_out), Rg.score(all_out=all_out)
    assert len(ro["refpos"]) > 0
    worst = compare_records(ro, rg)
    print(name, all_out, len(ro["refpos"]), {k: v for k, v in worst.items() if v})


@pytest.mark.parametrize("name", ["config2shape_5kb_300x", "umi_duplex_2kb_400x"])
def test_accumulate_is_repeatable(name, oracle_lib, gpu_lib):
    """A second accumulate on the same handle (what bench.py times) must reproduce the first: the transient
    bucket planes are not re-zeroed between calls, their consumers (P3b, P5b) have to leave them clean."""
    reads = synth.generate_region(**CASES[name])
    Ro = run_region(oracle_lib, reads)
    Rg = run_region(gpu_lib, reads)
    Rg.accumulate()
    Rg.accumulate()
    bad = diff_groups(Ro, Rg)
    assert not bad, "\n".join("%s: %d cells differ, e.g. %s" % (g, v[0], v[1]) for g, v in bad.items())


def test_reads_with_many_mismatches_and_indels(oracle_lib, gpu_lib):
    """Stress the rare-symbol paths: 3 % substitution errors, an InDel every 300 bp, 10 % clipped reads."""
    reads = synth.generate_region(region_len=4000, depth=120, seed=21, err_rate=0.03, indel_every=300, snv_every=150, clip_frac=0.1)
    Ro = run_region(oracle_lib, reads)
    Rg = run_region(gpu_lib, reads)
    bad = diff_groups(Ro, Rg)
    assert not bad, "\n".join("%s: %d cells differ, e.g. %s" % (g, v[0], v[1]) for g, v in bad.items())
    worst = compare_records(Ro.score(all_out=False), Rg.score(all_out=False))
    print({k: v for k, v in worst.items() if v})


def tumor_keys_from(records, every=2):
    """A plausible tumor-sample channel made from the tumor-only records of the same data: every `every`-th record becomes a key."""
    keys = []
    for i in range(0, len(records["refpos"]), every):
        sym = int(records["symbol"][i])
        is_indel = sym in (7, 8, 9, 10, 11, 12)
        keys.append((int(records["refpos"][i]), sym, int(records["cDP1x"][i]), int(records["CDP1x0"][i]), int(records["bAD"][i]), int(records["bDP"][i]),
                     int(records["tier2"][i]) | (i % 2), (1 + i % 4) if is_indel else 0,
                     int(records["cVQ1"][i]), int(records["cPCQ1"][i]), int(records["cDP2x"][i]), int(records["CDP2x0"][i]), int(records["cVQ2"][i]), int(records["cPCQ2"][i]),
                     int(records["bNMQ"][i]), int(records["vHGQ"][i]), int(records["DP"][i]) * (1 + 3 * (i % 3 == 0))))
    keys = sorted(set(keys), key=lambda k: (k[0], k[1]))
    return keys


@pytest.mark.parametrize("name", ["config1_10kb_30x", "umi_duplex_2kb_400x"])
def test_normal_sample_of_a_tn_pair(name, oracle_lib, gpu_lib):
    """SURVEY next-row N2: with vcf_tumor_fname provided only positions that carry a tumor record are scored, every symbol of them,
    and tpfa / the tier-2 flag / the InDel length come from the record (main.cpp:806-986, main.hpp:4297, 4475, 4804-4810)."""
    from uvc_amd import region
    reads = synth.generate_region(**CASES[name])
    keys = tumor_keys_from(run_region(oracle_lib, reads).score(all_out=False))
    assert len(keys) >= 8
    out = []
    for lib in (oracle_lib, gpu_lib):
        p = region.default_params(lib)
        p.tumor_vcf_is_provided = 1
        R = run_region(lib, reads, params=p)
        out.append(R.score(tumor_keys=keys))
        p.tn_syserr_norm_devqual = -1.0; p.outvar_flag = 63      # the other arm of the somatic quality + germline lines in the normal
        R2 = run_region(lib, reads, params=p)
        out.append(R2.score(tumor_keys=keys))
    compare_records(out[1], out[3])
    ro, rg = out[0], out[2]
    assert (ro["tkey"] >= 0).sum() >= len(keys) and ro["out"].sum() == (ro["tkey"] >= 0).sum() - ((ro["tkey"] >= 0) & (ro["symbol"] == 13)).sum()   # records with a tumor key are written (LINK_NN is blocked)
    assert len(set(ro["NLODV"][ro["out"] == 1].tolist())) > 1
    assert set(zip(ro["refpos"].tolist())) == set((k[0],) for k in keys)      # exactly the rescued positions
    assert len(ro["refpos"]) >= 14 * len(set(k[0] for k in keys))              # every symbol of both symbol types at a rescued position
    worst = compare_records(ro, rg)
    print(name, len(ro["refpos"]), {k: v for k, v in worst.items() if v})


VARIANTS = {
    "iontorrent": dict(platform=2),                                            # neighbouring-quality values, every fragment on the generic path
    "primer_gating": dict(set=dict(primerlen=12)),                             # amplicon primer window (main.hpp:1872-1875) gates P2 / P3
    "paired_primer_filter": dict(set=dict(primerlen=12, tn_is_paired=1, primer_flag=1)),
    "sscs_table": dict(set=dict(fam_flag=1)),                                  # PhredMutationTable cap in P3 (main.hpp:2758)
    "short_reads_low_thresholds": dict(set=dict(central_readlen=75, bias_thres_highBQ=10, bias_thres_PFBQ1=40, bias_thres_PFBQ2=45, fam_thres_highBQ_snv=5)),
    "bq_added": dict(set=dict(bq_phred_added_misma=6, bq_phred_added_indel=3, microadjust_padded_deletion_flag=3)),
    "fastq_only": dict(set=dict(inferred_is_vcf_generated=0)),                 # P1/P2/P3 skipped (main.hpp:3691)
    "germline_lines": dict(set=dict(outvar_flag=63, should_output_all_germline=1, vqual=5.0)),   # OUTVAR_GERMLINE: GERMLINE lines are written, so REF alleles are kept too
    "germline_default_gate": dict(set=dict(outvar_flag=63)),
    "normv_quals2": dict(set=dict(tn_syserr_norm_devqual=-1.0, min_a_ad=3, vad1=2, vdp1=50)),   # calc_binom_powlaw_syserr_normv_quals2 arm + the AD / DP keep rules
    "interfering_indel_at_the_limit": dict(set=dict(bias_thres_interfering_indel=10000)),   # the largest threshold the kernels take (dist_to_interfering_indel is 10000 on a simple read, main.hpp:1897): aP3 needs both ends 10000 away, the gap side still enters its bias block
}


@pytest.mark.gpu
def test_interfering_indel_threshold_beyond_the_distance_code_is_refused(gpu_lib):
    """Above 10000 the reference compares the threshold with differences of genome coordinates (the sentinels of a read's InDel list); the
    kernels carry the distance in 16 bits with '10000 or more' as one value: such a threshold is refused when the handle is made, not approximated."""
    from uvc_amd import region
    reads = synth.generate_region(seed=5, region_len=600, depth=10)
    p = region.default_params(gpu_lib)
    p.bias_thres_interfering_indel = 10001
    with pytest.raises(region.UvcError) as e:
        region.Region(gpu_lib, p, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    assert e.value.code == -3 and "above 10000" in str(e.value)   # UVCGPU_EUNSUPPORTED


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("name", ["nodedup_3kb_60x", "umi_duplex_2kb_400x"])
def test_parameter_variants(name, variant, oracle_lib, gpu_lib):
    """Non-default parameter arms of the hot path: planes bit-exact, records within tolerance."""
    from uvc_amd import region
    v = VARIANTS[variant]
    kw = dict(CASES[name]); kw["indel_every"] = 400; kw["clip_frac"] = 0.05
    reads = synth.generate_region(**kw)
    out = []
    for lib in (oracle_lib, gpu_lib):
        p = region.default_params(lib, platform=v.get("platform", 1))
        for k, val in v.get("set", {}).items():
            assert hasattr(p, k), k
            setattr(p, k, val)
        out.append(run_region(lib, reads, params=p))
    Ro, Rg = out
    bad = diff_groups(Ro, Rg)
    assert not bad, "\n".join("%s: %d cells differ, e.g. %s" % (g, vv[0], vv[1]) for g, vv in bad.items())
    if variant != "fastq_only":
        compare_records(Ro.score(all_out=False), Rg.score(all_out=False))


def test_score_request_options(oracle_lib, gpu_lib):
    """UvcScoreRequest arms: a sub-range (pos_beg / pos_end), the amplicon minABQ set (main.cpp:524-525), and host-supplied InDel
    alleles (several alleles of one (position, symbol) with their own bDPa / cDP0a / length, main.cpp:853-904)."""
    kw = dict(CASES["config2shape_5kb_300x"]); kw["indel_every"] = 300
    reads = synth.generate_region(**kw)
    Ro, Rg = run_region(oracle_lib, reads), run_region(gpu_lib, reads)
    base = Ro.score()
    indel = [i for i in range(len(base["refpos"])) if 7 <= base["symbol"][i] <= 12]
    assert len(indel) >= 3
    alleles = []
    for i in indel[:6]:
        pos, sym, b, c = int(base["refpos"][i]), int(base["symbol"][i]), int(base["bDPa"][i]), int(base["cDP0a"][i])
        alleles += [(pos, sym, max(1, b - 1), max(1, c - 1), 1 + (i % 3)), (pos, sym, 1, 1, 4 + (i % 5))]     # two alleles each
    alleles = sorted(set(alleles), key=lambda a: (a[0], a[1]))
    beg, end = reads["beg"] + 1000, reads["beg"] + 4000
    for kwargs in (dict(indel_alleles=alleles), dict(is_amplicon=True), dict(pos_beg=beg, pos_end=end), dict(pos_beg=beg, pos_end=end, all_out=True, indel_alleles=alleles, is_amplicon=True)):
        ro, rg = Ro.score(**kwargs), Rg.score(**kwargs)
        assert len(ro["refpos"]) > 0
        compare_records(ro, rg)
        if "pos_beg" in kwargs:
            assert ro["refpos"].min() >= beg - 1 and ro["refpos"].max() < end
    assert len(Ro.score(indel_alleles=alleles)["refpos"]) > len(base["refpos"])      # the extra alleles became extra records


def test_more_than_65535_fragments_on_one_position(oracle_lib, gpu_lib):
    """k_frag packs two 16-bit bucket counters per LDS word while no position is covered by 65 536 fragments or more; this pile-up
    (a 70 000-read amplicon stack) takes the 32-bit variant."""
    rng = np.random.default_rng(8)
    n, L, ref_len, beg = 70000, 60, 400, 7_000_000
    ref = rng.integers(0, 4, ref_len)
    start = 150 + rng.integers(0, 3, n)
    bases = ref[start[:, None] + np.arange(L)[None, :]]
    err = rng.random((n, L)) < 0.004
    bases = np.where(err, rng.integers(0, 4, (n, L)), bases).astype(np.uint8)
    reads = dict(n_reads=n, pos=(beg + start).astype(np.int32), mpos=np.full(n, -1, np.int32), isize=np.zeros(n, np.int32), flag=np.where(np.arange(n) % 2, 16, 0).astype(np.uint16),
                 mapq=np.full(n, 60, np.uint8), nm=np.full(n, -1, np.int32), l_qseq=np.full(n, L, np.int32), seq_off=(np.arange(n, dtype=np.int64) * L), cigar_off=np.arange(n, dtype=np.int64),
                 n_cigar=np.ones(n, np.int32), frag_id=np.arange(n, dtype=np.int32), fam_id=np.arange(n, dtype=np.int32), fam_strand=(np.arange(n) % 2).astype(np.uint8), n_fams=n,
                 fam_dflag=np.zeros(n, np.uint8), bases=bases.reshape(-1), quals=rng.choice([20, 30, 37], n * L).astype(np.uint8), cigars=np.full(n, (L << 4) | 0, np.uint32),
                 tid=1, beg=beg, end=beg + ref_len, refseq="".join("ACGT"[b] for b in ref))
    Ro, Rg = run_region(oracle_lib, reads), run_region(gpu_lib, reads)
    bad = diff_groups(Ro, Rg)
    assert not bad, "\n".join("%s: %d cells differ, e.g. %s" % (g, v[0], v[1]) for g, v in bad.items())
    assert Rg.fetch("FRAG")[:, 0].sum(axis=(0, 1)).max() >= 65536
    compare_records(Ro.score(), Rg.score())


def test_release_state(gpu_lib):
    """UvcScoreRequest::release_state: the planes are zeroed behind the scoring kernels instead of in front of the next accumulate;
    until then fetch / score refuse, the allele tables stay readable, and the next accumulate gives the same results."""
    from uvc_amd import region
    reads = synth.generate_region(**CASES["config2shape_5kb_300x"])
    R = run_region(gpu_lib, reads)
    planes = {g: R.fetch(g).copy() for g in ("SEG32", "FRAG", "FAM", "VQ", "PREP32")}
    rec = R.score(release_state=True)
    alleles = R.indel_alleles()
    for call in (lambda: R.fetch("SEG32"), lambda: R.score()):
        with pytest.raises(region.UvcError) as e:
            call()
        assert e.value.code == -5
    for _ in range(2):                       # released -> zeroed path, then the ordinary memset path
        R.accumulate()
        assert all(np.array_equal(R.fetch(g), planes[g]) for g in planes)
        rec2 = R.score()
        assert all(np.array_equal(rec[k], rec2[k]) for k in rec) and R.indel_alleles() == alleles
    # a buffer that is too small: UVCGPU_ENOMEM with the count, and the planes are still there for the second call (region.py retries)
    R.accumulate()
    rec3 = R.score(release_state=True, capacity=8)
    assert all(np.array_equal(rec[k], rec3[k]) for k in rec)


def test_reused_handle_same_length_equals_fresh_handles(oracle_lib, gpu_lib):
    """One handle over a run of regions of EQUAL length (what a tile stream does): the zero fill in front of an accumulate then skips the
    (plane family, symbol, block) parts the previous accumulate did not mark (RegionDev::dirty).  Every region's planes must equal the
    oracle's -- a stale cell of the previous region would show -- with and without release_state, and across UMI / InDel-dense / plain reads
    so that the marked set changes from region to region."""
    specs = [dict(region_len=9000, depth=60, seed=21, indel_every=150, snv_every=80),            # many rare symbols
             dict(region_len=9000, depth=40, seed=22, indel_every=100000, snv_every=100000, err_rate=0.0),   # almost none: stale cells of the first would survive a wrong skip
             dict(region_len=9000, depth=200, seed=23, umi=True),                                # FAMINFO / DUPLEX families
             dict(region_len=9000, depth=30, seed=24, err_rate=0.0, indel_every=100000),         # none again
             dict(region_len=9000, depth=50, seed=25, umi=True, indel_every=300)]
    from uvc_amd import region
    R = None
    for k, sp in enumerate(specs):
        reads = synth.generate_region(**sp)
        Ro = run_region(oracle_lib, reads)
        if R is None:
            R = region.Region(gpu_lib, region.default_params(gpu_lib), reads["tid"], reads["beg"], reads["end"], reads["refseq"])
        else:
            R.reset(reads["tid"], reads["beg"], reads["end"], reads["refseq"])
        R.set_reads(reads)
        R.accumulate()
        bad = diff_groups(Ro, R)
        assert not bad, "region %d: " % k + "\n".join("%s: %d cells differ, e.g. %s" % (g, v[0], v[1]) for g, v in bad.items())
        ro = Ro.score(all_out=True)
        rg = R.score(all_out=True, release_state=(k % 2 == 1))   # every other region hands its planes back with the score call (zeroed on the side stream)
        compare_records(ro, rg)
        Ro.close()
    R.close()
