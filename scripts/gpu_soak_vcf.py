#!/usr/bin/env python3
"""GPU-box helper (not a test of the suite): the record-line comparison of tests/test_vcf_text.py (the library's VCF text against the oracle's
values streamed through the reference's own streamAppendBcfFormat: record lines incl. bHap / cHap / c2Hap, InDel strings, MGVCF blocks,
ADDITIONAL_INDEL_CANDIDATE and GERMLINE lines) over many seeds for a given number of seconds.   python3 scripts/gpu_soak_vcf.py SECONDS [FIRST_SEED]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from uvc_amd import _ffi, region, synth  # noqa: E402
from test_gpu_fuzz import weird_region  # noqa: E402
from test_vcf_text import _load_ref_vcf, _oracle_lines, compare_lines  # noqa: E402
from test_gpu_parity import tumor_keys_from  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
glib = region.gpu_lib(); assert glib.dll.uvcgpu_init(0) == 0
olib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")
ref_vcf = _load_ref_vcf()
t0, n_ok, n_lines, n_tn, fails = time.time(), 0, 0, 0, []
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    umi, platform, all_out = bool(rng.integers(0, 2)), int(rng.choice([1, 1, 2])), bool(rng.integers(0, 2))
    if rng.random() < 0.5:
        reads = weird_region(seed, n_frag=int(rng.choice([60, 260, 600])), ref_len=int(rng.choice([300, 700, 1500])), umi=umi)
    else:
        reads = synth.generate_region(seed=seed, region_len=int(rng.choice([2000, 4000])), depth=int(rng.choice([40, 150, 400])), umi=umi, snv_every=int(rng.choice([150, 1000])),
                                      somatic_every=int(rng.choice([400, 10000])), indel_every=int(rng.choice([200, 800])), err_rate=float(rng.choice([1e-3, 1e-2])))
    sets = {}
    if rng.random() < 0.4: sets.update(outvar_flag=63)                       # GERMLINE / MGVCF / ADDITIONAL_INDEL_CANDIDATE lines
    if rng.random() < 0.2: sets.update(should_output_all_germline=1, vqual=5.0)
    try:
        R = []
        for lib in (olib, glib):
            P = region.default_params(lib, platform=platform)
            for k, v in sets.items(): setattr(P, k, v)
            r = region.Region(lib, P, reads["tid"], reads["beg"], reads["end"], reads["refseq"]); r.set_reads(reads); r.accumulate(); R.append(r)
        rg = R[1].score(all_out=all_out)
        mine = R[1].vcf_records("chrS", rg).splitlines()
        want = _oracle_lines(olib, ref_vcf, R[0], "chrS", all_out=all_out)
        assert len(mine) == len(want), ("line count", len(mine), len(want))
        compare_lines(mine, want)
        n_ok += 1; n_lines += len(want)
        if rng.random() < 0.4 and not all_out:
            # the normal-sample pass of a T/N pair on keys made from these tumor-only records; InDel keys carry "REF\tALT" strings (rescued InDel
            # records take their string from them, main.cpp:867-880)
            ro = R[0].score(all_out=False)
            keys = [k + (7 + i, 3 + i % 5, 2 * i) for i, k in enumerate(tumor_keys_from(ro, every=int(rng.integers(1, 4))))]
            if len(keys) >= 2:
                ras = []
                for k in keys:
                    ln = int(k[7])
                    if 7 <= k[1] <= 9: ras.append("A" + "C" * max(ln, 1) + "\tA")
                    elif 10 <= k[1] <= 12: ras.append("A\tA" + "G" * max(ln, 1))
                    else: ras.append("A\tC")
                RN = []
                for lib in (olib, glib):
                    P = region.default_params(lib, platform=platform); P.tumor_vcf_is_provided = 1
                    for kk, v in sets.items(): setattr(P, kk, v)
                    r = region.Region(lib, P, reads["tid"], reads["beg"], reads["end"], reads["refseq"]); r.set_reads(reads); r.accumulate(); RN.append(r)
                rgn = RN[1].score(tumor_keys=keys)
                mine = RN[1].vcf_records("chrS", rgn, tumor_keys=keys, tumor_ref_alt=ras).splitlines()
                want = _oracle_lines(olib, ref_vcf, RN[0], "chrS", tumor_keys=keys, tumor_ref_alt=ras)
                assert len(mine) == len(want), ("T/N line count", len(mine), len(want))
                compare_lines(mine, want)
                n_tn += 1; n_lines += len(want)
                for r in RN: r.close()
        for r in R: r.close()
    except (AssertionError, ValueError, region.UvcError) as e:
        detail = ""
        if isinstance(e, ValueError):   # a non-numeric tag differs: say which
            for lm, lw in zip(mine, want):
                cm, cw = lm.split("\t"), lw.split("\t")
                if len(cm) == len(cw) == 10 and cm[8] == cw[8]:
                    for k, a, b in zip(cm[8].split(":"), cm[9].split(":"), cw[9].split(":")):
                        if a != b and not all(x.lstrip("-").isdigit() for x in (a + "," + b).split(",")):
                            detail = " line %s:%s tag %s mine %r want %r" % (cm[0], cm[1], k, a[:200], b[:200]); break
                if detail: break
        fails.append(seed); print("FAIL seed", seed, dict(umi=umi, platform=platform, all_out=all_out, sets=sets), repr(e)[:300] + detail, flush=True)
    seed += 1
print("vcf soak: %d regions + %d normal-sample passes (%d lines) equal, %d FAILED %s in %.0f s" % (n_ok, n_tn, n_lines, len(fails), fails[:20], time.time() - t0))
sys.exit(1 if fails else 0)
