#!/usr/bin/env python3
"""CPU helper (not a test of the suite): the oracle against the chain of independent Python restatements (tests/golden/make_chain_golden.py) over many
fuzz seeds -- all 14 plane groups, the InDel allele rows, and the records of every symbol (all-out and under the default gate) incl. the calling step,
exactly.  No GPU.    python3 scripts/cpu_soak_chain.py SECONDS [FIRST_SEED] [tn]
tn: also the normal-sample pass of a T/N pair on tumor keys made from the chain's own default-gate records of the tumor pass."""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from uvc_amd import _ffi, region  # noqa: E402
from util import INT_GROUPS  # noqa: E402
spec = importlib.util.spec_from_file_location("mcg", os.path.join(ROOT, "tests", "golden", "make_chain_golden.py")); mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
import test_chain_golden as T  # noqa: E402

budget, seed = float(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
with_tn = len(sys.argv) > 3 and sys.argv[3] == "tn"
olib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")
t0, n_ok, n_rec, fails = time.time(), 0, 0, []
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    umi, platform = bool(rng.integers(0, 2)), int(rng.choice([1, 1, 2]))
    n_frag, ref_len = int(rng.choice([60, 120, 250])), int(rng.choice([250, 420, 700]))
    reads = mg.weird_region(seed, n_frag=n_frag, ref_len=ref_len, umi=umi)
    P = mg.params_for(platform, 0)
    fam_flag = int(rng.integers(0, 2)); P.fam_flag = fam_flag
    try:
        alleles, planes = mg.chain_planes(reads, P, platform, 0)
        Po = region.default_params(olib, platform=platform); Po.fam_flag = fam_flag
        R = region.Region(olib, Po, reads["tid"], reads["beg"], reads["end"], reads["refseq"]); R.set_reads(reads); R.accumulate()
        for g in INT_GROUPS:
            got = R.fetch(g)
            assert np.array_equal(got, planes[g]), (g, np.argwhere(got != planes[g])[:4].tolist())
        arows = mg.allele_rows(alleles, reads["refseq"], int(reads["beg"]))
        got_rows = {}
        for r in R.indel_alleles():
            x = r["refpos"] - reads["beg"]
            text = r["seq"] if r["seq"] is not None else reads["refseq"][x:x + r["len"]]
            got_rows[(r["refpos"], r["symbol"], r["strand"], text)] = (r["bAD1"], r["cAD1"], r["c2AD"], r["c2dAD"])
        assert got_rows == arows, ("allele rows", sorted(set(got_rows.items()) ^ set(arows.items()))[:4])
        recs = mg.chain_records(planes, reads, P, arows, all_out=True)
        n_rec += T.compare_with_chain(R.score(all_out=True), recs, True, True) if (R.score(all_out=True)["out"] != 0).sum() > 100 else 0
        gated = mg.chain_records(planes, reads, P, arows, all_out=False)
        if len(gated.get("refpos", [])) > 50 and (R.score(all_out=False)["out"] != 0).sum() > 10:
            n_rec += T.compare_with_chain(R.score(all_out=False), gated, True, False)
        R.close()
        if with_tn and len(gated.get("refpos", [])) >= 8:
            keys = mg.tumor_keys_from_chain(gated)
            Pn = mg.params_for(platform, 1); Pn.fam_flag = fam_flag
            al_n, planes_n = mg.chain_planes(reads, Pn, platform, 1)
            nrecs = mg.chain_records_normal(planes_n, reads, Pn, mg.allele_rows(al_n, reads["refseq"], int(reads["beg"])), keys)
            Pon = region.default_params(olib, platform=platform); Pon.fam_flag = fam_flag; Pon.tumor_vcf_is_provided = 1
            Rn = region.Region(olib, Pon, reads["tid"], reads["beg"], reads["end"], reads["refseq"]); Rn.set_reads(reads); Rn.accumulate()
            for g in INT_GROUPS:
                assert np.array_equal(Rn.fetch(g), planes_n[g]), ("normal", g)
            got = Rn.score(tumor_keys=keys)
            assert set(got["refpos"].tolist()) == set(int(k[0]) for k in keys), "rescued positions"
            if len(nrecs.get("refpos", [])) and (got["out"] != 0).sum() > 10:
                n_rec += T.compare_with_chain(got, nrecs, True, False)
            Rn.close()
        n_ok += 1
    except (AssertionError, KeyError, region.UvcError) as e:
        fails.append(seed); print("FAIL seed", seed, dict(umi=umi, platform=platform, n_frag=n_frag, ref_len=ref_len, fam_flag=fam_flag), repr(e)[:600], flush=True)
    seed += 1
print("chain soak (CPU): %d regions, %d records equal, %d FAILED %s in %.0f s" % (n_ok, n_rec, len(fails), fails[:20], time.time() - t0))
sys.exit(1 if fails else 0)
