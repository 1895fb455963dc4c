"""tests/golden/chain_*.npz: fuzzed reads + every per-position plane group of the accumulate path (rows a3 - a8) and the scored records of
the base symbols (rows a13 - a17 and the calling step behind them: 116 fields of gather, calc_DPv, sum_DPv, calc_qual, output_germline, NLODQ / TLODQ / QUAL / FILTER) as computed by the chain of independent Python restatements (tests/golden/make_chain_golden.py: no library produced a number in these files; parameters from the
reference's own defaults).  The oracle (CPU suite) and the HIP path through the C ABI (-m gpu) must both reproduce them bit for bit -- the
one parity line of this repository whose expected values come from neither of the two."""
import json
import os
import sys

import numpy as np
import pytest

from uvc_amd import region
from util import INT_GROUPS, run_region
from test_gpu_parity import EXACT_FIELDS, PCT_FIELDS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["chain_illumina_umi", "chain_illumina_plain", "chain_iontorrent_umi_normal", "chain_illumina_umi_deep", "chain_iontorrent_plain"]


def load(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    meta = json.loads(str(z["meta"]))
    reads = {k[len("reads__"):]: z[k] for k in z.files if k.startswith("reads__")}
    reads.update(tid=meta["tid"], beg=meta["beg"], end=meta["end"], refseq=meta["refseq"], n_reads=meta["n_reads"], n_fams=meta["n_fams"])
    return (reads, meta, {k[len("planes__"):]: z[k] for k in z.files if k.startswith("planes__")}, {k[len("records__"):]: z[k] for k in z.files if k.startswith("records__")},
            z["alleles__rows"], z["alleles__text"], {k[len("gated__"):]: z[k] for k in z.files if k.startswith("gated__")},
            ((z["normal__keys"], {k[len("normal__"):]: z[k] for k in z.files if k.startswith("normal__") and k != "normal__keys"}) if "normal__keys" in z.files else None))


def compare_with_chain(got, recs, exact_records, all_out):
    """Library records against the chain's.  A symbol without InDel string has one record per position; the records of an InDel symbol (one per
    majority allele, order among equal bAD1^2 * length left open by the reference's std::sort) are matched in (length, bDPa, cDP0a) order."""
    def order(rec):
        key = np.stack([rec["refpos"].astype(np.int64), rec["symbol"].astype(np.int64), rec["gapSa_len"].astype(np.int64), rec["bDPa"].astype(np.int64), rec["cDP0a"].astype(np.int64)])
        return np.lexsort(key[::-1])
    inner = (got["refpos"] >= recs["refpos"].min()) & (got["refpos"] <= recs["refpos"].max())
    og, ow = order(got), order(recs)
    first, cnt_g = {}, {}
    for j, i in enumerate(og):
        k = (int(got["refpos"][i]), int(got["symbol"][i]))
        first.setdefault(k, j); cnt_g[k] = cnt_g.get(k, 0) + 1
    idx, seen = [], {}
    for i in ow:
        k = (int(recs["refpos"][i]), int(recs["symbol"][i]))
        n = seen.get(k, 0); seen[k] = n + 1
        assert k in first, ("a record the chain expects is missing", k)
        idx.append(og[first[k] + n])
    idx = np.array(idx)
    recs = {k: v[ow] for k, v in recs.items()}
    assert np.array_equal(got["refpos"][idx], recs["refpos"]) and np.array_equal(got["symbol"][idx], recs["symbol"])
    assert all(cnt_g[k] == n for k, n in seen.items()), [(k, n, cnt_g[k]) for k, n in seen.items() if cnt_g[k] != n][:5]      # as many alleles per InDel symbol
    extra_keys = [k for k in cnt_g if k not in seen and recs["refpos"].min() <= k[0] <= recs["refpos"].max()]
    assert not extra_keys, ("records the chain does not expect", extra_keys[:5])          # the gate: exactly the chain's (position, symbol) set
    written = got["out"][idx] != 0
    assert written.sum() > (100 if all_out else 10)
    is_ref = recs["symbol"] == recs["refsymbol"]
    for k, want in recs.items():
        sel = np.ones(len(idx), dtype=bool)
        if k == "has_key":                              # (normal sample) the record was scored with its tumor key
            assert np.array_equal(got["tkey"][idx] >= 0, want != 0), k
            continue
        if k.startswith("call__"):                      # per-record values of the calling step: defined for the records that are written
            k, sel = k[len("call__"):], written
            if k == "keep" and not all_out:
                sel = sel & ~is_ref                     # a REF record under the default gate is kept when a GERMLINE line is (main.cpp:1117): not restated here
        if k == "QUAL":                                 # the record carries the bits of a float
            q = got[k][idx].view(np.float32).astype(np.float64)
            assert (np.abs(q - want)[sel] <= (1e-4 if exact_records else 1e-3) * np.maximum(1.0, np.abs(want[sel]))).all(), k
            continue
        g = got[k][idx].astype(np.int64)
        d = np.abs(g - want)
        if k.startswith("FTSpct") and not exact_records:
            d = np.max([np.abs(((g >> s) & 0xFF) - ((want >> s) & 0xFF)) for s in (0, 8, 16, 24)], axis=0)
        tol = 0 if (exact_records or k in EXACT_FIELDS) else (np.maximum(1, np.abs(want) // 100) if k in PCT_FIELDS else 1)
        d = np.where(sel, d - tol, -1)
        w = int(np.argmax(d))
        assert d.max() <= 0, (k, all_out, int(recs["refpos"][w]), int(recs["symbol"][w]), int(g[w]), int(want[w]))
    return len(idx)


def check(lib, name, exact_records):
    reads, meta, planes, recs, arows, z_text, gated, normal = load(name)
    assert sorted(planes) == sorted(INT_GROUPS)
    P = region.default_params(lib, platform=meta["platform"])
    P.tumor_vcf_is_provided = meta["normal"]
    R = run_region(lib, reads, params=P)
    bad = {}
    for g in INT_GROUPS:
        got, want = R.fetch(g), planes[g]
        assert got.shape == want.shape and got.dtype == want.dtype, (g, got.shape, want.shape, got.dtype, want.dtype)
        if not np.array_equal(got, want):
            idx = np.argwhere(got != want)
            bad[g] = (len(idx), [(tuple(int(v) for v in i), int(got[tuple(i)]), int(want[tuple(i)])) for i in idx[:5]])
    assert not bad, bad
    # the InDel allele rows (what fill_by_indel_info reads: fragment, family, cDP2 and duplex support of every allele per strand)
    want_rows = {}
    texts = str(z_text).split(";") if len(arows) else []
    for row, text in zip(arows, texts):
        want_rows[(int(row[0]), int(row[1]), int(row[2]), text)] = tuple(int(v) for v in row[4:8])
    got_rows = {}
    for r in R.indel_alleles():
        x = r["refpos"] - reads["beg"]
        text = r["seq"] if r["seq"] is not None else reads["refseq"][x:x + r["len"]]
        got_rows[(r["refpos"], r["symbol"], r["strand"], text)] = (r["bAD1"], r["cAD1"], r["c2AD"], r["c2dAD"])
    assert got_rows == want_rows, sorted(set(got_rows.items()) ^ set(want_rows.items()))[:6]
    if len(recs["refpos"]):
        # the scored records: gather -> calc_DPv -> sum_DPv -> calc_qual -> calling step of the restatements on the chain's planes, all-out and
        # under the default gate (main.cpp:835-841)
        assert compare_with_chain(R.score(all_out=True), recs, exact_records, True) > 2000
        assert compare_with_chain(R.score(all_out=False), gated, exact_records, False) > 50
    if normal is not None:
        # the normal sample of a T/N pair on tumor keys made from the chain's own tumor pass: only the keyed positions, every symbol there,
        # the keyed records with the key's tier-2 flag / tpfa / InDel length, NLODQ and the somatic quality through the T/N arm
        keys, nrecs = normal
        got = R.score(tumor_keys=[tuple(int(v) for v in k) for k in keys])
        assert set(got["refpos"].tolist()) == set(int(k[0]) for k in keys)
        assert compare_with_chain(got, nrecs, exact_records, False) > 1000
    R.close()


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_restatement_chain(name, oracle_lib):
    check(oracle_lib, name, exact_records=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_reproduces_the_restatement_chain(name, gpu_lib):
    check(gpu_lib, name, exact_records=False)      # quality fields within 1 Phred, x100 depths within 1 % (device libm), the rest exact


def _regenerate(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_chain_golden", os.path.join(ROOT, "tests", "golden", "make_chain_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    kw = mg.CASES[name]
    reads = mg.weird_region(kw["seed"], n_frag=kw["n_frag"], ref_len=kw["ref_len"], umi=kw["umi"])
    _, planes = mg.chain_planes(reads, mg.params_for(kw["platform"], kw["normal"]), kw["platform"], kw["normal"])
    return planes


@pytest.mark.parametrize("name", CASES)
def test_every_fixture_is_what_the_generator_writes(name):
    """All committed fixtures regenerated on the spot (pure Python, a few seconds each): a fixture that drifted from
    tests/golden/make_chain_golden.py would go unnoticed otherwise."""
    planes = _regenerate(name)
    _, _, gold, _, _, _, _, _ = load(name)
    for g in INT_GROUPS:
        assert np.array_equal(planes[g], gold[g]), g
