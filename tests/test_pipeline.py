"""BAM + FASTA -> records through the whole chain (uvc_amd/pipeline.py): the HIP libraries against the oracle libraries on files
written by tests/bamwriter.py, UMI and non-UMI.  The -m "not gpu" part runs the chain on the oracle alone (sanity of the glue)."""
import io

import numpy as np
import pytest

from uvc_amd import io as uio, pipeline, synth
import bamwriter


def make_files(d, umi):
    reads = synth.generate_region(seed=41 + umi, region_len=6000, depth=60 if not umi else 150, beg=30000, umi=bool(umi), snv_every=300, somatic_every=900, indel_every=500)
    rng = np.random.default_rng(5)
    umis = None
    if umi:   # duplex-structured UMIs in the read names, "name#ALPHA+BETA"
        umis = ["".join("ACGT"[i] for i in rng.integers(0, 4, 6)) + "+" + "".join("ACGT"[i] for i in rng.integers(0, 4, 6)) for _ in range(int(reads["n_fams"]))]
    recs = bamwriter.records_from_reads(reads, tid=0, umis=umis)
    chrom_len = reads["end"] + 5000
    ref = rng.integers(0, 4, chrom_len)
    seq = "".join("ACGT"[i] for i in ref)
    seq = seq[:reads["beg"]] + reads["refseq"] + seq[reads["end"]:]
    bamwriter.write_bam(str(d / ("u%d.bam" % umi)), [("chrT", chrom_len)], recs)
    bamwriter.write_fasta(str(d / ("u%d.fa" % umi)), [("chrT", seq)])
    return reads


def test_chain_on_the_oracle(tmp_path, oracle_lib):
    reads = make_files(tmp_path, 0)
    bam, fa = uio.Bam(str(tmp_path / "u0.bam")), uio.Fasta(str(tmp_path / "u0.fa"))
    res = pipeline.call_region(oracle_lib, bam, fa, "chrT", reads["beg"] + 500, reads["beg"] + 5500)
    rec = res["records"]
    assert res["n_reads"] > 1000 and res["rpos"] == (reads["beg"] + 500, reads["beg"] + 5500)
    assert rec["keep"].sum() >= 5 and (rec["refpos"] >= reads["beg"] + 499).all() and (rec["refpos"] <= reads["beg"] + 5500).all()
    out = io.StringIO(); pipeline.write_tsv(res, out)
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 + int(rec["keep"].sum()) and lines[1].split("\t")[0] == "chrT"
    assert pipeline.call_region(oracle_lib, bam, fa, "chrT", 100, 2000) is None          # nothing aligned there
    tiles = list(pipeline.call_contig(oracle_lib, bam, fa, "chrT", tile=2500))
    par = list(pipeline.call_contig(oracle_lib, str(tmp_path / "u0.bam"), str(tmp_path / "u0.fa"), "chrT", tile=2500, workers=3))   # three tiles in flight, own handles each
    assert [t["rpos"] for t in par] == [t["rpos"] for t in tiles] and all(np.array_equal(a["records"]["TLODQ"], b["records"]["TLODQ"]) for a, b in zip(par, tiles))
    assert len(tiles) == 3 and [t["rpos"][0] for t in tiles] == sorted(t["rpos"][0] for t in tiles)
    # a position scored in two different tilings gets the same record: every tile re-reads its own halo
    a = {(int(p), int(s)): int(q) for t in tiles for p, s, q in zip(t["records"]["refpos"], t["records"]["symbol"], t["records"]["TLODQ"])}
    b = {(int(p), int(s)): int(q) for p, s, q in zip(rec["refpos"], rec["symbol"], rec["TLODQ"])}
    common = [k for k in b if k in a and reads["beg"] + 600 < k[0] < reads["beg"] + 5400 and abs((k[0] - 30000) % 2500) > 150 and abs((k[0] - 30000) % 2500 - 2500) > 150]
    assert len(common) >= 50 and all(a[k] == b[k] for k in common)


@pytest.mark.gpu
@pytest.mark.parametrize("umi", [0, 1])
def test_chain_gpu_equals_oracle(tmp_path, umi, oracle_lib, gpu_lib):
    from test_gpu_parity import compare_records
    reads = make_files(tmp_path, umi)
    bam, fa = uio.Bam(str(tmp_path / ("u%d.bam" % umi))), uio.Fasta(str(tmp_path / ("u%d.fa" % umi)))
    ro = pipeline.call_region(oracle_lib, bam, fa, "chrT", reads["beg"] + 200, reads["beg"] + 5800, molecule_tag=0)
    rg = pipeline.call_region(gpu_lib, bam, fa, "chrT", reads["beg"] + 200, reads["beg"] + 5800, molecule_tag=0)
    assert (ro["n_reads"], ro["n_fams"], ro["rpos"], ro["ext"]) == (rg["n_reads"], rg["n_fams"], rg["rpos"], rg["ext"])
    assert ro["alleles"] == rg["alleles"]
    compare_records(ro["records"], rg["records"])
    assert ro["records"]["keep"].sum() >= 5


@pytest.mark.gpu
def test_a_reset_handle_equals_a_fresh_one(tmp_path, gpu_lib):
    """Tiles of different lengths through one handle (uvcgpu_region_reset: buffers kept while the region does not grow, reallocated when
    it does) give the records of a fresh handle per tile."""
    reads = make_files(tmp_path, 1)
    bam, fa = uio.Bam(str(tmp_path / "u1.bam")), uio.Fasta(str(tmp_path / "u1.fa"))
    b0 = reads["beg"]
    spans = [(b0 + 100, b0 + 1500), (b0 + 1500, b0 + 2100), (b0 + 2100, b0 + 5900), (b0 + 300, b0 + 900)]   # shrink, grow, shrink
    holder = {}
    for beg, end in spans:
        a = pipeline.call_region(gpu_lib, bam, fa, "chrT", beg, end, reuse=holder)
        b = pipeline.call_region(gpu_lib, bam, fa, "chrT", beg, end)
        assert a["rpos"] == b["rpos"] and a["alleles"] == b["alleles"]
        assert all(np.array_equal(a["records"][k], b["records"][k]) for k in a["records"]), (beg, end)
    tiles = list(pipeline.call_contig(gpu_lib, str(tmp_path / "u1.bam"), str(tmp_path / "u1.fa"), "chrT", b0, b0 + 6000, tile=1500, workers=2))
    assert len(tiles) == 4


@pytest.mark.gpu
def test_files_to_vcf(tmp_path, gpu_lib):
    """BAM + FASTA -> block-gzipped VCF: header of the library, record lines of every tile (uvcgpu_region_vcf_records), BGZF writer."""
    import gzip
    reads = make_files(tmp_path, 0)
    out = str(tmp_path / "calls.vcf.gz")
    b0 = reads["beg"]
    n = pipeline.write_vcf(gpu_lib, str(tmp_path / "u0.bam"), str(tmp_path / "u0.fa"), "chrT", b0, b0 + 6000, out, sample="T1", tile=2000)
    text = gzip.open(out, "rt").read().splitlines()
    head, body = [l for l in text if l.startswith("#")], [l for l in text if not l.startswith("#")]
    assert head[0] == "##fileformat=VCFv4.2" and head[-1].split("\t")[-1] == "T1" and any(l.startswith("##contig=<ID=chrT,") for l in head)
    assert n == len(body) >= 5
    declared = {l.split("ID=")[1].split(",")[0] for l in head if l.startswith("##FORMAT=")}
    pos = []
    n_symbolic = 0
    for l in body:
        c = l.split("\t")
        pos.append(int(c[1]))
        if c[4] in ("<NON_REF>", "<ADDITIONAL_INDEL_CANDIDATE>"):     # position-level lines: MGVCF blocks, InDel candidates
            assert c[7].split(";")[0] in ("MGVCF_BLOCK", "ADDITIONAL_INDEL_CANDIDATE") and set(c[8].split(":")) <= declared
            n_symbolic += 1
            continue
        assert len(c) == 10 and c[0] == "chrT" and c[6] in pipeline.FILTERS and c[7].startswith("ANY_VAR;")
        keys, vals = c[8].split(":"), c[9].split(":")
        assert len(keys) == len(vals) and set(keys) <= declared and keys[0] == "GT" and vals[0] == "./1"
    assert pos == sorted(pos) and n_symbolic >= 5          # 6 kb: one block line per 1000 positions that are scored
    # the same records as the table writer sees them
    bam, fa = uio.Bam(str(tmp_path / "u0.bam")), uio.Fasta(str(tmp_path / "u0.fa"))
    kept = sum(int(t["records"]["keep"].sum()) for t in pipeline.call_contig(gpu_lib, bam, fa, "chrT", b0, b0 + 6000, tile=2000))
    assert kept == len(body) - n_symbolic


@pytest.mark.gpu
def test_native_command_line_equals_the_python_chain(tmp_path, gpu_lib):
    """uvc_amd/csrc/uvc1-mi355x (the chain in C++, tiles in flight on threads) writes the record lines of uvc_amd/pipeline.py."""
    import gzip
    import os
    import subprocess
    from uvc_amd import _ffi
    exe = os.path.join(_ffi.ROOT, "uvc_amd", "csrc", "uvc1-mi355x")
    assert os.path.exists(exe), "build it: make -C uvc_amd/csrc"
    for umi in (0, 1):
        reads = make_files(tmp_path, umi)
        bam, fa = str(tmp_path / ("u%d.bam" % umi)), str(tmp_path / ("u%d.fa" % umi))
        b0 = reads["beg"]
        target = "chrT:%d-%d" % (b0 + 1, b0 + 6000)
        out_c, out_py = str(tmp_path / "c.vcf.gz"), str(tmp_path / "py.vcf.gz")
        r = subprocess.run([exe, bam, "-f", fa, "-o", out_c, "-s", "T1", "--targets", target, "--tile", "2000", "-t", "3", "--timing"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert "record lines" in r.stderr
        pipeline.write_vcf(gpu_lib, bam, fa, "chrT", b0, b0 + 6000, out_py, sample="T1", tile=2000)
        a, b = gzip.open(out_c, "rt").read().splitlines(), gzip.open(out_py, "rt").read().splitlines()
        assert [l for l in a if not l.startswith("##")] == [l for l in b if not l.startswith("##")]
        assert len([l for l in a if not l.startswith("#") and "ANY_VAR" in l]) >= 5
    # -R regions.bed: the same three tiles as BED lines
    bed = str(tmp_path / "r.bed")
    open(bed, "w").write("# comment\n" + "".join("chrT\t%d\t%d\n" % (b0 + k, b0 + k + 2000) for k in (0, 2000, 4000)))
    out_b = str(tmp_path / "b.vcf.gz")
    r = subprocess.run([exe, bam, "-f", fa, "-o", out_b, "-s", "T1", "-R", bed, "-t", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert gzip.open(out_b, "rt").read() == gzip.open(out_c, "rt").read()
    r = subprocess.run([exe, bam, "-f", fa, "-o", out_c, "--no-such-option"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "unknown option" in r.stderr
