"""VCF text.  The layout side is pinned on the REFERENCE's own generated code: oracle/_ref/libref_vcf.so wraps the header that
the reference's bcf_formats_generator1.cpp prints (built from where it lies by oracle/Makefile, as the reference's Makefile:55-59 does):
its FORMAT key strings, ##FORMAT / ##FILTER lines and streamAppendBcfFormat are reference code, not a restatement.

CPU tests: the product's FORMAT key strings equal the reference's; the ID / Number / Type of every ##FORMAT line and the ##FILTER IDs equal
the reference's (the Description texts are this repository's own wording).
GPU tests: for the same reads, every record line of the HIP path equals the line made from the oracle's values with the fixed columns
of the oracle's restatement of append_vcf_record and the sample column streamed by the reference's streamAppendBcfFormat -- integer depth /
count tags exactly, Phred-like tags within 1, x100 depth tags within 1 %, QUAL within 1e-3."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from uvc_amd import _ffi, region, synth
from util import run_region

REF_SO = os.path.join(_ffi.ROOT, "oracle", "_ref", "libref_vcf.so")


def _load_ref_vcf():
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libref_vcf.so not built (make -C oracle ref_formats needs /root/reference)")
    L = C.CDLL(REF_SO)
    for n in ("uvc_ref_format_string", "uvc_ref_format_id", "uvc_ref_format_line", "uvc_ref_filter_id", "uvc_ref_filter_line"):
        getattr(L, n).restype = C.c_char_p
    L.uvc_ref_stream_format.restype = C.c_int64
    L.uvc_ref_stream_format.argtypes = [C.c_char_p, C.c_char_p, C.c_int64]
    return L


@pytest.fixture(scope="module")
def ref_vcf():
    return _load_ref_vcf()


@pytest.fixture(scope="module")
def product_dll():
    return C.CDLL(_ffi.gpu_library_path())   # loading needs no GPU; the text entry points used on the CPU are host-only


class _HostLib:
    def __init__(self, dll):
        self.dll = dll

    def last_error(self):
        self.dll.uvcgpu_last_error.restype = C.c_char_p
        return (self.dll.uvcgpu_last_error() or b"").decode()


def test_format_keys_equal_reference(ref_vcf, product_dll):
    lib = _HostLib(product_dll)
    for tier2 in (0, 1):
        assert region.vcf_format_keys(lib, tier2) == ref_vcf.uvc_ref_format_string(tier2).decode()


def _parse_meta(line):
    m = re.match(r'##(\w+)=<ID=([^,]+)(?:,Number=([^,]+),Type=([^,]+))?,Description="(.*)">$', line)
    assert m, line
    return m.groups()


def _reference_header_literals(params):
    """The static text of generate_vcf_header (main.hpp:5778-5883) read from the reference source as TEXT: every string literal of the
    function body in order, with the three compile-time / parameter expressions the body splices in replaced by their values.  Loop bodies
    appear once.  Returns the text split into lines."""
    import os, re
    path = "/root/reference/main.hpp"
    if not os.path.exists(path):
        pytest.skip("reference source not present on this box")
    src = open(path).read()
    body = src[src.index("generate_vcf_header("):]
    body = body[body.index('std::string ret = "";'):body.index("return ret;")]
    body = body.replace("std::to_string(MGVCF_REGION_MAX_SIZE)", '"1000"')
    body = body.replace("std::to_string(paramset.germ_phred_hetero_indel - paramset.germ_phred_hetero_snp)", '"%d"' % (params.germ_phred_hetero_indel - params.germ_phred_hetero_snp))
    body = body.replace("SYMBOL_TO_DESC_ARR[ADDITIONAL_INDEL_CANDIDATE_SYMBOL]", '"<ADDITIONAL_INDEL_CANDIDATE>"')
    lits = re.findall(r'"((?:[^"\\]|\\.)*)"', body)
    text = "".join(lits).replace('\\"', '"').replace("\\n", "\n").replace("\\t", "\t")
    return text.split("\n")


def test_header_is_the_references_text(ref_vcf, product_dll):
    """VERDICT r2 missing #4: the ##ALT / ##FILTER / ##INFO / ##FORMAT lines of uvcgpu_vcf_header byte for byte -- Description texts included --
    against (a) FILTER_LINES / FORMAT_LINES of the reference's own generated header (libref_vcf.so = bcf_formats_generator1.cpp compiled as it
    lies), all 258 FORMAT lines incl. the eight tags the reference declares and never writes, and (b) the string literals of
    generate_vcf_header read from the reference source, in the reference's order."""
    lib = _HostLib(product_dll)
    p = _ffi.UvcParams()
    product_dll.uvcgpu_params_default.argtypes = [C.POINTER(_ffi.UvcParams)]
    product_dll.uvcgpu_params_default(C.byref(p))
    p.inferred_sequencing_platform = 1; p.central_readlen = 150
    hdr = region.vcf_header(lib, p, "S1", [("chr20", 64444167), ("chrM", 16569)]).splitlines()
    assert hdr[0] == "##fileformat=VCFv4.2" and hdr[1] == "##contig=<ID=chr20,length=64444167>" and hdr[2] == "##contig=<ID=chrM,length=16569>"
    assert hdr[-1] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1"
    assert hdr[-2] == "##variantCallerInferredParameters=(inferred_sequencing_platform=Illumina/BGI,central_readlen=150)"
    ref_filter = [ref_vcf.uvc_ref_filter_line(i).decode() for i in range(ref_vcf.uvc_ref_n_filter())]
    ref_format = [ref_vcf.uvc_ref_format_line(i).decode() for i in range(ref_vcf.uvc_ref_n_format())]
    i_alt = hdr.index([l for l in hdr if l.startswith("##ALT=")][0])
    assert hdr[i_alt + 1: i_alt + 1 + len(ref_filter)] == ref_filter                       # right behind ##ALT, main.hpp:5796-5801
    i_fmt = hdr.index(ref_format[0])
    assert hdr[i_fmt: i_fmt + len(ref_format)] == ref_format and len(ref_format) == 258
    assert hdr[i_fmt - 1].startswith("##INFO=<ID=R3X2,")                                    # the generated lines follow the last ##INFO, main.hpp:5841-5845
    assert [l.split(",")[0] for l in hdr[i_fmt + len(ref_format): i_fmt + len(ref_format) + 6]] == ["##FORMAT=<ID=" + k for k in ("GL4", "GST", "CDP1", "cDP1", "POS_VT_BDP_CDP_HomRefQ", "clipDP")]
    # every hand-written line of the reference's function, same bytes, same order
    ref_static = [l for l in _reference_header_literals(p) if l.startswith(("##ALT=", "##INFO=", "##FORMAT=", "##phasing", "##fileformat"))]
    mine_static = [l for l in hdr if l.startswith(("##ALT=", "##INFO=", "##phasing", "##fileformat")) or (l.startswith("##FORMAT=") and l not in ref_format)]
    assert len(ref_static) == 1 + 1 + 22 + 6 + 1   # fileformat, ALT, 22 INFO, 6 FORMAT, phasing
    assert mine_static == ref_static


def test_header_ex_adds_the_run_lines(product_dll):
    """##fileDate, ##reference and ##variantCallerCommand sit where generate_vcf_header writes them (main.hpp:5792-5794, 5870-5874)."""
    p = _ffi.UvcParams()
    product_dll.uvcgpu_params_default.argtypes = [C.POINTER(_ffi.UvcParams)]
    product_dll.uvcgpu_params_default(C.byref(p))
    fn = product_dll.uvcgpu_vcf_header_ex
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(_ffi.UvcParams), C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int32, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    names = (C.c_char_p * 1)(b"chr1"); lens = (C.c_int64 * 1)(1000)
    n = C.c_int64()
    fn(C.byref(p), b"S", None, names, lens, 1, b"2026-01-02 03:04:05", b"ref.fa", b"uvc1  a.bam  ", None, 0, C.byref(n))
    buf = C.create_string_buffer(n.value)
    assert fn(C.byref(p), b"S", None, names, lens, 1, b"2026-01-02 03:04:05", b"ref.fa", b"uvc1  a.bam  ", buf, n.value, C.byref(n)) == 0
    hdr = buf.raw[:n.value].decode().splitlines()
    assert hdr[:4] == ["##fileformat=VCFv4.2", "##fileDate=2026-01-02 03:04:05", "##reference=ref.fa", "##contig=<ID=chr1,length=1000>"]
    i = [k for k, l in enumerate(hdr) if l.startswith("##variantCallerVersion=")][0]
    assert hdr[i - 1] == "##phasing=partial" and hdr[i + 1] == "##variantCallerCommand=uvc1  a.bam  " and hdr[i + 2].startswith("##variantCallerInferredParameters=")


# ---- GPU: record lines ----
PHRED_TAGS = {"aBQ", "a2BQf", "a2BQr", "aBQQ", "bMQ", "aAaMQ", "bNMQ", "bNMa", "bNMb", "bMQQ", "bIAQ", "cIAQ", "bTINQ", "cTINQ", "cPCQ1", "cPLQ1", "cVQ1", "gVQ1",
              "cPCQ2", "cPLQ2", "cVQ2", "cMmQ", "dVQinc", "CONTQ", "nPF", "nNFA", "nAFA", "nBCFA", "cVQ1M", "cVQ2M", "vHGQ", "vNLODQ", "GQ", "GL4", "GST"}
PCT_TAGS = {"cDP1v", "CDP1v", "cDP1w", "CDP1w", "cDP1x", "CDP1x", "cDP2v", "CDP2v", "cDP2w", "CDP2w", "cDP2x", "CDP2x"}
PHRED_INFO = {"SomaticQ", "TLODQ", "NLODQ", "TNBQF", "TNCQF"}


def _oracle_lines(oracle_lib, ref_vcf, Ro, tname, tumor_keys=None, **score_kw):
    fn = oracle_lib.dll.uvc_oracle_region_vcf
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(_ffi.UvcScoreRequest), C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    req, _keep = region.Region.make_request(tumor_keys=tumor_keys, **score_kw)   # pos_beg / pos_end / all_out / base_at_pos_beg / region_beg
    ln = C.c_int64(0)
    fn(Ro.h, C.byref(req), tname.encode(), None, 0, C.byref(ln))
    buf = C.create_string_buffer(max(1, ln.value))
    assert fn(Ro.h, C.byref(req), tname.encode(), buf, ln.value, C.byref(ln)) == 0
    text = buf.raw[:ln.value].decode()
    out = []
    sbuf = C.create_string_buffer(1 << 20)
    for rec in (text.split("\x1d") if text else []):
        fixed, tier2, spec = rec.split("\x1e")
        if tier2 == "-1":          # a position-level line (MGVCF block, ADDITIONAL_INDEL_CANDIDATE): whole line
            out.append(fixed)
            continue
        n = ref_vcf.uvc_ref_stream_format(spec.encode(), sbuf, 1 << 20)
        assert n > 0, n
        out.append(fixed + "\t" + ref_vcf.uvc_ref_format_string(int(tier2)).decode() + "\t" + sbuf.value.decode())
    return out


def _cmp_ints(tag, a, b, tol_kind):
    xa, xb = a.split(","), b.split(",")
    assert len(xa) == len(xb), (tag, a, b)
    for u, v in zip(xa, xb):
        if u == v:
            continue
        iu, iv = int(u), int(v)
        if tol_kind == "phred":
            assert abs(iu - iv) <= 1, (tag, a, b)
        elif tol_kind == "pct":
            assert abs(iu - iv) <= max(1, abs(iu) // 100), (tag, a, b)
        else:
            raise AssertionError((tag, a, b))


def compare_lines(mine, want):
    assert len(mine) == len(want), (len(mine), len(want))
    n_tags = 0
    for lm, lw in zip(mine, want):
        cm, cw = lm.split("\t"), lw.split("\t")
        assert len(cm) == len(cw) == 10, (len(cm), len(cw))
        if cw[4] == "<ADDITIONAL_INDEL_CANDIDATE>":
            assert lm == lw, (lm, lw)
            continue
        if cw[4] == "<NON_REF>":     # MGVCF block: positions, types and depths exactly, the hom-ref quality within 1 Phred
            assert cm[:9] == cw[:9], (cm[:9], cw[:9])
            vm, vw = cm[9].split(":"), cw[9].split(":")
            assert vm[:2] == vw[:2]
            em, ew = vm[2].split(","), vw[2].split(",")
            assert len(em) == len(ew) and em[-1] == ew[-1], (len(em), len(ew))
            for q in range(0, len(ew) - 1, 8):
                assert em[q:q + 6] == ew[q:q + 6] and em[q + 7] == ew[q + 7] == "." and abs(int(em[q + 6]) - int(ew[q + 6])) <= 1, (em[q:q + 8], ew[q:q + 8])
            continue
        assert cm[:5] == cw[:5], (cm[:5], cw[:5])                               # CHROM POS ID REF ALT
        assert abs(float(cm[5]) - float(cw[5])) <= 1e-3 * max(1.0, abs(float(cw[5]))), (cm[5], cw[5])
        if abs(float(cm[5]) - round(float(cw[5]), -1)) > 0.01:                  # FILTER steps at multiples of 10
            assert cm[6] == cw[6], (cm[:7], cw[:7])
        im, iw = cm[7].split(";"), cw[7].split(";")
        assert [e.split("=")[0] for e in im] == [e.split("=")[0] for e in iw], (cm[7], cw[7])
        for em, ew in zip(im, iw):
            if em == ew:
                continue
            k = em.split("=")[0]
            assert k in PHRED_INFO, (em, ew)
            _cmp_ints(k, em.split("=")[1], ew.split("=")[1], "phred")
        assert cm[8] == cw[8]
        keys, vm, vw = cm[8].split(":"), cm[9].split(":"), cw[9].split(":")
        assert len(keys) == len(vm) == len(vw), (len(keys), len(vm), len(vw))
        for k, a, b in zip(keys, vm, vw):
            n_tags += 1
            if a == b:
                continue
            if k == "FTS":   # names must agree, the percentages within 1
                pa, pb = a.split("|"), b.split("|")
                assert [e.rsplit("-", 1)[0] for e in pa] == [e.rsplit("-", 1)[0] for e in pb], (a, b)
                for ea, eb in zip(pa, pb):
                    if "-" in ea:
                        assert abs(int(ea.rsplit("-", 1)[1]) - int(eb.rsplit("-", 1)[1])) <= 1, (a, b)
                continue
            _cmp_ints(k, a, b, "phred" if k in PHRED_TAGS else "pct" if k in PCT_TAGS else "exact")
    return n_tags


CASES = {
    "config1_10kb_30x": dict(region_len=10000, depth=30, seed=12345),
    "config2shape_5kb_300x": dict(region_len=5000, depth=300, seed=7),
    "umi_duplex_2kb_400x": dict(region_len=2000, depth=400, seed=11, umi=True),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_record_lines_match_oracle_and_reference_stream(name, oracle_lib, gpu_lib, ref_vcf):
    reads = synth.generate_region(**CASES[name])
    Ro, Rg = run_region(oracle_lib, reads), run_region(gpu_lib, reads)
    rg = Rg.score()
    mine = Rg.vcf_records("chr20", rg).splitlines()
    want = _oracle_lines(oracle_lib, ref_vcf, Ro, "chr20")
    assert len(want) > 0
    n = compare_lines(mine, want)
    print(name, len(mine), "lines", n, "tags compared")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 2, 3, 5, 7, 10])
def test_record_lines_of_weird_reads(seed, oracle_lib, gpu_lib, ref_vcf):
    """InDel-heavy, clipped, multi-allelic reads (the fuzz generator), every symbol written (all_out: REF alleles and both NN symbols too)."""
    from test_gpu_fuzz import run, weird_region
    reads = weird_region(seed, umi=(seed % 3 == 2))
    platform = 2 if seed % 4 == 3 else 1
    try:
        Ro, Rg = run(oracle_lib, reads, platform=platform), run(gpu_lib, reads, platform=platform)
    except region.UvcError:
        pytest.skip("shape refused (covered by test_gpu_fuzz)")
    rg = Rg.score(all_out=True)
    mine = Rg.vcf_records("chrF", rg).splitlines()
    want = _oracle_lines(oracle_lib, ref_vcf, Ro, "chrF", all_out=True)
    assert len(want) > 20 and any(l.split("\t")[4] == "<ADDITIONAL_INDEL_CANDIDATE>" for l in want)
    n_indel = sum(1 for l in want if len(l.split("\t")[3]) != len(l.split("\t")[4]) and not l.split("\t")[4].startswith("<"))
    compare_lines(mine, want)
    print(seed, len(want), "lines,", n_indel, "with an InDel string")


@pytest.mark.gpu
def test_record_lines_of_the_normal_sample(oracle_lib, gpu_lib, ref_vcf):
    """T/N: SOMATIC lines; tbDP / tDP / tAD / t2DP repeat the tumor record, nDP / nAD / n2AD are the normal's (main.hpp:6214-6224)."""
    from test_gpu_parity import tumor_keys_from
    reads = synth.generate_region(region_len=10000, depth=30, seed=12345)
    keys = [k + (7 + i, 3 + i % 5, 2 * i) for i, k in enumerate(tumor_keys_from(run_region(oracle_lib, reads).score(all_out=False)))]
    out = []
    for lib in (oracle_lib, gpu_lib):
        p = region.default_params(lib)
        p.tumor_vcf_is_provided = 1
        out.append(run_region(lib, reads, params=p))
    Ro, Rg = out
    rg = Rg.score(tumor_keys=keys)
    mine = Rg.vcf_records("chr20", rg, tumor_keys=keys).splitlines()
    want = _oracle_lines(oracle_lib, ref_vcf, Ro, "chr20", tumor_keys=keys)
    recs = [l for l in want if l.split("\t")[4] not in ("<NON_REF>", "<ADDITIONAL_INDEL_CANDIDATE>")]
    assert len(recs) > 0 and all(l.split("\t")[7].startswith("SOMATIC;") for l in recs) and len(recs) < len(want)
    compare_lines(mine, want)


@pytest.mark.gpu
def test_tumor_sample_column_is_carried_over(oracle_lib, gpu_lib, ref_vcf):
    """is_tumor_format_retrieved (the default): every line of the normal sample ends with the sample column of the tumor's own line --
    records: bcf1_to_string(tki.bcf1_record) (main.hpp:6269); MGVCF block / ADDITIONAL_INDEL_CANDIDATE lines: the tumor's line of that
    position when it has exactly one, else the fillers of main.cpp:739-757, 784-798."""
    from test_gpu_parity import tumor_keys_from
    reads = synth.generate_region(region_len=10000, depth=30, seed=12345)
    keys = [k + (7 + i, 3 + i % 5, 2 * i) for i, k in enumerate(tumor_keys_from(run_region(oracle_lib, reads).score(all_out=False)))]
    b0 = reads["beg"] - reads["beg"] % 1000
    zero = (0,) * 18
    keys += [(b0 + 1000, 15) + zero, (b0 + 3000, 15) + zero, (b0 + 3000, 15) + zero]            # tumor MGVCF lines: one at +1000, two at +3000, none elsewhere
    keys = sorted(keys, key=lambda k: (k[0], k[1]))
    cols = ["T%d:x,%d" % (i, k[1]) for i, k in enumerate(keys)]
    out = []
    for lib in (oracle_lib, gpu_lib):
        p = region.default_params(lib)
        p.tumor_vcf_is_provided = 1
        out.append(run_region(lib, reads, params=p))
    Ro, Rg = out
    rg = Rg.score(tumor_keys=keys)
    plain = Rg.vcf_records("chr20", rg, tumor_keys=keys).splitlines()
    mine = Rg.vcf_records("chr20", rg, tumor_keys=keys, tumor_sample_columns=cols).splitlines()
    assert len(mine) == len(plain) > 10
    n_rec = n_blk = 0
    for a, b in zip(plain, mine):
        c = b.split("\t")
        assert len(c) == 11 and "\t".join(c[:10]) == a
        if c[4] == "<NON_REF>":
            refpos = int(c[1]) - 1
            want = {b0 + 1000: cols[[k[:2] for k in keys].index((b0 + 1000, 15))], b0 + 3000: ".:.,.:-1"}.get(refpos, ".:.,.:.")
            assert c[10] == want, (refpos, c[10], want); n_blk += 1
        elif c[4] == "<ADDITIONAL_INDEL_CANDIDATE>":
            assert c[10] == ".:.,.:.,."
        else:
            vti = c[9].split(":")[c[8].split(":").index("VTI")].split(",")
            symbol = int(vti[1])
            refpos = int(c[1]) - 1 if symbol <= 5 else int(c[1])
            cand = [cols[i] for i, k in enumerate(keys) if k[:2] == (refpos, symbol)]
            assert c[10] in cand, (c[:5], c[10], cand); n_rec += 1
    assert n_rec > 0 and n_blk >= 5


@pytest.mark.gpu
def test_columns_equal_planes(gpu_lib):
    reads = synth.generate_region(region_len=2000, depth=100, seed=21, umi=True)
    Rg = run_region(gpu_lib, reads)
    pos = np.array([reads["beg"] + 5, reads["beg"] + 777, reads["end"] - 3, reads["beg"] - 10, reads["end"] + 50], dtype=np.int32)
    cols = Rg.fetch_columns(pos)
    assert (cols[3] == 0).all() and (cols[4] == 0).all()   # outside the region
    for g in ["PREP32", "PREP64", "THRES", "SEG32", "SEG64", "VQ", "BQSUM", "FRAG", "FAM", "FAMINFO32", "FAMINFO64", "DUPLEX"]:
        planes = Rg.fetch(g)
        flat = planes.reshape(-1, planes.shape[-1])
        base = Rg.column_base(g)
        for i in range(3):
            assert np.array_equal(cols[i, base:base + flat.shape[0]], flat[:, pos[i] - reads["beg"]].astype(np.int64)), g


@pytest.mark.gpu
def test_vcf_records_need_the_planes(gpu_lib):
    reads = synth.generate_region(region_len=1000, depth=60, seed=4)
    Rg = run_region(gpu_lib, reads)
    rg = Rg.score(release_state=True)
    with pytest.raises(region.UvcError):
        Rg.vcf_records("chr20", rg)


@pytest.mark.gpu
def test_record_lines_of_a_sub_range_and_of_thin_data(gpu_lib):
    """Scoring and writing a sub-range gives exactly the record lines of the full run that lie in it; a region with almost no reads
    writes position-level lines only (or nothing) without failing."""
    reads = synth.generate_region(region_len=6000, depth=80, seed=77)
    Rg = run_region(gpu_lib, reads)
    sym = ("<NON_REF>", "<ADDITIONAL_INDEL_CANDIDATE>")
    full = [l for l in Rg.vcf_records("c", Rg.score()).splitlines() if l.split("\t")[4] not in sym]
    lo, hi = reads["beg"] + 1500, reads["beg"] + 4200
    part_rec = Rg.score(pos_beg=lo, pos_end=hi)
    part = [l for l in Rg.vcf_records("c", part_rec, pos_beg=lo, pos_end=hi).splitlines() if l.split("\t")[4] not in sym]
    def zpos(l):   # the zerobased_pos iteration that wrote the line: SNV lines carry refpos + 1 = zpos, InDel lines refpos = zpos
        return int(l.split("\t")[1])
    want = [l for l in full if lo < zpos(l) < hi or (zpos(l) == lo and len(l.split("\t")[3]) != len(l.split("\t")[4]))]
    assert len(part) >= 3 and set(part) <= set(full)
    assert [l for l in part if lo + 1 < zpos(l) < hi - 1] == [l for l in want if lo + 1 < zpos(l) < hi - 1]
    thin = synth.generate_region(region_len=2500, depth=2, seed=5)
    Rt = run_region(gpu_lib, thin)
    rec = Rt.score()
    lines = Rt.vcf_records("c", rec).splitlines()
    assert all(len(l.split("\t")) == 10 for l in lines)
    assert sum(1 for l in lines if l.split("\t")[4] not in sym) == int(rec["keep"].sum())


HAP_CASES = {
    "dense_snvs_80x": dict(region_len=3000, depth=80, seed=5, snv_every=90, somatic_every=400, indel_every=350),
    "dense_umi_300x": dict(region_len=2500, depth=300, seed=6, snv_every=90, somatic_every=400, indel_every=300, umi=True),
    "dense_indels_120x": dict(region_len=2000, depth=120, seed=8, snv_every=150, somatic_every=0, indel_every=110),
}


def test_haplotype_links_on_the_oracle(oracle_lib):
    """fragments / families that carry several high-quality mutations are linked (updateHapMap, main.hpp:3596-3663)."""
    reads = synth.generate_region(**HAP_CASES["dense_umi_300x"])
    R = run_region(oracle_lib, reads)
    bq, fq, f2q = R.hap_links()
    assert len(bq) > 20 and len(fq) > 10 and len(f2q) > 5
    for links in (bq, fq, f2q):
        for muts, fr, other in links:
            assert len(muts) >= 2 and list(muts) == sorted(muts, key=lambda ps: (ps[0], 0 if ps[1] >= 6 else 1)) or len({p for p, _ in muts}) < len(muts)
            assert fr[0] + fr[1] >= 1 + len(muts)                                  # phasing_haplotype_min_ad + size
        assert sum(1 for l in links if l[2] != (-1, -1)) <= 3                      # phasing_haplotype_max_detail_cnt
    # families are fewer than fragments: the de-duplicated counts do not exceed the raw ones for a link both have
    raw = {l[0]: l[1] for l in bq}
    assert any(l[0] in raw and sum(l[1]) <= sum(raw[l[0]]) for l in fq)
    R.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(HAP_CASES))
def test_haplotype_links_match_the_oracle(name, oracle_lib, gpu_lib, ref_vcf):
    """bHap / cHap / c2Hap: the link vectors (uvcgpu_region_hap_links) equal the oracle's, and so do the strings of every written record."""
    reads = synth.generate_region(**HAP_CASES[name])
    Ro, Rg = run_region(oracle_lib, reads), run_region(gpu_lib, reads)
    lo, lg = Ro.hap_links(), Rg.hap_links()
    assert sum(len(l) for l in lo) > 10
    for w in range(3):
        assert lo[w] == lg[w], (w, next((a, b) for a, b in zip(lo[w] + [None], lg[w] + [None]) if a != b))
    rg = Rg.score()
    mine = Rg.vcf_records("chr20", rg).splitlines()
    want = _oracle_lines(oracle_lib, ref_vcf, Ro, "chr20")
    compare_lines(mine, want)
    n_linked = 0
    for l in mine:
        c = l.split("\t")
        if c[4].startswith("<N") or c[4].startswith("<ADD"):
            continue
        f = dict(zip(c[8].split(":"), c[9].split(":")))
        n_linked += f["bHap"] != "."
    assert n_linked >= 5
    Ro.close(); Rg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["config1_10kb_30x", "dense_indels_120x"])
def test_germline_lines(name, oracle_lib, gpu_lib, ref_vcf):
    """OUTVAR_GERMLINE (--outvar-flag bit 1): the GERMLINE lines of output_germline (main.hpp:5612-5775) -- genotype, GQ, the REF / ALT text
    of one or two alternative alleles (two InDel alleles share the longer REF), CDP1, the allele depths, GL4 and GST -- in front of the
    records of their position."""
    kw = dict(CASES, **HAP_CASES)[name]
    reads = synth.generate_region(**kw)
    out = []
    for lib in (oracle_lib, gpu_lib):
        p = region.default_params(lib)
        p.outvar_flag = 63
        out.append(run_region(lib, reads, params=p))
    Ro, Rg = out
    rg = Rg.score()
    mine = Rg.vcf_records("chr20", rg).splitlines()
    want = _oracle_lines(oracle_lib, ref_vcf, Ro, "chr20")
    germ = [l.split("\t") for l in want if l.split("\t")[7] == "GERMLINE"]
    assert len(germ) >= 5 and {c[9].split(":")[0] for c in germ} >= {"0/1"} and all(c[8] == "GT:GQ:HQ:FT:CDP1:cDP1:GL4:GST:note" for c in germ)
    if name == "dense_indels_120x":
        assert any(len(c[3]) != len(c[4].split(",")[0]) for c in germ)       # an InDel genotype with its REF / ALT text
    compare_lines(mine, want)
    Ro.close(); Rg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,outvar", [("config1_10kb_30x", None), ("config2shape_5kb_300x", None), ("dense_indels_120x", 63), ("umi_duplex_2kb_400x", 63)])
def test_kept_only_records_are_the_groups_the_writer_reads(name, outvar, gpu_lib):
    """UvcScoreRequest::kept_only: the (position, symbol type) groups with a written record or a GERMLINE line, whole, in order, with
    germ_ref / germ_alt1 / germ_alt2 re-based -- and the record writer produces the same text from them as from all records."""
    reads = synth.generate_region(**dict(CASES, **HAP_CASES)[name])
    p = region.default_params(gpu_lib)
    if outvar is not None:
        p.outvar_flag = outvar
    R = run_region(gpu_lib, reads, params=p)
    full = R.score()
    kept = R.score(kept_only=True, capacity=64)                  # the caller's buffer only has to hold the kept groups (the mirror grows it on ENOMEM)
    n = len(full["refpos"])
    is_base = full["symbol"] <= 5
    head = np.ones(n, bool)
    head[1:] = (full["refpos"][1:] != full["refpos"][:-1]) | (is_base[1:] != is_base[:-1])
    gid = np.cumsum(head) - 1
    written = ((full["keep"] == 1) & (full["out"] == 1)) | (full["germ_emit"] == 1)
    group_kept = np.zeros(gid.max() + 1, bool)
    group_kept[gid[written]] = True
    sel = np.nonzero(group_kept[gid])[0]
    assert 0 < len(sel) < n and len(kept["refpos"]) == len(sel)
    new_index = -np.ones(n, np.int64)
    new_index[sel] = np.arange(len(sel))
    for f in full:
        want = full[f][sel]
        if f in ("germ_ref", "germ_alt1", "germ_alt2"):
            want = np.where(want >= 0, new_index[np.maximum(want, 0)], -1)
            assert (want[full[f][sel] >= 0] >= 0).all()          # a genotype's records belong to its own group
        assert np.array_equal(kept[f], want), f
    assert R.vcf_records("chr20", kept) == R.vcf_records("chr20", full)
    # all-out: everything is written, nothing to drop but LINK_NN-only groups
    full_a, kept_a = R.score(all_out=True), R.score(all_out=True, kept_only=True)
    assert R.vcf_records("chr20", kept_a) == R.vcf_records("chr20", full_a) and len(kept_a["refpos"]) <= len(full_a["refpos"])
    R.close()


@pytest.mark.gpu
def test_symbolic_indel_alleles_are_written(oracle_lib, gpu_lib, ref_vcf):
    """-A with a low --vqual writes the records of InDel symbols nobody carries: ALT and gapSa are the symbolic allele then (the fallback of
    indel_get_majority, main.hpp:5417-5424).  Found by scripts/gpu_soak_vcf.py: gapSa used to be empty for them."""
    from test_gpu_fuzz import weird_region
    reads = weird_region(118, n_frag=60, ref_len=300)
    R = []
    for lib in (oracle_lib, gpu_lib):
        P = region.default_params(lib, platform=2)
        P.should_output_all_germline, P.vqual = 1, 5.0
        r = region.Region(lib, P, reads["tid"], reads["beg"], reads["end"], reads["refseq"]); r.set_reads(reads); r.accumulate(); R.append(r)
    mine = R[1].vcf_records("chrS", R[1].score(all_out=True)).splitlines()
    want = _oracle_lines(oracle_lib, ref_vcf, R[0], "chrS", all_out=True)
    sym = [l for l in want if l.split("\t")[4] in ("<LI1>", "<LI2>", "<LI3P>", "<LD1>", "<LD2>", "<LD3P>")]
    assert len(sym) > 50 and all(("," + l.split("\t")[4]) in l.split("\t")[9] for l in sym)
    compare_lines(mine, want)


@pytest.mark.gpu
def test_rescued_indel_strings_in_record_and_germline_lines(oracle_lib, gpu_lib, ref_vcf):
    """Normal sample of a T/N pair with GERMLINE lines on and the tumor records' "REF\\tALT" strings handed over: a rescued InDel record takes its
    string from the tumor record (main.cpp:867-880) in its own line AND in the GERMLINE line that names it (LAST(fmt.gapSa), main.hpp:5625-5700).
    Found by scripts/gpu_soak_vcf.py: the GERMLINE line used to carry the symbolic allele."""
    from test_gpu_fuzz import weird_region
    from test_gpu_parity import tumor_keys_from
    reads = weird_region(1019, n_frag=260, ref_len=700, umi=False)
    P0 = region.default_params(oracle_lib); P0.outvar_flag = 63
    Rt = region.Region(oracle_lib, P0, reads["tid"], reads["beg"], reads["end"], reads["refseq"]); Rt.set_reads(reads); Rt.accumulate()
    keys = [k + (7 + i, 3 + i % 5, 2 * i) for i, k in enumerate(tumor_keys_from(Rt.score(all_out=False), every=3))]
    ras = [("A" + "C" * max(int(k[7]), 1) + "\tA") if 7 <= k[1] <= 9 else ("A\tA" + "G" * max(int(k[7]), 1)) if 10 <= k[1] <= 12 else "A\tC" for k in keys]
    R = []
    for lib in (oracle_lib, gpu_lib):
        P = region.default_params(lib); P.tumor_vcf_is_provided, P.outvar_flag = 1, 63
        r = region.Region(lib, P, reads["tid"], reads["beg"], reads["end"], reads["refseq"]); r.set_reads(reads); r.accumulate(); R.append(r)
    mine = R[1].vcf_records("chrS", R[1].score(tumor_keys=keys), tumor_keys=keys, tumor_ref_alt=ras).splitlines()
    want = _oracle_lines(oracle_lib, ref_vcf, R[0], "chrS", tumor_keys=keys, tumor_ref_alt=ras)
    germ = [l.split("\t") for l in want if "GERMLINE" in l.split("\t")[7]]
    assert any(len(c[3]) != len(c[4]) and not c[4].startswith("<") and set(c[3][1:] + c[4][1:]) <= set("CG") for c in germ), "no GERMLINE line with a rescued InDel string in this case"
    compare_lines(mine, want)
