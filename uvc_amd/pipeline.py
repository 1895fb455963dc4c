"""BAM + FASTA -> scored records for one region: the chain process_batch runs (main.cpp:458-1193), on the C ABIs of this package.

    fetch reads (uvcio, replaces sam_itr_queryi / faidx_fetch_seq)        grouping.cpp:664, main.cpp:529-531
    family assignment (uvcgpu_group_families)                              grouping.cpp:608-997
    region bounds + reference (+-100 bp STR halo)                          main.cpp:523-552
    apply_bq_err_correction3 / updateByRegion3Aln / scoring + calling      main.cpp:567-1168

`lib` is the HIP library (`region.gpu_lib()`); tests also run the chain on the oracle library to compare.  Output: the reference's VCF
record lines (uvcgpu_vcf_header / uvcgpu_region_vcf_records, block-gzipped by libuvcio's BGZF writer) or a short tab-separated table.

    python -m uvc_amd.pipeline in.bam ref.fa chr20:1000000-1100000 --vcf out.vcf.gz --sample TUMOR
    python -m uvc_amd.pipeline in.bam ref.fa chr20 > out.tsv                       (the whole contig in 1 Mb tiles)
"""
import sys

import numpy as np

from . import _ffi, group, io as uio, region

MAX_INSERT_SIZE = 2000   # common.hpp:64
MAX_STR_N_BASES = 100    # common.hpp:63
SYMBOL_DESC = ["A", "C", "G", "T", "N", "*", "<LR>", "<LD3P>", "<LD2>", "<LD1>", "<LI3P>", "<LI2>", "<LI1>", "*"]
FILTERS = ["Q10", "Q20", "Q30", "Q40", "Q50", "Q60", "PASS"]


def call_region(lib, bam, fasta, chrom, beg, end, params=None, group_params=None, molecule_tag=0, disable_duplex=0, correct_bq=True, all_out=False, keep_handle=False, reuse=None, vcf=False,
                continues=False, has_next=False, region_beg=None, tumor_vcf=None, umi_struct=None):
    """Scores [beg, end) of `chrom`.  Returns None when no read passes the filters (process_batch returns -1, main.cpp:520-523), else a
    dict: records (field -> int32 array), alleles (InDel allele rows), score range, region handle (if keep_handle).
    Tiles of one stretch: the reference scores zerobased_pos rpos_beg .. rpos_end inclusive and skips the BASE sub-position of the first
    (main.cpp:608, 643), so two adjacent regions both write the LINK records of their shared end point.  Here every zerobased_pos has one
    owner: `has_next` (an adjacent tile [end, ..) follows) leaves zerobased_pos `end` to that tile, and `continues` (an adjacent tile
    [.., beg) is in front) makes this one score `beg` completely, BASE sub-position included (UvcScoreRequest::base_at_pos_beg) -- a run of
    tiles writes the records of one uncut region.  `region_beg`: begin of the BED line / contig range the run belongs to (incluBegPosition
    of main.cpp:655-656, default `beg`).  `tumor_vcf`: normal sample of a T/N pair -- the tumor pass's VCF as `uvc_amd.io.TumorVcf` (its records
    of this region become UvcScoreRequest::tumor_keys; `params.tumor_vcf_is_provided` must be set).  `umi_struct`: the in-read UMI pattern
    (the reference's environment variable ONE_STEP_UMI_STRUCT).
    `reuse`: a dict the caller keeps between calls; the region handle lives in it and is reset for every new region instead of being
    created and destroyed (its device buffers survive while the regions do not grow)."""
    import os, time
    timing = bool(os.environ.get("UVC_PIPELINE_TIMING"))
    laps = [("start", time.perf_counter())]
    lap = (lambda name: laps.append((name, time.perf_counter()))) if timing else (lambda name: None)
    tid = bam.tid(chrom)
    tlen = bam.refs[tid][1]
    cols = bam.fetch(tid, max(0, beg - MAX_INSERT_SIZE), end + MAX_INSERT_SIZE)
    lap("fetch")
    n = cols["n_alns"]
    if n == 0:
        return None
    kind, h = group._digest_batch(lib, cols["qnames"], molecule_tag, disable_duplex)
    if umi_struct:                                                                     # ONE_STEP_UMI_STRUCT: the in-read UMI of single-end reads (bam2umihash, grouping.cpp:787-792)
        kind = np.ascontiguousarray(kind)
        group.umi_in_read_batch(lib, umi_struct, cols, kind)
    gp = group_params if group_params is not None else group.default_params(lib, beg, end, platform=(params.inferred_sequencing_platform if params is not None else 1))
    gp.fetch_tbeg, gp.fetch_tend = beg, end
    g = group.group_families(lib, gp, dict(tid=cols["tid"], pos=cols["pos"], endpos=cols["endpos"], mtid=cols["mtid"], mpos=cols["mpos"], isize=cols["isize"], flag=cols["flag"], mapq=cols["mapq"],
                                           qname_hash31=h[0], qname_hash17=h[1], umi_hash31=h[2], umi_hash17=h[3], umi_kind=kind))
    lap("digest+group")
    if g["n_kept"] == 0:
        return None
    o = g["order"]
    bam_beg, bam_end = g["ext_beg"], g["ext_end"]                                      # bam_inclu_beg_pos / bam_exclu_end_pos
    rpos_beg, rpos_end = max(beg, bam_beg), min(end, bam_end)                          # main.cpp:523-524
    ext_beg = max(0, max(min(beg, bam_beg) - MAX_STR_N_BASES, 0))                      # main.cpp:525
    ext_end = min(tlen, max(end, bam_end) + MAX_STR_N_BASES)                           # main.cpp:526
    refseq = fasta.fetch(chrom, ext_beg, ext_end)
    reads = dict(n_reads=int(g["n_kept"]), pos=cols["pos"][o], mpos=cols["mpos"][o], isize=g["isize_norm"][o], flag=cols["flag"][o], mapq=cols["mapq"][o], nm=cols["nm"][o],
                 l_qseq=cols["l_qseq"][o], seq_off=cols["seq_off"][o], cigar_off=cols["cigar_off"][o], n_cigar=cols["n_cigar"][o],
                 frag_id=g["frag_id"], fam_id=g["fam_id"], fam_strand=g["fam_strand"], n_fams=int(g["n_fams"]), fam_dflag=g["fam_dflag"],
                 bases=cols["bases"], quals=cols["quals"], cigars=cols["cigars"])
    p = params if params is not None else region.default_params(lib)
    lap("columns+refseq")
    if reuse is not None and reuse.get("region") is not None:
        R = reuse["region"]; R.reset(tid, ext_beg, ext_end, refseq)
    else:
        R = region.Region(lib, p, tid, ext_beg, ext_end, refseq)
        if reuse is not None:
            reuse["region"] = R
    lap("region_create")
    R.set_reads(reads)
    lap("set_reads")
    if correct_bq:
        R.correct_bq()
    R.accumulate()
    is_amplicon = (g["n_amplicon"] * 2 > g["n_kept"])                                  # !is_by_capture, main.cpp:507-508
    last_excl = min(end, bam_end + 1) if has_next else min(rpos_end + 1, ext_end)      # zerobased_pos `end` belongs to the tile behind, if there is one
    skw = dict(pos_beg=rpos_beg, pos_end=last_excl, base_at_pos_beg=bool(continues and rpos_beg == beg and beg > ext_beg), region_beg=(beg if region_beg is None else region_beg))
    score_range = (skw["pos_beg"], skw["pos_end"])
    if score_range[1] <= score_range[0]:
        return None
    tk, tcols, tras = None, None, None
    if tumor_vcf is not None:                                                          # the tumor records inside this region (tkis_beg .. tkis_end, main.cpp:532-533)
        tk, tcols = tumor_vcf.fetch(tid, ext_beg, ext_end)
        tras = tumor_vcf.last_ref_alt
    rec = R.score(all_out=all_out, is_amplicon=bool(is_amplicon), tumor_keys=tk, **skw)
    lap("bq+accumulate+score")
    out = dict(records=rec, alleles=R.indel_alleles(), rpos=(rpos_beg, rpos_end), ext=(ext_beg, ext_end), n_reads=int(g["n_kept"]), n_fams=int(g["n_fams"]), chrom=chrom, refseq=refseq, score_range=score_range)
    if vcf:
        out["vcf"] = R.vcf_records(chrom, rec, tumor_keys=tk, tumor_sample_columns=tcols, tumor_ref_alt=tras, **skw)          # the record lines of append_vcf_record (uvcgpu_region_vcf_records), before the handle moves on
    if keep_handle:
        out["region"] = R
    elif reuse is None:
        R.close()
    lap("alleles+close")
    if timing:
        sys.stderr.write("[pipeline] " + ", ".join("%s %.1f ms" % (b[0], 1e3 * (b[1] - a[1])) for a, b in zip(laps, laps[1:])) + "\n")
    return out


def contig_tiles(beg, end, tile):
    """The fixed tiles of [beg, end) with their ownership flags (see call_region): list of dicts beg, end, continues, has_next, region_beg."""
    return [dict(beg=b, end=min(b + tile, end), continues=(b != beg), has_next=(b + tile < end), region_beg=beg) for b in range(beg, end, tile)]


def call_contig(lib, bam, fasta, chrom, beg=0, end=None, tile=1_000_000, workers=1, device=None, only=None, **kw):
    """Tiles [beg, end) of a contig (default: all of it) and yields the result of every tile that has reads, in order.  The reference cuts
    its regions by read and position counts (SamIter, grouping.cpp:28-67, 157-314) and scores the shared end point of two adjacent
    regions in both; here fixed tiles of the size the state slab is laid out for are used and every zerobased_pos has one owner
    (`continues` of call_region), so that the tiles of [beg, end) together give the records of one uncut region [beg, end).
    workers > 1: that many tiles in flight on host threads, each with its own file handles and region handle (the library calls release
    the GIL; the reference runs process_batch on `nthreads` OpenMP threads the same way, main.cpp:1478-1520).  `bam` / `fasta` may be
    paths or open handles; with workers > 1 they must be paths.  `device`: the GPU the worker threads bind to (uvcgpu_init is per host
    thread); None = the device the calling thread initialised the library with is NOT inherited, so pass it when it is not 0.
    `only`: indices into the tile list -- the shard of a multi-process run (uvc_amd.shard.plan_contiguous): ownership is a property of the
    list, not of who runs a tile, so the shards' outputs concatenate to the single-process output."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    paths = (bam, fasta) if isinstance(bam, str) else None
    if workers > 1 and paths is None:
        raise ValueError("workers > 1 needs the BAM / FASTA paths: every worker opens its own handles")
    local = threading.local()

    def handles():
        if paths is None:
            return bam, fasta
        if not hasattr(local, "h"):
            local.h = (uio.Bam(paths[0]), uio.Fasta(paths[1]))
        return local.h
    b0 = handles()[0]
    tlen = b0.refs[b0.tid(chrom)][1]
    end = tlen if end is None else min(end, tlen)
    starts = contig_tiles(beg, end, tile)
    if only is not None:
        starts = [starts[i] for i in only]

    def one(t):
        hb, hf = handles()
        if not hasattr(local, "reuse"):
            local.reuse = {}                                   # one region handle per worker, reset from tile to tile
            if device is not None and lib.prefix == "uvcgpu_" and lib.dll.uvcgpu_init(int(device)) != 0:   # hipSetDevice is per thread
                raise RuntimeError(lib.last_error())
        return call_region(lib, hb, hf, chrom, t["beg"], t["end"], reuse=local.reuse, continues=t["continues"], has_next=t["has_next"], region_beg=t["region_beg"], **kw)
    if workers <= 1:
        for b in starts:
            res = one(b)
            if res is not None:
                yield res
        return
    with ThreadPoolExecutor(max_workers=workers) as ex:
        for res in ex.map(one, starts):
            if res is not None:
                yield res


def write_tsv(res, fh, kept_only=True, header=True):
    rec, rows = res["records"], res["alleles"]
    ext_beg = res["ext"][0]
    if header:
        fh.write("#CHROM\tPOS\tREF\tALT\tQUAL\tFILTER\tSYMBOL\tDP\tAD\tbDP\tbAD\tcVQ1\tcVQ2\tTLODQ\tNLODQ\tGT_IDX\tGQ\n")
    q = rec["QUAL"].view(np.float32)
    for i in range(len(rec["refpos"])):
        if kept_only and not rec["keep"][i]:
            continue
        sym, refpos = int(rec["symbol"][i]), int(rec["refpos"][i])
        x = refpos - ext_beg
        if sym <= 5:
            pos, ref, alt = refpos + 1, res["refseq"][x], SYMBOL_DESC[sym]
        else:                                                                          # append_vcf_record, main.hpp:6066-6090
            pos, ref = refpos, (res["refseq"][x - 1] if x > 0 else "n")
            alt = ref
            row = rows[rec["gapSa"][i]] if rec["gapSa"][i] >= 0 else None
            if row is None: alt = SYMBOL_DESC[sym]
            elif row["seq"] is not None: alt = ref + row["seq"]
            else: ref = ref + res["refseq"][x:x + row["len"]]
        fh.write("%s\t%d\t%s\t%s\t%.6f\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n" % (
            res["chrom"], pos, ref, alt, q[i], FILTERS[int(rec["FILTER"][i])], SYMBOL_DESC[sym], rec["DP"][i], rec["AD"][i], rec["bDP"][i], rec["bAD"][i],
            rec["cVQ1"][i], rec["cVQ2"][i], rec["TLODQ"][i], rec["NLODQ"][i], rec["germ_GT"][i], rec["germ_GQ"][i]))


def write_vcf(lib, bam, fasta, chrom, beg, end, path, sample="SAMPLE", params=None, tumor_vcf=None, **kw):
    """BAM + FASTA -> VCF: the header (uvcgpu_vcf_header) and the record lines of every tile, through the BGZF writer when `path` ends in
    .gz (what the reference does with bgzf_write, main.cpp:1196-1215, 1571-1583), else as plain text ("-" = stdout).  Returns the number
    of record lines."""
    p = params if params is not None else region.default_params(lib)
    b = bam if not isinstance(bam, str) else uio.Bam(bam)
    header = region.vcf_header(lib, p, sample, [(name, ln) for name, ln in b.refs], tumor_sample=(tumor_vcf.sample if tumor_vcf is not None else None))
    if tumor_vcf is not None:
        kw["tumor_vcf"] = tumor_vcf
    sink = uio.BgzfWriter(path) if path.endswith(".gz") else (sys.stdout if path == "-" else open(path, "w"))
    n = 0
    try:
        sink.write(header)
        for res in call_contig(lib, bam, fasta, chrom, beg, end, vcf=True, params=params, **kw):
            sink.write(res["vcf"]); n += res["vcf"].count("\n")
    finally:
        if sink is not sys.stdout:
            sink.close()
    return n


def main(argv):
    args, vcf_path, sample = [], None, "SAMPLE"
    it = iter(argv[1:])
    for a in it:
        if a == "--vcf": vcf_path = next(it, None)
        elif a == "--sample": sample = next(it, "SAMPLE")
        else: args.append(a)
    argv = argv[:1] + args
    if len(argv) != 4:
        sys.stderr.write(__doc__); return 2
    chrom, _, rng = argv[3].partition(":")
    beg, end = ((int(v.replace(",", "")) for v in rng.split("-")) if rng else (0, None))
    lib = region.gpu_lib()
    if lib.dll.uvcgpu_init(0) != 0:
        raise RuntimeError(lib.last_error())
    bam, fasta = uio.Bam(argv[1]), uio.Fasta(argv[2])
    if vcf_path:
        n = write_vcf(lib, bam, fasta, chrom, beg, end, vcf_path, sample=sample)
        sys.stderr.write("%d records written to %s\n" % (n, vcf_path))
        return 0
    n = 0
    for res in call_contig(lib, bam, fasta, chrom, beg, end):
        write_tsv(res, sys.stdout, header=(n == 0)); n += 1
    if n == 0:
        sys.stderr.write("no reads pass the filters in %s\n" % argv[3]); return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
