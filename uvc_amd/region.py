"""Host-side mirror of the reference's per-region processing surface (process_batch, main.cpp:458-1193).

    region = Region(lib, params, tid, beg, end, refseq)   # Symbol2CountCoverageSet(tid, beg, end+1), main.cpp:569
    region.set_reads(reads)                               # the alns3 equivalent (UvcReadSoA)
    region.accumulate()                                   # updateByRegion3Aln, main.hpp:3665
    records = region.score(all_out=False)                 # BcfFormat_symbol* call group, main.cpp:648-967
    planes  = region.fetch("SEG32")                       # raw per-position state

`lib` is a `_ffi.Lib`.  The product path binds uvc_amd/csrc/libuvcgpu.so (hand-written HIP for
gfx950); there is NO CPU fallback here -- `gpu_lib()` raises when the extension is missing.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi


class UvcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("uvc error %d: %s" % (code, msg))
        self.code = code


_gpu_lib = None


def gpu_lib():
    """Loads libuvcgpu.so and initialises device 0.  Fails loudly when the extension or the GPU is missing."""
    global _gpu_lib
    if _gpu_lib is None:
        import os
        path = _ffi.gpu_library_path()
        if not os.path.exists(path):
            raise ImportError("HIP extension %s is not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        lib = _ffi.Lib(path, "uvcgpu_")
        lib.dll.uvcgpu_init.restype, lib.dll.uvcgpu_init.argtypes = C.c_int, [C.c_int]
        _gpu_lib = lib
    return _gpu_lib


def default_params(lib, platform=1, central_readlen=150, max_mapq=60):
    """Reference defaults + the platform deltas of CommandLineArgs::selfUpdateByPlatform (CmdLineArgs.cpp:113-134)."""
    p = _ffi.UvcParams()
    lib.call("params_default", C.byref(p))
    apply_platform(p, platform, central_readlen, max_mapq)
    return p


def apply_platform(p, platform, central_readlen, max_mapq):
    # CmdLineArgs.cpp:13-15, 113-134
    p.inferred_sequencing_platform = platform
    if p.central_readlen == 0:
        p.central_readlen = central_readlen
    p.inferred_maxMQ = max(p.inferred_maxMQ, max_mapq)
    if platform == 2:    # IonTorrent
        p.bq_phred_added_misma += 8
        for name, dec in (("fam_thres_highBQ_snv", 30), ("fam_thres_highBQ_indel", 30), ("bias_thres_PFBQ1", 30), ("bias_thres_PFBQ2", 30), ("bias_thres_highBQ", 13)):
            v = getattr(p, name)
            setattr(p, name, v - min(v, dec))
    elif platform == 1:  # Illumina
        p.syserr_minABQ_pcr_snv += 200
        p.syserr_minABQ_pcr_indel += 100
        p.syserr_minABQ_cap_snv += 200
        p.syserr_minABQ_cap_indel += 100


_READ_FIELDS = [("pos", np.int32), ("mpos", np.int32), ("isize", np.int32), ("flag", np.uint16), ("mapq", np.uint8), ("nm", np.int32),
                ("l_qseq", np.int32), ("seq_off", np.int64), ("cigar_off", np.int64), ("n_cigar", np.int32),
                ("frag_id", np.int32), ("fam_id", np.int32), ("fam_strand", np.uint8)]


_NT16 = np.array([1, 2, 4, 8, 15], dtype=np.uint8)   # A C G T N as BAM's 4-bit codes (seq_nt16_str "=ACMGRSVTWYHKDBN")


def compact_form(reads):
    """The same reads in the compact input form of UvcReadSoA: `bases4` = the bases as a BAM record holds them (two 4-bit codes per byte, high
    nibble first, every read on a byte boundary, reads back to back) instead of one byte per base, and no seq_off / cigar_off columns (the
    library derives them on the device).  Needs the reads back to back in read order, which is how every packer here lays them out."""
    n = int(reads["n_reads"])
    lq = np.asarray(reads["l_qseq"], dtype=np.int64)
    nc = np.asarray(reads["n_cigar"], dtype=np.int64)
    so = np.concatenate(([0], np.cumsum(lq)))[:n]
    co = np.concatenate(([0], np.cumsum(nc)))[:n]
    if not (np.array_equal(so, np.asarray(reads["seq_off"])) and np.array_equal(co, np.asarray(reads["cigar_off"]))):
        raise ValueError("compact_form needs reads that lie back to back in read order")
    codes = _NT16[np.minimum(np.asarray(reads["bases"], dtype=np.uint8), 4)]
    nb = (lq + 1) // 2
    bo = np.concatenate(([0], np.cumsum(nb)))
    if n and not (lq % 2).any():                                     # every read starts on an even base index: two neighbours per byte
        out4 = (codes[0::2] << 4) | codes[1::2]
    else:
        out4 = np.zeros(int(bo[-1]), dtype=np.uint8)
        if n:
            k = np.arange(int(lq.sum())) - np.repeat(so, lq)            # index of every base inside its read
            byte = np.repeat(bo[:n], lq) + k // 2
            hi = (k % 2 == 0)
            out4[byte[hi]] = codes[hi] << 4                              # each byte has one high and at most one low nibble
            out4[byte[~hi]] |= codes[~hi]
    c = {k_: v for k_, v in reads.items() if k_ not in ("bases", "seq_off", "cigar_off")}
    c["bases4"] = out4
    return c


def _fill_soa(reads, put):
    """UvcReadSoA from a dict of columns; `put(array, dtype) -> address` places a column (host or device).  Columns a compact dict leaves
    out (seq_off, cigar_off, bases) stay NULL."""
    soa = _ffi.UvcReadSoA()
    soa.struct_size = C.sizeof(_ffi.UvcReadSoA)
    soa.n_reads = int(reads["n_reads"])
    for name, dt in _READ_FIELDS:
        if name in ("seq_off", "cigar_off") and reads.get(name) is None:
            continue
        a = np.ascontiguousarray(reads[name], dtype=dt)
        assert a.shape == (soa.n_reads,), name
        setattr(soa, name, put(a, dt))
    soa.n_bases = int(np.asarray(reads["quals"]).size)
    soa.n_cigar_ops = int(np.asarray(reads["cigars"]).size)
    soa.quals = put(reads["quals"], np.uint8)
    soa.cigars = put(reads["cigars"], np.uint32)
    if reads.get("bases") is not None:
        soa.bases = put(reads["bases"], np.uint8)
    else:
        soa.bases4 = put(reads["bases4"], np.uint8)
        soa.n_bases4_bytes = int(np.asarray(reads["bases4"]).size)
    soa.n_fams = int(reads["n_fams"])
    soa.fam_dflag = put(reads["fam_dflag"], np.uint8)
    return soa


def pack_reads(reads):
    """dict of numpy arrays -> (UvcReadSoA, keepalive list)."""
    keep = []

    def put(a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data
    return _fill_soa(reads, put), keep


def device_reads(reads, device):
    """The UvcReadSoA columns of `reads` as torch tensors on `device` (torch is only the owner of the HBM here) -> (UvcReadSoA of device
    pointers, keepalive) for Region.set_reads_device: what a caller has whose decoder writes straight to the GPU.
    torch ships its own copy of the HIP runtime: initialise torch.cuda BEFORE libuvcgpu.so touches the device (bench.py does), the other
    order leaves torch without a GPU.  Without torch, allocate through the runtime the library links (tests/test_gpu_device_reads.py)."""
    import torch
    keep = []

    def put(a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        # torch has no uint16 / uint32 tensors everywhere: ship the bytes
        t = torch.from_numpy(a.view(np.uint8).reshape(-1)).to(device)
        keep.append(t)
        return t.data_ptr() if t.numel() else 0
    soa = _fill_soa(reads, put)
    torch.cuda.synchronize(device)
    return soa, keep


def vcf_format_keys(lib, tier2=False):
    fn = getattr(lib.dll, "uvcgpu_vcf_format_keys")
    fn.restype, fn.argtypes = C.c_char_p, [C.c_int32]
    return fn(int(tier2)).decode()


def vcf_header(lib, params, sample, contigs, tumor_sample=None):
    """##-lines and the #CHROM line (uvcgpu_vcf_header); contigs = [(name, length), ...]; tumor_sample: the second sample column of a
    normal-sample VCF that carries the tumor's FORMAT over (is_tumor_format_retrieved)."""
    fn = getattr(lib.dll, "uvcgpu_vcf_header")
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(_ffi.UvcParams), C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    names = (C.c_char_p * max(1, len(contigs)))(*[c[0].encode() for c in contigs])
    lens = (C.c_int64 * max(1, len(contigs)))(*[int(c[1]) for c in contigs])
    ln = C.c_int64(0)
    ts = tumor_sample.encode() if tumor_sample else None
    fn(C.byref(params), sample.encode(), ts, names, lens, len(contigs), None, 0, C.byref(ln))
    dst = C.create_string_buffer(max(1, ln.value))
    rc = fn(C.byref(params), sample.encode(), ts, names, lens, len(contigs), dst, ln.value, C.byref(ln))
    if rc:
        raise UvcError(rc, lib.last_error())
    return dst.raw[:ln.value].decode()


class Region:
    def __init__(self, lib, params, tid, beg, end, refseq):
        self.lib, self.tid, self.beg, self.end = lib, tid, beg, end
        self.npos = end - beg + 1
        self.h = C.c_void_p()
        self._params = params
        ref = refseq.encode() if isinstance(refseq, str) else bytes(refseq)
        if len(ref) != end - beg:
            raise ValueError("refseq must cover [beg, end)")
        self._check(lib.call("create", C.byref(self.h), C.byref(params), tid, beg, end, ref))

    def _check(self, rc):
        if rc != 0:
            raise UvcError(rc, self.lib.last_error())

    def reset(self, tid, beg, end, refseq):
        """Re-binds the handle to another region (uvcgpu_region_reset): streams and, if the region is not longer, device buffers are kept.
        Libraries without that entry point (the test oracle) get a fresh handle instead."""
        ref = refseq.encode() if isinstance(refseq, str) else bytes(refseq)
        if len(ref) != end - beg:
            raise ValueError("refseq must cover [beg, end)")
        fn = getattr(self.lib.dll, self.lib.prefix + "region_reset", None)
        if fn is None:
            self.close()
            self._check(self.lib.call("create", C.byref(self.h), C.byref(self._params), tid, beg, end, ref))
        else:
            fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_char_p]
            self._check(fn(self.h, tid, beg, end, ref))
        self.tid, self.beg, self.end, self.npos = tid, beg, end, end - beg + 1   # (the records buffer stays: score() replaces it when the capacity asked for changes)

    def set_reads(self, reads):
        soa, keep = pack_reads(reads)
        self._check(self.lib.call("set_reads", self.h, C.byref(soa)))

    def set_reads_device(self, dev):
        """uvcgpu_region_set_reads_device: `dev` = (UvcReadSoA of device pointers, keepalive) as `device_reads` makes it.  The arrays must
        outlive the reads of the handle."""
        soa, keep = dev
        fn = getattr(self.lib.dll, self.lib.prefix + "region_set_reads_device")
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(_ffi.UvcReadSoA)]
        self._dev_reads = keep
        self._check(fn(self.h, C.byref(soa)))

    def accumulate(self):
        self._check(self.lib.call("accumulate", self.h))
        if os.environ.get("UVCGPU_CHECK_PRESENCE") and hasattr(self.lib.dll, self.lib.prefix + "region_check_presence"):
            # the test suite's switch: the planes of every accumulate against the presence statement the scoring gather relies on
            n = C.c_int64(-1)
            fn = getattr(self.lib.dll, self.lib.prefix + "region_check_presence")
            fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]
            self._check(fn(self.h, C.byref(n)))
            if n.value != 0:
                raise AssertionError("uvcgpu_region_check_presence: %d (position, symbol) cells contradict the presence statement" % n.value)

    def correct_bq(self):
        """apply_bq_err_correction3 (grouping.cpp:459-543) on the library's copy of the base qualities."""
        fn = getattr(self.lib.dll, self.lib.prefix + "region_correct_bq")
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
        self._check(fn(self.h))

    def read_quals(self, n_bases):
        fn = getattr(self.lib.dll, self.lib.prefix + "region_read_quals")
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]
        out = np.empty(n_bases, dtype=np.uint8)
        self._check(fn(self.h, out.ctypes.data, n_bases))
        return out

    def fetch(self, group):
        gid, dt, shape = _ffi.FIELD_GROUPS[group]
        nbytes = self.lib.call("field_bytes", self.h, gid)
        if nbytes < 0:
            raise UvcError(-5, "field group %s not available" % group)
        out = np.empty(shape + (self.npos,), dtype=dt)
        assert out.nbytes == nbytes, (group, out.nbytes, nbytes)
        self._check(self.lib.call("fetch", self.h, gid, out.ctypes.data, out.nbytes))
        return out

    def indel_alleles(self):
        """The per-strand InDel allele rows fill_by_indel_info reads (instcode.hpp): list of dicts with the inserted sequence as text
        (insertions) or None (deletions: the deleted bases are refseq[refpos - beg : + len])."""
        fn = getattr(self.lib.dll, self.lib.prefix + "region_indel_alleles")
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        n, nb = C.c_int64(0), C.c_int64(0)
        rc = fn(self.h, None, 0, C.byref(n), None, 0, C.byref(nb))
        if rc not in (0, -6):
            self._check(rc)
        rows = (_ffi.UvcGapRow * max(1, n.value))()
        seq = np.zeros(max(1, nb.value), dtype=np.uint8)
        self._check(fn(self.h, rows, n.value, C.byref(n), seq.ctypes.data, nb.value, C.byref(nb)))
        out = []
        for i in range(n.value):
            r = rows[i]
            text = "".join("ACGTN"[b] for b in seq[r.seq_off:r.seq_off + r.len]) if r.seq_off >= 0 else None
            out.append(dict(refpos=r.refpos, symbol=r.symbol, strand=r.strand, len=r.len, seq=text, bAD1=r.bAD1, cAD1=r.cAD1, c2AD=r.c2AD, c2dAD=r.c2dAD))
        return out

    @staticmethod
    def make_request(all_out=False, pos_beg=-1, pos_end=-1, is_amplicon=False, indel_alleles=None, tumor_keys=None, release_state=False, base_at_pos_beg=False, region_beg=0,
                     tumor_sample_columns=None, tumor_ref_alt=None, kept_only=False):
        """UvcScoreRequest + the ctypes arrays it points into (keep both alive for the call)."""
        req = _ffi.UvcScoreRequest()
        req.pos_beg, req.pos_end, req.all_out, req.is_amplicon = pos_beg, pos_end, int(all_out), int(is_amplicon)
        req.release_state = int(release_state)   # the planes may be zeroed for the next accumulate as soon as the scoring kernels are done
        req.base_at_pos_beg, req.region_beg = int(base_at_pos_beg), int(region_beg)
        req.kept_only = int(kept_only)           # only the (position, symbol type) groups the record writer reads
        arr = None
        if indel_alleles:
            arr = (_ffi.UvcIndelAllele * len(indel_alleles))(*[_ffi.UvcIndelAllele(*a) for a in indel_alleles])
            req.n_indel_alleles, req.indel_alleles = len(indel_alleles), C.cast(arr, C.c_void_p)
        tk = None
        if tumor_keys is not None and len(tumor_keys):   # T/N: UvcTumorKey field tuples (or a ready ctypes array), sorted by (refpos, symbol)
            tk = tumor_keys if isinstance(tumor_keys, C.Array) else (_ffi.UvcTumorKey * len(tumor_keys))(*[_ffi.UvcTumorKey(*t) for t in tumor_keys])
            req.n_tumor_keys, req.tumor_keys = len(tk), C.cast(tk, C.c_void_p)
        cols = None
        if tk is not None and tumor_sample_columns:   # is_tumor_format_retrieved: the tumor's sample column of every key, appended by the record writer
            assert len(tumor_sample_columns) == len(tk)
            cols = (C.c_char_p * len(tk))(*[c.encode() if isinstance(c, str) else c for c in tumor_sample_columns])
            req.tumor_sample_columns = C.cast(cols, C.c_void_p)
        ras = None
        if tk is not None and tumor_ref_alt:          # "REF\tALT" of every key: the InDel strings of rescued InDel records (record writer)
            assert len(tumor_ref_alt) == len(tk)
            ras = (C.c_char_p * len(tk))(*[c.encode() if isinstance(c, str) else c for c in tumor_ref_alt])
            req.tumor_ref_alt = C.cast(ras, C.c_void_p)
        return req, (arr, tk, cols, ras)

    def hap_links(self):
        """hap_bq / hap_fq / hap_f2q of updateByRegion3Aln (uvcgpu_region_hap_links): three lists of (mutations, (fwd, rev), (other fwd, other rev))
        with mutations = ((refpos, symbol), ...)."""
        fn = getattr(self.lib.dll, self.lib.prefix + "region_hap_links")
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        n, m = C.c_int64(0), C.c_int64(0)
        rc = fn(self.h, None, 0, C.byref(n), None, 0, C.byref(m))
        if rc not in (0, -6):
            self._check(rc)
        links = (_ffi.UvcHapLink * max(1, n.value))()
        muts = np.zeros(max(1, m.value), dtype=np.int32)
        self._check(fn(self.h, links, n.value, C.byref(n), muts.ctypes.data, m.value, C.byref(m)))
        out = [[], [], []]
        for i in range(n.value):
            l = links[i]
            pairs = tuple((int(muts[l.mut_off + 2 * k]), int(muts[l.mut_off + 2 * k + 1])) for k in range(l.n_muts))
            out[l.which].append((pairs, (l.fr_cnt[0], l.fr_cnt[1]), (l.other_cnt[0], l.other_cnt[1])))
        return out

    def score(self, all_out=False, pos_beg=-1, pos_end=-1, is_amplicon=False, indel_alleles=None, capacity=None, copy=True, tumor_keys=None, release_state=False, base_at_pos_beg=False, region_beg=0,
              kept_only=False):
        req, _keep = self.make_request(all_out, pos_beg, pos_end, is_amplicon, indel_alleles, tumor_keys, release_state, base_at_pos_beg, region_beg, kept_only=kept_only)
        if capacity is None:
            npos = (pos_end - pos_beg) if pos_beg >= 0 else self.npos
            capacity = 14 * (npos + 1) if all_out else max(4096, 4 * (npos + 1))
        capacity = max(capacity, getattr(self, "_score_cap", 0))   # a handle that needed a larger buffer once asks for it at once the next time
        while True:
            buf = getattr(self, "_score_buf", None)   # reused across calls: the library fills n_records columns of every row
            if buf is None or buf.shape[1] != capacity:
                self._free_score_buf()
                buf = self._score_buf = self._alloc_score_buf(capacity)
            out = _ffi.UvcScoreOut(capacity, 0, buf.ctypes.data)
            rc = self.lib.call("score", self.h, C.byref(req), C.byref(out))
            if rc == -6 and out.n_records > capacity:
                capacity = self._score_cap = int(out.n_records) + int(out.n_records) // 8
                continue
            self._check(rc)
            # copy=False returns views into the handle's reusable buffer (valid until the next score() call)
            return {name: (buf[i, :out.n_records].copy() if copy else buf[i, :out.n_records]) for i, name in enumerate(_ffi.SCORE_FIELDS)}

    def fetch_columns(self, refpos):
        """Every plane value of the given positions (uvcgpu_region_fetch_columns): int64 [len(refpos), n_columns]; `column_base(group)`
        gives the first column of a plane group, the planes of a group follow in the order of `fetch(group)`."""
        fn = getattr(self.lib.dll, self.lib.prefix + "region_fetch_columns")
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        ncol = getattr(self.lib.dll, self.lib.prefix + "region_n_columns")
        ncol.restype, ncol.argtypes = C.c_int32, []
        pos = np.ascontiguousarray(refpos, dtype=np.int32)
        out = np.zeros((len(pos), ncol()), dtype=np.int64)
        self._check(fn(self.h, pos.ctypes.data, len(pos), out.ctypes.data))
        return out

    def column_base(self, group):
        fn = getattr(self.lib.dll, self.lib.prefix + "region_column_base")
        fn.restype, fn.argtypes = C.c_int32, [C.c_int32]
        return fn(_ffi.FIELD_GROUPS[group][0])

    def vcf_records(self, contig_name, records, tumor_keys=None, pos_beg=-1, pos_end=-1, base_at_pos_beg=False, region_beg=0, tumor_sample_columns=None, tumor_ref_alt=None):
        """The VCF lines (text) of the records `score()` returned that are written (out and keep set): uvcgpu_region_vcf_records.
        Needs the planes, i.e. a score call without release_state.  The keyword arguments repeat those of the score call."""
        fn = getattr(self.lib.dll, self.lib.prefix + "region_vcf_records")
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(_ffi.UvcScoreOut), C.POINTER(_ffi.UvcScoreRequest), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        n = len(records["refpos"])
        buf = np.ascontiguousarray(np.stack([np.asarray(records[name], dtype=np.int32) for name in _ffi.SCORE_FIELDS])) if n else np.zeros((_ffi.NUM_SCORE_FIELDS, 1), dtype=np.int32)
        so = _ffi.UvcScoreOut(max(n, 1), n, buf.ctypes.data)
        req, _keep = self.make_request(pos_beg=pos_beg, pos_end=pos_end, tumor_keys=tumor_keys, base_at_pos_beg=base_at_pos_beg, region_beg=region_beg, tumor_sample_columns=tumor_sample_columns, tumor_ref_alt=tumor_ref_alt)
        ln = C.c_int64(0)
        rc = fn(self.h, contig_name.encode(), C.byref(so), C.byref(req), None, 0, C.byref(ln))
        if rc not in (0, -6):
            self._check(rc)
        dst = C.create_string_buffer(max(1, ln.value))
        self._check(fn(self.h, contig_name.encode(), C.byref(so), C.byref(req), dst, ln.value, C.byref(ln)))
        return dst.raw[:ln.value].decode()

    def _alloc_score_buf(self, capacity):
        """[NUM_SCORE_FIELDS][capacity] int32 in page-locked memory of the library's own (uvcgpu_host_alloc), so that the D2H of the records runs
        at PCIe speed.  Not a pinned numpy buffer: heap pages that once were the source of a pageable upload can be mapped read-only for the
        GPU and come back from the allocator as such a buffer.  Libraries without the call (the oracle) get a plain array."""
        n = _ffi.NUM_SCORE_FIELDS * capacity
        alloc = getattr(self.lib.dll, self.lib.prefix + "host_alloc", None)
        if alloc is not None and not os.environ.get("UVC_NO_PIN"):
            alloc.restype, alloc.argtypes = C.c_int, [C.POINTER(C.c_void_p), C.c_int64]
            p = C.c_void_p()
            if alloc(C.byref(p), n * 4) == 0 and p.value:
                self._score_buf_host = p
                return np.ctypeslib.as_array((C.c_int32 * n).from_address(p.value)).reshape(_ffi.NUM_SCORE_FIELDS, capacity)
        self._score_buf_host = None
        return np.empty((_ffi.NUM_SCORE_FIELDS, capacity), dtype=np.int32)

    def _free_score_buf(self):
        p = getattr(self, "_score_buf_host", None)
        self._score_buf = None
        if p is not None and p.value:
            fn = getattr(self.lib.dll, self.lib.prefix + "host_free")
            fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
            fn(p)
        self._score_buf_host = None

    def close(self):
        self._free_score_buf()
        if self.h:
            self.lib.call("destroy", self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
