"""ctypes mirror of include/uvcio.h: BAM (+ BAI) and FASTA (+ .fai) readers of libuvcio.so -- the htslib calls of the reference's
ingest (sam_itr_queryi / sam_itr_next, faidx_fetch_seq; grouping.cpp:617-731, main.cpp:529-531) on zlib only."""
import ctypes as C
import os

import numpy as np

from . import _ffi


class UvcBamBatch(C.Structure):
    _fields_ = [("n_alns", C.c_int64), ("tid", C.c_void_p), ("pos", C.c_void_p), ("endpos", C.c_void_p), ("mtid", C.c_void_p), ("mpos", C.c_void_p), ("isize", C.c_void_p),
                ("flag", C.c_void_p), ("mapq", C.c_void_p), ("nm", C.c_void_p), ("l_qseq", C.c_void_p), ("n_cigar", C.c_void_p),
                ("seq_off", C.c_void_p), ("cigar_off", C.c_void_p), ("qname_off", C.c_void_p),
                ("n_bases", C.c_int64), ("bases", C.c_void_p), ("quals", C.c_void_p), ("n_cigar_ops", C.c_int64), ("cigars", C.c_void_p),
                ("n_qname_bytes", C.c_int64), ("qnames", C.c_void_p)]


_COLS = [("tid", np.int32), ("pos", np.int32), ("endpos", np.int32), ("mtid", np.int32), ("mpos", np.int32), ("isize", np.int32), ("flag", np.uint16), ("mapq", np.uint8),
         ("nm", np.int32), ("l_qseq", np.int32), ("n_cigar", np.int32), ("seq_off", np.int64), ("cigar_off", np.int64), ("qname_off", np.int64)]
_dll = None


def library_path():
    # UVCIO_LIBRARY: another build of the same library (scripts/cpu_sanitize.sh runs the reader tests on an AddressSanitizer build)
    return os.environ.get("UVCIO_LIBRARY") or os.path.join(_ffi.ROOT, "uvc_amd", "csrc", "libuvcio.so")


def dll():
    global _dll
    if _dll is None:
        if not os.path.exists(library_path()):
            raise ImportError("%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % library_path())
        d = C.CDLL(library_path())
        d.uvcio_last_error.restype = C.c_char_p
        d.uvcio_bam_open.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
        d.uvcio_bam_n_refs.argtypes = [C.c_void_p]
        d.uvcio_bam_ref_name.restype, d.uvcio_bam_ref_name.argtypes = C.c_char_p, [C.c_void_p, C.c_int32]
        d.uvcio_bam_ref_len.restype, d.uvcio_bam_ref_len.argtypes = C.c_int64, [C.c_void_p, C.c_int32]
        d.uvcio_bam_has_index.argtypes = [C.c_void_p]
        d.uvcio_bam_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.POINTER(UvcBamBatch)]
        d.uvcio_bam_close.argtypes = [C.c_void_p]
        d.uvcio_fasta_open.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
        d.uvcio_fasta_seq_len.restype, d.uvcio_fasta_seq_len.argtypes = C.c_int64, [C.c_void_p, C.c_char_p]
        d.uvcio_fasta_fetch.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_char_p]
        d.uvcio_fasta_close.argtypes = [C.c_void_p]
        d.uvcio_bam_region_bytes.restype, d.uvcio_bam_region_bytes.argtypes = C.c_int64, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64]
        d.uvcio_plan_shards.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        d.uvcio_bgzf_concat.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int32]
        d.uvcio_tumor_vcf_open.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.POINTER(C.c_char_p), C.c_int32, C.c_int32]
        d.uvcio_tumor_vcf_sample_name.restype, d.uvcio_tumor_vcf_sample_name.argtypes = C.c_char_p, [C.c_void_p]
        d.uvcio_tumor_vcf_n_records.restype, d.uvcio_tumor_vcf_n_records.argtypes = C.c_int64, [C.c_void_p]
        d.uvcio_tumor_vcf_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        d.uvcio_tumor_vcf_close.argtypes = [C.c_void_p]
        _dll = d
    return _dll


def _check(rc):
    if rc != 0:
        raise IOError("uvcio error %d: %s" % (rc, dll().uvcio_last_error().decode()))


class _Names:
    """read names of a batch, decoded on demand (a list of two million Python strings is the slowest part of a fetch)"""

    def __init__(self, raw, off):
        self.raw, self.off = raw, off

    def __len__(self):
        return len(self.off)

    def __getitem__(self, i):
        o = int(self.off[i])
        return self.raw[o:self.raw.index(b"\0", o)].decode()

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class Bam:
    def __init__(self, path):
        self.h = C.c_void_p()
        _check(dll().uvcio_bam_open(C.byref(self.h), path.encode()))
        self.refs = [(dll().uvcio_bam_ref_name(self.h, i).decode(), dll().uvcio_bam_ref_len(self.h, i)) for i in range(dll().uvcio_bam_n_refs(self.h))]
        self.has_index = bool(dll().uvcio_bam_has_index(self.h))

    def tid(self, name):
        return [n for n, _ in self.refs].index(name)

    def fetch(self, tid, beg, end):
        """Alignments overlapping [beg, end) of reference `tid`, in file order: dict of numpy columns (copies) + the read names."""
        b = UvcBamBatch()
        _check(dll().uvcio_bam_fetch(self.h, tid, beg, end, C.byref(b)))
        n = b.n_alns

        def col(ptr, dt, cnt):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(cnt,)).copy() if cnt else np.zeros(0, dt)
        out = {name: col(getattr(b, name), dt, n) for name, dt in _COLS}
        out["bases"] = col(b.bases, np.uint8, b.n_bases); out["quals"] = col(b.quals, np.uint8, b.n_bases); out["cigars"] = col(b.cigars, np.uint32, b.n_cigar_ops)
        raw = C.string_at(b.qnames, b.n_qname_bytes) if b.n_qname_bytes else b""
        out["qnames_raw"] = raw                      # NUL-terminated names back to back, qname_off[i] = start of the i-th
        out["qnames"] = _Names(raw, out["qname_off"])
        out["n_alns"] = n
        return out

    def close(self):
        if self.h:
            dll().uvcio_bam_close(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Fasta:
    def __init__(self, path):
        self.h = C.c_void_p()
        _check(dll().uvcio_fasta_open(C.byref(self.h), path.encode()))

    def seq_len(self, name):
        return dll().uvcio_fasta_seq_len(self.h, name.encode())

    def fetch(self, name, beg, end):
        buf = C.create_string_buffer(max(1, end - beg))
        _check(dll().uvcio_fasta_fetch(self.h, name.encode(), beg, end, buf))
        return buf.raw[:end - beg].decode()

    def close(self):
        if self.h:
            dll().uvcio_fasta_close(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class UvcRegionCut(C.Structure):
    _fields_ = [("tid", C.c_int32), ("beg", C.c_int32), ("end", C.c_int32), ("flag", C.c_int32), ("batch", C.c_int32), ("n_reads", C.c_int64)]


class BgzfWriter:
    """uvcio_bgzf_write_*: the block-gzipped stream the reference writes its VCF through (main.cpp:1196-1215)."""

    def __init__(self, path, level=6):
        d = dll()
        d.uvcio_bgzf_write_open.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int32]
        d.uvcio_bgzf_write.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        d.uvcio_bgzf_write_close.argtypes = [C.c_void_p]
        self.h = C.c_void_p()
        _check(d.uvcio_bgzf_write_open(C.byref(self.h), path.encode(), level))

    def write(self, data):
        b = data.encode() if isinstance(data, str) else bytes(data)
        _check(dll().uvcio_bgzf_write(self.h, b, len(b)))

    def close(self):
        if self.h:
            h, self.h = self.h, C.c_void_p()
            _check(dll().uvcio_bgzf_write_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def plan_regions(tid, pos, endpos, flag, target_lens, nthreads=1, mem_per_thread_mb=1536):
    """SamIter::iternext without a BED file (grouping.cpp:225-312) over alignment columns in file order: the blocks the reference hands
    to process_batch, as dicts (tid, beg, end, flag, batch, n_reads)."""
    d = dll()
    d.uvcio_plan_regions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    a = [np.ascontiguousarray(tid, dtype=np.int32), np.ascontiguousarray(pos, dtype=np.int32), np.ascontiguousarray(endpos, dtype=np.int32), np.ascontiguousarray(flag, dtype=np.uint16)]
    tl = np.ascontiguousarray(target_lens, dtype=np.int64)
    n = C.c_int64(0)
    rc = d.uvcio_plan_regions(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, len(a[0]), tl.ctypes.data, len(tl), nthreads, mem_per_thread_mb, None, 0, C.byref(n))
    if rc not in (0, -6):
        _check(rc)
    out = (UvcRegionCut * max(1, n.value))()
    _check(d.uvcio_plan_regions(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, len(a[0]), tl.ctypes.data, len(tl), nthreads, mem_per_thread_mb, out, n.value, C.byref(n)))
    return [dict(tid=c.tid, beg=c.beg, end=c.end, flag=c.flag, batch=c.batch, n_reads=c.n_reads) for c in out[:n.value]]


def plan_regions_stream(tid, pos, endpos, flag, target_lens, nthreads=1, mem_per_thread_mb=1536, piece=1000):
    """The same cuts through the streaming form (uvcio_planner_*): the columns are fed `piece` alignments at a time, cuts are taken as they appear."""
    d = dll()
    d.uvcio_planner_open.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int32, C.c_int32, C.c_int64]
    d.uvcio_planner_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    d.uvcio_planner_finish.argtypes = [C.c_void_p]
    d.uvcio_planner_take.restype, d.uvcio_planner_take.argtypes = C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64]
    d.uvcio_planner_close.restype, d.uvcio_planner_close.argtypes = None, [C.c_void_p]
    a = [np.ascontiguousarray(tid, dtype=np.int32), np.ascontiguousarray(pos, dtype=np.int32), np.ascontiguousarray(endpos, dtype=np.int32), np.ascontiguousarray(flag, dtype=np.uint16)]
    tl = np.ascontiguousarray(target_lens, dtype=np.int64)
    h = C.c_void_p()
    _check(d.uvcio_planner_open(C.byref(h), tl.ctypes.data, len(tl), nthreads, mem_per_thread_mb))
    cuts, buf = [], (UvcRegionCut * 64)()

    def take():
        while True:
            k = d.uvcio_planner_take(h, buf, 64)
            cuts.extend(dict(tid=c.tid, beg=c.beg, end=c.end, flag=c.flag, batch=c.batch, n_reads=c.n_reads) for c in buf[:k])
            if k < 64:
                break
    try:
        for i in range(0, len(a[0]), max(1, piece)):
            s = [x[i:i + piece] for x in a]
            _check(d.uvcio_planner_feed(h, s[0].ctypes.data, s[1].ctypes.data, s[2].ctypes.data, s[3].ctypes.data, len(s[0])))
            take()
        _check(d.uvcio_planner_finish(h))
        take()
    finally:
        d.uvcio_planner_close(h)
    return cuts


class TumorVcf:
    """The tumor VCF of a T/N pair as the normal pass reads it (uvcio_tumor_vcf_*: rescue_variants_from_vcf, main.cpp:183-398)."""

    def __init__(self, path, contig_names, is_tumor_format_retrieved=True):
        self.h = C.c_void_p()
        names = (C.c_char_p * max(1, len(contig_names)))(*[n.encode() for n in contig_names])
        _check(dll().uvcio_tumor_vcf_open(C.byref(self.h), path.encode(), names, len(contig_names), int(is_tumor_format_retrieved)))
        self.sample = dll().uvcio_tumor_vcf_sample_name(self.h).decode()
        self.n_records = dll().uvcio_tumor_vcf_n_records(self.h)

    def fetch(self, tid, pos_beg, pos_end):
        """-> (ctypes array of UvcTumorKey or None, list of sample-column strings): the records with pos_beg <= symbolpos <= pos_end;
        `last_ref_alt` holds their "REF\tALT" strings."""
        keys, cols, ras, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64(0)
        _check(dll().uvcio_tumor_vcf_fetch(self.h, tid, pos_beg, pos_end, C.byref(keys), C.byref(cols), C.byref(ras), C.byref(n)))
        self.last_ref_alt = []
        if n.value == 0:
            return None, []
        arr = (_ffi.UvcTumorKey * n.value).from_address(keys.value)
        texts = [s.decode() for s in (C.c_char_p * n.value).from_address(cols.value)]
        self.last_ref_alt = [s.decode() for s in (C.c_char_p * n.value).from_address(ras.value)]   # "REF\tALT" of the same records
        return arr, texts

    def close(self):
        if self.h:
            dll().uvcio_tumor_vcf_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
