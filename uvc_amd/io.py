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
    return os.path.join(_ffi.ROOT, "uvc_amd", "csrc", "libuvcio.so")


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
