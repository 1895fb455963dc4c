"""Host-side mirror of the family-assignment pass (bamfname_to_strand_to_familyuid_to_reads, grouping.cpp:608-997) over the
C ABI of include/uvcgroup.h.  `lib` is a `_ffi.Lib`: libuvcgpu.so (HIP) or, in tests only, the oracle."""
import ctypes as C
import os
import re

import numpy as np

from . import _ffi


def _read_def():
    ints, dbls = [], []
    with open(os.path.join(_ffi.ROOT, "include", "uvc_group_params.def")) as fh:
        for line in fh:
            m = re.match(r"UVC_G([ID])\((\w+),\s*([^)]*)\)", line)
            if m:
                (ints if m.group(1) == "I" else dbls).append((m.group(2), eval(m.group(3))))
    return ints, dbls


GROUP_INTS, GROUP_DBLS = _read_def()


class UvcGroupParams(C.Structure):
    _fields_ = ([("struct_size", C.c_int32), ("fetch_tbeg", C.c_int32), ("fetch_tend", C.c_int32), ("end2end", C.c_int32), ("inferred_sequencing_platform", C.c_int32)]
                + [(n, C.c_int32) for n, _ in GROUP_INTS] + [("pad_", C.c_int32)] + [(n, C.c_double) for n, _ in GROUP_DBLS])


_IN = [("tid", np.int32), ("pos", np.int32), ("endpos", np.int32), ("mtid", np.int32), ("mpos", np.int32), ("isize", np.int32), ("flag", np.uint16), ("mapq", np.uint8),
       ("qname_hash31", np.uint64), ("qname_hash17", np.uint64), ("umi_hash31", np.uint64), ("umi_hash17", np.uint64), ("umi_kind", np.uint8)]


class UvcGroupInput(C.Structure):
    _fields_ = [("n_alns", C.c_int64)] + [(n, C.c_void_p) for n, _ in _IN]


class UvcGroupOut(C.Structure):
    _fields_ = [("filter_reason", C.c_void_p), ("isize_norm", C.c_void_p), ("order", C.c_void_p), ("fam_id", C.c_void_p), ("frag_id", C.c_void_p),
                ("fam_strand", C.c_void_p), ("fam_dflag", C.c_void_p), ("fam_idflag", C.c_void_p),
                ("n_kept", C.c_int64), ("n_fams", C.c_int32), ("n_frags", C.c_int32),
                ("extended_inclu_beg_pos", C.c_int32), ("extended_exclu_end_pos", C.c_int32), ("n_amplicon", C.c_int64), ("n_visited_qnames", C.c_int64)]


def _fn(lib, name, restype, argtypes):
    f = getattr(lib.dll, lib.prefix + name)
    f.restype, f.argtypes = restype, argtypes
    return f


def default_params(lib, fetch_tbeg, fetch_tend, platform=1):
    p = UvcGroupParams()
    _fn(lib, "group_params_default", None, [C.POINTER(UvcGroupParams)])(C.byref(p))
    p.fetch_tbeg, p.fetch_tend, p.inferred_sequencing_platform = fetch_tbeg, fetch_tend, platform
    return p


def strnhash(lib, s, base=31, n=None):
    b = s.encode() if isinstance(s, str) else bytes(s)
    return _fn(lib, "strnhash", C.c_uint64, [C.c_char_p, C.c_size_t, C.c_uint64])(b, len(b) if n is None else n, base)


def hash2hash(lib, a, b):
    return _fn(lib, "hash2hash", C.c_uint64, [C.c_uint64, C.c_uint64])(a, b)


def qname_digest(lib, qname, molecule_tag=0, disable_duplex=0):
    """-> (umi_kind, qname_hash31, qname_hash17, umi_hash31, umi_hash17)"""
    v = [C.c_uint64() for _ in range(4)]
    f = _fn(lib, "qname_digest", C.c_int, [C.c_char_p, C.c_int, C.c_int] + [C.POINTER(C.c_uint64)] * 4)
    k = f(qname.encode(), molecule_tag, disable_duplex, *[C.byref(x) for x in v])
    return (k,) + tuple(x.value for x in v)


def qname_digest_batch(lib, names):
    """list of read names -> (umi_kind uint8[n], hashes uint64[4][n]: qname31, qname17, umi31, umi17), in one library call"""
    return _digest_batch(lib, names, 0, 0)


def _digest_batch(lib, names, molecule_tag, disable_duplex):
    n = len(names)
    if hasattr(names, "raw"):                        # uvc_amd.io batch: the names are already one buffer + offsets
        raw, off = names.raw, np.ascontiguousarray(names.off, dtype=np.int64)
    else:
        raw = b"".join(q.encode() + b"\0" for q in names)
        off = np.zeros(n, np.int64)
        if n > 1:
            off[1:] = np.cumsum([len(q.encode()) + 1 for q in names[:-1]])
    h = np.zeros((4, max(n, 1)), np.uint64); kind = np.zeros(max(n, 1), np.uint8)
    f = _fn(lib, "qname_digest_batch", C.c_int, [C.c_char_p, C.c_void_p, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 5)
    rc = f(raw, off.ctypes.data, n, molecule_tag, disable_duplex, h[0].ctypes.data, h[1].ctypes.data, h[2].ctypes.data, h[3].ctypes.data, kind.ctypes.data)
    if rc != 0:
        raise RuntimeError("qname_digest_batch failed: %d" % rc)
    return kind[:n], h[:, :n]


def group_families(lib, params, cols):
    """cols: dict of the UvcGroupInput columns (numpy).  Returns a dict of numpy outputs (trimmed to n_kept / n_fams)."""
    n = len(cols["pos"])
    keep = [np.ascontiguousarray(cols[k], dtype=dt) for k, dt in _IN]
    inp = UvcGroupInput(n, *[a.ctypes.data for a in keep])
    o = dict(filter_reason=np.zeros(n, np.int32), isize_norm=np.zeros(n, np.int32), order=np.zeros(n, np.int32), fam_id=np.zeros(n, np.int32), frag_id=np.zeros(n, np.int32),
             fam_strand=np.zeros(n, np.uint8), fam_dflag=np.zeros(n, np.uint8), fam_idflag=np.zeros(n, np.uint8))
    out = UvcGroupOut(*[o[k].ctypes.data for k in ("filter_reason", "isize_norm", "order", "fam_id", "frag_id", "fam_strand", "fam_dflag", "fam_idflag")])
    rc = _fn(lib, "group_families", C.c_int, [C.POINTER(UvcGroupParams), C.POINTER(UvcGroupInput), C.POINTER(UvcGroupOut)])(C.byref(params), C.byref(inp), C.byref(out))
    if rc != 0:
        raise RuntimeError("group_families failed: %d %s" % (rc, lib.last_error()))
    k, f = out.n_kept, out.n_fams
    res = {a: o[a][:k] for a in ("order", "fam_id", "frag_id", "fam_strand")}
    res.update(filter_reason=o["filter_reason"], isize_norm=o["isize_norm"], fam_dflag=o["fam_dflag"][:f], fam_idflag=o["fam_idflag"][:f],
               n_kept=k, n_fams=f, n_frags=out.n_frags, ext_beg=out.extended_inclu_beg_pos, ext_end=out.extended_exclu_end_pos,
               n_amplicon=out.n_amplicon, n_visited_qnames=out.n_visited_qnames)
    return res


def umi_in_read_batch(lib, umi_struct, cols, umi_kind):
    """bam2umihash (grouping.cpp:569-606): marks in `umi_kind` (uint8 array, edited in place) the unpaired reads of a fetched batch (`cols` =
    uvc_amd.io.Bam.fetch columns) that carry the in-read UMI pattern `umi_struct` (ONE_STEP_UMI_STRUCT); returns the hashes of their UMI letters."""
    n = len(umi_kind)
    h = np.zeros(max(n, 1), np.uint64)
    keep = [np.ascontiguousarray(cols["bases"], np.uint8), np.ascontiguousarray(cols["seq_off"], np.int64), np.ascontiguousarray(cols["l_qseq"], np.int32), np.ascontiguousarray(cols["flag"], np.uint16)]
    assert umi_kind.dtype == np.uint8 and umi_kind.flags["C_CONTIGUOUS"]
    f = _fn(lib, "umi_in_read_batch", C.c_int, [C.c_char_p] + [C.c_void_p] * 4 + [C.c_int64, C.c_void_p, C.c_void_p])
    rc = f(umi_struct.encode(), keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data, keep[3].ctypes.data, n, umi_kind.ctypes.data, h.ctypes.data)
    if rc != 0:
        raise RuntimeError("umi_in_read_batch failed: %d" % rc)
    return h[:n]
