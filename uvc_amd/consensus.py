"""ctypes mirror of include/uvcconsensus.h: the insertion / soft-clip consensus blocks of a batch of reads (SURVEY row a9,
ConsensusBlockSet of main_consensus.hpp:116-225 as P4 fills it, main.hpp:2875-2911).  Host code inside libuvcgpu.so: no GPU needed."""
import ctypes as C

import numpy as np

from . import region

ROW = 8                                  # A C G T N, BASE_NN, BQ sum, fragments
TYPES = ("softclip_left_to_right", "ins", "softclip_right_to_left")


class UvcConBlock(C.Structure):
    _fields_ = [("fam_id", C.c_int32), ("strand", C.c_int32), ("type", C.c_int32), ("refpos", C.c_int32), ("len", C.c_int32), ("n_fragments", C.c_int32), ("row_off", C.c_int64)]


class UvcConBase(C.Structure):
    _fields_ = [("base", C.c_char), ("quality", C.c_int8), ("pad_", C.c_int16), ("family_size", C.c_int32), ("family_identity", C.c_int32)]


class UvcConBlockRequest(C.Structure):
    _fields_ = [("min_fragments", C.c_int32), ("tid", C.c_int32), ("curr_beg", C.c_int32), ("curr_end", C.c_int32),
                ("prev_tid", C.c_int32), ("prev_beg", C.c_int32), ("prev_end", C.c_int32), ("reserved_", C.c_int32)]


def _collect(call):
    nb, nr = C.c_int64(0), C.c_int64(0)
    rc = call(None, 0, C.byref(nb), None, 0, C.byref(nr))
    if rc not in (0, -6):
        raise region.UvcError(rc, "consensus blocks")
    blocks = (UvcConBlock * max(1, nb.value))()
    rows = np.zeros((max(1, nr.value // ROW), ROW), dtype=np.int32)
    rc = call(blocks, nb.value, C.byref(nb), rows.ctypes.data, rows.size, C.byref(nr))
    if rc != 0:
        raise region.UvcError(rc, "consensus blocks")
    out = []
    for i in range(nb.value):
        b = blocks[i]
        out.append(dict(fam_id=b.fam_id, strand=b.strand, type=b.type, refpos=b.refpos, n_fragments=b.n_fragments, rows=rows[b.row_off:b.row_off + b.len].copy()))
    return out


def family_blocks(lib, params, reads, min_fragments=1, curr=None, prev=None):
    """uvcgpu_consensus_blocks: the family-level blocks of every (family, strand) unit with at least `min_fragments` fragments whose span
    overlaps curr = (beg, end) and not prev = (tid, beg, end)."""
    soa, keep = region.pack_reads(reads)
    fn = getattr(lib.dll, lib.prefix + "consensus_blocks")
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    cb, ce = curr if curr is not None else (0, 2**31 - 1)
    pt, pb, pe = prev if prev is not None else (-1, 0, 0)
    req = UvcConBlockRequest(int(min_fragments), int(reads["tid"]), int(cb), int(ce), int(pt), int(pb), int(pe), 0)
    return _collect(lambda b, nbc, nb, r, nrc, nr: fn(C.byref(params), C.byref(soa), C.byref(req), b, nbc, nb, r, nrc, nr))


def fragment_blocks(lib, params, reads, first_read, n_reads):
    """uvcgpu_consensus_blocks_of_fragment: what incByPosSeqQual leaves for the reads [first_read, first_read + n_reads) of one fragment."""
    soa, keep = region.pack_reads(reads)
    fn = getattr(lib.dll, lib.prefix + "consensus_blocks_of_fragment")
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    return _collect(lambda b, nbc, nb, r, nrc, nr: fn(C.byref(params), C.byref(soa), int(first_read), int(n_reads), b, nbc, nb, r, nrc, nr))


def block_to_seq(lib, rows, right_to_left=False, trim=None):
    """consensusBlockToSeqQual, after ConsensusBlock_trim(perc_dp, n_consec) when trim = (perc_dp, n_consec): list of (base, quality,
    family_size, family_identity)."""
    rows = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, ROW)
    fn = getattr(lib.dll, lib.prefix + "consensus_block_to_seq")
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
    out = (UvcConBase * max(1, len(rows)))()
    n = C.c_int32(0)
    p, k = trim if trim is not None else (-1, 0)
    rc = fn(rows.ctypes.data, len(rows), int(bool(right_to_left)), int(p), int(k), out, C.byref(n))
    if rc != 0:
        raise region.UvcError(rc, "consensus_block_to_seq")
    return [(out[i].base.decode(), int(out[i].quality), int(out[i].family_size), int(out[i].family_identity)) for i in range(n.value)]
