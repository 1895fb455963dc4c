"""SURVEY row a10 pinned on the reference itself.

oracle/_ref/libref_molid.so is the reference's MolecularID.cpp + MolecularID.hpp + Hash.hpp (+ common.cpp), compiled from where they lie
by `make -C oracle ref_molid` (they do not need htslib) behind a C wrapper.  The oracle's strnhash / hash2hash and its family key
(createKey with the strings replaced by hash pairs) are checked against it: equal hashes, the same key fields kept per dedup_idflag, and
-- what decides which reads share a family -- key equality exactly where the reference's operator< calls two keys equivalent.
The HIP family assignment is compared with the oracle in tests/test_group.py (-m gpu), and its digest entry points here.
"""
import ctypes as C
import itertools
import os

import numpy as np
import pytest

from uvc_amd import _ffi, group

REF_SO = os.path.join(_ffi.ROOT, "oracle", "_ref", "libref_molid.so")
M64 = (1 << 64) - 1

pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libref_molid.so not built (needs /root/reference at build time)")


@pytest.fixture(scope="module")
def ref():
    dll = C.CDLL(REF_SO)
    dll.ref_strnhash.restype, dll.ref_strnhash.argtypes = C.c_uint64, [C.c_char_p, C.c_size_t, C.c_uint64]
    dll.ref_strhash.restype, dll.ref_strhash.argtypes = C.c_uint64, [C.c_char_p, C.c_uint64]
    dll.ref_hash2hash.restype, dll.ref_hash2hash.argtypes = C.c_uint64, [C.c_uint64, C.c_uint64]
    bc = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    dll.ref_molid_key.restype, dll.ref_molid_key.argtypes = C.c_uint64, bc + [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    dll.ref_molid_less.restype, dll.ref_molid_less.argtypes = C.c_int, bc + bc
    return dll


STRINGS = ["", "a", "read/1#ACGT+TTGA", "x#AC#tail", "noumi", "q#A", "#", "longer_name_with_many_chars_0123456789#ACGTAC+GTTGCA#z",
           "A00123:45:HXXXXXXX:1:1101:1000:2000", "\x7f~}|", "zzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzz"]


def test_hashes_equal_the_reference(ref, oracle_lib):
    rng = np.random.default_rng(5)
    cases = STRINGS + ["".join(chr(int(c)) for c in rng.integers(33, 127, int(rng.integers(1, 80)))) for _ in range(200)]
    for s in cases:
        b = s.encode("latin-1")
        for base in (31, 17):
            assert group.strnhash(oracle_lib, b, base) == ref.ref_strnhash(b, len(b), base) == ref.ref_strhash(b, base)
            for n in (0, 1, 3, len(b) // 2):
                assert group.strnhash(oracle_lib, b, base, n=n) == ref.ref_strnhash(b, n, base)
    for a, b in [(0, 0), (1, 2), (12345678901234567, 98765), (M64, M64), (M64 - 5, 77)] + [tuple(int(v) for v in rng.integers(0, 1 << 63, 2)) for _ in range(100)]:
        assert group.hash2hash(oracle_lib, a, b) == ref.ref_hash2hash(a, b)


def _oracle_key(lib, bc):
    begtid, beg, endtid, end, qname, umi, dflag, idflag = bc
    fn = lib.dll.uvc_oracle_molecular_key
    fn.restype = None
    fn.argtypes = [C.c_int] * 4 + [C.c_uint64] * 4 + [C.c_int, C.c_int, C.POINTER(C.c_int64)]
    out = (C.c_int64 * 10)()
    h = lambda s, base: group.strnhash(lib, s, base)
    # the caller hands over the hash pair of an absent UMI as (0, 0) = the hashes of the empty string (include/uvcgroup.h)
    fn(begtid, beg, endtid, end, h(qname, 31), h(qname, 17), h(umi, 31), h(umi, 17), dflag, idflag, out)
    return tuple(out)


def _barcodes(rng, n):
    qn = [b"r%d" % i for i in range(6)] + [b"r1#ACG+TTA", b""]
    um = [b"", b"ACGT+TTGA", b"ACGT+TTGC", b"TTGA+ACGT", b"A"]
    out = []
    for _ in range(n):
        out.append((int(rng.integers(0, 2)), int(rng.integers(100, 103)), int(rng.integers(0, 2)), int(rng.integers(100, 103)),
                    qn[int(rng.integers(0, len(qn)))], um[int(rng.integers(0, len(um)))], int(rng.choice([0, 1, 3, 7, 8])),
                    int(rng.choice([0x0, 0x1, 0x2, 0x3, 0x4, 0x7, 0x8, 0x9, 0xA, 0xB, 0xF]))))
    return out


def test_key_fields_follow_createKey(ref, oracle_lib):
    rng = np.random.default_rng(11)
    for bc in _barcodes(rng, 400):
        out4, ql, ul = (C.c_int32 * 4)(), C.c_int32(), C.c_int32()
        ref.ref_molid_key(*bc, out4, C.byref(ql), C.byref(ul))
        k = _oracle_key(oracle_lib, bc)
        assert tuple(out4) == k[:4], (bc, tuple(out4), k)
        # the strings createKey keeps are the ones whose hashes the oracle keeps (the empty string hashes to 0)
        assert (ql.value > 0) == (k[4] != 0 or k[5] != 0), bc
        assert (ul.value > 0) == (k[6] != 0 or k[7] != 0), bc
        assert k[8:] == (bc[6], bc[7])


def test_key_equality_is_the_reference_equivalence(ref, oracle_lib):
    """Two alignments join one family iff neither key is less than the other (std::map<MolecularBarcode, ...>, grouping.cpp:939)."""
    rng = np.random.default_rng(12)
    bcs = _barcodes(rng, 160)
    keys = [_oracle_key(oracle_lib, bc) for bc in bcs]
    n_equal = 0
    for (a, ka), (b, kb) in itertools.combinations(zip(bcs, keys), 2):
        lt, gt = ref.ref_molid_less(*a, *b), ref.ref_molid_less(*b, *a)
        assert not (lt and gt)
        assert (ka == kb) == (not lt and not gt), (a, b)
        n_equal += (ka == kb)
        # the leading (tid, pos) fields order the keys as the reference orders them
        if ka[:4] != kb[:4]:
            assert (ka[:4] < kb[:4]) == bool(lt), (a, b)
    assert n_equal > 10   # the case generator does produce families


def test_calcHash_is_a_function_of_the_key(ref):
    """calcHash only sees the key: barcodes with equal keys have equal hashvalue, so the final tie-break of operator< never splits a family."""
    rng = np.random.default_rng(13)
    seen = {}
    for bc in _barcodes(rng, 600):
        out4, ql, ul = (C.c_int32 * 4)(), C.c_int32(), C.c_int32()
        h = ref.ref_molid_key(*bc, out4, C.byref(ql), C.byref(ul))
        idflag = bc[7]
        key = (tuple(out4), bc[4] if idflag & 4 else b"", bc[5] if idflag & 8 else b"", bc[6], idflag)
        assert seen.setdefault(key, h) == h


@pytest.mark.gpu
def test_gpu_digest_hashes_equal_the_reference(ref, gpu_lib):
    for s in STRINGS:
        b = s.encode("latin-1")
        if b"\x00" in b:
            continue
        for base in (31, 17):
            assert group.strnhash(gpu_lib, b, base) == ref.ref_strhash(b, base)
    assert group.hash2hash(gpu_lib, 12345678901234567, 98765) == ref.ref_hash2hash(12345678901234567, 98765)
