"""uvcgpu_region_set_reads_device: the read columns handed over in HBM (a decoder that writes to the device, the bench's resident inputs)
must give exactly what uvcgpu_region_set_reads gives for the same columns on the host -- also when the base / quality columns are
sub-arrays that start at an odd address (the packing kernel then takes its byte-wise form)."""
import zlib

import numpy as np
import pytest

from uvc_amd import _ffi, region, synth

pytestmark = pytest.mark.gpu
GROUPS = ("PREP32", "SEG32", "SEG64", "VQ", "BQSUM", "FRAG", "FAM", "FAMINFO32", "DUPLEX")


class DeviceColumns:
    """The UvcReadSoA columns of `reads` in device memory, allocated through the HIP runtime the library itself links (torch brings its own
    copy of the runtime, which does not find the GPU once another copy has initialised it)."""
    def __init__(self, reads, misalign=0):
        import ctypes as C
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]; self.hip.hipFree.argtypes = [C.c_void_p]
        self.ptrs = []
        soa = _ffi.UvcReadSoA()
        soa.struct_size = C.sizeof(_ffi.UvcReadSoA)
        soa.n_reads = int(reads["n_reads"])

        def put(a, dt, shift=0):
            a = np.ascontiguousarray(a, dtype=dt)
            p = C.c_void_p()
            assert self.hip.hipMalloc(C.byref(p), a.nbytes + 64) == 0
            self.ptrs.append(p)
            if a.nbytes:
                assert self.hip.hipMemcpy(p.value + shift, a.ctypes.data, a.nbytes, 1) == 0
            return p.value + shift
        for name, dt in region._READ_FIELDS:
            if reads.get(name) is not None:   # the compact form has no seq_off / cigar_off
                setattr(soa, name, put(reads[name], dt))
        for name, dt, cnt in (("bases", np.uint8, "n_bases"), ("quals", np.uint8, "n_bases"), ("cigars", np.uint32, "n_cigar_ops")):
            if reads.get(name) is None:
                continue
            setattr(soa, name, put(reads[name], dt, misalign if name != "cigars" else 0))
            setattr(soa, cnt, int(np.asarray(reads[name]).size))
        if reads.get("bases") is None:
            soa.bases4 = put(reads["bases4"], np.uint8, misalign)
            soa.n_bases4_bytes = int(reads["bases4"].size)
        soa.n_fams = int(reads["n_fams"])
        soa.fam_dflag = put(reads["fam_dflag"], np.uint8)
        self.soa = soa

    def free(self):
        for p in self.ptrs:
            self.hip.hipFree(p)


def planes_and_records(R):
    R.accumulate()
    sums = {g: zlib.crc32(R.fetch(g).tobytes()) for g in GROUPS}
    rec = R.score()
    return sums, rec


@pytest.mark.parametrize("kw", [dict(seed=21, region_len=6000, depth=120), dict(seed=22, region_len=2500, depth=300, umi=True)])
def test_device_columns_equal_host_columns(kw, gpu_lib):
    reads = synth.generate_region(**kw)
    p = region.default_params(gpu_lib)
    Rh = region.Region(gpu_lib, p, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    Rh.set_reads(reads)
    want_sums, want_rec = planes_and_records(Rh)
    # aligned device columns, then bases and qualities as sub-arrays that start one byte into their allocations
    for misalign in (0, 1):
        cols = DeviceColumns(reads, misalign)
        assert (cols.soa.bases % 8 != 0) == (misalign == 1)
        Rd = region.Region(gpu_lib, p, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
        Rd.set_reads_device((cols.soa, cols))
        sums, rec = planes_and_records(Rd)
        assert sums == want_sums and all(np.array_equal(rec[k], want_rec[k]) for k in rec), misalign
        Rd.close(); cols.free()
    Rh.close()


@pytest.mark.parametrize("kw", [dict(seed=31, region_len=5000, depth=100), dict(seed=32, region_len=2000, depth=300, umi=True), dict(weird=7)])
def test_compact_columns_equal_plain_columns(kw, gpu_lib):
    """UvcReadSoA::bases4 (BAM's 4-bit codes, reads back to back) with seq_off / cigar_off left NULL: the offsets are prefix sums on the
    device, the bases go through seq_nt16_int there -- same planes, same records, from host columns and from device columns."""
    if "weird" in kw:   # the fuzz generator: read lengths of every parity (each read starts on a byte boundary of bases4), InDels, clips
        from test_gpu_fuzz import weird_region
        reads = weird_region(kw["weird"])
        assert (np.asarray(reads["l_qseq"]) % 2 == 1).any()
    else:
        reads = synth.generate_region(**kw)
    p = region.default_params(gpu_lib)
    Rh = region.Region(gpu_lib, p, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    Rh.set_reads(reads)
    want_sums, want_rec = planes_and_records(Rh)
    comp = region.compact_form(reads)
    assert "bases" not in comp and comp["bases4"].size < reads["bases"].size * 0.51
    Rc = region.Region(gpu_lib, p, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    Rc.set_reads(comp)
    sums, rec = planes_and_records(Rc)
    assert sums == want_sums and all(np.array_equal(rec[k], want_rec[k]) for k in rec)
    for misalign in (0, 1):
        cols = DeviceColumns(comp, misalign)
        Rc.reset(reads["tid"], reads["beg"], reads["end"], reads["refseq"])
        Rc.set_reads_device((cols.soa, cols))
        sums, rec = planes_and_records(Rc)
        assert sums == want_sums and all(np.array_equal(rec[k], want_rec[k]) for k in rec), misalign
        cols.free()
    # a packed column that is too short for the read lengths is refused, not read past its end
    bad = dict(comp); bad["bases4"] = comp["bases4"][: comp["bases4"].size // 2]
    with pytest.raises(region.UvcError) as e:
        Rc.reset(reads["tid"], reads["beg"], reads["end"], reads["refseq"]); Rc.set_reads(bad)
    assert e.value.code == -2
    Rc.close(); Rh.close()
