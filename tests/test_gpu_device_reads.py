"""uvcgpu_region_set_reads_device: the read columns handed over in HBM (a decoder that writes to the device, the bench's resident inputs)
must give exactly what uvcgpu_region_set_reads gives for the same columns on the host -- also when the base / quality columns are
sub-arrays that start at an odd address (the packing kernel then takes its byte-wise form)."""
import zlib

import numpy as np
import pytest

from uvc_amd import _ffi, region, synth

pytestmark = pytest.mark.gpu
GROUPS = ("PREP32", "SEG32", "SEG64", "VQ", "BQSUM", "FRAG", "FAM", "FAMINFO32", "DUPLEX")


def planes_and_records(R):
    R.accumulate()
    sums = {g: zlib.crc32(R.fetch(g).tobytes()) for g in GROUPS}
    rec = R.score()
    return sums, rec


@pytest.mark.parametrize("kw", [dict(seed=21, region_len=6000, depth=120), dict(seed=22, region_len=2500, depth=300, umi=True)])
def test_device_columns_equal_host_columns(kw, gpu_lib):
    import torch
    reads = synth.generate_region(**kw)
    p = region.default_params(gpu_lib)
    Rh = region.Region(gpu_lib, p, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    Rh.set_reads(reads)
    want_sums, want_rec = planes_and_records(Rh)
    dev = torch.device("cuda", 0)
    Rd = region.Region(gpu_lib, p, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    Rd.set_reads_device(region.device_reads(reads, dev))
    sums, rec = planes_and_records(Rd)
    assert sums == want_sums and all(np.array_equal(rec[k], want_rec[k]) for k in rec)
    # the same columns again, bases and qualities as sub-arrays that start one byte into their allocations
    soa, keep = region.device_reads(reads, dev)
    b = np.ascontiguousarray(reads["bases"], np.uint8); q = np.ascontiguousarray(reads["quals"], np.uint8)
    tb = torch.from_numpy(np.concatenate([[255], b]).astype(np.uint8)).to(dev); tq = torch.from_numpy(np.concatenate([[255], q]).astype(np.uint8)).to(dev)
    soa.bases = tb.data_ptr() + 1; soa.quals = tq.data_ptr() + 1
    assert soa.bases % 8 != 0
    Ro = region.Region(gpu_lib, p, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
    Ro.set_reads_device((soa, keep + [tb, tq]))
    sums, rec = planes_and_records(Ro)
    assert sums == want_sums and all(np.array_equal(rec[k], want_rec[k]) for k in rec)
    for R in (Rh, Rd, Ro):
        R.close()
