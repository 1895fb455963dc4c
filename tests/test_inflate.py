"""BGZF inflate on the device (uvc_amd/csrc/uvc_inflate.hip, include/uvcgpu.h: uvcgpu_bgzf_inflate) against zlib.
CPU part: the decoder core (uvc_inflate_core.h, the body of the GPU thread) compiled for the host by tests/native/inflate_core_host.cpp."""
import ctypes
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def payloads():
    """(name, uncompressed bytes, raw DEFLATE stream) over the block types and code shapes zlib can produce"""
    rng = np.random.default_rng(1)
    out = []
    for n in (0, 1, 2, 10, 100, 1000, 0xff00):
        datas = {"random": bytes(rng.integers(0, 256, n, dtype=np.uint8)), "2bit": bytes(rng.integers(0, 4, n, dtype=np.uint8)),
                 "periodic": (b"ACGTTGCA" * (n // 8 + 1))[:n], "reads": bytes(rng.choice(np.frombuffer(b"ACGT!#$%&IIIIFFFF", np.uint8), n))}
        for dn, data in datas.items():
            for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                    (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
                c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
                out.append(("%s_%d_l%d_s%d" % (dn, n, level, strategy), data, c.compress(data) + c.flush()))
    return out


@pytest.fixture(scope="module")
def host_core(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("native") / "inflate_core_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "native", "inflate_core_host.cpp")])
    dll = ctypes.CDLL(so)
    dll.inflate_core_host.restype = ctypes.c_int
    dll.inflate_core_host.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    return dll


def test_decoder_core_on_the_host_equals_zlib(host_core):
    for name, data, comp in payloads():
        out = np.zeros(max(1, len(data)), np.uint8)
        assert host_core.inflate_core_host(comp, len(comp), out.ctypes.data, len(data)) == 0, name
        assert out[:len(data)].tobytes() == data, name


def test_decoder_core_flags_corrupt_streams(host_core):
    rng = np.random.default_rng(3)
    data = bytes(rng.integers(0, 8, 5000, dtype=np.uint8))
    c = zlib.compressobj(6, zlib.DEFLATED, -15); comp = c.compress(data) + c.flush()
    out = np.zeros(6000, np.uint8)
    for i in range(0, len(comp), 5):                  # a flipped byte anywhere: an error code or other bytes, never a crash or a silent match
        b = bytearray(comp); b[i] ^= 0x55
        rc = host_core.inflate_core_host(bytes(b), len(b), out.ctypes.data, 5000)
        assert rc != 0 or out[:5000].tobytes() != data, i
    assert host_core.inflate_core_host(comp[:len(comp) // 2], len(comp) // 2, out.ctypes.data, 5000) != 0     # truncated input
    assert host_core.inflate_core_host(comp, len(comp), out.ctypes.data, 4000) != 0                            # ISIZE too small
    assert host_core.inflate_core_host(comp, len(comp), out.ctypes.data, 6000) != 0                            # ISIZE too large


def _gpu_inflate(lib, comps, sizes):
    fn = lib.dll.uvcgpu_bgzf_inflate
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    comp = np.frombuffer(b"".join(comps) + b"\0" * 8, np.uint8).copy()
    in_len = np.array([len(c) for c in comps], np.int32); in_off = np.concatenate([[0], np.cumsum(in_len[:-1])]).astype(np.int64)
    out_len = np.array(sizes, np.int32); out_off = (np.concatenate([[0], np.cumsum(out_len[:-1])]) + 100).astype(np.int64)   # the first 100 bytes are the caller's
    out = np.full(int(out_len.sum()) + 200, 0xAB, np.uint8)
    rc = fn(None, comp.ctypes.data, len(comp), in_off.ctypes.data, in_len.ctypes.data, out_off.ctypes.data, out_len.ctypes.data, len(comps), out.ctypes.data, len(out))
    return rc, out, out_off


@pytest.fixture(params=["lane per block", "wave per block", "wave per block, 8 waves per SIMD"])
def kernel_form(request, monkeypatch):
    """uvc_inflate.hip has three kernels, chosen by UVCGPU_INFLATE_WAVE at every call: 0 k_bgzf_inflate, 1 k_bgzf_inflate_wave, 8 (default) k_bgzf_inflate_wave8"""
    if request.param.startswith("wave per block"): monkeypatch.setenv("UVCGPU_INFLATE_WAVE", "8" if "8" in request.param else "1")
    else: monkeypatch.setenv("UVCGPU_INFLATE_WAVE", "0")
    return request.param


@pytest.mark.gpu
def test_device_inflate_equals_zlib(gpu_lib, kernel_form):
    cases = payloads()
    rc, out, off = _gpu_inflate(gpu_lib, [c for _, _, c in cases], [len(d) for _, d, _ in cases])
    assert rc == 0, gpu_lib.last_error()
    for (name, data, _), o in zip(cases, off):
        assert out[o:o + len(data)].tobytes() == data, name
    total = sum(len(d) for _, d, _ in cases)
    assert (out[:100] == 0xAB).all() and (out[100 + total:] == 0xAB).all()          # nothing outside the blocks' range is written


@pytest.mark.gpu
def test_device_inflate_of_a_bam_and_a_corrupt_block(gpu_lib, tmp_path, kernel_form):
    import bamwriter
    from uvc_amd import synth
    reads = synth.generate_region(seed=77, region_len=20000, depth=200)
    path = str(tmp_path / "t.bam")
    bamwriter.write_bam(path, [("chrT", int(reads["end"]) + 1000)], bamwriter.records_from_reads(reads))
    raw = open(path, "rb").read()
    comps, sizes, want = [], [], []
    off = 0
    while off + 18 <= len(raw):
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        xlen = struct.unpack_from("<H", raw, off + 10)[0]
        payload = raw[off + 12 + xlen: off + bsize - 8]
        comps.append(payload); sizes.append(struct.unpack_from("<I", raw, off + bsize - 4)[0]); want.append(zlib.decompress(payload, -15))
        off += bsize
    assert len(comps) > 70                                        # more than one wave of blocks
    rc, out, o = _gpu_inflate(gpu_lib, comps, sizes)
    assert rc == 0, gpu_lib.last_error()
    for k in range(len(comps)):
        assert out[o[k]:o[k] + sizes[k]].tobytes() == want[k], k
    bad = list(comps); b = bytearray(bad[40]); b[len(b) // 2] ^= 0x10; bad[40] = bytes(b)
    rc, out, o = _gpu_inflate(gpu_lib, bad, sizes)
    if rc == 0:
        assert out[o[40]:o[40] + sizes[40]].tobytes() != want[40]   # a flip that still decodes to ISIZE bytes is the CRC's to find
    else:
        assert "block 40" in gpu_lib.last_error()
