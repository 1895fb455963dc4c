"""The C-ABI library loads without a GPU and exports every entry point include/uvcgpu.h declares
(no compute calls here), and it refuses to run without a device instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess

import pytest

from uvc_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dll():
    path = _ffi.gpu_library_path()
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "uvc_amd", "csrc"), "-j4"])
    return C.CDLL(path)


def declared_symbols():
    names = set()
    for hdr in ("uvcgpu.h", "uvcgroup.h", "uvcconsensus.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(uvcgpu_[a-z_0-9]+)\s*\(", txt))
    return sorted(names)


def test_every_declared_symbol_is_exported(dll):
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(dll, n), n


def test_reader_library_exports_its_header():
    """libuvcio.so (host-only file readers) exports every entry point of include/uvcio.h"""
    from uvc_amd import io as uio
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "uvcio.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(uvcio_[a-z_0-9]+)\s*\(", txt)))
    assert len(names) >= 10
    for n in names:
        assert hasattr(uio.dll(), n), n


def test_no_cpu_fallback(dll):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    dll.uvcgpu_init.restype = C.c_int
    dll.uvcgpu_last_error.restype = C.c_char_p
    assert dll.uvcgpu_init(0) == _ffi.ENUMS["UVCGPU_EDEVICE"]
    assert b"no CPU fallback" in dll.uvcgpu_last_error()
    # creating a region needs a HIP stream: must fail loudly, not compute on the host
    p = _ffi.UvcParams()
    dll.uvcgpu_params_default(C.byref(p))
    h = C.c_void_p()
    dll.uvcgpu_region_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(_ffi.UvcParams), C.c_int32, C.c_int32, C.c_int32, C.c_char_p]
    assert dll.uvcgpu_region_create(C.byref(h), C.byref(p), 0, 0, 8, b"ACGTACGT") != 0


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "uvc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                # no path, symbol prefix, import or attribute access of the checker (prose may mention it)
                assert "liboracle" not in src and "uvc_oracle_" not in src and "import oracle" not in src and "oracle/" not in src and "oracle.library_path" not in src, os.path.join(dirpath, f)
