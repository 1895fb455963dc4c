"""TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

`oracle.library_path()`: where `make -C oracle` puts the CPU restatement (liboracle.so, prefix `uvc_oracle_`).  It is bound with the same
ctypes declarations as the product library (uvc_amd._ffi.Lib) because it exports the same C ABI under another prefix."""
import os


def library_path():
    # UVC_ORACLE_LIBRARY: another build of the same checker (scripts/cpu_sanitize.sh: AddressSanitizer + UBSan)
    return os.environ.get("UVC_ORACLE_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboracle.so")
