"""uvc_amd -- MI355X-native hot path of the UVC variant caller (accumulate + score).

Only what the path needs: the HIP kernels + C ABI (csrc/), the ctypes binding (_ffi), the host
mirror of the reference's per-region surface (region), and the synthetic-region generator (synth).
"""
from .region import Region, UvcError, default_params, gpu_lib  # noqa: F401
from .synth import generate_region  # noqa: F401
