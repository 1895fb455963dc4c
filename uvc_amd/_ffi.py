"""ctypes mirror of include/uvcgpu.h (struct layouts are generated from include/uvc_params.def).

`Lib` binds any shared library that exports the uvcgpu.h entry points under a prefix: the product library
(uvc_amd/csrc/libuvcgpu.so, prefix `uvcgpu_`) and, in tests / smoke / bench cpu_baseline only, the checker of the
same ABI (whose path lives with it, outside this package -- nothing here knows where it is).
"""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read_params_def():
    ints, dbls = [], []
    with open(os.path.join(ROOT, "include", "uvc_params.def")) as fh:
        for line in fh:
            m = re.match(r"UVC_P([ID])\((\w+),\s*(.*)\)\s*$", line)
            if not m:
                continue
            (ints if m.group(1) == "I" else dbls).append((m.group(2), eval(m.group(3))))
    return ints, dbls


PARAM_INTS, PARAM_DBLS = _read_params_def()


class UvcParams(C.Structure):
    _fields_ = ([("struct_size", C.c_int32), ("reserved_", C.c_int32)]
                + [(n, C.c_int32) for n, _ in PARAM_INTS] + [("pad_to_8_", C.c_int32)]
                + [(n, C.c_double) for n, _ in PARAM_DBLS])


class UvcReadSoA(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("reserved_", C.c_int32),
        ("n_reads", C.c_int64),
        ("pos", C.c_void_p), ("mpos", C.c_void_p), ("isize", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p),
        ("nm", C.c_void_p), ("l_qseq", C.c_void_p), ("seq_off", C.c_void_p), ("cigar_off", C.c_void_p), ("n_cigar", C.c_void_p),
        ("frag_id", C.c_void_p), ("fam_id", C.c_void_p), ("fam_strand", C.c_void_p),
        ("n_bases", C.c_int64), ("bases", C.c_void_p), ("quals", C.c_void_p),
        ("n_cigar_ops", C.c_int64), ("cigars", C.c_void_p),
        ("n_fams", C.c_int32), ("fam_dflag", C.c_void_p),
        ("bases4", C.c_void_p), ("n_bases4_bytes", C.c_int64),
    ]


class UvcIndelAllele(C.Structure):
    _fields_ = [("refpos", C.c_int32), ("symbol", C.c_int32), ("bDPa", C.c_int32), ("cDP0a", C.c_int32), ("indel_len", C.c_int32)]


class UvcTumorKey(C.Structure):
    _fields_ = [("refpos", C.c_int32), ("symbol", C.c_int32), ("cDP1x", C.c_int32), ("CDP1x", C.c_int32), ("bDP", C.c_int32), ("BDP", C.c_int32),
                ("tier2", C.c_int32), ("indel_len", C.c_int32)]
    _fields_ += [(n, C.c_int32) for n in ("cVQ1", "cPCQ1", "cDP2x", "CDP2x", "cVQ2", "cPCQ2", "bNMQ", "vHGQ", "tDP", "tAD0", "tAD1", "t2DP")]


class UvcGapRow(C.Structure):
    _fields_ = [("refpos", C.c_int32), ("symbol", C.c_int32), ("strand", C.c_int32), ("len", C.c_int32), ("seq_off", C.c_int64),
                ("bAD1", C.c_int32), ("cAD1", C.c_int32), ("c2AD", C.c_int32), ("c2dAD", C.c_int32)]


class UvcHapLink(C.Structure):
    _fields_ = [("which", C.c_int32), ("n_muts", C.c_int32), ("mut_off", C.c_int64), ("fr_cnt", C.c_int32 * 2), ("other_cnt", C.c_int32 * 2)]


class UvcScoreRequest(C.Structure):
    _fields_ = [("pos_beg", C.c_int32), ("pos_end", C.c_int32), ("all_out", C.c_int32), ("is_amplicon", C.c_int32),
                ("n_indel_alleles", C.c_int64), ("indel_alleles", C.c_void_p), ("n_tumor_keys", C.c_int64), ("tumor_keys", C.c_void_p),
                ("release_state", C.c_int32), ("base_at_pos_beg", C.c_int32), ("region_beg", C.c_int32), ("kept_only", C.c_int32),
                ("tumor_sample_columns", C.c_void_p), ("tumor_ref_alt", C.c_void_p)]


class UvcScoreOut(C.Structure):
    _fields_ = [("capacity", C.c_int64), ("n_records", C.c_int64), ("fields", C.c_void_p)]


def _parse_enums():
    """Pulls every `NAME = value` / implicit enumerator of include/uvcgpu.h into a dict."""
    txt = open(os.path.join(ROOT, "include", "uvcgpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for body in re.findall(r"enum\s*\w*\s*\{(.*?)\}", txt, flags=re.S):
        val = -1
        for item in body.split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, v = [t.strip() for t in item.split("=")]
                val = int(eval(v, {}, out))
            else:
                name = item
                val += 1
            out[name] = val
    return out


ENUMS = _parse_enums()
NSYM = ENUMS["UVC_NUM_SYMBOLS"]

# field group -> (dtype, planes per position as a shape prefix)
import numpy as _np
FIELD_GROUPS = {
    "PREP32": (ENUMS["UVC_F_PREP32"], _np.int32, (ENUMS["UVC_NPREP32"],)),
    "PREP64": (ENUMS["UVC_F_PREP64"], _np.int64, (ENUMS["UVC_NPREP64"],)),
    "THRES": (ENUMS["UVC_F_THRES"], _np.int32, (ENUMS["UVC_NTHRES"],)),
    "SEG32": (ENUMS["UVC_F_SEG32"], _np.int32, (ENUMS["UVC_NSEG32"], NSYM)),
    "SEG64": (ENUMS["UVC_F_SEG64"], _np.int64, (ENUMS["UVC_NSEG64"], NSYM)),
    "VQ": (ENUMS["UVC_F_VQ"], _np.int32, (ENUMS["UVC_NVQ"], NSYM)),
    "BQSUM": (ENUMS["UVC_F_BQSUM"], _np.int32, (NSYM,)),
    "FRAG": (ENUMS["UVC_F_FRAG"], _np.int32, (2, ENUMS["UVC_NFRAG"], NSYM)),
    "FAM": (ENUMS["UVC_F_FAM"], _np.int32, (2, ENUMS["UVC_NFAM"], NSYM)),
    "FAMINFO32": (ENUMS["UVC_F_FAMINFO32"], _np.int32, (ENUMS["UVC_NFAMINFO32"], NSYM)),
    "FAMINFO64": (ENUMS["UVC_F_FAMINFO64"], _np.int64, (ENUMS["UVC_NFAMINFO64"], NSYM)),
    "DUPLEX": (ENUMS["UVC_F_DUPLEX"], _np.int32, (ENUMS["UVC_NDUPLEX"], NSYM)),
    "RTR": (ENUMS["UVC_F_RTR"], _np.int32, (ENUMS["UVC_NRTR"],)),
    "BAQ": (ENUMS["UVC_F_BAQ"], _np.int64, (2,)),
}
SCORE_FIELDS = [k[len("UVC_O_"):] for k, v in sorted(((k, v) for k, v in ENUMS.items() if k.startswith("UVC_O_")), key=lambda kv: kv[1])]
NUM_SCORE_FIELDS = ENUMS["UVC_NUM_SCORE_FIELDS"]


class Lib:
    """Binds one shared library exporting the uvcgpu.h entry points under `prefix`."""

    def __init__(self, path, prefix):
        self.path, self.prefix = path, prefix
        self.dll = C.CDLL(path)
        f = self._f
        f("params_default", None, [C.POINTER(UvcParams)])
        f("last_error", C.c_char_p, [])
        create = "region_create" if prefix == "uvcgpu_" else "create"
        self.n = dict(create=create,
                      set_reads="region_set_reads" if prefix == "uvcgpu_" else "set_reads",
                      accumulate="region_accumulate" if prefix == "uvcgpu_" else "accumulate",
                      field_bytes="region_field_bytes" if prefix == "uvcgpu_" else "field_bytes",
                      fetch="region_fetch" if prefix == "uvcgpu_" else "fetch",
                      score="region_score" if prefix == "uvcgpu_" else "score",
                      destroy="region_destroy" if prefix == "uvcgpu_" else "destroy")
        f(self.n["create"], C.c_int, [C.POINTER(C.c_void_p), C.POINTER(UvcParams), C.c_int32, C.c_int32, C.c_int32, C.c_char_p])
        f(self.n["set_reads"], C.c_int, [C.c_void_p, C.POINTER(UvcReadSoA)])
        f(self.n["accumulate"], C.c_int, [C.c_void_p])
        f(self.n["field_bytes"], C.c_int64, [C.c_void_p, C.c_int32])
        f(self.n["fetch"], C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64])
        f(self.n["score"], C.c_int, [C.c_void_p, C.POINTER(UvcScoreRequest), C.POINTER(UvcScoreOut)])
        f(self.n["destroy"], None, [C.c_void_p])

    def _f(self, name, restype, argtypes):
        fn = getattr(self.dll, self.prefix + name)
        fn.restype, fn.argtypes = restype, argtypes
        return fn

    def call(self, key, *args):
        return getattr(self.dll, self.prefix + self.n.get(key, key))(*args)

    def last_error(self):
        e = getattr(self.dll, self.prefix + "last_error")()
        return e.decode() if e else ""


def gpu_library_path():
    # UVCGPU_LIBRARY: another build of the same ABI (A/B measurements of a kernel variant on one GPU box: scripts/gpu_ab_*.sh)
    return os.environ.get("UVCGPU_LIBRARY") or os.path.join(ROOT, "uvc_amd", "csrc", "libuvcgpu.so")
