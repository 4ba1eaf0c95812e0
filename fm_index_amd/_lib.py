"""ctypes binding of libfmx.so (include/fmx.h).  Fails loudly when the HIP
library is missing: there is no Python / CPU fallback for any query."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfmx.so")
_lib = None

OK = 0
ERR_TEXT_START_ZERO = 1
ERR_TEXT_END_ZERO = 2
ERR_SYMBOL_RANGE = 3
ERR_ARG = 4
ERR_UNSUPPORTED = 5
ERR_HIP = 6
ERR_NO_LOCATE = 7
KIND_FM = 0
KIND_RLFM = 1
NO_LOCATE = 0xFFFFFFFF
FLAG_KEEP_SA = 1
FLAG_PAIR_INDEX = 2

# every symbol include/fmx.h declares: (name, restype, argtypes)
_V, _U64, _U32, _I, _D = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_double
SYMBOLS = [
    ("fmx_last_error", C.c_char_p, []),
    ("fmx_error_message", C.c_char_p, [_I]),
    ("fmx_build", _I, [_V, _U64, _U32, _U64, _U32, _U32, _U32, _I, C.POINTER(_V)]),
    ("fmx_build_dev", _I, [_V, _U64, _U32, _U64, _U32, _U32, _U32, _I, C.POINTER(_V)]),
    ("fmx_free", None, [_V]),
    ("fmx_len", _U64, [_V]),
    ("fmx_index_bytes", _U64, [_V]),
    ("fmx_max_character", _U64, [_V]),
    ("fmx_kind", _U32, [_V]),
    ("fmx_level", _U32, [_V]),
    ("fmx_device", _I, [_V]),
    ("fmx_get_l", _U64, [_V, _U64]),
    ("fmx_lf_map", _U64, [_V, _U64]),
    ("fmx_lf_map2", _U64, [_V, _U64, _U64]),
    ("fmx_get_sa", _U64, [_V, _U64]),
    ("fmx_get_l_batch_dev", _I, [_V, _V, _U64, _V, _V]),
    ("fmx_lf_map_batch_dev", _I, [_V, _V, _U64, _V, _V]),
    ("fmx_lf_map2_batch_dev", _I, [_V, _V, _V, _U64, _V, _V]),
    ("fmx_get_sa_batch_dev", _I, [_V, _V, _U64, _V, _V]),
    ("fmx_get_l_batch", _I, [_V, _V, _U64, _V]),
    ("fmx_lf_map_batch", _I, [_V, _V, _U64, _V]),
    ("fmx_lf_map2_batch", _I, [_V, _V, _V, _U64, _V]),
    ("fmx_get_sa_batch", _I, [_V, _V, _U64, _V]),
    ("fmx_count_batch_dev", _I, [_V, _V, _V, _U64, _V, _V, _V, _V, _V]),
    ("fmx_count_batch", _I, [_V, _V, _V, _U64, _V, _V, _V, _V]),
    ("fmx_stream_status", _I, [_V]),
    ("fmx_locate_batch_dev", _I, [_V, _V, _V, _U64, _V, _U64, _V, _V]),
    ("fmx_locate_batch", _I, [_V, _V, _V, _U64, _V, _V]),
    ("fmx_offsets_dev", _I, [_V, _V, _V, _U64, _V, _V]),
    ("fmx_set_timing", None, [_V, _I]),
    ("fmx_last_kernel_ms", _D, [_V]),
    ("fmx_last_steps", _U64, [_V]),
    ("fmx_build_ms", _D, [_V]),
    ("fmx_export_bwt", _I, [_V, _V]),
    ("fmx_export_cs", _I, [_V, _V]),
    ("fmx_export_sa_samples", _I, [_V, _V]),
    ("fmx_num_samples", _U64, [_V]),
    ("fmx_export_sa", _I, [_V, _V]),
    ("fmx_verify_sa", _I, [_V, C.POINTER(_U64)]),
    ("fmx_num_runs", _U64, [_V]),
    ("fmx_sym_bytes", _U32, [_V]),
    ("fmx_has_pair_index", _I, [_V]),
]


def build_library(force=False):
    """Compile fm_index_amd/libfmx.so for gfx950 with hipcc (in-tree)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", csrc])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "fm_index_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib
