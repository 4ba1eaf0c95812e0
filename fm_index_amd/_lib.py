"""ctypes binding of libfmx.so (include/fmx.h).  Fails loudly when the HIP
library is missing: there is no Python / CPU fallback for any query."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# FMX_LIB=<path> selects another build of the same ABI (e.g. the range-checked libfmx_debug.so)
LIB_PATH = os.environ.get("FMX_LIB") or os.path.join(_HERE, "libfmx.so")
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
KIND_MULTI = 2
NO_LOCATE = 0xFFFFFFFF
FLAG_KEEP_SA = 1
FLAG_PAIR_INDEX = 2
FLAG_KMER_TABLE = 4
FLAG_TEXT_ORDER = 8
FLAG_ROW_ORDER = 16
FLAG_FORCE_WIDE = 32
FLAG_NO_WALK_RECORDS = 64
FLAG_AUTO = 128
FLAG_KEEP_SCRATCH = 256
FLAG_RUN_TABLE = 512
FLAG_PLAIN = 1024

# every symbol include/fmx.h declares: (name, restype, argtypes)
_V, _U64, _U32, _I, _D = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_double
SYMBOLS = [
    ("fmx_last_error", C.c_char_p, []),
    ("fmx_error_message", C.c_char_p, [_I]),
    ("fmx_release_scratch", None, []),
    ("fmx_build", _I, [_V, _U64, _U32, _U64, _U32, _U32, _U32, _I, C.POINTER(_V)]),
    ("fmx_build_dev", _I, [_V, _U64, _U32, _U64, _U32, _U32, _U32, _I, C.POINTER(_V)]),
    ("fmx_free", None, [_V]),
    ("fmx_save", _I, [_V, C.c_char_p]),
    ("fmx_load", _I, [C.c_char_p, _I, C.POINTER(_V)]),
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
    ("fmx_get_f", _U64, [_V, _U64]),
    ("fmx_fl_map", _U64, [_V, _U64]),
    ("fmx_get_l_batch_dev", _I, [_V, _V, _U64, _V, _V]),
    ("fmx_lf_map_batch_dev", _I, [_V, _V, _U64, _V, _V]),
    ("fmx_lf_map2_batch_dev", _I, [_V, _V, _V, _U64, _V, _V]),
    ("fmx_get_sa_batch_dev", _I, [_V, _V, _U64, _V, _V]),
    ("fmx_get_f_batch_dev", _I, [_V, _V, _U64, _V, _V]),
    ("fmx_fl_map_batch_dev", _I, [_V, _V, _U64, _V, _V]),
    ("fmx_get_l_batch", _I, [_V, _V, _U64, _V]),
    ("fmx_lf_map_batch", _I, [_V, _V, _U64, _V]),
    ("fmx_lf_map2_batch", _I, [_V, _V, _V, _U64, _V]),
    ("fmx_get_sa_batch", _I, [_V, _V, _U64, _V]),
    ("fmx_get_f_batch", _I, [_V, _V, _U64, _V]),
    ("fmx_fl_map_batch", _I, [_V, _V, _U64, _V]),
    ("fmx_extract_batch_dev", _I, [_V, _V, _U64, _U64, _I, _V, _V, _V, _V]),
    ("fmx_extract_batch", _I, [_V, _V, _U64, _U64, _I, _V, _V, _V]),
    ("fmx_count_batch_dev", _I, [_V, _V, _V, _U64, _V, _V, _V, _V, _V]),
    ("fmx_count_batch", _I, [_V, _V, _V, _U64, _V, _V, _V, _V]),
    ("fmx_stream_status", _I, [_V]),
    ("fmx_locate_batch_dev", _I, [_V, _V, _V, _U64, _V, _U64, _V, _V]),
    ("fmx_locate_batch", _I, [_V, _V, _V, _U64, _V, _V]),
    ("fmx_offsets_dev", _I, [_V, _V, _V, _U64, _V, _V]),
    ("fmx_locate_workspace_bytes", _U64, [_V, _U64]),
    ("fmx_locate_batch_ws_dev", _I, [_V, _V, _V, _U64, _V, _U64, _V, _V, _U64, _V]),
    ("fmx_offsets_workspace_bytes", _U64, [_U64]),
    ("fmx_offsets_ws_dev", _I, [_V, _V, _V, _U64, _V, _V, _U64, _V]),
    ("fmx_replicate", _I, [_V, _I, C.POINTER(_V)]),
    ("fmx_shard_range", None, [_U64, _U32, _U32, C.POINTER(_U64), C.POINTER(_U64)]),
    ("fmx_count_batch_multi", _I, [_V, _U32, _V, _V, _U64, _V, _V, _V, _V]),
    ("fmx_count_batch_multi_resident", _I, [_V, _U32, _V, _V, _U64, _V, _V, _V, _V]),
    ("fmx_locate_batch_multi", _I, [_V, _U32, _V, _V, _U64, _V, _V]),
    ("fmx_set_timing", None, [_V, _I]),
    ("fmx_last_kernel_ms", _D, [_V]),
    ("fmx_series_kernel_ms", _D, [_V]),
    ("fmx_last_steps", _U64, [_V]),
    ("fmx_build_ms", _D, [_V]),
    ("fmx_export_bwt", _I, [_V, _V]),
    ("fmx_export_cs", _I, [_V, _V]),
    ("fmx_export_sa_samples", _I, [_V, _V]),
    ("fmx_export_sa_samples64", _I, [_V, _V]),
    ("fmx_num_samples", _U64, [_V]),
    ("fmx_export_sa", _I, [_V, _V]),
    ("fmx_verify_sa", _I, [_V, C.POINTER(_U64)]),
    ("fmx_num_runs", _U64, [_V]),
    ("fmx_pieces_count", _U64, [_V]),
    ("fmx_piece_id", _U64, [_V, _U64]),
    ("fmx_piece_id_batch_dev", _I, [_V, _V, _U64, _V, _V]),
    ("fmx_piece_id_batch", _I, [_V, _V, _U64, _V]),
    ("fmx_match_counts_dev", _I, [_V, _V, _V, _U64, _I, _V, _V]),
    ("fmx_match_counts", _I, [_V, _V, _V, _U64, _I, _V]),
    ("fmx_match_rows_dev", _I, [_V, _V, _V, _U64, _I, _V, _V, _V]),
    ("fmx_match_rows", _I, [_V, _V, _V, _U64, _I, _V, _V]),
    ("fmx_sym_bytes", _U32, [_V]),
    ("fmx_has_pair_index", _I, [_V]),
    ("fmx_is_wide", _I, [_V]),
    ("fmx_text_order", _I, [_V]),
    ("fmx_walk_records", _I, [_V]),
    ("fmx_kmer_k", _U32, [_V]),
]


def csrc_hash():
    """content hash of the HIP sources (the .git directory does not travel to the GPU box):
    profiles/traffic.json entries are only trusted for the source tree they were measured on."""
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(_HERE, "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def build_library(force=False):
    """Compile fm_index_amd/libfmx.so for gfx950 with hipcc (in-tree)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", csrc, "all"])
    # the census library (line log for bench.py's byte model) and the measurement library (FMX_VARIANT
    # switches for A/B runs and for tests that force a code path the shipped dispatch would not pick):
    # measurement only, never the product path -- real file targets (a second call rebuilds nothing), and
    # a failure in either does not fail the product build
    rc = subprocess.call(["make", "-s", "-j4", "-C", csrc, "census", "measure"])
    if rc != 0:
        import warnings
        warnings.warn("fm_index_amd: the measurement-only libraries (libfmx_census.so / libfmx_measure.so) "
                      "did not build (make rc %d); the product library is unaffected" % rc)
    return LIB_PATH


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (soname
    libamdhip64.so.7, the soname libfmx.so needs).  If torch is imported first, libfmx binds to
    that copy and everything -- including torch streams handed to the *_dev entry points -- lives
    in one runtime.  If libfmx were loaded first it would pull /opt/rocm's copy and a later
    `import torch` would bring up a second runtime (which then finds no device).  So when a torch
    wheel with a bundled runtime is installed, load that copy first (torch itself is NOT
    imported).  FMX_HIP_RUNTIME=<path> overrides; FMX_HIP_RUNTIME=system skips this."""
    forced = os.environ.get("FMX_HIP_RUNTIME")
    if forced == "system":
        return
    path = forced
    if not path:
        try:
            import importlib.util
            spec = importlib.util.find_spec("torch")
            if spec and spec.submodule_search_locations:
                cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
                if os.path.exists(cand):
                    path = cand
        except Exception:  # noqa: BLE001
            path = None
    if path:
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "fm_index_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        _preload_hip_runtime()
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib
