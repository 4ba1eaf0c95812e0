"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

The C file oracle/fm_oracle.c restates the reference algorithm
(/root/reference/src/{wrapper,fm_index,rlfmi}.rs, suffix_array/sample.rs);
this module only marshals numpy arrays into it.  It is the checker for the
HIP path and the `cpu_baseline` row of bench.py -- never a product code path.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)


class _Backend(C.Structure):
    _fields_ = [("self", C.c_void_p), ("get_l", C.c_void_p), ("lf_map", C.c_void_p),
                ("lf_map2", C.c_void_p), ("len", C.c_void_p), ("get_sa", C.c_void_p),
                ("get_f", C.c_void_p), ("fl_map", C.c_void_p), ("piece_id", C.c_void_p),
                ("pieces_count", C.c_uint64), ("max_character", C.c_uint64)]


def build(native=False):
    """Compile the oracle. native=True: -march=native build in a temp dir (cpu_baseline)."""
    if native:
        out = os.path.join(tempfile.gettempdir(), "libfm_oracle_native_%d.so" % os.getuid())
        subprocess.check_call(["make", "-s", "-C", _HERE, "native", "OUT=" + out])
        return out
    subprocess.check_call(["make", "-s", "-C", _HERE, "libfm_oracle.so"])
    return os.path.join(_HERE, "libfm_oracle.so")


def _bind(lib):
    lib.orc_error_message.restype = C.c_char_p
    lib.orc_error_message.argtypes = [C.c_int]
    lib.orc_max_bits.restype = C.c_uint32
    lib.orc_max_bits.argtypes = [C.c_uint64]
    lib.orc_validate_text.argtypes = [C.c_void_p, C.c_uint64]
    lib.orc_suffix_array_naive.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_suffix_array.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_bucket_start.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    lib.orc_bwt.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.orc_fm_new.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    lib.orc_fm_from_bwt.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_uint64,
                                    C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_fm_from_bwt64.argtypes = lib.orc_fm_from_bwt.argtypes
    lib.orc_fm_free.argtypes = [C.c_void_p]
    lib.orc_fm_backend.restype = _Backend
    lib.orc_fm_backend.argtypes = [C.c_void_p]
    lib.orc_fm_heap_bytes.restype = C.c_uint64
    lib.orc_fm_heap_bytes.argtypes = [C.c_void_p]
    lib.orc_rlfm_new.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    lib.orc_rlfm_from_bwt.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_uint64,
                                      C.c_void_p, C.c_int]
    lib.orc_rlfm_from_bwt64.argtypes = lib.orc_rlfm_from_bwt.argtypes
    lib.orc_rlfm_free.argtypes = [C.c_void_p]
    lib.orc_rlfm_backend.restype = _Backend
    lib.orc_rlfm_backend.argtypes = [C.c_void_p]
    bp = C.POINTER(_Backend)
    lib.orc_count_batch.argtypes = [bp, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_locate_batch.argtypes = [bp, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                     C.c_void_p, C.c_int]
    lib.orc_lf_map2_batch.argtypes = [bp, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_lf_map_batch.argtypes = [bp, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_get_l_batch.argtypes = [bp, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_get_sa_batch.argtypes = [bp, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_get_f_batch.argtypes = [bp, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_piece_id_batch.argtypes = [bp, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_match_rows.restype = C.c_uint64
    lib.orc_match_rows.argtypes = [bp, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
    lib.orc_multi_new.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    lib.orc_multi_free.argtypes = [C.c_void_p]
    lib.orc_multi_backend.restype = _Backend
    lib.orc_multi_backend.argtypes = [C.c_void_p]
    lib.orc_fl_map_batch.argtypes = [bp, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_naive_search.restype = C.c_uint64
    lib.orc_naive_search.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                     C.c_uint64]
    lib.orc_max_threads.restype = C.c_int
    lib.orc_team_size.restype = C.c_int
    lib.orc_team_size.argtypes = [C.c_int]
    lib.orc_set_thread_spread.argtypes = [C.c_int]
    lib.orc_set_thread_spread.restype = None
    # wide symbols (u16 / u32 texts as uint32 arrays)
    lib.orc_suffix_array_naive_w.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_suffix_array_w.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_fm_new_w.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    lib.orc_rlfm_new_w.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_uint64, C.c_int]
    lib.orc_count_batch_w.argtypes = [bp, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_naive_search_w.restype = C.c_uint64
    lib.orc_naive_search_w.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                       C.c_uint64]
    return lib


def lib(native=False):
    global _LIB
    if native:
        return _bind(C.CDLL(build(native=True)))
    if _LIB is None:
        path = os.path.join(_HERE, "libfm_oracle.so")
        src = os.path.join(_HERE, "fm_oracle.c")
        if (not os.path.exists(path)) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        _LIB = _bind(C.CDLL(path))
    return _LIB


class OracleError(Exception):
    """Mirrors Error::InvalidText(msg) (src/error.rs:3-6)."""

    def __init__(self, code, msg):
        super().__init__("invalid text: " + msg)
        self.code = code
        self.msg = msg


def _u8(a):
    if isinstance(a, (bytes, bytearray)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _is_wide(a):
    return isinstance(a, np.ndarray) and a.dtype.itemsize > 1


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def pack_patterns(patterns):
    """list of byte strings -> (flat u8, offsets u64[npat+1])."""
    off = np.zeros(len(patterns) + 1, dtype=np.uint64)
    if len(patterns):
        off[1:] = np.cumsum([len(p) for p in patterns], dtype=np.uint64)
    flat = np.frombuffer(b"".join(bytes(p) for p in patterns), dtype=np.uint8).copy()
    if flat.size == 0:
        flat = np.zeros(1, dtype=np.uint8)
    return flat, off


def suffix_array(text, naive=False):
    if _is_wide(text):
        t = _u32(text)
        sa = np.zeros(max(len(t), 1), dtype=np.uint32)
        (lib().orc_suffix_array_naive_w if naive else lib().orc_suffix_array_w)(_p(t), len(t), _p(sa))
        return sa[:len(t)]
    t = _u8(text)
    sa = np.zeros(max(len(t), 1), dtype=np.uint32)
    (lib().orc_suffix_array_naive if naive else lib().orc_suffix_array)(_p(t), len(t), _p(sa))
    return sa[:len(t)]


def bucket_start(text, max_character):
    t = _u8(text)
    cs = np.zeros(max_character + 1, dtype=np.uint64)
    lib().orc_bucket_start(_p(t), len(t), max_character, _p(cs))
    return cs


def bwt(text, sa):
    t = _u8(text)
    out = np.zeros(max(len(t), 1), dtype=np.uint8)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    lib().orc_bwt(_p(t), len(t), _p(sa), _p(out))
    return out[:len(t)]


def naive_search(text, pattern):
    """NaiveSearchIndex::search (tests/testutil/mod.rs:62-86): ascending positions."""
    if _is_wide(text):
        t, p = _u32(text), _u32(pattern)
        cap = max(len(t), 1)
        out = np.zeros(cap, dtype=np.uint64)
        k = lib().orc_naive_search_w(_p(t), len(t), _p(p), len(p), _p(out), cap)
        return out[:k].copy()
    t, p = _u8(text), _u8(pattern)
    cap = max(len(t), 1)
    out = np.zeros(cap, dtype=np.uint64)
    k = lib().orc_naive_search(_p(t), len(t), _p(p), len(p), _p(out), cap)
    return out[:k].copy()


class OracleIndex:
    """FMIndex / FMIndexWithLocate / RLFMIndex / RLFMIndexWithLocate on the CPU oracle."""

    def __init__(self, text=None, max_character=255, level=None, kind="fm", _lib=None,
                 _from_bwt=None):
        self._l = _lib or lib()
        self.wide = False
        self.kind = kind
        self.max_character = int(max_character)
        h = C.c_void_p()
        lvl = -1 if level is None else int(level)
        if _from_bwt is not None:
            b, cs, samples = _from_bwt
            b = _u8(b)
            cs = np.ascontiguousarray(cs, dtype=np.uint64)
            big = len(b) >= (1 << 32)        # sample values need more than 32 bits (the reference is usize)
            sdt = np.uint64 if big else np.uint32
            sarr = np.ascontiguousarray(samples, dtype=sdt) if samples is not None else None
            sp = _p(sarr) if sarr is not None else None
            if kind == "rlfm":   # run-length structure of the same L column (rlfmi.rs:41-96)
                fn = self._l.orc_rlfm_from_bwt64 if big else self._l.orc_rlfm_from_bwt
                rc = fn(C.byref(h), _p(b), len(b), self.max_character, sp, lvl)
            else:
                fn = self._l.orc_fm_from_bwt64 if big else self._l.orc_fm_from_bwt
                rc = fn(C.byref(h), _p(b), len(b), self.max_character, _p(cs), sp, lvl)
        elif _is_wide(text):
            t = _u32(text)
            self._text_keep = t
            self.wide = True
            new = self._l.orc_fm_new_w if kind == "fm" else self._l.orc_rlfm_new_w
            rc = new(C.byref(h), _p(t), len(t), self.max_character, lvl)
        else:
            t = _u8(text)
            self._text_keep = t
            new = {"fm": self._l.orc_fm_new, "rlfm": self._l.orc_rlfm_new,
                   "multi": self._l.orc_multi_new}[kind]
            rc = new(C.byref(h), _p(t), len(t), self.max_character, lvl)
        if rc != 0:
            raise OracleError(rc, self._l.orc_error_message(rc).decode())
        self._h = h
        self._b = {"fm": self._l.orc_fm_backend, "rlfm": self._l.orc_rlfm_backend,
                   "multi": self._l.orc_multi_backend}[kind](h)
        self.has_locate = level is not None

    @classmethod
    def from_bwt(cls, bwt_arr, cs, max_character, samples=None, level=None, native=False, kind="fm"):
        return cls(max_character=max_character, level=level, kind=kind,
                   _lib=lib(native=True) if native else None, _from_bwt=(bwt_arr, cs, samples))

    def close(self):
        if getattr(self, "_h", None):
            {"fm": self._l.orc_fm_free, "rlfm": self._l.orc_rlfm_free,
             "multi": self._l.orc_multi_free}[self.kind](self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def heap_bytes(self):
        return int(self._l.orc_fm_heap_bytes(self._h)) if self.kind == "fm" else 0

    # --- batched driver -------------------------------------------------
    def count_batch(self, flat, off, s0e0=None, nthreads=1, want_steps=False):
        flat = _u32(flat) if self.wide else _u8(flat)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        npat = len(off) - 1
        s = np.zeros(max(npat, 1), dtype=np.uint64)
        e = np.zeros(max(npat, 1), dtype=np.uint64)
        st = np.zeros(max(npat, 1), dtype=np.uint64)
        se = None if s0e0 is None else np.ascontiguousarray(s0e0, dtype=np.uint64)
        fn = self._l.orc_count_batch_w if self.wide else self._l.orc_count_batch
        rc = fn(C.byref(self._b), _p(flat), _p(off), npat,
                                     None if se is None else _p(se), _p(s), _p(e), _p(st),
                                     nthreads)
        if rc != 0:
            raise OracleError(rc, self._l.orc_error_message(rc).decode())
        if want_steps:
            return s[:npat], e[:npat], st[:npat]
        return s[:npat], e[:npat]

    def set_thread_spread(self, on):
        """pin the threads of count_batch one per CPU, spread over the allowed CPUs (cpu_baseline)"""
        self._l.orc_set_thread_spread(1 if on else 0)

    def team_size(self, nthreads):
        """threads an OpenMP team of `nthreads` really gets on this box"""
        return int(self._l.orc_team_size(int(nthreads)))

    def locate_batch(self, s, e, nthreads=1):
        s = np.ascontiguousarray(s, dtype=np.uint64)
        e = np.ascontiguousarray(e, dtype=np.uint64)
        cnt = (e - s).astype(np.uint64)
        off = np.zeros(len(s) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(cnt, dtype=np.uint64)
        pos = np.zeros(max(int(off[-1]), 1), dtype=np.uint64)
        self._l.orc_locate_batch(C.byref(self._b), _p(s), _p(e), len(s), _p(off), _p(pos), nthreads)
        return off, pos[:int(off[-1])]

    # --- reference-shaped convenience ------------------------------------
    def search(self, pattern, s0e0=None):
        if self.wide:
            flat = _u32(pattern) if len(pattern) else np.zeros(1, dtype=np.uint32)
            off = np.array([0, len(pattern)], dtype=np.uint64)
        else:
            flat, off = pack_patterns([pattern])
        se = None if s0e0 is None else np.array(s0e0, dtype=np.uint64)
        s, e = self.count_batch(flat, off, se)
        return int(s[0]), int(e[0])

    def count(self, pattern):
        s, e = self.search(pattern)
        return e - s

    def locate(self, pattern):
        s, e = self.search(pattern)
        _, pos = self.locate_batch([s], [e])
        return [int(x) for x in pos]

    def __len__(self):
        fn = C.CFUNCTYPE(C.c_uint64, C.c_void_p)(self._b.len)
        return int(fn(self._b.self))

    def lf_map2(self, c, i):
        c = np.ascontiguousarray(c, dtype=np.uint64)
        i = np.ascontiguousarray(i, dtype=np.uint64)
        out = np.zeros(max(len(i), 1), dtype=np.uint64)
        self._l.orc_lf_map2_batch(C.byref(self._b), _p(c), _p(i), len(i), _p(out))
        return out[:len(i)]

    def lf_map(self, i):
        i = np.ascontiguousarray(i, dtype=np.uint64)
        out = np.zeros(max(len(i), 1), dtype=np.uint64)
        self._l.orc_lf_map_batch(C.byref(self._b), _p(i), len(i), _p(out))
        return out[:len(i)]

    def get_l(self, i):
        i = np.ascontiguousarray(i, dtype=np.uint64)
        out = np.zeros(max(len(i), 1), dtype=np.uint64)
        self._l.orc_get_l_batch(C.byref(self._b), _p(i), len(i), _p(out))
        return out[:len(i)]

    def get_sa(self, i):
        i = np.ascontiguousarray(i, dtype=np.uint64)
        out = np.zeros(max(len(i), 1), dtype=np.uint64)
        self._l.orc_get_sa_batch(C.byref(self._b), _p(i), len(i), _p(out))
        return out[:len(i)]

    def get_f(self, i):
        i = np.ascontiguousarray(i, dtype=np.uint64)
        out = np.zeros(max(len(i), 1), dtype=np.uint64)
        self._l.orc_get_f_batch(C.byref(self._b), _p(i), len(i), _p(out))
        return out[:len(i)]

    def fl_map(self, i):
        i = np.ascontiguousarray(i, dtype=np.uint64)
        out = np.zeros(max(len(i), 1), dtype=np.uint64)
        self._l.orc_fl_map_batch(C.byref(self._b), _p(i), len(i), _p(out))
        return out[:len(i)]

    # --- multi-pieces (multi_pieces.rs) ---
    def pieces_count(self):
        return int(self._b.pieces_count)

    def piece_id(self, i):
        i = np.ascontiguousarray(i, dtype=np.uint64)
        out = np.zeros(max(len(i), 1), dtype=np.uint64)
        self._l.orc_piece_id_batch(C.byref(self._b), _p(i), len(i), _p(out))
        return out[:len(i)]

    def match_rows(self, s, e, prefix_only=False):
        cap = max(int(e) - int(s), 1)
        out = np.zeros(cap, dtype=np.uint64)
        k = self._l.orc_match_rows(C.byref(self._b), int(s), int(e), int(prefix_only), _p(out), cap)
        return out[:k].copy()
