"""fm_index_amd -- host-side mirror of the reference's Search / Match /
MatchWithLocate surface (reference: src/frontend.rs:26-98, 110-243) over the
C ABI of libfmx.so (include/fmx.h).  Same names, argument meaning and error
behaviour as the Rust crate for the count / locate path:

    text  = Text(b"mississippi\\0")            # Text::new            (text.rs:28-33)
    index = FMIndexWithLocate(text, 2)         # FMIndexWithLocate::new (frontend.rs:213-221)
    s = index.search(b"ssi")                   # SearchIndex::search  (frontend.rs:30-34)
    s.count()                                  # Search::count        (frontend.rs:80)
    [m.locate() for m in s.iter_matches()]     # MatchWithLocate::locate (frontend.rs:96-98)
    index.search(b"i").search(b"ss")           # refinement prepends  (wrapper.rs:99-124)

plus the batched forms the GPU exists for (`search_many`, `locate_many`).
Every query runs in HIP kernels; importing works without a GPU, calling does not.
"""
import ctypes as C

import numpy as np

from . import _lib as L

__all__ = ["Text", "Error", "FMIndex", "FMIndexWithLocate", "RLFMIndex", "RLFMIndexWithLocate",
           "FMIndexMultiPieces", "FMIndexMultiPiecesWithLocate", "Search", "Match", "SearchBatch",
           "Replicas", "pack_patterns"]


class Error(Exception):
    """Error::InvalidText(msg) (src/error.rs:3-15); other ABI failures carry their code."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def _check(rc):
    if rc != L.OK:
        raise Error(rc, L.lib().fmx_last_error().decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_DTYPES = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}


def _sym(x, dtype=None):
    """symbols as a contiguous array: bytes/str -> u8; unsigned arrays keep their width
    (Character = u8 / u16 / u32 / u64, character.rs:38-42) unless `dtype` is given.  Anything else
    (Python lists, signed or object arrays) gets the narrowest unsigned width that holds its
    maximum; a negative value, or a value that does not fit a forced `dtype`, is refused with
    ERR_SYMBOL_RANGE instead of wrapping (the reference's typed Character cannot hold it at all)."""
    if isinstance(x, str):
        x = x.encode("latin-1")
    if isinstance(x, (bytes, bytearray, memoryview)):
        x = np.frombuffer(bytes(x), dtype=np.uint8)
    x = np.asarray(x)
    if x.dtype.kind == "u" and x.dtype.itemsize in _DTYPES:
        if dtype is None or np.dtype(dtype) == x.dtype:
            return np.ascontiguousarray(x)
    elif x.size and x.dtype.kind not in "iub":
        raise Error(L.ERR_SYMBOL_RANGE, "symbols must be non-negative integers")
    lo = int(x.min()) if x.size else 0
    hi = int(x.max()) if x.size else 0
    if lo < 0:
        raise Error(L.ERR_SYMBOL_RANGE, "negative symbol %d" % lo)
    if dtype is None:
        dtype = np.uint8 if hi <= 0xFF else (np.uint16 if hi <= 0xFFFF else
                                             (np.uint32 if hi <= 0xFFFFFFFF else np.uint64))
    elif hi > int(np.iinfo(dtype).max):
        raise Error(L.ERR_SYMBOL_RANGE, "symbol %d does not fit %s" % (hi, np.dtype(dtype).name))
    return np.ascontiguousarray(x, dtype=dtype)


def _u8(x):
    return _sym(x, np.uint8)


def pack_patterns(patterns, dtype=np.uint8):
    """list of symbol strings -> (flat symbols of `dtype`, offsets u64[npat+1])."""
    pats = [_sym(p, dtype) for p in patterns]
    off = np.zeros(len(pats) + 1, dtype=np.uint64)
    if pats:
        off[1:] = np.cumsum([len(p) for p in pats], dtype=np.uint64)
    flat = np.concatenate(pats) if pats and int(off[-1]) else np.zeros(1, dtype=dtype)
    return np.ascontiguousarray(flat, dtype=dtype), off


class Text:
    """Text (text.rs:11-64): the symbols INCLUDING the trailing 0, plus max_character."""

    def __init__(self, text, max_character=None):
        self._t = _sym(text)
        if max_character is None:  # Text::new: C::max_value() (text.rs:28-33); the engine caps
            max_character = min(int(np.iinfo(self._t.dtype).max), (1 << 26) - 1)  # tables at 2^26
        self._max = int(max_character)

    @classmethod
    def with_max_character(cls, text, max_character):  # text.rs:44-49
        return cls(text, max_character)

    def text(self):
        return self._t

    def max_character(self):
        return self._max


def _flags(keep_sa, pair_index, kmer_table, sampling, force_wide=False, walk_records=True, auto=False,
           keep_scratch=False, run_table=False, plain=False):
    """build flags of include/fmx.h; sampling: None (the builder's choice), "text" or "row"
    (FMX_FLAG_TEXT_ORDER / FMX_FLAG_ROW_ORDER: which rows carry a suffix-array sample)."""
    if sampling not in (None, "text", "row"):
        raise ValueError("sampling must be None, 'text' or 'row'")
    return ((L.FLAG_KEEP_SA if keep_sa else 0) | (L.FLAG_PAIR_INDEX if pair_index else 0) |
            (L.FLAG_KMER_TABLE if kmer_table else 0) | (L.FLAG_TEXT_ORDER if sampling == "text" else 0) |
            (L.FLAG_ROW_ORDER if sampling == "row" else 0) | (L.FLAG_FORCE_WIDE if force_wide else 0) |
            (0 if walk_records else L.FLAG_NO_WALK_RECORDS) | (L.FLAG_AUTO if auto else 0) |
            (L.FLAG_KEEP_SCRATCH if keep_scratch else 0) | (L.FLAG_RUN_TABLE if run_table else 0) |
            (L.FLAG_PLAIN if plain else 0))


class _Index:
    _kind = L.KIND_FM

    def __init__(self, text, level=None, device=0, keep_sa=False, pair_index=False, kmer_table=False,
                 sampling=None, force_wide=False, walk_records=True, auto=False, run_table=False, plain=False):
        """plain=True: FMX_FLAG_PLAIN -- no count accelerators unless asked for by name (the default DNA-like FM index of
        2^24+ symbols gets the pair index and the k-mer start table when the device has room; same results)"""
        if not isinstance(text, Text):
            text = Text(text)
        self._lib = L.lib()
        self._h = C.c_void_p()
        t = text.text()
        self._dtype = t.dtype
        lvl = L.NO_LOCATE if level is None else int(level)
        rc = self._lib.fmx_build(_p(t) if len(t) else None, len(t), t.dtype.itemsize,
                                 text.max_character(),
                                 self._kind, lvl,
                                 _flags(keep_sa, pair_index, kmer_table, sampling, force_wide, walk_records, auto,
                                        run_table=run_table, plain=plain),
                                 device, C.byref(self._h))
        _check(rc)

    @classmethod
    def from_device_text(cls, d_text_ptr, n, max_character, level=None, device=0, keep_sa=False,
                         pair_index=False, sym_bytes=1, kmer_table=False, sampling=None, force_wide=False,
                         walk_records=True, auto=False, keep_scratch=False, run_table=False, plain=False):
        """text already resident in HBM (e.g. a torch uint8 tensor's data_ptr())."""
        self = cls.__new__(cls)
        self._lib = L.lib()
        self._h = C.c_void_p()
        self._dtype = np.dtype(_DTYPES[sym_bytes])
        lvl = L.NO_LOCATE if level is None else int(level)
        _check(self._lib.fmx_build_dev(C.c_void_p(d_text_ptr), n, sym_bytes, max_character, cls._kind, lvl,
                                       _flags(keep_sa, pair_index, kmer_table, sampling, force_wide, walk_records, auto,
                                              keep_scratch, run_table, plain),
                                       device, C.byref(self._h)))
        return self

    def save(self, path):
        """write the flat index file (header + HBM arrays)."""
        _check(self._lib.fmx_save(self._h, str(path).encode()))

    @classmethod
    def load(cls, path, device=0):
        """upload a saved index to `device` without rebuilding."""
        self = cls.__new__(cls)
        self._lib = L.lib()
        self._h = C.c_void_p()
        _check(self._lib.fmx_load(str(path).encode(), device, C.byref(self._h)))
        self._dtype = np.dtype(_DTYPES[int(self._lib.fmx_sym_bytes(self._h))])
        return self

    # -- SearchIndex (frontend.rs:26-44) --
    def search(self, pattern):
        return Search(self, None, None).search(pattern)

    def match_rows_many(self, s, e, prefix_only=False):
        """rows iter_matches() visits for every [s, e): (offsets, rows) (wrapper.rs:203-217)."""
        s = np.ascontiguousarray(s, dtype=np.uint64)
        e = np.ascontiguousarray(e, dtype=np.uint64)
        cnt = np.zeros(max(len(s), 1), dtype=np.uint64)
        _check(self._lib.fmx_match_counts(self._h, _p(s), _p(e), len(s), int(prefix_only), _p(cnt)))
        off = np.zeros(len(s) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(cnt[:len(s)], dtype=np.uint64)
        rows = np.zeros(max(int(off[-1]), 1), dtype=np.uint64)
        _check(self._lib.fmx_match_rows(self._h, _p(s), _p(e), len(s), int(prefix_only), _p(off), _p(rows)))
        return off, rows[:int(off[-1])]

    def len(self):
        return int(self._lib.fmx_len(self._h))

    __len__ = len

    def heap_size(self):
        return int(self._lib.fmx_index_bytes(self._h))

    # -- batched --
    def search_many(self, patterns=None, flat=None, off=None, s0e0=None):
        """count for a batch: returns SearchBatch (s, e, counts as numpy u64)."""
        if flat is None:
            flat, off = pack_patterns(patterns, self._dtype)
        flat = _sym(flat, self._dtype)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        npat = len(off) - 1
        s = np.zeros(max(npat, 1), dtype=np.uint64)
        e = np.zeros(max(npat, 1), dtype=np.uint64)
        c = np.zeros(max(npat, 1), dtype=np.uint64)
        se = None if s0e0 is None else np.ascontiguousarray(s0e0, dtype=np.uint64)
        _check(self._lib.fmx_count_batch(self._h, _p(flat), _p(off), npat, _p(se), _p(s), _p(e),
                                         _p(c)))
        return SearchBatch(self, s[:npat], e[:npat], c[:npat])

    def locate_many(self, s, e):
        """(offsets u64[npat+1], positions u64[total]) in the reference's iteration order."""
        s = np.ascontiguousarray(s, dtype=np.uint64)
        e = np.ascontiguousarray(e, dtype=np.uint64)
        off = np.zeros(len(s) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(e - s, dtype=np.uint64)
        pos = np.zeros(max(int(off[-1]), 1), dtype=np.uint64)
        _check(self._lib.fmx_locate_batch(self._h, _p(s), _p(e), len(s), _p(off), _p(pos)))
        return off, pos[:int(off[-1])]

    # -- the backend trait, batched (backend.rs:9-15, 29-31) --
    def _scalar(self, fn, *arrs):
        arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in arrs]
        k = len(arrs[-1])
        out = np.zeros(max(k, 1), dtype=np.uint64)
        _check(fn(self._h, *[_p(a) for a in arrs], k, _p(out)))
        return out[:k]

    def get_l(self, i):
        return self._scalar(self._lib.fmx_get_l_batch, i)

    def lf_map(self, i):
        return self._scalar(self._lib.fmx_lf_map_batch, i)

    def lf_map2(self, c, i):
        return self._scalar(self._lib.fmx_lf_map2_batch, c, i)

    def get_sa(self, i):
        return self._scalar(self._lib.fmx_get_sa_batch, i)

    def get_f(self, i):
        return self._scalar(self._lib.fmx_get_f_batch, i)

    def fl_map(self, i):
        return self._scalar(self._lib.fmx_fl_map_batch, i)

    def extract_many(self, rows, length, forward=False):
        """iter_chars_backward / iter_chars_forward for many rows in one launch (wrapper.rs:142-183):
        (syms[nrows, length], lens[nrows], next_rows[nrows]).  Forward on a multi-pieces index ends
        at the piece end: lens < length there, the unused slots are 0 and next is 2^64-1."""
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        k, length = len(rows), int(length)
        dt = _DTYPES[int(self._lib.fmx_sym_bytes(self._h))]
        syms = np.zeros((max(k, 1), max(length, 1)), dtype=dt)
        lens = np.zeros(max(k, 1), dtype=np.uint64)
        nxt = np.zeros(max(k, 1), dtype=np.uint64)
        if length:
            syms = np.zeros((max(k, 1), length), dtype=dt)
        _check(self._lib.fmx_extract_batch(self._h, _p(rows), k, length, 1 if forward else 0,
                                           _p(syms), _p(lens), _p(nxt)))
        return syms[:k, :length], lens[:k], nxt[:k]

    # -- export / checks --
    def export_bwt(self):
        out = np.zeros(max(self.len(), 1), dtype=_DTYPES[int(self._lib.fmx_sym_bytes(self._h))])
        _check(self._lib.fmx_export_bwt(self._h, _p(out)))
        return out[:self.len()]

    def export_cs(self):
        out = np.zeros(int(self._lib.fmx_max_character(self._h)) + 1, dtype=np.uint64)
        _check(self._lib.fmx_export_cs(self._h, _p(out)))
        return out

    def export_sa_samples(self):
        """SOSampledSuffixArray's payload SA[k << level] (sample.rs:33-37); uint64 for texts of 2^32 symbols and more"""
        k = int(self._lib.fmx_num_samples(self._h))
        if self.is_wide():
            out = np.zeros(max(k, 1), dtype=np.uint64)
            _check(self._lib.fmx_export_sa_samples64(self._h, _p(out)))
            return out[:k]
        out = np.zeros(max(k, 1), dtype=np.uint32)
        _check(self._lib.fmx_export_sa_samples(self._h, _p(out)))
        return out[:k]

    def export_sa(self):
        out = np.zeros(max(self.len(), 1), dtype=np.uint32)
        _check(self._lib.fmx_export_sa(self._h, _p(out)))
        return out[:self.len()]

    def verify_sa(self):
        v = C.c_uint64(0)
        _check(self._lib.fmx_verify_sa(self._h, C.byref(v)))
        return int(v.value)

    def has_pair_index(self):
        return bool(self._lib.fmx_has_pair_index(self._h))

    def is_wide(self):
        """served by the 64-bit engine (n >= 2^32 - 16, or FMX_FLAG_FORCE_WIDE)?"""
        return bool(self._lib.fmx_is_wide(self._h))

    def walk_records(self):
        """does the index carry walk records (text-order DNA index, levels 1..3; FMX_FLAG_NO_WALK_RECORDS)?"""
        return bool(self._lib.fmx_walk_records(self._h))

    def text_order(self):
        """suffix-array samples kept in text order (FMX_FLAG_TEXT_ORDER or the builder's default)?"""
        return bool(self._lib.fmx_text_order(self._h))

    def kmer_k(self):
        """k of the opt-in k-mer start table (0 = none)."""
        return int(self._lib.fmx_kmer_k(self._h))

    def level(self):
        lv = int(self._lib.fmx_level(self._h))
        return None if lv == L.NO_LOCATE else lv

    def handle(self):
        return self._h

    def device(self):
        return int(self._lib.fmx_device(self._h))

    def replicate(self, device=None):
        """fmx_replicate: a second handle with its own copy of every HBM array, on `device` (default: this index's);
        device-to-device copies, nothing is rebuilt ("index replicated on every GPU")."""
        other = self.__class__.__new__(self.__class__)
        other._lib, other._h, other._dtype = self._lib, C.c_void_p(), self._dtype
        _check(self._lib.fmx_replicate(self._h, self.device() if device is None else int(device), C.byref(other._h)))
        return other

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.fmx_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Replicas:
    """One batch over several replicas of an index from ONE host caller (include/fmx.h: fmx_count_batch_multi /
    fmx_locate_batch_multi; BASELINE config 5).  Patterns are independent (wrapper.rs:103-124), so pattern k of N goes to
    replica floor(k * G / N) and every shard's results land in place in the caller's arrays: the same SearchBatch /
    (offsets, positions) as search_many / locate_many on one handle, bit for bit."""

    def __init__(self, indexes):
        self.indexes = list(indexes)
        assert self.indexes, "at least one index"
        self._lib = self.indexes[0]._lib
        self._dtype = self.indexes[0]._dtype

    @classmethod
    def of(cls, index, devices):
        """`index` stays replica 0; one more replica per entry of `devices`"""
        return cls([index] + [index.replicate(d) for d in devices])

    def __len__(self):
        return len(self.indexes)

    def _handles(self):
        return (C.c_void_p * len(self.indexes))(*[ix.handle().value for ix in self.indexes])

    def shard_range(self, nitems, r):
        lo, hi = C.c_uint64(0), C.c_uint64(0)
        self._lib.fmx_shard_range(nitems, len(self.indexes), r, C.byref(lo), C.byref(hi))
        return int(lo.value), int(hi.value)

    def search_many(self, patterns=None, flat=None, off=None, s0e0=None):
        if flat is None:
            flat, off = pack_patterns(patterns, self._dtype)
        flat = _sym(flat, self._dtype)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        npat = len(off) - 1
        s = np.zeros(max(npat, 1), dtype=np.uint64)
        e = np.zeros(max(npat, 1), dtype=np.uint64)
        c = np.zeros(max(npat, 1), dtype=np.uint64)
        se = None if s0e0 is None else np.ascontiguousarray(s0e0, dtype=np.uint64)
        _check(self._lib.fmx_count_batch_multi(self._handles(), len(self.indexes), _p(flat), _p(off), npat, _p(se),
                                               _p(s), _p(e), _p(c)))
        return SearchBatch(self, s[:npat], e[:npat], c[:npat])

    def locate_many(self, s, e):
        s = np.ascontiguousarray(s, dtype=np.uint64)
        e = np.ascontiguousarray(e, dtype=np.uint64)
        off = np.zeros(len(s) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(e - s, dtype=np.uint64)
        pos = np.zeros(max(int(off[-1]), 1), dtype=np.uint64)
        _check(self._lib.fmx_locate_batch_multi(self._handles(), len(self.indexes), _p(s), _p(e), len(s), _p(off), _p(pos)))
        return off, pos[:int(off[-1])]

    def close(self, keep_first=False):
        for ix in self.indexes[1 if keep_first else 0:]:
            ix.close()
        self.indexes = self.indexes[:1] if keep_first else []


class FMIndex(_Index):
    """FMIndex::new(&text) (frontend.rs:195-203) -- count only."""
    _kind = L.KIND_FM

    def __init__(self, text, device=0, keep_sa=False, pair_index=False, kmer_table=False, force_wide=False, auto=False,
                 plain=False):
        super().__init__(text, None, device, keep_sa, pair_index, kmer_table, None, force_wide, auto=auto, plain=plain)


class FMIndexWithLocate(_Index):
    """FMIndexWithLocate::new(&text, level) (frontend.rs:205-221)."""
    _kind = L.KIND_FM

    def __init__(self, text, level, device=0, keep_sa=False, pair_index=False, kmer_table=False, sampling=None,
                 force_wide=False, walk_records=True, auto=False, plain=False):
        super().__init__(text, level, device, keep_sa, pair_index, kmer_table, sampling, force_wide, walk_records, auto,
                         plain=plain)


class RLFMIndex(_Index):
    """RLFMIndex::new(&text) (frontend.rs:223-231)."""
    _kind = L.KIND_RLFM

    def __init__(self, text, device=0, keep_sa=False, kmer_table=False, force_wide=False, plain=False):
        super().__init__(text, None, device, keep_sa, False, kmer_table, None, force_wide, plain=plain)


class RLFMIndexWithLocate(_Index):
    """RLFMIndexWithLocate::new(&text, level) (frontend.rs:233-243)."""
    _kind = L.KIND_RLFM

    def __init__(self, text, level, device=0, keep_sa=False, kmer_table=False, sampling=None, walk_records=True,
                 force_wide=False, run_table=False, plain=False):
        """run_table=True: FMX_FLAG_RUN_TABLE -- the run table whatever the text's runs-per-row ratio (the builder adds
        it by itself when r <= n / 4)"""
        super().__init__(text, level, device, keep_sa, False, kmer_table, sampling, force_wide, walk_records,
                         run_table=run_table, plain=plain)


class FMIndexMultiPieces(_Index):
    """FMIndexMultiPieces::new(&text) (frontend.rs:245-253): several \\0-separated pieces."""
    _kind = L.KIND_MULTI

    def __init__(self, text, device=0, keep_sa=False, kmer_table=False, force_wide=False):
        super().__init__(text, None, device, keep_sa, False, kmer_table, None, force_wide)

    def piece_id(self, i):
        return self._scalar(self._lib.fmx_piece_id_batch, i)

    def pieces_count(self):
        return int(self._lib.fmx_pieces_count(self._h))

    # SearchIndexWithMultiPieces (frontend.rs:46-68; wrapper.rs:57-82)
    def search_prefix(self, pattern):
        return Search(self, None, None, True).search(pattern)

    def search_suffix(self, pattern):
        return Search(self, 0, self.pieces_count(), False).search(pattern)

    def search_exact(self, pattern):
        return Search(self, 0, self.pieces_count(), True).search(pattern)


class FMIndexMultiPiecesWithLocate(FMIndexMultiPieces):
    """FMIndexMultiPiecesWithLocate::new(&text, level) (frontend.rs:255-267)."""

    def __init__(self, text, level, device=0, keep_sa=False, kmer_table=False, sampling=None, force_wide=False):
        _Index.__init__(self, text, level, device, keep_sa, False, kmer_table, sampling, force_wide)


class Search:
    """Search (frontend.rs:70-84): an SA interval [s, e) that can be refined."""

    def __init__(self, index, s, e, match_prefix_only=False):
        self._ix = index
        self._s = s
        self._e = e
        self._prefix = match_prefix_only     # wrapper.rs:22, 208

    def search(self, pattern):  # wrapper.rs:103-124: prepends `pattern`
        flat, off = pack_patterns([pattern], self._ix._dtype)
        se = None if self._s is None else np.array([self._s, self._e], dtype=np.uint64)
        b = self._ix.search_many(flat=flat, off=off, s0e0=se)
        return Search(self._ix, int(b.s[0]), int(b.e[0]), self._prefix)

    def count(self):  # wrapper.rs:132-134
        return self._e - self._s

    def get_range(self):  # wrapper.rs:126-129 (test-only in the reference)
        return (self._s, self._e)

    def iter_matches(self):  # wrapper.rs:137-139, 203-217: rows s..e-1 ascending
        if not self._prefix:
            for i in range(self._s, self._e):
                yield Match(self._ix, i)
        else:  # match_prefix_only: only rows whose L symbol is the end marker (wrapper.rs:208)
            _, rows = self._ix.match_rows_many([self._s], [self._e], True)
            for i in rows:
                yield Match(self._ix, int(i))

    def locate_all(self):
        """iter_matches().map(|m| m.locate()).collect() in one kernel launch."""
        if self._prefix:
            _, rows = self._ix.match_rows_many([self._s], [self._e], True)
            return [int(x) for x in self._ix.get_sa(rows)]
        _, pos = self._ix.locate_many([self._s], [self._e])
        return [int(x) for x in pos]

    def piece_ids(self):
        """iter_matches().map(|m| m.piece_id()).collect() (multi-pieces index)."""
        _, rows = self._ix.match_rows_many([self._s], [self._e], self._prefix)
        return [int(x) for x in self._ix.piece_id(rows)]


class Match:
    """Match / MatchWithLocate (frontend.rs:86-98)."""

    def __init__(self, index, i):
        self._ix = index
        self._i = i

    def locate(self):  # wrapper.rs:238-242 -> get_sa
        lib = self._ix._lib
        v = int(lib.fmx_get_sa(self._ix._h, self._i))
        if v == 0xFFFFFFFFFFFFFFFF:
            raise Error(L.ERR_NO_LOCATE, lib.fmx_last_error().decode())
        return v

    def piece_id(self):  # MatchWithPieceId::piece_id (frontend.rs:100-104)
        return int(self._ix._lib.fmx_piece_id(self._ix._h, self._i))

    _CHUNK = 64  # characters fetched per launch by the iterators below

    def iter_chars_forward(self):  # wrapper.rs:175-183: get_f then fl_map (stops at None)
        i = self._i
        while True:
            syms, lens, nxt = self._ix.extract_many([i], self._CHUNK, forward=True)
            for c in syms[0, :int(lens[0])]:
                yield int(c)
            i = int(nxt[0])
            if i == 0xFFFFFFFFFFFFFFFF:      # fl_map -> None: `?` ends the iterator (wrapper.rs:180)
                return

    def iter_chars_backward(self):  # wrapper.rs:154-161: get_l then lf_map
        i = self._i
        while True:
            syms, _, nxt = self._ix.extract_many([i], self._CHUNK, forward=False)
            for c in syms[0]:
                yield int(c)
            i = int(nxt[0])


class SearchBatch:
    """Result of search_many: per-pattern (s, e, count) arrays (numpy u64)."""

    def __init__(self, index, s, e, counts):
        self._ix = index
        self.s = s
        self.e = e
        self.counts = counts

    def count(self):
        return self.counts

    def locate(self):
        return self._ix.locate_many(self.s, self.e)
