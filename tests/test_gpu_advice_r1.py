"""Regression tests for the round-1 advisor findings: interior pattern offsets are validated in the
kernels (every entry point, any index, small and large batches), host-pointer calls of different
threads never see each other's status, and a corrupt index file is refused, not trusted."""
import ctypes as C
import os
import struct
import threading

import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import workload as W
from oracle import fm_oracle as O

pytestmark = pytest.mark.gpu


def _idx(kind="fm", n=60000, **kw):
    if kind == "rlfm":
        t = W.repetitive_text_np(n, 5, base_len=1 << 10)
        return t, F.RLFMIndexWithLocate(F.Text(t), 2, **kw)
    if kind == "bytes":
        t = W.byte_text_np(n, 4)
        return t, F.FMIndexWithLocate(F.Text(t), 2, **kw)
    t = W.dna_text_np(n, 21)
    return t, F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2, **kw)


@pytest.mark.parametrize("kind,kw", [("fm", {}), ("fm", {"pair_index": True}), ("fm", {"kmer_table": True}),
                                     ("bytes", {}), ("rlfm", {}), ("rlfm", {"kmer_table": True})])
@pytest.mark.parametrize("npat", [64, (1 << 17) + 77])
def test_interior_offsets_are_checked_in_the_kernel(kind, kw, npat):
    t, idx = _idx(kind, **kw)
    flat, off, _ = W.substring_patterns_np(t, npat, 7, 3)
    good = idx.search_many(flat=flat, off=off)
    for victim, value in ((npat // 2 + 1, int(off[-1]) + 5), (3, 0), (npat - 1, 1 << 40)):
        bad = off.copy()
        bad[victim] = value                       # backwards, or far behind the pattern buffer
        with pytest.raises(F.Error) as ei:
            idx.search_many(flat=flat, off=bad)
        assert ei.value.code == F._lib.ERR_ARG
    again = idx.search_many(flat=flat, off=off)    # nothing sticks to the handle
    assert (again.s == good.s).all() and (again.e == good.e).all()


def test_interior_offsets_are_checked_by_the_dev_entry_point():
    import torch
    t, idx = _idx("fm")
    flat, off, _ = W.substring_patterns_np(t, 5000, 7, 3)
    bad = off.copy()
    bad[2501] = off[-1] + np.uint64(1 << 33)
    lib = idx._lib
    dev = torch.device("cuda", 0)
    d_flat = torch.from_numpy(flat).to(dev)
    d_s = torch.zeros(5000, dtype=torch.int64, device=dev)
    d_e = torch.zeros(5000, dtype=torch.int64, device=dev)
    for o, want in ((bad, F._lib.ERR_ARG), (off, 0)):
        d_off = torch.from_numpy(o.view(np.int64)).to(dev)
        rc = lib.fmx_count_batch_dev(idx.handle(), C.c_void_p(d_flat.data_ptr()), C.c_void_p(d_off.data_ptr()),
                                     5000, None, C.c_void_p(d_s.data_ptr()), C.c_void_p(d_e.data_ptr()), None, None)
        assert rc == 0
        torch.cuda.synchronize()
        assert lib.fmx_stream_status(idx.handle()) == want
    oi = O.OracleIndex(t, 4)
    os_, oe = oi.count_batch(flat, off)
    assert (d_s.cpu().numpy().view(np.uint64) == os_).all() and (d_e.cpu().numpy().view(np.uint64) == oe).all()


def test_threads_never_see_each_others_status():
    """thread A keeps sending a pattern with an out-of-range symbol, thread B valid patterns, on ONE
    handle: A must get FMX_ERR_SYMBOL_RANGE every time, B never (each host-pointer call owns its
    status word)."""
    t, idx = _idx("fm")
    flat, off, _ = W.substring_patterns_np(t, 300, 9, 5)
    want = idx.search_many(flat=flat, off=off)
    badflat = flat.copy()
    badflat[17] = 9                                # max_character is 4
    res = {"a_ok": 0, "a_err": 0, "b_ok": 0, "b_err": 0, "b_wrong": 0}
    stop = threading.Event()

    def a():
        for _ in range(400):
            try:
                idx.search_many(flat=badflat, off=off)
                res["a_ok"] += 1
            except F.Error as ex:
                res["a_err"] += ex.code == F._lib.ERR_SYMBOL_RANGE
        stop.set()

    def b():
        while not stop.is_set():
            try:
                r = idx.search_many(flat=flat, off=off)
                res["b_ok"] += 1
                res["b_wrong"] += not ((r.s == want.s).all() and (r.e == want.e).all())
            except F.Error:
                res["b_err"] += 1
    ta, tb = threading.Thread(target=a), threading.Thread(target=b)
    ta.start(); tb.start(); ta.join(); tb.join()
    assert res["a_err"] == 400 and res["a_ok"] == 0, res
    assert res["b_err"] == 0 and res["b_wrong"] == 0 and res["b_ok"] > 0, res


def test_null_arguments_are_refused():
    t, idx = _idx("fm", n=5000)
    lib = idx._lib
    i = np.array([1, 2], dtype=np.uint64)
    out = np.zeros(2, dtype=np.uint64)
    assert lib.fmx_lf_map_batch(idx.handle(), None, 2, F._p(out)) == F._lib.ERR_ARG
    assert lib.fmx_lf_map_batch(idx.handle(), F._p(i), 2, None) == F._lib.ERR_ARG
    assert lib.fmx_match_rows(idx.handle(), F._p(i), F._p(i), 2, 0, None, F._p(out)) == F._lib.ERR_ARG
    assert lib.fmx_match_counts(idx.handle(), F._p(i), F._p(i), 2, 0, None) == F._lib.ERR_ARG
    off = np.array([0, 2], dtype=np.uint64)
    assert lib.fmx_count_batch(idx.handle(), None, F._p(off), 1, None, F._p(out), F._p(out), None) == F._lib.ERR_ARG


def test_corrupt_index_files_are_refused(tmp_path):
    t, idx = _idx("rlfm", n=40000, kmer_table=True)
    path = str(tmp_path / "x.fmx")
    idx.save(path)
    raw = bytearray(open(path, "rb").read())
    # no device address of the saving process is in the file: a second build of the same text lives
    # at other addresses (the first index is still alive), yet its file is byte-identical
    hdr = 8 + 4 + 4 + 5 * 8 + 4 * 4
    dev_bytes = struct.unpack_from("<I", raw, 12)[0]
    t2, idx_b = _idx("rlfm", n=40000, kmer_table=True)
    path_b = str(tmp_path / "x2.fmx")
    idx_b.save(path_b)
    assert open(path_b, "rb").read() == bytes(raw), "the file depends on where the arrays were allocated"
    idx_b.close()
    lib = idx._lib

    def load(buf):
        p = str(tmp_path / "y.fmx")
        open(p, "wb").write(bytes(buf))
        h = C.c_void_p()
        rc = lib.fmx_load(p.encode(), 0, C.byref(h))
        if rc == 0:
            lib.fmx_free(h)
        return rc
    assert load(raw) == 0
    # every 4-byte field of the device struct, overwritten with hostile values: never a crash, and
    # either refused or (for fields that carry no size) harmless
    refused = 0
    for o in range(hdr, hdr + dev_bytes, 4):
        for val in (0xFFFFFFFF, 200, 0x7FFFFFF0):
            bad = bytearray(raw)
            struct.pack_into("<I", bad, o, val)
            if bad == raw:
                continue
            refused += load(bad) != 0
    assert refused > 60          # sizes, formats, lengths, levels, counts ...; the rest carries no size
    # header fields
    for o in range(8, hdr, 4):                       # version .. level_requested
        bad = bytearray(raw)
        struct.pack_into("<I", bad, o, 0x7FFFFFF1)
        harmless = 48 <= o < 56 or o == 68            # `bytes` is only reported back; level_requested is history
        assert load(bad) != 0 or harmless, o
    assert load(raw[:len(raw) // 2]) != 0            # truncated
    assert load(raw + bytes(16)) != 0               # trailing bytes
    idx2 = type(idx).load(path) if hasattr(type(idx), "load") else None
    if idx2 is not None:
        flat, off, _ = W.substring_patterns_np(t, 200, 6, 3)
        a, b = idx.search_many(flat=flat, off=off), idx2.search_many(flat=flat, off=off)
        assert (a.s == b.s).all() and (a.e == b.e).all()
