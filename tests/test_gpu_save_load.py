"""Flat index file: a loaded index answers exactly like the one that was saved."""
import numpy as np
import pytest

import fm_index_amd as F
from fm_index_amd import workload as W

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["fm", "fm_pair", "rlfm", "fm_bytes", "fm_u16", "multi"])
def test_save_load_roundtrip(tmp_path, kind):
    if kind in ("fm", "fm_pair"):
        t = W.dna_text_np(50000, 3)
        idx = F.FMIndexWithLocate(F.Text.with_max_character(t, 4), 2, pair_index=(kind == "fm_pair"))
        flat, off, _ = W.substring_patterns_np(t, 2000, 11, 5)
    elif kind == "rlfm":
        t = W.repetitive_text_np(50000, 5, base_len=256)
        idx = F.RLFMIndexWithLocate(F.Text(t), 3)
        flat, off, _ = W.substring_patterns_np(t, 2000, 9, 6)
    elif kind == "multi":
        t = W.byte_text_np(30000, 4)
        t[np.arange(500, 29000, 977)] = 0            # several pieces
        idx = F.FMIndexMultiPiecesWithLocate(F.Text(t), 2)
        flat, off, _ = W.substring_patterns_np(t, 1500, 3, 9)
    elif kind == "fm_bytes":
        t = W.byte_text_np(50000, 4)
        idx = F.FMIndexWithLocate(F.Text(t), 1)
        flat, off, _ = W.substring_patterns_np(t, 2000, 3, 7)
    else:
        t = ((W.splitmix64_np(9, 0, 20000) % np.uint64(3000)) + np.uint64(1)).astype(np.uint16)
        t[-1] = 0
        idx = F.FMIndexWithLocate(F.Text.with_max_character(t, 3000), 2)
        flat, off, _ = W.substring_patterns_np(t, 1000, 3, 8)
        flat = flat.astype(np.uint16) if False else t[np.add.outer(
            (W.splitmix64_np(8, 0, 1000) % np.uint64(20000 - 1 - 3)).astype(np.int64), np.arange(3))].reshape(-1)
    a = idx.search_many(flat=flat, off=off)
    aoff, apos = a.locate()
    path = tmp_path / "index.fmx"
    idx.save(path)
    cls = F.RLFMIndexWithLocate if kind == "rlfm" else \
        (F.FMIndexMultiPiecesWithLocate if kind == "multi" else F.FMIndexWithLocate)
    idx2 = cls.load(path)
    assert idx2.len() == idx.len() and idx2.level() == idx.level()
    assert idx2.heap_size() == idx.heap_size()
    assert idx2.has_pair_index() == idx.has_pair_index()
    b = idx2.search_many(flat=flat, off=off)
    assert (a.s == b.s).all() and (a.e == b.e).all()
    boff, bpos = b.locate()
    assert (aoff == boff).all() and (apos == bpos).all()
    rows = np.arange(0, idx.len(), 97)
    assert (idx.lf_map(rows) == idx2.lf_map(rows)).all()
    assert (idx.export_cs() == idx2.export_cs()).all()
    if kind == "multi":
        assert idx2.pieces_count() == idx.pieces_count() > 5
        assert (idx.piece_id(rows) == idx2.piece_id(rows)).all()


def test_load_rejects_garbage(tmp_path):
    p = tmp_path / "bad.fmx"
    p.write_bytes(b"not an index" * 100)
    with pytest.raises(F.Error):
        F.FMIndex.load(p)
    with pytest.raises(F.Error):
        F.FMIndex.load(tmp_path / "missing.fmx")
