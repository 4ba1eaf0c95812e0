"""Rows and positions beyond 2^31: n = 2^31 + 2^20, where every 32-bit quantity of the
engine that is a row or a text position uses its top bit -- and, on the run-length index, where the match flag
of the endpoint-per-lane rank rounds can no longer ride in bit 31 of the group sum (fmx_ep.h, `wide`).
Same protocol as tests/test_gpu_fullsize.py: size-independent properties on the device (every located position
holds its pattern, the pattern's source position is among its hits) and bit-identity of (s, e), of the trait
methods at rows >= 2^31 and of the ordered position lists with the CPU oracle, which is fed the exported BWT /
L column and the exported samples (reference semantics: fm_index.rs:82-140, rlfmi.rs:122-190, sample.rs:21-60).
`python tests/test_gpu_beyond_2g.py` prints the JSON summary kept under profiles/."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fm_index_amd as F  # noqa: E402
from fm_index_amd import _lib as L  # noqa: E402
from fm_index_amd import workload as W  # noqa: E402

pytestmark = pytest.mark.gpu

N = (1 << 31) + (1 << 20)


def _run(kind):
    import torch
    from oracle import fm_oracle as O
    torch.cuda.empty_cache()
    dev = torch.device("cuda", 0)
    lib = L.lib()
    dna = kind == "fm"
    text = W.dna_text_torch(N, 11, dev) if dna else W.byte_text_torch(N, 12, dev)
    maxc, m, level = (4, 28, 2) if dna else (255, 12, 2)
    cls = F.FMIndexWithLocate if dna else F.RLFMIndexWithLocate
    index = cls.from_device_text(text.data_ptr(), N, maxc, level=level)
    h = index.handle()
    assert index.len() == N
    # patterns = substrings of the text from (A) uniform positions, (B) positions whose suffix sorts into the top
    # 2^20 rows, i.e. beyond row 2^31 -- suffixes that start with six 4s (DNA: 4^-6 of the rows) or with 255 and
    # a symbol >= 226 (bytes: 30 / 255^2) --, (C) positions beyond 2^31 in the text
    win = text[:1 << 28]
    if dna:
        hi = win[:-8] == 4
        for j in range(1, 6):
            hi &= win[j:j - 8] == 4
    else:
        hi = (win[:-8] == 255) & (win[1:-7] >= 226)
    src_b = torch.nonzero(hi).flatten()[:1 << 14]
    assert src_b.numel() >= 1 << 12
    del hi, win
    src_a = W.umod_torch(W.splitmix64_torch(13, 0, 1 << 15, dev), N - 1 - m)
    src_c = (1 << 31) + W.umod_torch(W.splitmix64_torch(14, 0, 1 << 14, dev), (1 << 20) - 1 - m)
    src = torch.cat([src_b, src_a, src_c])
    npat = int(src.numel())
    pat = text[src[:, None] + torch.arange(m, dtype=torch.int64, device=dev)[None, :]].reshape(-1).contiguous()
    off = (torch.arange(npat + 1, dtype=torch.int64, device=dev) * m).contiguous()
    s = torch.empty(npat, dtype=torch.int64, device=dev)
    e = torch.empty(npat, dtype=torch.int64, device=dev)
    c = torch.empty(npat, dtype=torch.int64, device=dev)
    assert lib.fmx_count_batch_dev(h, C.c_void_p(pat.data_ptr()), C.c_void_p(off.data_ptr()), npat, None,
                                   C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), C.c_void_p(c.data_ptr()),
                                   None) == 0
    torch.cuda.synchronize()
    assert lib.fmx_stream_status(h) == 0
    assert bool((c >= 1).all())
    rows_hi = int((e > (1 << 31)).sum().item())
    assert rows_hi >= int(src_b.numel())                # every (B) pattern's interval lies beyond row 2^31
    d_off = torch.empty(npat + 1, dtype=torch.int64, device=dev)
    assert lib.fmx_offsets_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat,
                               C.c_void_p(d_off.data_ptr()), None) == 0
    total = int(d_off[-1].item())
    d_pos = torch.empty(total, dtype=torch.int64, device=dev)
    assert lib.fmx_locate_batch_dev(h, C.c_void_p(s.data_ptr()), C.c_void_p(e.data_ptr()), npat,
                                    C.c_void_p(d_off.data_ptr()), total, C.c_void_p(d_pos.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert lib.fmx_stream_status(h) == 0
    hit = torch.repeat_interleave(torch.arange(npat, device=dev), c)
    ok = torch.ones(total, dtype=torch.bool, device=dev)
    for j in range(m):
        ok &= text[d_pos + j] == pat.view(npat, m)[hit, j]
    assert bool(ok.all())                               # every located position holds the pattern
    found = torch.zeros(npat, dtype=torch.bool, device=dev)
    found[hit[d_pos == src[hit]]] = True
    assert bool(found.all())                            # and the source position is among the hits
    pos_hi = int((d_pos >= (1 << 31)).sum().item())
    assert pos_hi >= int(src_c.numel())
    # ---- the oracle, from the exported L column and the exported samples ----
    oi = O.OracleIndex.from_bwt(index.export_bwt(), index.export_cs(), maxc, samples=index.export_sa_samples(),
                                level=level, kind="fm" if dna else "rlfm")
    k = 1 << 12
    so, eo = oi.count_batch(pat[:k * m].cpu().numpy(), np.arange(k + 1, dtype=np.uint64) * np.uint64(m), nthreads=16)
    assert (so == s[:k].cpu().numpy().view(np.uint64)).all()
    assert (eo == e[:k].cpu().numpy().view(np.uint64)).all()
    ooff, opos = oi.locate_batch(so[:1024], eo[:1024], nthreads=16)
    assert (opos == d_pos[:int(ooff[-1])].cpu().numpy().view(np.uint64)).all()
    # trait methods at rows in [2^31, n): lf_map2 with i == n included
    rows = (np.uint64(1 << 31) + W.splitmix64_np(21, 0, 2048) % np.uint64(1 << 20)).astype(np.uint64)
    syms = (np.uint64(1) + W.splitmix64_np(22, 0, 2048) % np.uint64(maxc)).astype(np.uint64)
    rows2 = rows.copy()
    rows2[0] = N
    assert (index.lf_map2(syms, rows2) == oi.lf_map2(syms, rows2)).all()
    assert (index.lf_map(rows) == oi.lf_map(rows)).all() and (index.get_l(rows) == oi.get_l(rows)).all()
    assert (index.get_sa(rows[:256]) == oi.get_sa(rows[:256])).all()
    oi.close()
    out = {"kind": kind, "n": N, "level": level, "patterns": npat, "pattern_len": m, "hits": total,
           "intervals_with_e_beyond_2^31": rows_hi, "positions_beyond_2^31": pos_hi,
           "max_row": int(e.max().item()), "max_position": int(d_pos.max().item()),
           "oracle_patterns_identical": k, "oracle_located_patterns_identical": 1024,
           "trait_rows_checked_beyond_2^31": int(len(rows)), "build_ms": round(float(lib.fmx_build_ms(h)), 1),
           "index_bytes": index.heap_size(), "text_order": index.text_order()}
    index.close()
    del text, pat, d_pos
    torch.cuda.empty_cache()
    return out


def test_fm_dna_rows_and_positions_beyond_2g():
    _run("fm")


def test_rlfm_byte_text_rows_beyond_2g():
    _run("rlfm")


if __name__ == "__main__":
    for kind_ in (sys.argv[1:] or ["fm", "rlfm"]):
        print(json.dumps(_run(kind_)))
