"""Build / query / free cycles must give all HBM back (the handle owns every array;
`Drop` in the crate, `fmx_free` here), including builds that fail validation."""
import numpy as np
import pytest
import torch

import fm_index_amd as F
from fm_index_amd import workload as W

pytestmark = pytest.mark.gpu


def _cycle(seed):
    t = W.byte_text_np(40000, seed)
    for make in (lambda: F.FMIndexWithLocate(F.Text(t), 2),
                 lambda: F.FMIndex(F.Text(t)),
                 lambda: F.RLFMIndexWithLocate(F.Text(t), 1),
                 lambda: F.FMIndexWithLocate(F.Text.with_max_character(((t & 3) + (t != 0)).astype(np.uint8), 4), 2,
                                             pair_index=True)):
        idx = make()
        flat = np.frombuffer(bytes([1, 2, 3, 1]) * 64, dtype=np.uint8)
        off = np.arange(0, 257, 4, dtype=np.uint64)
        b = idx.search_many(flat=flat, off=off)
        if idx.level() is not None:
            b.locate()
        idx.close()
    tm = t.copy()
    tm[1000::3000] = 0
    idx = F.FMIndexMultiPiecesWithLocate(F.Text(tm), 2)
    idx.search_many([b"ab", b"\x01"]).locate()
    idx.close()
    bad = t.copy()
    bad[-1] = 7                                           # no terminator -> FMX_ERR_TEXT_END_ZERO
    with pytest.raises(F.Error):
        F.FMIndex(F.Text(bad))
    bad = t.copy()
    bad[0] = 0
    with pytest.raises(F.Error):
        F.FMIndex(F.Text(bad))


def test_build_query_free_cycles_return_all_memory():
    dev = torch.device("cuda", 0)
    for s in range(5):                                    # let pools / code objects settle
        _cycle(s)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    for s in range(40):
        _cycle(100 + s)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    # 40 cycles x 7 builds; one leaked 40 kB text per build would already be > 11 MB
    assert free0 - free1 < (4 << 20), (free0, free1)


def test_big_index_memory_is_returned():
    dev = torch.device("cuda", 0)
    n = 1 << 26
    d_text = W.dna_text_torch(n, 3, dev)
    F.FMIndexWithLocate.from_device_text(d_text.data_ptr(), n, 4, 2).close()   # runtime pools settle
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(dev)
    for _ in range(3):
        idx = F.FMIndexWithLocate.from_device_text(d_text.data_ptr(), n, 4, 2)
        assert idx.heap_size() > n
        idx.close()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(dev)
    assert free0 - free1 < (8 << 20), (free0, free1)


def test_release_scratch_gives_the_builder_cache_back():
    """Large builds keep their temporaries in a process-wide cache (a hipMalloc of memory the process has cycled through
    costs ~30 ms per GiB on this runtime) -- at most 32 GiB or an eighth of the device -- and fmx_release_scratch()
    returns all of it (and the idle small-build buffers) to the driver."""
    from fm_index_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda", 0)
    lib.fmx_release_scratch()
    torch.cuda.synchronize()
    free0, total = torch.cuda.mem_get_info(dev)
    n = 1 << 26
    d_text = W.dna_text_torch(n, 5, dev)
    torch.cuda.synchronize()
    free_text, _ = torch.cuda.mem_get_info(dev)
    F.FMIndex.from_device_text(d_text.data_ptr(), n, 4).close()
    F.FMIndex(F.Text.with_max_character(W.dna_text_np(3000, 1), 4)).close()        # a small build: leased buffer
    torch.cuda.synchronize()
    held, _ = torch.cuda.mem_get_info(dev)
    assert free_text - held > 8 * n                      # the suffix sort's buffers are still ours ...
    assert free_text - held <= min(32 << 30, total // 8) + (64 << 20)
    lib.fmx_release_scratch()
    torch.cuda.synchronize()
    after, _ = torch.cuda.mem_get_info(dev)
    assert free_text - after < (8 << 20), (free_text, after)   # ... until the hook is called
    # and the next build simply allocates again
    F.FMIndex.from_device_text(d_text.data_ptr(), n, 4).close()
    lib.fmx_release_scratch()
    del d_text
