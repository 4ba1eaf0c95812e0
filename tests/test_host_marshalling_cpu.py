"""Host-side marshalling that needs no GPU: symbols are never truncated silently."""
import numpy as np
import pytest

import fm_index_amd as F


def test_symbols_keep_the_narrowest_width_that_holds_them():
    assert F.Text([1, 200, 2, 0]).text().dtype == np.uint8
    t = F.Text([1, 300, 2, 0])                    # used to wrap to [1, 44, 2, 0]
    assert t.text().dtype == np.uint16 and t.text().tolist() == [1, 300, 2, 0]
    assert F.Text(np.array([1, 70000, 0], dtype=np.int64)).text().dtype == np.uint32
    assert F.Text(np.array([5, 1, 0], dtype=np.uint32)).text().dtype == np.uint32   # unsigned arrays keep theirs


def test_negative_and_oversized_symbols_are_refused():
    with pytest.raises(F.Error) as ei:
        F.Text([1, -3, 0])
    assert ei.value.code == F._lib.ERR_SYMBOL_RANGE
    with pytest.raises(F.Error) as ei:
        F.pack_patterns([[1, 2, 300]], np.uint8)   # a u8 index asked for symbol 300
    assert ei.value.code == F._lib.ERR_SYMBOL_RANGE
    with pytest.raises(F.Error):
        F.Text(np.array([1.5, 0.0]))
    flat, off = F.pack_patterns([[1, 2, 255], b"ab"], np.uint8)
    assert flat.tolist() == [1, 2, 255, 97, 98] and off.tolist() == [0, 3, 5]
