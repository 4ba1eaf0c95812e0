"""Replays tests/golden/naive_fixtures.json (brute-force scan + suffix sort by definition,
no index structure involved):
  * CPU: the oracle's FM and RLFM paths must reproduce it (pins the fast oracle on a third opinion);
  * -m gpu: the HIP path, FMIndexWithLocate and RLFMIndexWithLocate at levels 0-4, must reproduce it
    with NO oracle in the loop."""
import hashlib
import json
import os

import numpy as np
import pytest

from fm_index_amd import workload as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "naive_fixtures.json")) as f:
    CASES = json.load(f)["cases"]


def text_of(c):
    t = (W.splitmix64_np(c["seed"], 0, c["n"]) % np.uint64(c["sigma"])).astype(np.uint8) + 1
    t[-1] = 0
    assert hashlib.sha1(t.tobytes()).hexdigest() == c["text_sha1"], "the text generator changed"
    return t


def check_positions(ent, pos):
    pos = np.asarray(pos, dtype=np.int64)
    assert len(pos) == ent["count"]
    if "positions" in ent:
        assert pos.tolist() == ent["positions"]
    elif ent["count"]:
        h = len(ent["positions_head"])
        assert pos[:h].tolist() == ent["positions_head"] and pos[-h:].tolist() == ent["positions_tail"]
        assert hashlib.sha1(pos.astype("<u8").tobytes()).hexdigest() == ent["positions_sha1"]


def test_fixture_file_shape():
    assert len(CASES) == 28 and {c["sigma"] for c in CASES} == {2, 4, 8, 255}
    assert {c["n"] for c in CASES} == {1 << k for k in range(10, 17)}
    assert sum(e["count"] == 0 for c in CASES for e in c["patterns"]) > 50      # absent patterns are in
    assert sum(e["count"] > 24 for c in CASES for e in c["patterns"]) > 50      # and wide intervals


@pytest.mark.parametrize("kind", ["fm", "rlfm"])
def test_oracle_reproduces_the_naive_fixtures(kind):
    from oracle import fm_oracle as O
    for c in CASES:
        if c["n"] > (1 << 14):
            continue                       # CPU budget: the big cases are replayed on the GPU
        t = text_of(c)
        oi = O.OracleIndex(t, c["sigma"], level=2, kind=kind)
        for ent in c["patterns"]:
            p = bytes.fromhex(ent["pattern"])
            s, e = oi.search(p)
            assert e - s == ent["count"], (c["n"], c["sigma"], ent["pattern"])
            if ent["count"]:
                assert (s, e) == (ent["s"], ent["e"])
                _, pos = oi.locate_batch([s], [e])
                check_positions(ent, pos)
        oi.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["fm", "rlfm"])
@pytest.mark.parametrize("level", [0, 1, 2, 3, 4])
def test_gpu_reproduces_the_naive_fixtures(kind, level):
    import fm_index_amd as F
    for c in CASES:
        if level in (1, 3, 4) and c["n"] not in (1 << 10, 1 << 13, 1 << 16):
            continue                       # every level sees small, medium and large; 0 and 2 see all
        t = text_of(c)
        cls = F.FMIndexWithLocate if kind == "fm" else F.RLFMIndexWithLocate
        idx = cls(F.Text.with_max_character(t, c["sigma"]), level)
        pats = [bytes.fromhex(e["pattern"]) for e in c["patterns"]]
        b = idx.search_many(pats)
        off, pos = b.locate()
        for k, ent in enumerate(c["patterns"]):
            assert int(b.counts[k]) == ent["count"], (c["n"], c["sigma"], ent["pattern"])
            if ent["count"]:
                assert (int(b.s[k]), int(b.e[k])) == (ent["s"], ent["e"])
            else:
                assert int(b.s[k]) == int(b.e[k])
            check_positions(ent, pos[int(off[k]):int(off[k + 1])])
        idx.close()
