"""Runs the reference's README example through the C++ host mirror (fm_index.hpp -> C ABI)."""
import subprocess

import pytest

from test_abi_cpu import build_c_example, build_readme_example

pytestmark = pytest.mark.gpu


def test_readme_example_cpp(golden, tmp_path):
    exe = build_readme_example(tmp_path)
    txt = tmp_path / "lorem.bin"
    txt.write_bytes(golden["readme"]["text"].encode("latin-1"))
    mp = tmp_path / "twinkle.bin"
    mp.write_bytes(golden["multi_pieces_example"]["text"].encode("latin-1"))
    out = subprocess.run([exe, str(txt), str(mp)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "count 4"
    assert lines[1] == "positions 246 12 300 103"          # README.md:64, in this order
    assert lines[2] == "forward " + golden["readme"]["forward_20_from_match_3"]
    assert lines[3] == "backward " + golden["readme"]["backward_16_from_first_match"]
    assert lines[4] == "refined 4"
    assert lines[5] == "star 4"                             # examples/multi_pieces.rs:33
    assert lines[6] == "suffix 0 1 2"                       # examples/multi_pieces.rs:80-87
    assert lines[7] == "prefix 0"                           # examples/multi_pieces.rs:70-77
    assert lines[8] == "error invalid text: the given text must end with exactly one zero character"
    assert lines[9] == "sharded same"                       # fmx_replicate + fmx_count_batch_multi through the C++ mirror


def test_plain_c_example_runs(tmp_path):
    """tests/c/abi_example.c: build, count, locate and extract through the ABI from a C99 program."""
    out = subprocess.run([build_c_example(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok len=12 ")
