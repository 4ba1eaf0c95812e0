"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/fmx.h declares, and fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "fmx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fmx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    from fm_index_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build_library()
    so = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(so, n), "libfmx.so does not export " + n
    bound = {s[0] for s in _lib.SYMBOLS}
    assert set(names) == bound, (set(names) ^ bound)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import fm_index_amd as F
    with pytest.raises(F.Error) as ei:
        F.FMIndex(b"abc\x00")
    assert ei.value.code == F._lib.ERR_HIP  # loud failure, not a silent CPU path


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under fm_index_amd/ may reference it."""
    pkg = os.path.join(ROOT, "fm_index_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "fm_oracle" not in txt and "import oracle" not in txt and \
                    "from oracle" not in txt, os.path.join(dp, f)


def test_error_messages_match_reference(golden):
    from fm_index_amd import _lib
    l = _lib.lib()
    msgs = {c["message"] for c in golden["invalid_texts"]["cases"]}
    got = {l.fmx_error_message(1).decode(), l.fmx_error_message(2).decode()}
    assert got == {"invalid text: " + m for m in msgs}  # error.rs:11 Display format


def build_readme_example(tmpdir):
    """g++ build of the C++ host mirror's README example against libfmx.so (C ABI only)."""
    import subprocess
    from fm_index_amd import _lib
    exe = os.path.join(str(tmpdir), "readme_example")
    subprocess.check_call([
        "g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
        "-I" + os.path.join(ROOT, "fm_index_amd", "host"),
        os.path.join(ROOT, "tests", "cpp", "readme_example.cpp"), "-o", exe,
        "-L" + os.path.dirname(_lib.LIB_PATH), "-lfmx",
        "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH)])
    return exe


def build_c_example(tmpdir, name="abi_example"):
    """gcc (C99) build of tests/c/<name>.c against libfmx.so: the ABI from plain C."""
    import subprocess
    from fm_index_amd import _lib
    exe = os.path.join(str(tmpdir), name)
    subprocess.check_call([
        "gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "c", name + ".c"), "-o", exe,
        "-L" + os.path.dirname(_lib.LIB_PATH), "-lfmx",
        "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH)])
    return exe


def test_plain_c_program_compiles_and_links(tmp_path):
    assert os.path.exists(build_c_example(tmp_path))
    assert os.path.exists(build_c_example(tmp_path, "abi_multi"))     # the multi-replica entry points from C99


def test_cpp_host_mirror_compiles_and_links(tmp_path):
    exe = build_readme_example(tmp_path)
    assert os.path.exists(exe)


def test_header_is_plain_c99(tmp_path):
    """include/fmx.h is the boundary a Rust / C / Go binding generator reads: it must compile as
    strict C99 on its own (no C++-isms, no HIP or torch types)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "fmx.h"\nint main(void) { return fmx_error_message(0) == 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc,
                           "-fsyntax-only", str(src)])
    code = re.sub(r"/\*.*?\*/", "", open(os.path.join(inc, "fmx.h")).read(), flags=re.S)
    assert "torch" not in code and "hipStream_t" not in code and "#include <hip" not in code


def test_shard_range_is_the_partition_of_the_python_harness():
    """fmx_shard_range (pure host code: callable without a GPU) = fm_index_amd.sharding.shard_range for every (N, G, r):
    the one-process C-ABI path and the one-process-per-GPU path cut a batch at the same patterns; the shards are
    contiguous, cover the batch and differ by at most one pattern."""
    import ctypes as C
    from fm_index_amd import _lib, sharding
    lib = _lib.lib()
    lo, hi = C.c_uint64(0), C.c_uint64(0)
    for n in (0, 1, 2, 7, 8, 9, 1000, 8193, 8388608, (1 << 40) + 12345):
        for g in (1, 2, 3, 5, 8, 64):
            prev, sizes = 0, []
            for r in range(g):
                lib.fmx_shard_range(n, g, r, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == sharding.shard_range(n, r, g), (n, g, r)
                assert lo.value == prev and hi.value >= lo.value
                prev = hi.value
                sizes.append(hi.value - lo.value)
            assert prev == n and max(sizes) - min(sizes) <= 1
    lib.fmx_shard_range(10, 4, 9, C.byref(lo), C.byref(hi))      # r >= G: empty
    assert (lo.value, hi.value) == (0, 0)
    # the multi-replica entry points refuse an empty or NULL handle list before anything touches a device
    assert lib.fmx_count_batch_multi(None, 0, None, None, 3, None, None, None, None) == _lib.ERR_ARG
    hs = (C.c_void_p * 2)(None, None)
    assert lib.fmx_count_batch_multi(hs, 2, None, None, 3, None, None, None, None) == _lib.ERR_ARG
    assert lib.fmx_locate_batch_multi(hs, 2, None, None, 3, None, None) == _lib.ERR_ARG
    out = C.c_void_p()
    assert lib.fmx_replicate(None, 0, C.byref(out)) == _lib.ERR_ARG and not out.value
