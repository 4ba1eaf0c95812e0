"""integration/rust (the `gpu` feature a maintainer adds to ajalab/fm-index) cannot be compiled
here -- no Rust toolchain -- so its binding is checked structurally: every `extern "C"` item of
src/gpu/ffi.rs against the prototype of the same name in include/fmx.h (name, arity, width and
constness of every parameter, return type), both parsed independently of the generator."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "integration", "rust")

C_TO_RUST = {
    "uint64_t": "u64", "uint32_t": "u32", "int": "c_int", "double": "f64", "void": None,
    "const char*": "*const c_char", "void*": "*mut c_void", "const void*": "*const c_void",
    "const uint64_t*": "*const u64", "uint64_t*": "*mut u64", "uint32_t*": "*mut u32",
    "fmx_index*": "*mut FmxIndex", "const fmx_index*": "*const FmxIndex", "fmx_index**": "*mut *mut FmxIndex",
    "fmx_index*const*": "*const *mut FmxIndex", "const void*const*": "*const *const c_void",
    "const uint64_t*const*": "*const *const u64",
}


def c_prototypes():
    txt = open(os.path.join(ROOT, "include", "fmx.h")).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = "\n".join(ln for ln in txt.splitlines() if not ln.lstrip().startswith("#"))
    protos = {}
    for stmt in txt.split(";"):
        m = re.search(r"\b(fmx_\w+)\s*\((.*)\)\s*$", stmt.strip(), flags=re.S)
        if not m or "typedef" in stmt:
            continue
        name = m.group(1)
        ret = stmt.strip()[:m.start(1)].replace('extern "C" {', "").strip()
        params = []
        body = " ".join(m.group(2).split())
        if body and body != "void":
            for p in body.split(","):
                t = re.sub(r"\w+$", "", p.strip()).strip()          # drop the parameter name
                params.append(t.replace(" *", "*").replace("* ", "*"))
        protos[name] = (ret.replace(" *", "*"), params)
    return protos


def rust_externs():
    txt = open(os.path.join(RUST, "src", "gpu", "ffi.rs")).read()
    block = txt[txt.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block, flags=re.S):
        params = [p.split(":", 1)[1].strip() for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = (m.group(3).strip() if m.group(3) else None, params)
    return out


def test_every_abi_function_is_bound_with_the_same_shape():
    c, r = c_prototypes(), rust_externs()
    assert len(c) >= 60
    assert set(c) == set(r), (set(c) ^ set(r))
    for name, (ret, params) in c.items():
        rret, rparams = r[name]
        assert C_TO_RUST[ret] == rret, (name, ret, rret)
        assert [C_TO_RUST[p] for p in params] == rparams, (name, params, rparams)


def test_ffi_rs_is_what_the_generator_makes_of_the_current_header():
    sys.path.insert(0, RUST)
    try:
        import gen_ffi
    finally:
        sys.path.pop(0)
    assert gen_ffi.generate() == open(os.path.join(RUST, "src", "gpu", "ffi.rs")).read(), \
        "include/fmx.h changed: run python integration/rust/gen_ffi.py"


def test_constants_match_the_header():
    h = open(os.path.join(ROOT, "include", "fmx.h")).read()
    r = open(os.path.join(RUST, "src", "gpu", "ffi.rs")).read()
    for m in re.finditer(r"^#define (FMX_(?:ERR|KIND|FLAG|OK|NO)\w*) (\S+)", h, flags=re.M):
        val = int(m.group(2).rstrip("u"), 0)
        rm = re.search(r"pub const %s: \w+ = (\S+);" % m.group(1), r)
        assert rm and int(rm.group(1), 0) == val, m.group(1)


def test_the_shim_only_calls_functions_that_exist_with_the_right_arity():
    r = rust_externs()
    for f in ("backend.rs", "batch.rs"):
        src = open(os.path.join(RUST, "src", "gpu", f)).read()
        for m in re.finditer(r"ffi::(fmx_\w+)\s*\(", src):
            name = m.group(1)
            assert name in r, (f, name)
            depth, i, args, cur = 1, m.end(), 0, ""
            while depth:                      # count top-level commas of the call
                ch = src[i]
                depth += ch in "([{"
                depth -= ch in ")]}"
                if depth == 1 and ch == ",":
                    args += bool(cur.strip())
                    cur = ""
                elif depth >= 1:
                    cur += ch
                i += 1
            args += bool(cur.strip())
            assert args == len(r[name][1]), (f, name, args, len(r[name][1]))
    # the trait methods of src/backend.rs:5-31 are all implemented
    b = open(os.path.join(RUST, "src", "gpu", "backend.rs")).read()
    for meth in ("fn get_l", "fn lf_map(", "fn lf_map2", "fn get_f", "fn fl_map", "fn len", "fn get_sa",
                 "fn search_range", "fn heap_size", "fn match_rows", "fn piece_id", "fn pieces_count"):
        assert meth in b, meth


def test_the_overlay_covers_the_multi_pieces_backend():
    """GpuIndexKind::Multi, impl HasMultiPieces over fmx_piece_id / fmx_pieces_count
    (backend.rs:34-40, multi_pieces.rs:201-224), fl_map -> None where the ABI returns the all-ones value
    (multi_pieces.rs:176-187), and the match_prefix_only rows over fmx_match_counts / fmx_match_rows
    (wrapper.rs:57-82, 203-217)."""
    b = open(os.path.join(RUST, "src", "gpu", "backend.rs")).read()
    assert re.search(r"Multi\s*=\s*ffi::FMX_KIND_MULTI", b)
    assert "impl<C: GpuCharacter> HasMultiPieces for GpuBackend<C>" in b
    assert "use crate::backend::{HasMultiPieces, HasPosition, SearchIndexBackend};" in b and "use crate::piece::PieceId;" in b
    body = b[b.index("impl<C: GpuCharacter> HasMultiPieces"):]
    body = body[:body.index("\n}\n") + 3]
    assert "ffi::fmx_piece_id(" in body and "ffi::fmx_pieces_count(" in body and "PieceId::from(" in body
    fl = b[b.index("fn fl_map"):]
    fl = fl[:fl.index("\n    }\n") + 7]
    assert "ffi::fmx_fl_map_batch(" in fl and "u64::MAX" in fl and "None" in fl and "Some(" in fl
    mr = b[b.index("fn match_rows"):]
    mr = mr[:mr.index("\n    }\n") + 7]
    assert "ffi::fmx_match_counts(" in mr and "ffi::fmx_match_rows(" in mr and "Box::new(s..e)" in mr
    # the trait signatures are the reference's (backend.rs:34-40), when the checkout is here
    ref = "/root/reference/src/backend.rs"
    if os.path.exists(ref):
        r = open(ref).read()
        for sig in ("fn piece_id(&self, i: usize) -> PieceId", "fn pieces_count(&self) -> usize",
                    "fn fl_map(&self, i: usize) -> Option<usize>"):
            assert sig in r and sig in b, sig
    # the patch gives the reference the two provided methods the overrides need
    p = open(os.path.join(RUST, "gpu-backend.patch")).read()
    assert "+    fn search_range(" in p and "+    fn match_rows<'a>(" in p
    assert "+            rows: backend.match_rows(i, e, match_prefix_only)," in p


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference checkout not present")
def test_the_patch_applies_to_the_reference(tmp_path):
    import shutil
    shutil.copytree("/root/reference/src", tmp_path / "src")
    p = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(RUST, "gpu-backend.patch")],
                       cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()
