"""Builds tests/cpp/test_host_mirror.cpp (C++ host mirror of the reference interface,
ronkathon_amd/host/ronkathon.hpp, over the C ABI) and runs it on the GPU."""
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.bin")


def _src_sha(*paths):
    h = hashlib.sha256()
    for p_ in paths:
        with open(p_, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _current(exe, *srcs):
    """a built test binary travels with the snapshot to the GPU box (like the .so files): it is used there only if it was built
    from the sources as they are now (their hash sits beside it), otherwise rebuilt -- a stale binary once failed the GPU suite
    on an expectation that had been corrected in the source"""
    try:
        with open(exe + ".sha") as f:
            return os.path.exists(exe) and f.read().strip() == _src_sha(*srcs)
    except OSError:
        return False


def _stamp(exe, *srcs):
    with open(exe + ".sha", "w") as f:
        f.write(_src_sha(*srcs))


def build():
    src = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")
    hdr = os.path.join(ROOT, "ronkathon_amd", "host", "ronkathon.hpp")
    lib = os.path.join(ROOT, "ronkathon_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", EXE, src, "-L" + lib, "-lronk_ntt", "-Wl,-rpath," + lib,
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    _stamp(EXE, src, hdr)
    return EXE


def test_cpp_host_mirror_compiles():
    """CPU: the header and the test translate and link against the library (no compute)."""
    build()


@pytest.mark.gpu
def test_cpp_host_mirror_on_gpu():
    src = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")
    exe = EXE if _current(EXE, src, os.path.join(ROOT, "ronkathon_amd", "host", "ronkathon.hpp")) else build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


# ---- the Rust shim's FFI call sequence, replayed in C (rust/ronk-goldilocks/src/polynomial.rs)
REPLAY = os.path.join(ROOT, "tests", "cpp", "test_rust_ffi_replay.bin")


def build_replay():
    src = os.path.join(ROOT, "tests", "cpp", "test_rust_ffi_replay.c")
    lib, orc = os.path.join(ROOT, "ronkathon_amd"), os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-o", REPLAY, src, "-L" + lib, "-lronk_ntt", "-L" + orc, "-lronk_oracle",
                           "-Wl,-rpath," + lib, "-Wl,-rpath," + orc, "-Wl,-rpath-link,/opt/rocm/lib"])
    _stamp(REPLAY, src, os.path.join(ROOT, "include", "ronk_ntt.h"))
    return REPLAY


# Rust FFI type -> the C type it must face in include/ronk_ntt.h (after normalisation: no parameter names, no spaces)
RUST_TO_C = {
    "u64": "uint64_t", "u32": "uint32_t", "usize": "size_t", "c_int": "int",
    "*const u64": "constuint64_t*", "*mut u64": "uint64_t*", "*mut c_int": "int*", "*const c_int": "constint*",
    "*mut c_void": "void*", "*const c_void": "constvoid*", "*mut *mut c_void": "void**",
    "*const *const u64": "constuint64_t*const*", "*const *mut u64": "uint64_t*const*",
    "*mut RonkPlan": "ronk_plan*", "*const RonkPlan": "constronk_plan*", "*mut *mut RonkPlan": "ronk_plan**",
    "*const RonkPlanOpts": "constronk_plan_opts*",
    "*mut RonkShardedPlan": "ronk_sharded_plan*", "*const RonkShardedPlan": "constronk_sharded_plan*",
    "*mut *mut RonkShardedPlan": "ronk_sharded_plan**",
    "*const c_char": "constchar*",
}


def _c_param_type(decl):
    """`const uint64_t* d_in` -> `constuint64_t*`; `uint64_t out[8]` -> `uint64_t*` (array parameters decay)"""
    import re
    decl = decl.strip()
    arr = re.search(r"\[\d*\]$", decl)
    if arr:
        decl = decl[:arr.start()]
    m = re.match(r"^(.*?)(\w+)$", decl.strip())          # the last identifier is the parameter name
    ty = m.group(1) if m and m.group(1).strip() else decl
    ty = ty.replace(" ", "")
    return ty + ("*" if arr else "")


def test_rust_ffi_replay_compiles_and_shim_sources_match_the_header():
    """CPU: the replay links against the library, and every `extern "C"` item of the Rust shim names a symbol the header
    declares with the same number of parameters, the same parameter TYPES and the same return type.  Also: the shim calls no
    private item of the reference (its test once called `quotient_and_remainder`, src/polynomial/mod.rs:170)."""
    import re
    build_replay()
    hdr = open(os.path.join(ROOT, "include", "ronk_ntt.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    ffi = open(os.path.join(ROOT, "rust", "ronk-goldilocks", "src", "ffi.rs")).read()
    ffi = re.sub(r"//[^\n]*", "", ffi)
    decls = re.findall(r"pub fn (ronk_\w+)\(([^)]*)\)\s*->\s*([^;]+);", ffi)
    assert len(decls) >= 40
    for name, params, ret in decls:
        m = re.search(r"^(int|const char\*) %s\(([^;]*?)\);" % name, hdr, re.S | re.M)
        assert m, "the shim binds %s, which the header does not declare" % name
        assert RUST_TO_C[ret.strip()] == m.group(1).replace(" ", ""), (name, ret)
        rust_types = [q.split(":", 1)[1].strip() for q in params.split(",") if q.strip()]
        c_params = m.group(2).strip()
        c_types = [] if c_params in ("", "void") else [_c_param_type(q) for q in c_params.split(",")]
        assert len(rust_types) == len(c_types), (name, rust_types, c_types)
        for i, (rt, ct) in enumerate(zip(rust_types, c_types)):
            assert rt in RUST_TO_C, (name, i, rt)
            assert RUST_TO_C[rt] == ct, "%s parameter %d: Rust `%s` faces C `%s`" % (name, i, rt, ct)
    # ronk_plan_opts field for field
    m = re.search(r"typedef struct ronk_plan_opts \{(.*?)\} ronk_plan_opts;", hdr, re.S)
    c_fields = [re.sub(r"\s+", " ", f.strip()) for f in m.group(1).split(";") if f.strip()]
    assert c_fields == ["int tile_log2_columns", "int twiddle_matrix_log2_max", "int in_flight", "int split_log2_rows", "int three_pass_from_log2", "int reserved[3]"]
    r_fields = re.search(r"pub struct RonkPlanOpts \{(.*?)\}", ffi, re.S).group(1)
    assert re.findall(r"pub (\w+):\s*([^,]+),", r_fields) == [("tile_log2_columns", "c_int"), ("twiddle_matrix_log2_max", "c_int"),
                                                              ("in_flight", "c_int"), ("split_log2_rows", "c_int"), ("three_pass_from_log2", "c_int"), ("reserved", "[c_int; 3]")]
    # visibility desk-check: every method the shim's sources call on the reference's Polynomial is `pub` there
    # (src/polynomial/mod.rs:98 new, :113 degree, :133 evaluate, :240 dft, :273 fft, :358 Lagrange new, :382 evaluate,
    # :430 ifft); the private ones (:101 trim_zeros, :170 quotient_and_remainder, :295 fft_recursive, :456 ifft_recursive)
    # must not appear as calls
    srcs = ""
    for f in ("polynomial.rs", "device.rs", "field.rs", "bn254.rs"):
        srcs += re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "rust", "ronk-goldilocks", "src", f)).read())
    for private in ("quotient_and_remainder(", "trim_zeros(", "fft_recursive(", "ifft_recursive("):
        assert "." + private not in srcs and "::" + private not in srcs, private


@pytest.mark.gpu
def test_rust_ffi_replay_on_gpu():
    exe = REPLAY if _current(REPLAY, os.path.join(ROOT, "tests", "cpp", "test_rust_ffi_replay.c"), os.path.join(ROOT, "include", "ronk_ntt.h")) else build_replay()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
