"""Builds tests/cpp/test_host_mirror.cpp (C++ host mirror of the reference interface,
ronkathon_amd/host/ronkathon.hpp, over the C ABI) and runs it on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.bin")


def build():
    src = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")
    lib = os.path.join(ROOT, "ronkathon_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", EXE, src, "-L" + lib, "-lronk_ntt", "-Wl,-rpath," + lib,
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    return EXE


def test_cpp_host_mirror_compiles():
    """CPU: the header and the test translate and link against the library (no compute)."""
    build()


@pytest.mark.gpu
def test_cpp_host_mirror_on_gpu():
    exe = EXE if os.path.exists(EXE) else build()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


# ---- the Rust shim's FFI call sequence, replayed in C (rust/ronk-goldilocks/src/polynomial.rs)
REPLAY = os.path.join(ROOT, "tests", "cpp", "test_rust_ffi_replay.bin")


def build_replay():
    src = os.path.join(ROOT, "tests", "cpp", "test_rust_ffi_replay.c")
    lib, orc = os.path.join(ROOT, "ronkathon_amd"), os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-o", REPLAY, src, "-L" + lib, "-lronk_ntt", "-L" + orc, "-lronk_oracle",
                           "-Wl,-rpath," + lib, "-Wl,-rpath," + orc, "-Wl,-rpath-link,/opt/rocm/lib"])
    return REPLAY


def test_rust_ffi_replay_compiles_and_shim_sources_match_the_header():
    """CPU: the replay links against the library, and every `extern "C"` item of the Rust shim names a symbol the header
    declares with the same number of parameters."""
    import re
    build_replay()
    hdr = open(os.path.join(ROOT, "include", "ronk_ntt.h")).read()
    ffi = open(os.path.join(ROOT, "rust", "ronk-goldilocks", "src", "ffi.rs")).read()
    decls = re.findall(r"pub fn (ronk_\w+)\(([^)]*)\)", ffi)
    assert len(decls) >= 10
    for name, params in decls:
        m = re.search(r"^(?:int|const char\*) %s\(([^;]*?)\);" % name, hdr, re.S | re.M)
        assert m, "the shim binds %s, which the header does not declare" % name
        n_rust = len([q for q in params.split(",") if q.strip()])
        c_params = m.group(1).strip()
        n_c = 0 if c_params in ("", "void") else len(c_params.split(","))
        assert n_rust == n_c, (name, n_rust, n_c)


@pytest.mark.gpu
def test_rust_ffi_replay_on_gpu():
    exe = REPLAY if os.path.exists(REPLAY) else build_replay()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
