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
