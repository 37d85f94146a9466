"""rust/ronk-goldilocks/in_tree/ronkathon.patch: the edit that makes ronkathon's own `poly.fft()` / `a * b` / `a / b` call
sites reach the GPU for `Goldilocks` (row N3 of SURVEY.md 8f).  No rustc here, so what can be checked is checked:
the patch applies cleanly to the reference (`git apply --check` on a scratch copy), it is the one the generator produces
from the crate's current sources, and the vendored FFI block is the crate's FFI block."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "rust", "ronk-goldilocks", "in_tree", "ronkathon.patch")
REF = "/root/reference"


def new_file_body(patch, path):
    """content of a file the patch creates (lines after its hunk header, '+' stripped)"""
    marker = "+++ b/" + path + "\n"
    i = patch.index(marker)
    j = patch.find("\ndiff --git ", i)
    body = patch[i + len(marker): j + 1 if j >= 0 else len(patch)]
    lines = body.splitlines(keepends=True)
    assert lines[0].startswith("@@ -0,0 +1,")
    assert all(ln.startswith("+") for ln in lines[1:])
    return "".join(ln[1:] for ln in lines[1:])


def test_patch_vendors_the_crate_ffi_verbatim():
    patch = open(PATCH).read()
    ffi = open(os.path.join(ROOT, "rust", "ronk-goldilocks", "src", "ffi.rs")).read()
    assert new_file_body(patch, "src/algebra/field/goldilocks/ffi.rs") == ffi
    gpu = new_file_body(patch, "src/algebra/field/goldilocks/gpu.rs")
    assert "ronkathon::" not in gpu and "use super::{" in gpu and "fn fft_gpu" in gpu
    disp = new_file_body(patch, "src/polynomial/dispatch.rs")
    for item in ("default fn fft_impl", "default fn ifft_impl", "default fn dft_impl", "default fn evaluate_impl",
                 "default fn quotient_and_remainder_impl", "Accelerated::fft_gpu(self)", "Accelerated::mul_gpu(&self, &rhs)"):
        assert item in disp, item
    assert "unimplemented!" not in patch


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is only present in the build container")
def test_patch_applies_to_the_reference_and_is_current(tmp_path):
    scratch = tmp_path / "ronkathon"
    scratch.mkdir()
    shutil.copytree(os.path.join(REF, "src"), scratch / "src")
    shutil.copy(os.path.join(REF, "Cargo.toml"), scratch / "Cargo.toml")
    subprocess.check_call(["git", "init", "-q", "."], cwd=scratch)
    subprocess.check_call(["git", "apply", "--check", PATCH], cwd=scratch)
    subprocess.check_call(["git", "apply", PATCH], cwd=scratch)
    mod = (scratch / "src" / "polynomial" / "mod.rs").read_text()
    for name in ("evaluate_reference", "quotient_and_remainder_reference", "dft_reference", "fft_reference", "ifft_reference"):
        assert "fn " + name in mod
    assert "default fn mul" in (scratch / "src" / "polynomial" / "arithmetic.rs").read_text()
    assert "pub mod goldilocks;" in (scratch / "src" / "algebra" / "field" / "mod.rs").read_text()
    for f in ("build.rs", "src/polynomial/dispatch.rs", "src/algebra/field/goldilocks/mod.rs", "src/algebra/field/goldilocks/ffi.rs",
              "src/algebra/field/goldilocks/gpu.rs"):
        assert (scratch / f).is_file()
    # the committed patch is what the generator makes from the crate's sources today (written to a scratch file: the tracked
    # patch is only read)
    fresh = tmp_path / "fresh.patch"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_in_tree_patch.py"), REF, "--output=%s" % fresh],
                          stdout=subprocess.DEVNULL)
    assert fresh.read_text() == open(PATCH).read(), "rust sources changed: re-run tools/make_in_tree_patch.py and commit the patch"
