"""SURVEY.md section 5 (sanitizer builds): `make sanitize` compiles the oracle under ASan+UBSan and the host-compilable
product sources -- csrc/gl64.h, csrc/ntt_tile.h + csrc/ntt_small.h + csrc/plan.h through the fiber emulator (UBSan), csrc/bn254.h (ASan+UBSan) --, then runs them
(any report aborts: -fno-sanitize-recover).  CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not shutil.which("gcc") or not shutil.which("make"), reason="needs gcc + make")
def test_make_sanitize():
    out = subprocess.run(["make", "-C", ROOT, "-s", "sanitize"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("OK") >= 7 and "bn254 sanitize run ok" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
