"""A short seeded run of tools/fuzz_gpu.py inside the GPU suite: every C-ABI family against the oracle on random sizes, batches,
planner options, moduli, alignments and in-place forms (the long runs: profiles/r04_fuzz_gpu.txt)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_seeded_fuzz_sweep():
    env = dict(os.environ, FUZZ_SEED="4", FUZZ_SECONDS="36")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py")], env=env, capture_output=True, text=True,
                       timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    assert r.returncode == 0, tail
    assert "mismatches: 0" in r.stdout, tail
