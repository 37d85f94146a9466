import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the product library (hipcc cross-compiles
    gfx950 without a GPU) and the oracle once, exactly as __graft_entry__.build() does.  On the GPU box the
    prebuilt files travel with the snapshot and nothing is rebuilt."""
    import shutil
    import subprocess
    need = [os.path.join(ROOT, "ronkathon_amd", "libronk_ntt.so"), os.path.join(ROOT, "oracle", "libronk_oracle.so")]
    on_gpu_box = os.path.exists("/dev/kfd") and all(os.path.exists(f) for f in need)   # snapshot ships the built files
    if on_gpu_box or os.environ.get("RONK_NO_REBUILD") == "1":
        return
    if shutil.which("hipcc") and shutil.which("make"):
        # incremental: a no-op when nothing changed, and never a stale binary after an edit under csrc/
        subprocess.check_call(["make", "-C", ROOT, "-j8", "-s"])
    elif not all(os.path.exists(f) for f in need):
        raise RuntimeError("built libraries missing and no hipcc/make to build them")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests skip (instead of erroring) when no HIP device is visible, e.g. a plain `pytest` in the CPU-only
    container; on the GPU box nothing is skipped."""
    try:
        import ronkathon_amd
        have = ronkathon_amd.device_count() >= 1
    except Exception:  # noqa: BLE001
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="needs a HIP device (MI355X)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def refvec():
    with open(os.path.join(GOLDEN, "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def glvec():
    with open(os.path.join(GOLDEN, "goldilocks_derived.json")) as f:
        return json.load(f)


def splitmix_field(seed, n, p=0xFFFFFFFF00000001):
    """i.i.d. uniform on [0,p) by rejection from SplitMix64 (numpy, vectorised)."""
    import numpy as np
    out = np.empty(0, dtype=np.uint64)
    state = np.uint64(seed)
    with np.errstate(over="ignore"):
        while out.size < n:
            m = max(1024, int((n - out.size) * 1.01) + 16)
            idx = np.arange(1, m + 1, dtype=np.uint64)
            z = state + idx * np.uint64(0x9E3779B97F4A7C15)
            state = z[-1]
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            if p < (1 << 62):
                z = z % np.uint64(p)  # small test fields: reduce instead of rejecting
            out = np.concatenate([out, z[z < np.uint64(p)]])
    return out[:n].copy()
