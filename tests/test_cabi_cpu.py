"""CPU-side checks of the boundary: the library loads, exports every symbol include/ronk_ntt.h
declares, its host-side integer logic agrees with the oracle, and without a GPU every compute
entry point fails loudly (no CPU fallback).  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    from ronkathon_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "ronk_ntt.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(ronk_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = C.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libronk_ntt.so does not export %s" % name
    assert declared == L.EXPORTS, "ctypes table and header disagree: %s" % (set(declared) ^ set(L.EXPORTS))


def test_host_logic_matches_oracle():
    from ronkathon_amd import _lib as L
    for p in (2, 3, 5, 17, 101, 127, 257, 65537, 3 * 2**30 + 1, L.GOLDILOCKS_P):
        assert L.lib.ronk_check_prime(p) == 0
        try:
            g = L.out_scalar(L.lib.ronk_primitive_element, p)
        except L.RonkPanic as e:
            # the reference's heuristic panics "generator not found" for P = 3 (its loop never runs)
            assert e.code == L.ERR_NO_GENERATOR
            with pytest.raises(orc.OraclePanic):
                orc.find_primitive_element(p)
            continue
        if p == L.GOLDILOCKS_P:
            assert g == 7
        elif p == 2:
            assert g == 1
        else:
            assert g == orc.find_primitive_element(p)
        for n in (1, 2, 3, 4, 5, 8, 16, 64, 2**16):
            if (p - 1) % n == 0:
                assert L.out_scalar(L.lib.ronk_root_of_unity, p, g, n) == orc.primitive_root_of_unity(p, g, n)
            else:
                with pytest.raises(L.RonkPanic) as e:
                    L.out_scalar(L.lib.ronk_root_of_unity, p, g, n)
                assert e.value.code == L.ERR_NO_ROOT
    for p in (100, 4, 9, 2**32 + 1 + 0, 0xFFFFFFFF00000001 - 2):
        assert (L.lib.ronk_check_prime(p) == 0) == orc.is_prime(p)
    assert L.lib.ronk_strerror(-1) == b"n must divide p^q - 1"
    assert L.lib.ronk_strerror(-4) == b"input is not a prime number"


def test_no_cpu_fallback():
    from ronkathon_amd import _lib as L
    if L.device_count() > 0:
        pytest.skip("a GPU is present")
    a = np.arange(4, dtype=np.uint64)
    out = np.empty_like(a)
    assert L.lib.ronk_vec_add(101, L.ptr(a), L.ptr(a), L.ptr(out), 4) == L.ERR_NO_DEVICE
    assert L.lib.ronk_dft(101, 2, L.ptr(a), L.ptr(out), 4) == L.ERR_NO_DEVICE
    h = C.c_void_p()
    assert L.lib.ronk_plan_create(C.byref(h), L.GOLDILOCKS_P, 7, 10, 1, -1) == L.ERR_NO_DEVICE
    # argument errors are reported before the device is needed, with the reference's panic condition
    assert L.lib.ronk_plan_create(C.byref(h), 101, 2, 3, 1, -1) == L.ERR_NO_ROOT
    assert L.lib.ronk_plan_create(C.byref(h), 100, 2, 1, 1, -1) == L.ERR_NOT_PRIME
    import ronkathon_amd as R
    with pytest.raises(R.RonkPanic) as e:
        R.Polynomial.new(R.PlutoBaseField, [1, 2, 3, 4]).fft()
    assert e.value.code == L.ERR_NO_DEVICE


def test_host_mirror_scalar_surface(refvec):
    import ronkathon_amd as R
    for name, op in (("field_add", "__add__"), ("field_sub", "__sub__"), ("field_mul", "__mul__")):
        for p, a, b, r in refvec[name]["cases"]:
            F = R.PrimeField(p)
            assert getattr(F(a), op)(F(b)) == F(r)
    for p, a, e, r in refvec["field_pow"]["cases"]:
        assert R.PrimeField(p)(a).pow(e) == R.PrimeField(p)(r)
    for p, a, r in refvec["field_inverse"]["cases"]:
        assert R.PrimeField(p)(a).inverse() == R.PrimeField(p)(r)
    assert R.PlutoBaseField(0).inverse() is None
    with pytest.raises(R.RonkPanic):
        R.PlutoBaseField(3) / R.PlutoBaseField(0)
    for p, a, r in refvec["field_halve"]["cases"]:
        assert R.PrimeField(p)(a) / R.PrimeField(p)(2) == R.PrimeField(p)(r)
    with pytest.raises(R.RonkPanic):
        R.PrimeField(100)
    assert [int(R.PrimeField(p).PRIMITIVE_ELEMENT) for p in (101, 17, 127)] == [2, 14, 3]
    assert int(R.GoldilocksField.PRIMITIVE_ELEMENT) == 7
    for p, n, w in refvec["roots_of_unity"]["cases"]:
        assert int(R.PrimeField(p).primitive_root_of_unity(n)) == w


def test_minor_trait_surface_of_the_mirror():
    """SURVEY.md 8(b) "minor surface": Display, From<i32>, FromStr, Distribution -- host-only, no device needed"""
    import numpy as np
    from ronkathon_amd.field import PrimeField
    from ronkathon_amd.polynomial import Polynomial, Monomial, Lagrange
    F = PrimeField(101)
    assert F(-3) == F(98) and F.from_str(" 205 ") == F(3) and str(F(7)) == "7"          # prime/mod.rs:125-127, :250-270
    rng = np.random.default_rng(1)
    xs = [F.sample(rng) for _ in range(3)]                                               # 28-bit draws, rejection
    assert all(0 <= int(x) < 101 for x in xs)
    G = PrimeField(0xFFFFFFFF00000001)
    assert all(int(G.sample(rng)) < (1 << 28) for _ in range(20))                        # the sampler never exceeds 28 bits
    p = Polynomial.__new__(Polynomial)
    p.field, p.basis, p.coefficients = F, Monomial(), np.array([1, 2, 3], dtype=np.uint64)
    assert str(p) == "1 + 2x^1 + 3x^2"                                                     # polynomial/mod.rs:326-342
    lag = Lagrange.__new__(Lagrange)
    lag.nodes = np.array([1, 84, 16], dtype=np.uint64)
    p.basis = lag
    assert str(p) == "1*l_1(x) + 2*l_84(x) + 3*l_16(x)"                                    # polynomial/mod.rs:487-501


def test_argument_errors_of_the_round_3_entry_points_need_no_device():
    """ronk_plan_opts / the many-arrays entry points / the exchange selector: invalid arguments are refused before any device
    work (so this runs in the CPU-only container), and the new error code has a text"""
    from ronkathon_amd import _lib as L
    h = C.c_void_p()
    for bad in (0, 3, -2):
        opts = L.PlanOpts(-1, -1, bad)
        assert L.lib.ronk_plan_create_opts(C.byref(h), L.GOLDILOCKS_P, 7, 10, 1, -1, C.byref(opts)) == L.ERR_INVALID
    assert L.lib.ronk_plan_create_opts(C.byref(h), 101, 2, 3, 1, -1, None) == L.ERR_NO_ROOT          # NULL options = defaults
    assert L.lib.ronk_plan_in_flight(None) == L.ERR_INVALID
    assert L.lib.ronk_ntt_forward_many_dev(None, None, None, 2, None) == L.ERR_INVALID
    devs = (C.c_int * 2)(0, 1)
    assert L.lib.ronk_sharded_plan_create_ex(C.byref(h), 20, 0, devs, 2, 0, 7) == L.ERR_INVALID       # unknown exchange
    assert L.lib.ronk_sharded_plan_exchange(None) == L.ERR_INVALID
    assert L.ERR_RCCL == -12 and L.lib.ronk_strerror(L.ERR_RCCL) == b"RCCL error"
    opts = L.PlanOpts()
    assert (opts.tile_log2_columns, opts.twiddle_matrix_log2_max, opts.in_flight, opts.split_log2_rows, opts.three_pass_from_log2, list(opts.reserved)) == (-1, -1, -1, 0, 0, [0] * 3)
    assert C.sizeof(L.PlanOpts) == 32                                                                  # 8 ints, as in the header


def test_product_does_not_reach_the_oracle():
    """the oracle is test infrastructure: no source of the product (the package, the C ABI header, the Rust crate, the library's
    link line) names it, and the built library has no dependency on libronk_oracle.so"""
    import subprocess
    hits = []
    for top in ("ronkathon_amd", "include", os.path.join("rust", "ronk-goldilocks", "src")):
        for dirpath, _dirs, files in os.walk(os.path.join(ROOT, top)):
            if "__pycache__" in dirpath:
                continue
            for fn in files:
                if fn.endswith((".so", ".pyc", ".o")):
                    continue
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                for needle in ("import oracle", "from oracle", "ronk_oracle", "orc_"):
                    if re.search(r"\b" + re.escape(needle), text):
                        hits.append((os.path.relpath(os.path.join(dirpath, fn), ROOT), needle))
    assert not hits, hits
    from ronkathon_amd import _lib as L
    needed = subprocess.run(["readelf", "-d", L.LIB_PATH], capture_output=True, text=True).stdout
    assert "NEEDED" in needed and "oracle" not in needed
    mk = open(os.path.join(ROOT, "Makefile")).read()
    link = [ln for ln in mk.splitlines() if "-shared" in ln]
    assert link and all("oracle" not in ln for ln in link)
