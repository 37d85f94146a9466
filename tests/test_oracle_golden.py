"""Pin the CPU oracle on every golden vector the reference's own tests hold for the path
(SURVEY.md section 8c), plus first-principles Goldilocks vectors.  CPU only."""
import os

import numpy as np
import pytest

import oracle as orc
from oracle import OraclePanic

GP, GG = orc.GOLDILOCKS_P, orc.GOLDILOCKS_G


def gen(p):
    return orc.find_primitive_element(p)


def test_field_kats(refvec):
    for p, a, b, r in refvec["field_add"]["cases"]:
        assert orc.add(p, orc.new(p, a), orc.new(p, b)) == r
    for p, a, b, r in refvec["field_sub"]["cases"]:
        assert orc.sub(p, orc.new(p, a), orc.new(p, b)) == r
    for p, a, b, r in refvec["field_mul"]["cases"]:
        assert orc.mul(p, a, b) == r
    for p, a, e, r in refvec["field_pow"]["cases"]:
        assert orc.pow_(p, a, e) == r
    for p, a, r in refvec["field_inverse"]["cases"]:
        assert orc.inverse(p, a) == r
        assert orc.mul(p, orc.inverse(p, a), a) == 1
    for p, a in refvec["field_inverse"]["panics"]:
        with pytest.raises(OraclePanic):
            orc.inverse(p, a)
    for p, a, r in refvec["field_halve"]["cases"]:
        assert orc.div(p, a, 2) == r


def test_field_ext_sqrt(refvec):
    """FieldExt::sqrt / euler_criterion (prime/mod.rs:142-226): the reference's rstest cases, its list of the quadratic
    residues of GF(101), and the root pair's order / squares over every field of the tests"""
    v = refvec["field_sqrt"]
    for p, a, r0, r1 in v["cases"]:
        assert orc.sqrt(p, a) == (r0, r1)
    for p, a in v["panics"]:
        with pytest.raises(OraclePanic) as e:
            orc.sqrt(p, a)
        assert e.value.code == -13
    assert [a for a in range(1, 101) if orc.euler_criterion(101, a)] == v["residues_101"]
    assert not orc.euler_criterion(101, 0)
    for p in (17, 101, 127, GP, 0xFFFFFFFC00000001):
        rng = np.random.default_rng(p % 1000)
        xs = [int(x) % (p - 1) + 1 for x in rng.integers(1, 2**63, size=64)]
        for x in xs:
            y = x * x % p
            assert orc.sqrt(p, y) == (min(x, p - x), max(x, p - x))
        ys = np.array([x * x % p for x in xs], dtype=np.uint64)
        r0, r1 = orc.vec_sqrt(p, ys)
        assert all(int(a) * int(a) % p == int(y) and int(a) + int(b) == p for a, b, y in zip(r0, r1, ys))
        assert orc.vec_euler(p, ys).tolist() == [1] * 64


def test_prime_and_generator(refvec):
    assert not orc.is_prime(refvec["non_prime_panics"]["p"])
    for p in refvec["generator"]["primes"]:
        assert orc.is_prime(p)
        g = gen(p)
        val, counter = g, 1
        while val != 1:
            val = orc.mul(p, val, g)
            counter += 1
        assert counter == p - 1
    # the heuristic gives 2 / 14 / 3 and a NON-generator 3 for Goldilocks
    assert [gen(101), gen(17), gen(127), gen(GP)] == [2, 14, 3, 3]
    assert orc.pow_(GP, 3, (GP - 1) // 2) == 1          # 3 is a quadratic residue
    assert orc.pow_(GP, 7, (GP - 1) // 2) == GP - 1     # 7 is not
    assert orc.is_prime(GP)


def test_field_exhaustive_identities():
    # prime/mod.rs:346-384, prime/arithmetic.rs:221-236
    for p in (17, 101):
        for i in range(p):
            assert orc.add(p, i, 0) == i and orc.mul(p, i, 1) == i and orc.mul(p, i, 0) == 0
            assert orc.add(p, i, orc.neg(p, i)) == 0
            if i:
                assert orc.inverse(p, orc.inverse(p, i)) == i
        assert orc.new(p, p) == 0 and orc.new(p, 0) == 0


def test_roots(refvec):
    for p, n in refvec["no_root"]["cases"]:
        with pytest.raises(OraclePanic) as e:
            orc.primitive_root_of_unity(p, gen(p), n)
        assert e.value.code == -1
        with pytest.raises(OraclePanic):
            orc.dft(p, gen(p), [1, 2, 3])
    for p, n, w in refvec["roots_of_unity"]["cases"]:
        assert orc.primitive_root_of_unity(p, gen(p), n) == w


def test_polynomial_kats(refvec):
    v = refvec
    for p, c, x, y in v["poly_eval"]["cases"]:
        assert orc.poly_eval(p, c, x) == y
    d = v["poly_dft"]
    assert orc.dft(d["p"], gen(d["p"]), d["in"]).tolist() == d["out"]
    d = v["poly_fft"]
    assert orc.fft(d["p"], gen(d["p"]), d["in"]).tolist() == d["out"]
    d = v["poly_ifft_roundtrip"]
    g = gen(d["p"])
    assert orc.ifft(d["p"], g, orc.fft(d["p"], g, d["in"])).tolist() == d["in"]
    d = v["lagrange_eval"]
    g = gen(d["p"])
    assert orc.lagrange_eval(d["p"], orc.dft(d["p"], g, d["coeffs"]), orc.lagrange_nodes(d["p"], g, 4), d["x"]) == d["y"]
    assert orc.degree(v["degree"]["c"]) == v["degree"]["degree"]
    assert orc.leading_coefficient(v["leading_coefficient"]["c"]) == v["leading_coefficient"]["lc"]
    d = v["pow_mult"]
    assert orc.pow_mult(d["p"], d["c"], d["d2"], d["coeff"]).tolist() == d["out"]
    d = v["poly_add"]
    assert orc.poly_add(d["p"], d["a"], d["b"]).tolist() == d["out"]
    for a, b, r in v["poly_sub"]["cases"]:
        assert orc.poly_sub(101, a, b).tolist() == r
    d = v["poly_neg"]
    assert orc.poly_neg(d["p"], d["a"]).tolist() == d["out"]
    for a, b, q in v["poly_div"]["cases"]:
        assert orc.poly_divrem(101, a, b)[0].tolist() == q
    for a, b, r in v["poly_rem"]["cases"]:
        assert orc.poly_divrem(101, a, b)[1].tolist() == r
    for a, b, c in v["poly_mul"]["cases"]:
        assert orc.poly_mul(101, a, b).tolist() == c


def test_plonk_lagrange_polys_to_coefficient_form(refvec):
    """row N3: the compiler's selector / permutation polynomials are Lagrange-basis values on the 4th roots of
    unity of F_17 (src/compiler/program.rs:350-420); `ifft` (polynomial/mod.rs:430-484) takes them to coefficient form.  The
    reference holds the VALUES, not the coefficients, so the oracle's ifft is pinned here by the definition it inverts:
    evaluating the coefficients at omega^i (the O(D^2) `evaluate`, mod.rs:133-139) returns the reference's values."""
    d = refvec["plonk_lagrange_polys"]
    p, n = d["p"], d["n"]
    g = gen(p)
    nodes = orc.lagrange_nodes(p, g, n)
    assert int(nodes[1]) == 13                       # omega_4 of F_17 as the compiler uses it (program.rs:59-63)
    for name, vals in d["cases"].items():
        c = orc.ifft(p, g, vals)
        assert [orc.poly_eval(p, c, int(w)) for w in nodes] == vals, name
        assert orc.fft(p, g, c).tolist() == vals and orc.dft(p, g, c).tolist() == vals, name
        assert [orc.lagrange_eval(p, vals, nodes, int(w)) for w in (2, 3, 5, 7)] == [orc.poly_eval(p, c, w) for w in (2, 3, 5, 7)], name


def test_callers(refvec):
    d = refvec["rs_encode"]
    xs, ys = orc.rs_encode(d["p"], gen(d["p"]), d["msg"], d["n"])
    assert xs.tolist() == d["x"] and ys.tolist() == d["y"]
    # encode IS a size-n DFT of the zero-padded message
    assert orc.dft(d["p"], gen(d["p"]), orc.poly_from(d["msg"], d["n"])).tolist() == d["y"]
    xs7, ys7 = orc.rs_encode(127, 3, [1, 2, 3], 7)
    assert orc.dft(127, 3, orc.poly_from([1, 2, 3], 7)).tolist() == ys7.tolist()
    d = refvec["kzg_open_quotient"]
    assert orc.kzg_open_quotient(d["p"], d["coeffs"], d["z"]).tolist() == d["quot"]
    d = refvec["rs_decode"]
    for msg in d["cases"]:
        xs, ys = orc.rs_encode(d["p"], gen(d["p"]), msg, d["n"])
        assert orc.rs_decode(d["p"], xs, ys, len(msg)).tolist() == msg
        # any k surviving coordinates interpolate the same message (erasures)
        assert orc.rs_decode(d["p"], xs[[6, 1, 4, 0, 3]], ys[[6, 1, 4, 0, 3]], len(msg)).tolist() == msg
    with pytest.raises(orc.OraclePanic) as e:          # coincident nodes: numerator / ZERO -> unwrap on None
        orc.rs_decode(127, [1, 1, 2], [3, 4, 5], 3)
    assert e.value.code == -2
    assert orc.rs_decode(127, [], [], 0).tolist() == []


def test_curve_and_kzg_commit(refvec):
    """the reference's curve and KZG vectors pin the oracle's curve arithmetic"""
    v = refvec["curve"]
    c = orc.Curve(v["p"], v["nr"], v["a"], v["b"])
    g = v["g"]
    assert orc.curve_add(c, g, g) == v["multiples_of_g"]["2"]                      # doubling through Add
    acc = g
    for k in range(2, 18):
        acc = orc.curve_add(c, acc, g)
        if str(k) in v["multiples_of_g"]:
            assert acc == v["multiples_of_g"][str(k)], k
        assert orc.curve_mul(c, g, k) == acc
    assert acc == orc.INFINITY and orc.curve_mul(c, g, 17) == orc.INFINITY          # order 17
    assert orc.curve_mul(c, g, 0) == orc.INFINITY
    assert orc.curve_add(c, g, orc.INFINITY) == g and orc.curve_add(c, orc.INFINITY, g) == g
    assert orc.curve_add(c, g, v["multiples_of_g"]["16"]) == orc.INFINITY           # g + (-g)
    assert orc.curve_add(c, v["g2"], v["g2"]) == v["two_g2"]                        # extension-field doubling
    assert orc.curve_is_on_curve(c, v["g2"]) and not orc.curve_is_on_curve(c, v["false_point"])
    srs = [orc.curve_mul(c, g, pow(v["tau"], i, v["scalar_order"])) for i in range(7)]   # kzg/setup.rs:12-40
    assert srs == v["g1_srs"]
    k = refvec["kzg_commit"]
    for case in k["cases"]:
        assert orc.kzg_commit(c, case["coeffs"], srs) == case["commit"]
    o = k["opening"]
    quot = orc.kzg_open_quotient(v["scalar_order"], o["coeffs"], o["z"])
    assert orc.kzg_commit(c, quot.tolist(), srs) == o["open"]                         # kzg::open = commit(quotient)
    with pytest.raises(orc.OraclePanic) as e:
        orc.kzg_commit(c, [1, 2], [v["false_point"], g])
    assert e.value.code == -11
    with pytest.raises(orc.OraclePanic) as e:                                         # srs shorter than the coefficients
        orc.kzg_commit(c, [1, 2, 3], [g, g])
    assert e.value.code == -6


def test_reference_quirks():
    # Lagrange evaluate AT a node: the fold's `return c` replaces the accumulator and the product
    # with l(x) == 0 gives ZERO (polynomial/mod.rs:382-415)
    nodes = orc.lagrange_nodes(101, 2, 4)
    assert orc.lagrange_eval(101, [10, 79, 99, 18], nodes, int(nodes[1])) == 0
    # long division: the loop guard uses the divisor's UNTRIMMED length (mod.rs:186-188)
    q, r = orc.poly_divrem(101, [1, 2, 3], [1, 1, 0, 0])
    assert q.tolist() == [0, 0, 0] and r.tolist() == [1, 2, 3]
    # zero divisor: rposition(..).unwrap() panics
    with pytest.raises(OraclePanic):
        orc.poly_divrem(101, [1, 2, 3], [0, 0])
    # pow(0, 0) == ONE; inverse(0) is None
    assert orc.pow_(101, 0, 0) == 1
    # fft on a non power of two is a compile-time error in the reference
    with pytest.raises(OraclePanic) as e:
        orc.fft(127, 3, [1, 2, 3])
    assert e.value.code == -3
    # power of two that does not divide p-1: 8 over F_101
    with pytest.raises(OraclePanic) as e:
        orc.fft(101, 2, list(range(8)))
    assert e.value.code == -1


def test_goldilocks_derived(glvec):
    v = glvec
    assert v["p"] == GP
    for k, w in v["roots"].items():
        assert orc.primitive_root_of_unity(GP, GG, 1 << int(k)) == w
    for k, ni in v["n_inverse"].items():
        assert orc.inverse(GP, 1 << int(k)) == ni
    assert orc.fft(GP, GG, [1, 2, 3, 4]).tolist() == v["dft_1234"]
    for case in v["dft_random"]:
        assert orc.dft(GP, GG, case["in"]).tolist() == case["out"]
        assert orc.fft(GP, GG, case["in"]).tolist() == case["out"]
        assert orc.ifft(GP, GG, case["out"]).tolist() == case["in"]
    for case in v["dft_non_pow2"]:
        assert orc.dft(GP, GG, case["in"]).tolist() == case["out"]
    for case in v["mul_random"]:
        assert orc.poly_mul(GP, case["a"], case["b"]).tolist() == case["out"]
    e = v["field_edge"]
    vals = e["values"]
    for i, a in enumerate(vals):
        for j, b in enumerate(vals):
            assert orc.add(GP, a, b) == e["add"][i][j]
            assert orc.sub(GP, a, b) == e["sub"][i][j]
            assert orc.mul(GP, a, b) == e["mul"][i][j]
        if a:
            assert orc.inverse(GP, a) == e["inv"][i]


def test_generic_prime64_derived():
    """the oracle on the generic odd 64-bit primes of the Montgomery tile path (round 5), pinned like Goldilocks: vectors computed
    from the definitions with Python integers (tests/golden/make_prime64_vectors.py; no oracle, no library).  The GPU suite then
    compares the library with the oracle over these same fields (tests/test_gpu_mont.py)."""
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prime64_derived.json")) as f:
        fields = json.load(f)["fields"]
    assert [(e["p"], e["g"]) for e in fields] == [(0xFFFFFFFC00000001, 10), (0x3A00000000000001, 3), (0xC0000001, 5),
                                                   (0xFFFFFFFF00000001, 343), (0xFFFFFFFF00000001, 7)]
    for e in fields:
        p, g = e["p"], e["g"]
        assert orc.is_prime(p)
        for k, w in e["roots"].items():
            assert orc.primitive_root_of_unity(p, g, 1 << int(k)) == w
        for k, ni in e["n_inverse"].items():
            assert orc.inverse(p, (1 << int(k)) % p) == ni
        assert orc.fft(p, g, [1, 2, 3, 4]).tolist() == e["dft_1234"]
        for case in e["dft_random"]:
            assert orc.dft(p, g, case["in"]).tolist() == case["out"]
            assert orc.fft(p, g, case["in"]).tolist() == case["out"]
            assert orc.ifft(p, g, case["out"]).tolist() == case["in"]
        for case in e["mul_random"]:
            assert orc.poly_mul(p, case["a"], case["b"]).tolist() == case["out"]
        for case in e["divrem_random"]:
            q, r = orc.poly_divrem(p, case["a"], case["b"])
            assert q.tolist() == case["quot"] and r.tolist() == case["rem"]
        for case in e["evaluate"]:
            assert orc.poly_eval(p, case["c"], case["x"]) == case["out"]
        for case in e["lagrange_evaluate"]:
            assert orc.lagrange_nodes(p, g, len(case["nodes"])).tolist() == case["nodes"]
            assert orc.lagrange_eval(p, case["values"], case["nodes"], case["x"]) == case["out"]
        for case in e["reed_solomon"]:
            xs, ys = orc.rs_encode(p, g, case["msg"], case["n"])
            assert xs.tolist() == case["xs"] and ys.tolist() == case["ys"]
            assert orc.rs_decode(p, xs, ys, len(case["msg"])).tolist() == case["msg"]
        ed = e["field_edge"]
        vals = ed["values"]
        for i, a in enumerate(vals):
            for j, b in enumerate(vals):
                assert orc.add(p, a, b) == ed["add"][i][j]
                assert orc.sub(p, a, b) == ed["sub"][i][j]
                assert orc.mul(p, a, b) == ed["mul"][i][j]
            if a:
                assert orc.inverse(p, a) == ed["inv"][i]


def test_self_consistency_large():
    # dft == fft == definition on moderately large n; ifft(fft) == id at 2^16 (config 2)
    from conftest import splitmix_field
    x = splitmix_field(0x5EED0002, 1 << 10)
    assert np.array_equal(orc.dft(GP, GG, x), orc.fft(GP, GG, x))
    x = splitmix_field(0x5EED0002, 1 << 16)
    assert np.array_equal(orc.ifft(GP, GG, orc.fft(GP, GG, x)), x)
    # NTT-multiply == schoolbook (config 1 shape: 17 x 17 over F_101 is schoolbook only)
    a = splitmix_field(1, 17, 101); b = splitmix_field(2, 17, 101)
    c = orc.poly_mul(101, a, b)
    assert c.size == 33
    a = splitmix_field(3, 100); b = splitmix_field(4, 157)
    fa = orc.fft(GP, GG, orc.poly_from(a, 256)); fb = orc.fft(GP, GG, orc.poly_from(b, 256))
    c = orc.ifft(GP, GG, orc.vec_mul(GP, fa, fb))
    assert np.array_equal(c[:256], orc.poly_mul(GP, a, b))


def test_bn254_oracle_pins():
    """oracle/bn254.py against public constants and first principles (the reference has no 254-bit curve: 'derived' pins)"""
    from oracle import bn254 as o
    assert o.P == 21888242871839275222246405745257275088696311157297823662689037894645226208583
    assert o.R == 21888242871839275222246405745257275088548364400416034343698204186575808495617
    assert o.on_curve(o.G) and o.add(o.G, o.G) == o.TWO_G and o.mul(2, o.G) == o.TWO_G
    assert o.mul(o.R, o.G) is None and o.mul(o.R - 1, o.G) == o.neg(o.G) and o.mul(o.R + 5, o.G) == o.mul(5, o.G)
    assert o.add(o.G, o.neg(o.G)) is None and o.add(None, o.G) == o.G
    pts = o.multiples(20)
    assert all(o.on_curve(p) for p in pts) and pts[6] == o.mul(7, o.G)
    ks = [3, 0, 2**255 + 12345, o.R - 1, 1] + list(range(5, 20))
    want = o.mul(sum(k * (i + 1) for i, k in enumerate(ks)) % o.R, o.G)
    assert o.msm(pts, ks) == want
