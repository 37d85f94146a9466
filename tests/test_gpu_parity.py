"""Parity of the HIP path (through the C ABI) against the oracle, the reference's golden vectors
and size-independent properties.  Bit-exact: all integer work.  Needs a real MI355X (-m gpu)."""
import os

import numpy as np
import pytest

from conftest import splitmix_field

pytestmark = pytest.mark.gpu

GP, GG = 0xFFFFFFFF00000001, 7


@pytest.fixture(scope="module")
def R():
    import ronkathon_amd as R
    assert R.device_count() >= 1
    return R


@pytest.fixture(scope="module")
def orc():
    import oracle
    return oracle


def adversarial(n):
    """all-zero, all-(p-1), single 1 at index 0 / n-1"""
    z = np.zeros(n, dtype=np.uint64)
    m = np.full(n, GP - 1, dtype=np.uint64)
    e0 = z.copy(); e0[0] = 1
    e1 = z.copy(); e1[n - 1] = 1
    return [z, m, e0, e1]


# ---------------------------------------------------------------- reference golden vectors on the GPU path
def test_field_kats_on_gpu(R, refvec):
    for name, fn in (("field_add", "vec_add"), ("field_sub", "vec_sub"), ("field_mul", "vec_mul")):
        for p in (17, 101):
            F = R.PrimeField(p)
            cases = [c for c in refvec[name]["cases"] if c[0] == p]
            a = [c[1] % p for c in cases]; b = [c[2] % p for c in cases]
            assert getattr(F, fn)(a, b).tolist() == [c[3] for c in cases]
    for p, a, e, r in refvec["field_pow"]["cases"]:
        assert R.PrimeField(p).vec_pow([a], e).tolist() == [r]
    for p in (17, 101):
        F = R.PrimeField(p)
        cases = [c for c in refvec["field_inverse"]["cases"] if c[0] == p]
        assert F.vec_inv([c[1] for c in cases]).tolist() == [c[2] for c in cases]
        with pytest.raises(R.RonkPanic) as e:
            F.vec_inv([1, 0, 2])
        assert e.value.code == -2
        halves = [c for c in refvec["field_halve"]["cases"] if c[0] == p]
        inv2 = int(F.vec_inv([2])[0])
        assert F.vec_mul([c[1] for c in halves], [inv2] * len(halves)).tolist() == [c[2] for c in halves]
        # exhaustive identities (prime/mod.rs:346-384)
        allv = np.arange(p, dtype=np.uint64)
        assert np.array_equal(F.vec_add(allv, F.vec_neg(allv)), np.zeros(p, dtype=np.uint64))
        nz = allv[1:]
        assert np.array_equal(F.vec_inv(F.vec_inv(nz)), nz)
        assert np.array_equal(F.vec_mul(nz, F.vec_inv(nz)), np.ones(p - 1, dtype=np.uint64))


def test_field_ext_sqrt_on_gpu(R, orc, refvec):
    """FieldExt::sqrt / euler_criterion over arrays (ronk_vec_sqrt / ronk_vec_euler; reference prime/mod.rs:142-226): the
    reference's rstest cases and residue list on the GPU, then whole arrays against the oracle's Tonelli-Shanks over F_101,
    Goldilocks (2-adicity 32: the longest loop) and a Montgomery prime above 2^63; the host mirror's scalar form beside it"""
    v = refvec["field_sqrt"]
    F = R.PlutoBaseField
    r0, r1 = F.vec_sqrt([c[1] for c in v["cases"]])
    assert r0.tolist() == [c[2] for c in v["cases"]] and r1.tolist() == [c[3] for c in v["cases"]]
    for p, a, x0, x1 in v["cases"]:
        assert tuple(int(t) for t in R.PrimeField(p)(a).sqrt()) == (x0, x1)
    with pytest.raises(R.RonkPanic) as e:
        F.vec_sqrt([4, 2, 5])
    assert e.value.code == -13
    with pytest.raises(R.RonkPanic):
        F(2).sqrt()
    assert F.vec_euler(np.arange(101, dtype=np.uint64)).tolist() == [1 if a in v["residues_101"] else 0 for a in range(101)]
    for p in (101, GP, 0xFFFFFFFC00000001, 29 * 2**57 + 1):
        Fp = R.PrimeField(p)
        x = splitmix_field(0x5EED0500 + p % 97, 4096, p)
        x[0] = 0; x[1] = 1; x[2] = p - 1
        y = Fp.vec_mul(x, x)
        r0, r1 = Fp.vec_sqrt(y)
        o0, o1 = orc.vec_sqrt(p, y)
        assert np.array_equal(r0, o0) and np.array_equal(r1, o1), p
        assert np.array_equal(Fp.vec_euler(x), orc.vec_euler(p, x)), p
        assert tuple(int(t) for t in Fp(int(y[7])).sqrt()) == (int(o0[7]), int(o1[7]))


def test_polynomial_kats_on_gpu(R, refvec):
    F = R.PlutoBaseField
    v = refvec
    P = R.Polynomial
    poly = P.new(F, [1, 2, 3, 4])
    for p, c, x, y in v["poly_eval"]["cases"]:
        assert P.new(R.PrimeField(p), c).evaluate(x) == R.PrimeField(p)(y)
    assert poly.dft().coefficients.tolist() == v["poly_dft"]["out"]
    assert poly.fft().coefficients.tolist() == v["poly_fft"]["out"]
    assert poly.fft().ifft() == poly
    assert poly.dft().basis.nodes.tolist() == [1, 10, 100, 91]
    assert poly.dft().evaluate(v["lagrange_eval"]["x"]) == F(v["lagrange_eval"]["y"])
    assert poly.degree() == 3 and poly.leading_coefficient() == F(4)
    assert poly.pow_mult(2, 5).coefficients.tolist() == v["pow_mult"]["out"]
    a, b = P.new(F, [1, 2, 3, 4]), P.new(F, [5, 6, 7, 8, 9])
    assert (b + a).coefficients.tolist() == v["poly_add"]["out"]
    for x, y, r in v["poly_sub"]["cases"]:
        assert (P.new(F, x) - P.new(F, y)).coefficients.tolist() == r
    assert (-a).coefficients.tolist() == v["poly_neg"]["out"]
    for x, y, q in v["poly_div"]["cases"]:
        assert (P.new(F, x) / P.new(F, y)).coefficients.tolist() == q
    for x, y, r in v["poly_rem"]["cases"]:
        assert (P.new(F, x) % P.new(F, y)).coefficients.tolist() == r
    for x, y, c in v["poly_mul"]["cases"]:
        assert (P.new(F, x) * P.new(F, y)).coefficients.tolist() == c
    # no roots of unity => panic (polynomial/tests.rs:46-55)
    with pytest.raises(R.RonkPanic) as e:
        P.new(F, [1, 2, 3]).dft()
    assert e.value.code == -1
    with pytest.raises(R.RonkPanic) as e:
        P.new(R.PrimeField(127), [1, 2, 3]).fft()
    assert e.value.code == -3
    with pytest.raises(R.RonkPanic) as e:
        P.new(F, list(range(8))).fft()
    assert e.value.code == -1


def test_plonk_lagrange_polys_through_ifft_on_gpu(R, refvec, orc):
    """row N3: the Lagrange-basis selector / permutation polynomials of src/compiler/program.rs:350-420 (F_17, omega_4 = 13)
    through `ronk_ifft` / `ronk_fft` on the generic-prime path, against the oracle and against the values the reference holds"""
    from ronkathon_amd import _lib as L
    d = refvec["plonk_lagrange_polys"]
    p, n = d["p"], d["n"]
    g = int(orc.find_primitive_element(p))
    nodes = orc.lagrange_nodes(p, g, n)
    plan = L.Plan(p, g, 2)
    for name, vals in d["cases"].items():
        v = np.array(vals, dtype=np.uint64)
        c = np.zeros(n, dtype=np.uint64)
        L.check(L.lib.ronk_ifft(p, g, L.ptr(v), L.ptr(c), n))
        assert np.array_equal(c, orc.ifft(p, g, vals)), name
        assert np.array_equal(plan.inverse(v), c), name                     # the plan form of the same call
        back, nd = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        L.check(L.lib.ronk_fft(p, g, L.ptr(c), L.ptr(back), L.ptr(nd), n))
        assert back.tolist() == vals and nd.tolist() == [int(w) for w in nodes] and int(nd[1]) == 13, name
        # coefficient form evaluated at the nodes on the GPU = the reference's values
        import ctypes as C
        for i, w in enumerate(nodes):
            y = C.c_uint64(0)
            L.check(L.lib.ronk_poly_eval(p, L.ptr(c), n, int(w), C.byref(y)))
            assert int(y.value) == vals[i], name
    plan.close()


def test_callers_on_gpu(R, refvec, orc):
    from ronkathon_amd.callers import Message, kzg_open_quotient
    d = refvec["rs_encode"]
    xs, ys = Message(R.PrimeField(d["p"]), d["msg"]).encode(d["n"])
    assert xs.tolist() == d["x"] and ys.tolist() == d["y"]
    xs, ys = Message(R.PrimeField(127), [1, 2, 3]).encode(7)
    ox, oy = orc.rs_encode(127, 3, [1, 2, 3], 7)
    assert xs.tolist() == ox.tolist() and ys.tolist() == oy.tolist()
    d = refvec["kzg_open_quotient"]
    assert kzg_open_quotient(R.PrimeField(d["p"]), d["coeffs"], d["z"]).tolist() == d["quot"]
    # reference quirks (oracle header): untrimmed-length loop guard, zero divisor panic, Lagrange eval at a node
    F = R.PlutoBaseField
    q, r = R.Polynomial.new(F, [1, 2, 3]).quotient_and_remainder(R.Polynomial.new(F, [1, 1, 0, 0]))
    assert q.coefficients.tolist() == [0, 0, 0] and r.coefficients.tolist() == [1, 2, 3]
    with pytest.raises(R.RonkPanic) as e:
        R.Polynomial.new(F, [1, 2, 3]) / R.Polynomial.new(F, [0, 0])
    assert e.value.code == -6
    lag = R.Polynomial.new(F, [1, 2, 3, 4]).dft()
    assert lag.evaluate(10) == F(0)


def test_divrem_eval_random_vs_oracle(R, orc):
    for p, g in ((101, 2), (GP, GG)):
        F = R.PrimeField(p)
        for d, d2, seed in ((40, 7, 1), (300, 300, 2), (513, 2, 3), (64, 65, 4)):
            a = splitmix_field(seed, d, p); b = splitmix_field(seed + 100, d2, p)
            if b[-1] == 0:
                b[-1] = 1
            q, r = R.Polynomial.new(F, a).quotient_and_remainder(R.Polynomial.new(F, b))
            oq, orr = orc.poly_divrem(p, a, b)
            assert np.array_equal(q.coefficients, oq) and np.array_equal(r.coefficients, orr)
            x = int(splitmix_field(seed + 7, 1, p)[0])
            assert int(R.Polynomial.new(F, a).evaluate(x)) == orc.poly_eval(p, a, x)
    a = splitmix_field(11, 100000)
    assert int(R.Polynomial.new(R.GoldilocksField, a).evaluate(12345)) == orc.poly_eval(GP, a, 12345)


# ---------------------------------------------------------------- Goldilocks: derived vectors and the oracle
def test_goldilocks_derived(R, glvec):
    F = R.GoldilocksField
    v = glvec
    assert F.PRIMITIVE_ELEMENT == F(7)
    for k, w in v["roots"].items():
        assert int(F.primitive_root_of_unity(1 << int(k))) == w
    e = v["field_edge"]; vals = e["values"]
    A = np.repeat(np.array(vals, dtype=np.uint64), len(vals)); B = np.tile(np.array(vals, dtype=np.uint64), len(vals))
    assert F.vec_add(A, B).tolist() == [x for row in e["add"] for x in row]
    assert F.vec_sub(A, B).tolist() == [x for row in e["sub"] for x in row]
    assert F.vec_mul(A, B).tolist() == [x for row in e["mul"] for x in row]
    nzv = [x for x in vals if x]
    assert F.vec_inv(nzv).tolist() == [x for x in e["inv"] if x is not None]
    assert R.Polynomial.new(F, [1, 2, 3, 4]).fft().coefficients.tolist() == v["dft_1234"]
    for case in v["dft_random"] + v["dft_non_pow2"]:
        assert R.Polynomial.new(F, case["in"]).dft().coefficients.tolist() == case["out"]
    for case in v["dft_random"]:
        assert R.Polynomial.new(F, case["in"]).fft().coefficients.tolist() == case["out"]
        assert R.Polynomial.new_lagrange(F, case["out"]).ifft().coefficients.tolist() == case["in"]
    for case in v["mul_random"]:
        assert (R.Polynomial.new(F, case["a"]) * R.Polynomial.new(F, case["b"])).coefficients.tolist() == case["out"]


@pytest.mark.parametrize("k", list(range(0, 21)))
def test_ntt_vs_oracle_all_sizes(R, orc, k):
    from ronkathon_amd import _lib as L
    n = 1 << k
    plan = L.Plan(GP, GG, k)
    cases = [splitmix_field(0x5EED0000 + k, n)] + (adversarial(n) if k in (4, 8, 11, 13, 16) else [])
    for x in cases:
        y = plan.forward(x)
        assert np.array_equal(y, orc.fft(GP, GG, x)), "forward 2^%d" % k
        assert np.array_equal(plan.inverse(y), x), "roundtrip 2^%d" % k
        assert np.array_equal(plan.inverse(x), orc.ifft(GP, GG, x)), "inverse 2^%d" % k
    plan.close()


@pytest.mark.parametrize("k,batch", [(4, 1000), (6, 33), (8, 64), (10, 7), (12, 17), (13, 5), (16, 3)])
def test_batched_ragged_vs_oracle(R, orc, k, batch):
    from ronkathon_amd import _lib as L
    n = 1 << k
    plan = L.Plan(GP, GG, k, batch)
    x = splitmix_field(77 + k, n * batch)
    y = plan.forward(x)
    for b in range(batch):
        assert np.array_equal(y[b * n:(b + 1) * n], orc.fft(GP, GG, x[b * n:(b + 1) * n])), (k, b)
    assert np.array_equal(plan.inverse(y), x)
    plan.close()


# ---------------------------------------------------------------- plan variants: what bench.py times and what the planner picks
@pytest.mark.parametrize("k", [20, 21, 22, 23])
def test_tuned_tile_widths_vs_oracle(R, orc, k):
    """ronk_plan_create_tuned(tile_log2_columns = c), c = 0..3 (bench.py's throughput line runs c = 2 plans on two
    streams), plus the default plan: forward AND inverse against the oracle's fft / ifft (reference
    src/polynomial/mod.rs:273-323, :430-484); adversarial vectors at 2^22."""
    from ronkathon_amd import _lib as L
    n = 1 << k
    x = splitmix_field(0x5EED0200 + k, n)
    cases = [x] + (adversarial(n) if k == 22 else [])
    refs = [(c, orc.fft(GP, GG, c), orc.ifft(GP, GG, c)) for c in cases]
    for c in (-1, 0, 1, 2, 3):
        plan = L.Plan(GP, GG, k, tile_log2_columns=c)
        for xin, yf, yi in refs:
            assert np.array_equal(plan.forward(xin), yf), ("forward", k, c)
            assert np.array_equal(plan.inverse(xin), yi), ("inverse", k, c)
        plan.close()


@pytest.mark.parametrize("k,batch", [(19, 32), (20, 16), (22, 4), (23, 3)])
def test_planner_default_for_big_batches_vs_oracle(R, orc, k, batch):
    """the planner's many_tiles branch (batch*n >= 2^24 with 2^10 / 2^11-row passes -> 8192-coefficient tiles) and a
    batched three-pass plan: every polynomial against the oracle, forward and inverse"""
    from ronkathon_amd import _lib as L
    n = 1 << k
    x = splitmix_field(0x5EED0300 + k, n * batch)
    plan = L.Plan(GP, GG, k, batch)
    y = plan.forward(x)
    z = plan.inverse(x)
    for b in range(batch):
        assert np.array_equal(y[b * n:(b + 1) * n], orc.fft(GP, GG, x[b * n:(b + 1) * n])), ("forward", k, b)
        if b in (0, batch - 1):
            assert np.array_equal(z[b * n:(b + 1) * n], orc.ifft(GP, GG, x[b * n:(b + 1) * n])), ("inverse", k, b)
    assert np.array_equal(plan.inverse(y), x)
    plan.close()


class _DevArr:
    """a device array of uint64 through the library's own helpers (no torch in this file)"""

    def __init__(self, host=None, n=0):
        import ctypes as C
        from ronkathon_amd import _lib as L
        self.L, self.n = L, int(host.size if host is not None else n)
        self.h = C.c_void_p()
        L.check(L.lib.ronk_dev_alloc(C.byref(self.h), self.n * 8))
        if host is not None:
            a = L.arr(host)
            L.check(L.lib.ronk_memcpy_h2d(self.h, L.ptr(a), self.n * 8))

    @property
    def ptr(self):
        return self.h.value

    def get(self):
        out = np.empty(self.n, dtype=np.uint64)
        self.L.check(self.L.lib.ronk_dev_sync())
        self.L.check(self.L.lib.ronk_memcpy_d2h(self.L.ptr(out), self.h, self.n * 8))
        return out

    def free(self):
        if self.h:
            self.L.lib.ronk_dev_free(self.h)
            self.h = None


@pytest.mark.parametrize("k,batch", [(19, 2), (20, 3), (21, 2), (22, 2), (22, 5)])
def test_in_flight_lanes_vs_oracle(R, orc, k, batch):
    """ronk_plan_opts::in_flight: the second half of the batch runs on the plan's side stream (event fork / join); 1, 2 and
    auto give the oracle's values for every polynomial, forward and inverse (reference src/polynomial/mod.rs:273-323,
    :430-484), odd batches included (halves of different size)"""
    from ronkathon_amd import _lib as L
    n = 1 << k
    x = splitmix_field(0x5EED0700 + 31 * k + batch, n * batch)
    want_f = [orc.fft(GP, GG, x[b * n:(b + 1) * n]) for b in range(batch)]
    want_i = orc.ifft(GP, GG, x[(batch - 1) * n:])
    for lanes in (1, 2, -1):
        plan = L.Plan(GP, GG, k, batch, in_flight=lanes)
        assert plan.in_flight() == (2 if lanes == 2 else 1)      # automatic = 1
        y = plan.forward(x)
        for b in range(batch):
            assert np.array_equal(y[b * n:(b + 1) * n], want_f[b]), ("forward", k, batch, lanes, b)
        z = plan.inverse(x)
        assert np.array_equal(z[(batch - 1) * n:], want_i), ("inverse", k, batch, lanes)
        assert np.array_equal(plan.inverse(y), x)
        plan.close()


def test_in_flight_outside_window_is_one(R):
    from ronkathon_amd import _lib as L
    for k, batch in ((16, 8), (12, 64), (23, 2), (22, 1), (22, 4)):
        plan = L.Plan(GP, GG, k, batch)
        assert plan.in_flight() == 1
        plan.close()


@pytest.mark.parametrize("k,count", [(19, 5), (22, 4), (16, 3), (11, 3)])
def test_forward_many_dev_vs_oracle(R, orc, k, count):
    """ronk_ntt_forward_many_dev / _inverse_many_dev: `count` unrelated device arrays in one call on a batch-1 plan with two
    lanes (own scratch per lane); sizes outside the two-lane window run the arrays one after the other"""
    from ronkathon_amd import _lib as L
    n = 1 << k
    plan = L.Plan(GP, GG, k, 1, in_flight=2)
    xs = [splitmix_field(0x5EED0800 + 7 * k + i, n) for i in range(count)]
    din = [_DevArr(x) for x in xs]
    dout = [_DevArr(n=n) for _ in xs]
    for rep in range(2):     # twice: the second call reuses both scratches behind the first
        plan.forward_many_dev([d.ptr for d in din], [d.ptr for d in dout])
    ys = [d.get() for d in dout]
    for i in range(count):
        assert np.array_equal(ys[i], orc.fft(GP, GG, xs[i])), (k, i)
    plan.forward_many_dev([d.ptr for d in dout], [d.ptr for d in dout], inverse=True)   # in place
    for i in range(count):
        assert np.array_equal(dout[i].get(), xs[i]), ("roundtrip", k, i)
    for d in din + dout:
        d.free()
    plan.close()


def test_batch16_of_2_22_vs_oracle(R, orc):
    """16 polynomials of 2^22 coefficients through ONE plan handle (one launch pair, and two lanes of 8): every polynomial
    against the oracle"""
    from ronkathon_amd import _lib as L
    k, batch = 22, 16
    n = 1 << k
    x = splitmix_field(0x5EED0900, n * batch)
    want = [orc.fft(GP, GG, x[b * n:(b + 1) * n]) for b in range(batch)]
    for lanes in (-1, 2):
        plan = L.Plan(GP, GG, k, batch, in_flight=lanes)
        assert plan.in_flight() == (2 if lanes == 2 else 1)
        y = plan.forward(x)
        for b in range(batch):
            assert np.array_equal(y[b * n:(b + 1) * n], want[b]), (lanes, b)
        plan.close()


def test_config3_full_vector_2_22(R, orc):
    """BASELINE configs[2], element for element: the product of two 2^21-coefficient polynomials (NTT size 2^22) against
    ifft(fft(a) * fft(b)) computed by the oracle (three oracle transforms) -- every one of the 2^22 - 1 coefficients"""
    F = R.GoldilocksField
    d = 1 << 21
    a = splitmix_field(0x5EED0A11, d); b = splitmix_field(0x5EED0A12, d)
    prod = (R.Polynomial.new(F, a) * R.Polynomial.new(F, b)).coefficients
    z = np.zeros(d, dtype=np.uint64)
    fa, fb = orc.fft(GP, GG, np.concatenate([a, z])), orc.fft(GP, GG, np.concatenate([b, z]))
    want = orc.ifft(GP, GG, orc.vec_mul(GP, fa, fb))
    assert int(want[-1]) == 0
    assert np.array_equal(prod, want[:2 * d - 1])


@pytest.mark.parametrize("d,d2", [(1 << 19, 1 << 19), (300001, 7), (1, (1 << 20) - 3), ((1 << 21) + 5, (1 << 21) - 4), (1 << 20, 3 << 19),
                                  (1 << 20, 1 << 20), (700001, 900000)])
def test_fused_multiply_ragged_lengths(R, orc, d, d2):
    """the fused middle of the multiply (csrc/ntt_mul.h; NTT sizes 2^20, 2^21 -- the inverse plan split the other way round -- and 2^22): every product coefficient against the
    oracle's ifft(fft(a) * fft(b)) for operand lengths that are not powers of two, the extreme 1 x long case, and twice in
    a row through the cached plans (src/polynomial/arithmetic.rs:97-119: D + D2 - 1 coefficients)"""
    F = R.GoldilocksField
    a = splitmix_field(0x5EED0F00 + d % 97, d); b = splitmix_field(0x5EED0F80 + d2 % 89, d2)
    m = d + d2 - 1
    n = 1 << (m - 1).bit_length()
    want = orc.ifft(GP, GG, orc.vec_mul(GP, orc.fft(GP, GG, np.concatenate([a, np.zeros(n - d, dtype=np.uint64)])),
                                        orc.fft(GP, GG, np.concatenate([b, np.zeros(n - d2, dtype=np.uint64)]))))
    assert not want[m:].any()
    for _ in range(2):
        prod = (R.Polynomial.new(F, a) * R.Polynomial.new(F, b)).coefficients
        assert prod.size == m and np.array_equal(prod, want[:m])


def test_config4_all_rows_vs_oracle(R, orc):
    """BASELINE configs[3]: ALL 1024 rows of the batched 2^16 transform against the oracle, forward and inverse (the oracle
    rows run on a thread pool: its C code releases the GIL)"""
    import concurrent.futures as cf
    import os
    from ronkathon_amd import _lib as L
    n, batch = 1 << 16, 1024
    x = splitmix_field(0x5EED0044, n * batch)
    plan = L.Plan(GP, GG, 16, batch)
    y = plan.forward(x)
    yi = plan.inverse(x)
    plan.close()

    def row_ok(b):
        xb = x[b * n:(b + 1) * n]
        return (np.array_equal(y[b * n:(b + 1) * n], orc.fft(GP, GG, xb)) and
                np.array_equal(yi[b * n:(b + 1) * n], orc.ifft(GP, GG, xb)))
    with cf.ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 4)) as ex:
        bad = [b for b, ok in enumerate(ex.map(row_ok, range(batch))) if not ok]
    assert not bad, bad[:8]


def test_plan_seen_on_several_streams(R, orc):
    """the scratch guard (transform_dev): a plan used on stream A, then on B (A destroyed in between), then on the null
    stream -- every result correct, no call fails because of the dead handle"""
    import ctypes as C
    from ronkathon_amd import _lib as L
    hip = C.CDLL("libamdhip64.so")
    k = 20
    n = 1 << k
    plan = L.Plan(GP, GG, k, 1, in_flight=1)
    x = splitmix_field(0x5EED0B00, n)
    want = orc.fft(GP, GG, x)
    din, dout = _DevArr(x), _DevArr(n=n)
    sa, sb = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamCreate(C.byref(sa)) == 0 and hip.hipStreamCreate(C.byref(sb)) == 0
    plan.forward_dev(din.ptr, dout.ptr, sa.value)
    assert hip.hipStreamSynchronize(sa) == 0
    assert np.array_equal(dout.get(), want)
    assert hip.hipStreamDestroy(sa) == 0
    for s_ in (sb.value, 0, sb.value, sb.value, 0):
        plan.forward_dev(din.ptr, dout.ptr, s_)
    assert np.array_equal(dout.get(), want)
    assert hip.hipStreamDestroy(sb) == 0
    plan.forward_dev(din.ptr, dout.ptr, 0)
    assert np.array_equal(dout.get(), want)
    din.free(); dout.free(); plan.close()


def test_field_mul_carry_paths_on_gpu(R, orc):
    """gl64::mul / mul_2exp use v_mad_u64_u32's carry-out and hand-written borrow chains (csrc/gl64.h): products of
    edge values that hit every wrap/borrow branch, through the element-wise C ABI, against the oracle."""
    F = R.GoldilocksField
    e = np.array([0, 1, 2, 0xFFFFFFFF, 0x100000000, 0x100000001, 0xFFFFFFFE00000001, 0xFFFFFFFEFFFFFFFF,
                  0xFFFFFFFF00000000, GP - 1, GP - 2, GP - 0xFFFFFFFF, GP - 0x100000000, GP // 2, GP // 2 + 1,
                  0x0000FFFF00000001, 0xFFFF0000FFFF0000, 0x00000001FFFFFFFF, 0xFFFFFFFDFFFFFFFF, 0x8000000000000000,
                  0x7FFFFFFFFFFFFFFF, 0xFFFFFFFE00000000], dtype=np.uint64)
    rnd = splitmix_field(0xC0FFEE, 4096 - e.size)
    v = np.concatenate([e, rnd])
    a = np.repeat(v[:64], 64); b = np.tile(v[:64], 64)
    a = np.concatenate([a, v]); b = np.concatenate([b, v[::-1]])
    want_mul = np.array([orc.mul(GP, int(p), int(q)) for p, q in zip(a, b)], dtype=np.uint64)
    want_add = np.array([orc.add(GP, int(p), int(q)) for p, q in zip(a, b)], dtype=np.uint64)
    want_sub = np.array([orc.sub(GP, int(p), int(q)) for p, q in zip(a, b)], dtype=np.uint64)
    assert np.array_equal(F.vec_mul(a, b), want_mul)
    assert np.array_equal(F.vec_add(a, b), want_add)
    assert np.array_equal(F.vec_sub(a, b), want_sub)
    # a 16-point transform of edge values exercises every shift twiddle (mul_2exp<K>) on the same inputs
    from ronkathon_amd import _lib as L
    plan = L.Plan(GP, GG, 4, batch=v.size // 16)
    assert np.array_equal(plan.forward(v), np.concatenate([orc.fft(GP, GG, v[i:i + 16]) for i in range(0, v.size, 16)]))
    plan.close()


def test_config2_roundtrip_2_16(R, orc):
    """BASELINE configs[1]: forward+inverse NTT, degree 2^16, bit-exact round trip"""
    from ronkathon_amd import _lib as L
    x = splitmix_field(0x5EED0002, 1 << 16)
    plan = L.Plan(GP, GG, 16)
    y, nodes = plan.forward(x, nodes=True)
    assert np.array_equal(y, orc.fft(GP, GG, x))
    assert np.array_equal(nodes, orc.lagrange_nodes(GP, GG, 1 << 16))
    assert np.array_equal(plan.inverse(y), x)
    plan.close()


def test_config3_full_size_2_22(R, orc):
    """BASELINE configs[2] at full size: NTT 2^22 vs the oracle, and polynomial multiply with NTT size
    2^22 checked through size-independent properties (evaluation homomorphism, linearity)."""
    from ronkathon_amd import _lib as L
    n = 1 << 22
    x = splitmix_field(0x5EED0003, n)
    plan = L.Plan(GP, GG, 22)
    y = plan.forward(x)
    assert np.array_equal(y, orc.fft(GP, GG, x))
    assert np.array_equal(plan.inverse(y), x)
    # linearity: NTT(a + b) == NTT(a) + NTT(b)
    b = splitmix_field(0x5EED0033, n)
    F = R.GoldilocksField
    assert np.array_equal(plan.forward(F.vec_add(x, b)), F.vec_add(y, plan.forward(b)))
    plan.close()
    a, c = x[: n // 2], b[: n // 2]
    prod = (R.Polynomial.new(F, a) * R.Polynomial.new(F, c)).coefficients
    assert prod.size == n - 1
    for pt in (2, 0xDEADBEEFCAFE, GP - 1):
        lhs = orc.poly_eval(GP, prod, pt)
        assert lhs == orc.mul(GP, orc.poly_eval(GP, a, pt), orc.poly_eval(GP, c, pt))
    # schoolbook cross-check of the low and high ends (exact): c_0, c_1, c_{m-1}
    assert int(prod[0]) == orc.mul(GP, int(a[0]), int(c[0]))
    assert int(prod[-1]) == orc.mul(GP, int(a[-1]), int(c[-1]))
    assert int(prod[1]) == orc.add(GP, orc.mul(GP, int(a[0]), int(c[1])), orc.mul(GP, int(a[1]), int(c[0])))


def test_config3_variant_ntt_size_2_23(R, orc):
    """SURVEY.md 8(d) C3 variant: 2^22-coefficient operands -> NTT size 2^23 (the implicit padding, the fused pointwise
    product and the truncated store on the 2^23 plans); ragged operand lengths too.  EVERY coefficient against the oracle's
    product ifft(fft(a) * fft(b)) of the zero-padded operands -- three oracle transforms of 2^23 -- plus the exact end
    coefficients by the schoolbook definition (src/polynomial/arithmetic.rs:97-119)."""
    F = R.GoldilocksField
    N = 1 << 23
    for da, db in ((1 << 22, 1 << 22), ((1 << 22) + 12345, (1 << 21) - 7)):
        a = splitmix_field(0x5EED0A00 + da % 97, da); b = splitmix_field(0x5EED0B00 + db % 89, db)
        prod = (R.Polynomial.new(F, a) * R.Polynomial.new(F, b)).coefficients
        m = da + db - 1
        assert prod.size == m
        pa = np.zeros(N, dtype=np.uint64); pa[:da] = a
        pb = np.zeros(N, dtype=np.uint64); pb[:db] = b
        want = orc.ifft(GP, GG, orc.vec_mul(GP, orc.fft(GP, GG, pa), orc.fft(GP, GG, pb)))
        assert not want[m:].any()
        assert np.array_equal(prod, want[:m]), (da, db)
        assert int(prod[0]) == orc.mul(GP, int(a[0]), int(b[0]))
        assert int(prod[-1]) == orc.mul(GP, int(a[-1]), int(b[-1]))


def test_poly_mul_vs_schoolbook(R, orc):
    F = R.GoldilocksField
    for d, d2 in ((1, 1), (17, 17), (64, 1), (100, 157), (1000, 3000), (5000, 5000)):
        a = splitmix_field(d, d); b = splitmix_field(d2 + 1, d2)
        got = (R.Polynomial.new(F, a) * R.Polynomial.new(F, b)).coefficients
        assert np.array_equal(got, orc.poly_mul(GP, a, b)), (d, d2)
    # config 1: F_101, 17 x 17 coefficients (degree 16 x 16) -- schoolbook kernel
    a = splitmix_field(5, 17, 101); b = splitmix_field(6, 17, 101)
    got = (R.Polynomial.new(R.PlutoBaseField, a) * R.Polynomial.new(R.PlutoBaseField, b)).coefficients
    assert got.size == 33 and np.array_equal(got, orc.poly_mul(101, a, b))


def test_config4_batched_1024_x_2_16(R, orc):
    """BASELINE configs[3]: 1024 polynomials x 2^16; spot-check rows against the oracle, round-trip all"""
    from ronkathon_amd import _lib as L
    n, batch = 1 << 16, 1024
    x = splitmix_field(0x5EED0004, n * batch)
    plan = L.Plan(GP, GG, 16, batch)
    y = plan.forward(x)
    for b in (0, 1, 511, 1023):
        assert np.array_equal(y[b * n:(b + 1) * n], orc.fft(GP, GG, x[b * n:(b + 1) * n]))
    assert np.array_equal(plan.inverse(y), x)
    plan.close()


def test_generic_prime_pow2_ntt(R, orc):
    """a generic odd prime with 2-adicity: p = 3*2^30+1 (Montgomery radix-2 path), and Goldilocks with g != 7"""
    from ronkathon_amd import _lib as L
    p = 3 * 2**30 + 1
    g = orc.find_primitive_element(p)
    for k in (0, 1, 3, 10, 14):
        x = splitmix_field(k, 1 << k, p)
        plan = L.Plan(p, g, k)
        y = plan.forward(x)
        assert np.array_equal(y, orc.fft(p, g, x))
        assert np.array_equal(plan.inverse(y), x)
        plan.close()
    x = splitmix_field(3, 1 << 12)
    plan = L.Plan(GP, 3, 12)   # the reference's heuristic "generator" for Goldilocks
    assert np.array_equal(plan.forward(x), orc.fft(GP, 3, x))
    plan.close()
    with pytest.raises(R.RonkPanic) as e:
        L.Plan(101, 2, 3)      # 8 does not divide 100
    assert e.value.code == -1
    with pytest.raises(R.RonkPanic) as e:
        L.Plan(100, 2, 1)
    assert e.value.code == -4


def test_dist_fourstep_single_gpu(R, orc):
    """multi-GPU four-step phases executed rank by rank on ONE GPU (the exchange is a host copy)"""
    from ronkathon_amd.dist import fourstep_single_process
    for log2n, world, inv in ((12, 1, False), (16, 2, False), (18, 4, True), (20, 8, False), (26, 8, False)):
        x = splitmix_field(0x5EED0005 + log2n, 1 << log2n)
        got = fourstep_single_process(x, world, inverse=inv)
        ref = orc.ifft(GP, GG, x) if inv else orc.fft(GP, GG, x)
        assert np.array_equal(got, ref), (log2n, world, inv)


def test_sharded_transform_inside_the_library(R, orc):
    """ronk_sharded_*: the whole four-step (phase 1 in column chunks, peer-copy exchange on copy streams, phase 2) behind
    the C ABI.  Logical ranks share the one GPU here (devices = [0]*W: a peer copy to the same device is a plain copy), so
    the stream/event choreography, the chunked block layout and the scatter/gather run exactly as on a node; with >= 2
    devices the same test also runs on distinct ones."""
    from ronkathon_amd import _lib as L
    ndev = R.device_count()
    cases = [(12, 1, 1, False), (16, 2, 0, False), (16, 4, 2, True), (18, 8, 1, False), (20, 8, 4, False), (20, 4, 0, True),
             (26, 8, 2, False)]
    for log2n, W, chunks, inv in cases:
        x = splitmix_field(0x5EED0500 + log2n + W, 1 << log2n)
        ref = orc.ifft(GP, GG, x) if inv else orc.fft(GP, GG, x)
        layouts = [[0] * W] + ([[g % ndev for g in range(W)]] if ndev >= 2 else [])
        for devs in layouts:
            sp = L.ShardedPlan(log2n, devs, inverse=inv, chunks=chunks)
            assert sp.R * sp.C == 1 << log2n and sp.per_rank * W == 1 << log2n
            assert np.array_equal(sp.transform(x), ref), (log2n, W, chunks, inv, devs)
            if log2n <= 20:
                assert np.array_equal(sp.transform(x), ref)     # second call reuses the send/receive buffers (event guards)
            sp.close()
    with pytest.raises(R.RonkPanic) as e:
        L.ShardedPlan(10, [0] * 8)          # 32 columns over 8 ranks: fewer than 16 per rank
    assert e.value.code == -9
    with pytest.raises(R.RonkPanic) as e:
        L.ShardedPlan(16, [0, 99])
    assert e.value.code == -7


def test_sharded_rccl_exchange(R, orc):
    """RONK_EXCHANGE_RCCL: the exchange as ncclGroup{ncclSend, ncclRecv} through a dlopen()ed librccl -- on every visible GPU
    (one rank per device; with a single GPU: one rank that sends to itself), chunked and unchunked, twice per plan; logical
    ranks sharing a device are refused.  Same values as the peer-copy mesh and the oracle."""
    from ronkathon_amd import _lib as L
    ndev = R.device_count()
    W = 1
    while W * 2 <= ndev and W < 8:
        W *= 2
    for log2n, chunks in ((16, 1), (20, 2), (22, 0)):
        x = splitmix_field(0x5EED0650 + log2n, 1 << log2n)
        ref = orc.fft(GP, GG, x)
        sp = L.ShardedPlan(log2n, list(range(W)), chunks=chunks, exchange=L.EXCHANGE_RCCL)
        assert L.lib.ronk_sharded_plan_exchange(sp.h) == L.EXCHANGE_RCCL
        assert np.array_equal(sp.transform(x), ref), (log2n, W, chunks)
        assert np.array_equal(sp.transform(x), ref)
        sp.close()
        sm = L.ShardedPlan(log2n, list(range(W)), chunks=chunks)
        assert L.lib.ronk_sharded_plan_exchange(sm.h) == L.EXCHANGE_MESH
        assert np.array_equal(sm.transform(x), ref)
        sm.close()
    with pytest.raises(R.RonkPanic) as e:
        L.ShardedPlan(16, [0, 0], exchange=L.EXCHANGE_RCCL)
    assert e.value.code == -9


def test_sharded_device_api_pipelines_calls(R, orc):
    """ronk_ntt_sharded_dev: device-resident column blocks in, [C][R/W] blocks out; two transforms enqueued back to back
    on the plan's own streams, then one sync"""
    import ctypes as C
    from ronkathon_amd import _lib as L
    from ronkathon_amd.dist import scatter_input, place_output
    log2n, W = 18, 4
    sp = L.ShardedPlan(log2n, [0] * W, chunks=2)
    xs = [splitmix_field(0x5EED0600 + i, 1 << log2n) for i in range(2)]
    per = sp.per_rank
    bufs = []
    for x in xs:
        din, dout = [], []
        for g in range(W):
            a, b = C.c_void_p(), C.c_void_p()
            L.check(L.lib.ronk_dev_alloc(C.byref(a), per * 8)); L.check(L.lib.ronk_dev_alloc(C.byref(b), per * 8))
            loc = scatter_input(x, g, W)
            L.check(L.lib.ronk_memcpy_h2d(a, L.ptr(loc), per * 8))
            din.append(a.value); dout.append(b.value)
        bufs.append((din, dout))
    for din, dout in bufs:
        sp.transform_dev(din, dout)
    sp.sync()
    for x, (din, dout) in zip(xs, bufs):
        got = np.zeros(1 << log2n, dtype=np.uint64)
        for g in range(W):
            o = np.empty(per, dtype=np.uint64)
            L.check(L.lib.ronk_memcpy_d2h(L.ptr(o), dout[g], per * 8))
            place_output(got, o, g, W)
        assert np.array_equal(got, orc.fft(GP, GG, x))
        for q in din + dout:
            L.lib.ronk_dev_free(q)
    sp.close()


def test_fast_general_division_newton_vs_oracle(R, orc):
    """ronk_poly_divrem for non-linear divisors over Goldilocks: Newton inversion on the NTT path (O(n log n)) against the
    oracle's statement-by-statement restatement of quotient_and_remainder (reference src/polynomial/mod.rs:170-225),
    including its quirks: with trailing zero coefficients in the divisor the reference's loop (which compares with and
    indexes over the divisor's UNTRIMMED length) stops early or panics -- same code from the library."""
    from ronkathon_amd import _lib as L
    import ctypes as C

    def divrem(a, b):
        a, b = L.arr(a), L.arr(b)
        q, r = np.empty_like(a), np.empty_like(a)
        L.check(L.lib.ronk_poly_divrem(GP, L.ptr(a), a.size, L.ptr(b), b.size, L.ptr(q), L.ptr(r)))
        return q, r

    def z(v, k):
        return np.concatenate([v, np.zeros(k, dtype=np.uint64)])

    cases = [
        (splitmix_field(1, 40000), splitmix_field(2, 9000)),                       # plain
        (splitmix_field(3, 30000), z(splitmix_field(4, 3000), 500)),               # divisor with trailing zeros: early stop
        (z(splitmix_field(5, 20000), 1234), splitmix_field(6, 700)),               # dividend with leading zeros
        (z(splitmix_field(7, 6000), 4000), z(splitmix_field(8, 101), 7899)),       # short dividend, D >= D2: ONE step
        (splitmix_field(9, 8192), splitmix_field(10, 4096)),
        (splitmix_field(11, 5000), np.concatenate([np.zeros(99, dtype=np.uint64), np.ones(1, dtype=np.uint64), np.zeros(100, dtype=np.uint64)])),  # x^99
    ]
    for a, b in cases:
        try:
            oq, o_r = orc.poly_divrem(GP, a, b)
        except orc.OraclePanic as e:
            with pytest.raises(R.RonkPanic) as e2:
                divrem(a, b)
            assert e2.value.code == e.code, (a.size, b.size)
            continue
        q, r = divrem(a, b)
        assert np.array_equal(q, oq) and np.array_equal(r, o_r), (a.size, b.size)
    # 2^20 / 2^19: a == q b + r at random points, deg r < deg b, and the top quotient coefficients exactly
    a, b = splitmix_field(21, 1 << 20), splitmix_field(22, 1 << 19)
    q, r = divrem(a, b)
    assert not r[b.size - 1:].any()
    for pt in (5, 0xABCDEF0123456789 % GP):
        lhs = orc.poly_eval(GP, a, pt)
        rhs = orc.add(GP, orc.mul(GP, orc.poly_eval(GP, q, pt), orc.poly_eval(GP, b, pt)), orc.poly_eval(GP, r, pt))
        assert lhs == rhs
    top = orc.div(GP, int(a[-1]), int(b[-1]))
    assert int(q[a.size - b.size]) == top and not q[a.size - b.size + 1:].any()


def test_device_resident_general_division_takes_the_newton_form(R, orc):
    """ronk_poly_divrem_dev (what the shim's DevicePoly::div_rem calls): for a Goldilocks divisor of >= 64 coefficients and a
    quotient of >= 2048 it reads the degrees back and runs the Newton form -- the single-block long division would need
    d * d2 steps (minutes at 2^20 / 2^19).  Same cases, same oracle as the host-pointer form above, plus d_rem aliasing d_a and
    the status word."""
    import time

    import torch
    from ronkathon_amd import _lib as L

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()

    def host(t):
        return t.cpu().numpy().view(np.uint64)

    def z(v, k):
        return np.concatenate([v, np.zeros(k, dtype=np.uint64)])

    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()

    def divrem(a, b, alias=False, stream=None):
        da, db = dev(a), dev(b)
        dq = torch.full((a.size,), -1, dtype=torch.int64, device="cuda")
        dr = da if alias else torch.full((a.size,), -1, dtype=torch.int64, device="cuda")
        status.fill_(77)
        torch.cuda.synchronize()
        L.check(L.lib.ronk_poly_divrem_dev(GP, da.data_ptr(), a.size, db.data_ptr(), b.size, dq.data_ptr(), dr.data_ptr(),
                                           status.data_ptr(), stream.cuda_stream if stream is not None else None))
        torch.cuda.synchronize()
        return host(dq), host(dr), int(status.item())

    cases = [
        (splitmix_field(1, 40000), splitmix_field(2, 9000)),                       # Newton
        (splitmix_field(3, 30000), z(splitmix_field(4, 3000), 500)),               # ragged divisor: long division, early stop
        (z(splitmix_field(5, 20000), 1234), splitmix_field(6, 700)),               # dividend with leading zeros: Newton
        (z(splitmix_field(7, 6000), 4000), z(splitmix_field(8, 101), 7899)),       # short dividend: long division, ONE step
        (splitmix_field(9, 8192), splitmix_field(10, 4096)),                       # Newton, quotient of 4097
        (splitmix_field(12, 2200), splitmix_field(13, 64)),                        # just inside the window
        (splitmix_field(14, 2100), splitmix_field(15, 63)),                        # just outside (divisor too short)
        (np.zeros(5000, dtype=np.uint64), splitmix_field(16, 100)),                # zero dividend
    ]
    for i, (a, b) in enumerate(cases):
        try:
            oq, o_r = orc.poly_divrem(GP, a, b)
            code = 0
        except orc.OraclePanic as e:
            oq = o_r = None
            code = e.code
        for alias, stream in ((False, None), (True, st)):
            q, r, got = divrem(a, b, alias, stream)
            assert got == code, (i, got, code)
            if code == 0:
                assert np.array_equal(q, oq) and np.array_equal(r, o_r), (i, alias)
    # 2^20 / 2^19 on the device: a == q b + r at random points, in well under a second
    a, b = splitmix_field(21, 1 << 20), splitmix_field(22, 1 << 19)
    t0 = time.perf_counter()
    q, r, got = divrem(a, b)
    assert got == 0 and time.perf_counter() - t0 < 20.0
    assert not r[b.size - 1:].any()
    for pt in (5, 0xABCDEF0123456789 % GP):
        lhs = orc.poly_eval(GP, a, pt)
        rhs = orc.add(GP, orc.mul(GP, orc.poly_eval(GP, q, pt), orc.poly_eval(GP, b, pt)), orc.poly_eval(GP, r, pt))
        assert lhs == rhs


def test_scan_paths_fused_and_three_kernel(R, orc):
    """evaluate / division by a linear factor: the fused kernels (one / two launches, up to 2^23 coefficients for the
    division) and the three-kernel form with the serial carry scan beyond that -- remainder == evaluation == oracle,
    quotient through p(t) == q(t)(t - z) + r at a random point, sizes either side of the switch and ragged chunk tails"""
    import ctypes as C
    from ronkathon_amd import _lib as L
    z = 0x0123456789ABCDEF % GP
    for d in (1, 7, 2047, 2048, 2049, 4096 * 3 + 5, (1 << 20) + 123, (1 << 23) - 1, (1 << 23) + 17):
        x = splitmix_field(0x5CA0 + d % 1000, d)
        dc, dq, dr = C.c_void_p(), C.c_void_p(), C.c_void_p()
        for h, nbytes in ((dc, d * 8), (dq, d * 8), (dr, 8)):
            L.check(L.lib.ronk_dev_alloc(C.byref(h), nbytes))
        L.check(L.lib.ronk_memcpy_h2d(dc, L.ptr(x), d * 8))
        L.check(L.lib.ronk_poly_div_linear_dev(GP, dc, d, GP - z, 1, dq, dr, None))
        L.check(L.lib.ronk_dev_sync())
        q = np.empty(d, dtype=np.uint64); r = np.empty(1, dtype=np.uint64)
        L.check(L.lib.ronk_memcpy_d2h(L.ptr(q), dq, d * 8)); L.check(L.lib.ronk_memcpy_d2h(L.ptr(r), dr, 8))
        val = orc.poly_eval(GP, x, z)
        assert int(r[0]) == val, d
        assert int(q[d - 1]) == 0
        t = 0xFEEDFACE12345 % GP
        assert orc.poly_eval(GP, x, t) == orc.add(GP, orc.mul(GP, orc.poly_eval(GP, q, t), orc.sub(GP, t, z)), val), d
        if d <= 5000:
            assert np.array_equal(q, orc.kzg_open_quotient(GP, x, z)), d
        L.check(L.lib.ronk_poly_eval_dev(GP, dc, d, z, dr, None))
        L.check(L.lib.ronk_dev_sync())
        L.check(L.lib.ronk_memcpy_d2h(L.ptr(r), dr, 8))
        assert int(r[0]) == val, d
        for h in (dc, dq, dr):
            L.lib.ronk_dev_free(h)


@pytest.mark.parametrize("env", [{}, {"RONK_ONEPASS_DIV": "1"}, {"RONK_ONEPASS_DIV": "1", "RONK_LB_TEST_FLAGS": "1"},
                                 {"RONK_NO_ONEPASS_SCANS": "1"}, {"RONK_NO_FUSED_SCANS": "1"},
                                 {"RONK_LINDIV": "l"}, {"RONK_LINDIV": "0"}, {"RONK_LB_TEST_FLAGS": "1"}, {"RONK_LINDIV_ONE": "0"}, {"RONK_LINDIV_ONE_PL": "4"},
                                 {"RONK_LINDIV_ONE_PL": "8"}, {"RONK_LINDIV_ONE_PL": "8", "RONK_LB_TEST_FLAGS": "1"},
                                 {"RONK_LINDIV_ONE_MAXCH": "512"}])
def test_scan_onepass_variants(R, env):
    """the one-launch evaluate / linear division (look-back through an agent-coherent array), the same with every wait
    forced to give up (the recompute-from-coefficients path that makes the waits bounded), the two- and three-launch
    forms; the lane-scan division (lindiv_kernels.h): {} = the default, ONE launch of 1024-lane workgroups up to 2^23 coefficients
    (round 6, lindiv_one_kernel: 4 coefficients per lane below 1.5 M coefficients, 8 above -- RONK_LINDIV_ONE_PL forces one of them at
    every size; RONK_LINDIV_ONE_MAXCH=512: only while every chunk is resident; RONK_LB_TEST_FLAGS=1: its recompute path) and two
    launches beyond / in place / unaligned / for z = 0; RONK_LINDIV_ONE=0 two launches everywhere, with 16-byte reads of the lanes' runs
    or with the
    LDS image both ways ("l"; what unaligned dividends get anyway), "0" = the scan_kernels.h division: each in its own
    process, since the library reads its knobs once"""
    import subprocess, sys
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "scan_subprocess_check.py")],
                         capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0 and "scan check ok" in out.stdout, out.stdout[-500:] + out.stderr[-1500:]


def test_determinism(R):
    from ronkathon_amd import _lib as L
    x = splitmix_field(1234, 1 << 20)
    plan = L.Plan(GP, GG, 20)
    a = plan.forward(x); b = plan.forward(x)
    assert np.array_equal(a, b)
    plan.close()


def test_device_pointer_api_inplace_and_streams(R, orc):
    """_dev entry points on caller-owned device memory: out-of-place, in-place, on a side stream"""
    import torch
    from ronkathon_amd import _lib as L
    for k, batch in ((4, 1000), (5, 77), (6, 300), (9, 40), (10, 5), (16, 2), (20, 1)):   # 4, 5: staged I/O tiles
        n = 1 << k
        x = splitmix_field(900 + k, n * batch)
        ref = np.concatenate([orc.fft(GP, GG, x[b * n:(b + 1) * n]) for b in range(batch)])
        plan = L.Plan(GP, GG, k, batch)
        dx = torch.from_numpy(x.view(np.int64)).cuda()
        dy = torch.empty_like(dx)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            plan.forward_dev(dx.data_ptr(), dy.data_ptr(), side.cuda_stream)
        side.synchronize()
        assert np.array_equal(dy.cpu().numpy().view(np.uint64), ref)
        plan.forward_dev(dx.data_ptr(), dx.data_ptr(), 0)          # in place
        torch.cuda.synchronize()
        assert np.array_equal(dx.cpu().numpy().view(np.uint64), ref)
        plan.inverse_dev(dx.data_ptr(), dx.data_ptr(), 0)          # in place back
        torch.cuda.synchronize()
        assert np.array_equal(dx.cpu().numpy().view(np.uint64), x)
        plan.close()
    # device-resident multiply, repeated (plan cache + event ordering of the cached scratch)
    a = splitmix_field(31, 3000); b = splitmix_field(32, 5000)
    da = torch.from_numpy(a.view(np.int64)).cuda(); db = torch.from_numpy(b.view(np.int64)).cuda()
    dout = torch.empty(a.size + b.size - 1, dtype=torch.int64, device="cuda")
    ref = orc.poly_mul(GP, a, b)
    for _ in range(3):
        L.check(L.lib.ronk_poly_mul_dev(GP, GG, da.data_ptr(), a.size, db.data_ptr(), b.size, dout.data_ptr(), 0))
    torch.cuda.synchronize()
    assert np.array_equal(dout.cpu().numpy().view(np.uint64), ref)
    F = R.GoldilocksField
    dz = torch.empty_like(da)
    L.check(L.lib.ronk_vec_mul_dev(GP, da.data_ptr(), da.data_ptr(), dz.data_ptr(), a.size, 0))
    torch.cuda.synchronize()
    assert np.array_equal(dz.cpu().numpy().view(np.uint64), orc.vec_mul(GP, a, a))


def test_one_plan_many_streams_is_safe(R, orc):
    """one plan (one scratch buffer) driven from four streams at once without any host synchronisation in between:
    the library orders the calls on the plan by events (include/ronk_ntt.h, "re-entrant"), so every result is exact"""
    import torch
    from ronkathon_amd import _lib as L
    k = 20
    n = 1 << k
    plan = L.Plan(GP, GG, k)
    xs = [splitmix_field(0x57A0 + i, n) for i in range(4)]
    refs = [orc.fft(GP, GG, x) for x in xs]
    dxs = [torch.from_numpy(x.view(np.int64)).cuda() for x in xs]
    dys = [torch.empty_like(d) for d in dxs]
    streams = [torch.cuda.Stream() for _ in range(4)]
    torch.cuda.synchronize()
    for rep in range(6):
        for i in range(4):
            plan.forward_dev(dxs[i].data_ptr(), dys[i].data_ptr(), streams[i].cuda_stream)
    torch.cuda.synchronize()
    for i in range(4):
        assert np.array_equal(dys[i].cpu().numpy().view(np.uint64), refs[i]), i
    plan.close()


def test_hipgraph_capture_of_dev_entry_points(R, orc):
    """the _dev entry points only enqueue kernels on the caller's stream, so they can be captured in a hipGraph
    (here through torch.cuda.CUDAGraph) and replayed; results equal the oracle on every replay"""
    import torch
    from ronkathon_amd import _lib as L
    k = 14
    n = 1 << k
    x = splitmix_field(4242, n)
    dx = torch.from_numpy(x.view(np.int64)).cuda()
    dy = torch.zeros_like(dx); dz = torch.zeros_like(dx)
    plan = L.Plan(GP, GG, k, 1)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        plan.forward_dev(dx.data_ptr(), dy.data_ptr(), s.cuda_stream)   # warm-up outside the capture
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        st = torch.cuda.current_stream().cuda_stream
        plan.forward_dev(dx.data_ptr(), dy.data_ptr(), st)
        plan.inverse_dev(dy.data_ptr(), dz.data_ptr(), st)
    ref = orc.fft(GP, GG, x)
    for rep in range(3):
        dy.zero_(); dz.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(dy.cpu().numpy().view(np.uint64), ref)
        assert np.array_equal(dz.cpu().numpy().view(np.uint64), x)
        x = splitmix_field(4243 + rep, n)                    # new input, same graph
        dx.copy_(torch.from_numpy(x.view(np.int64)))
        ref = orc.fft(GP, GG, x)
    del g
    plan.close()


def test_hipgraph_capture_of_device_division(R, orc):
    """ronk_poly_divrem_dev inside a stream capture: a shape that would take the Newton form (degree probe = one stream
    synchronisation, impossible while capturing) keeps the long-division kernel, so the call stays capturable; replays equal the
    oracle, and the same call outside the capture (Newton) gives the same values"""
    import torch
    from ronkathon_amd import _lib as L
    d, d2 = 2300, 100
    a, b = splitmix_field(771, d), splitmix_field(772, d2)
    da = torch.from_numpy(a.view(np.int64)).cuda(); db = torch.from_numpy(b.view(np.int64)).cuda()
    dq = torch.zeros(d, dtype=torch.int64, device="cuda"); dr = torch.zeros(d, dtype=torch.int64, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    L.check(L.lib.ronk_poly_divrem_dev(GP, da.data_ptr(), d, db.data_ptr(), d2, dq.data_ptr(), dr.data_ptr(), status.data_ptr(), s.cuda_stream))
    s.synchronize()
    oq, o_r = orc.poly_divrem(GP, a, b)
    assert int(status.item()) == 0 and np.array_equal(dq.cpu().numpy().view(np.uint64), oq) and np.array_equal(dr.cpu().numpy().view(np.uint64), o_r)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        st = torch.cuda.current_stream().cuda_stream
        L.check(L.lib.ronk_poly_divrem_dev(GP, da.data_ptr(), d, db.data_ptr(), d2, dq.data_ptr(), dr.data_ptr(), status.data_ptr(), st))
    for rep in range(2):
        dq.fill_(-1); dr.fill_(-1); status.fill_(9)
        g.replay()
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        assert np.array_equal(dq.cpu().numpy().view(np.uint64), oq) and np.array_equal(dr.cpu().numpy().view(np.uint64), o_r)
        a = splitmix_field(773 + rep, d)
        da.copy_(torch.from_numpy(a.view(np.int64)))
        oq, o_r = orc.poly_divrem(GP, a, b)
    del g


def test_plan_cache_eviction_and_threads(R, orc):
    """more distinct one-shot sizes than cache entries, then concurrent callers (the reference's
    `cargo test` runs tests on parallel threads; the library must be re-entrant)"""
    import threading
    F = R.GoldilocksField
    for k in range(4, 16):
        x = splitmix_field(40 + k, 1 << k)
        p = R.Polynomial.new(F, x)
        assert np.array_equal(p.fft().coefficients, orc.fft(GP, GG, x))
    errs = []

    def worker(seed):
        try:
            for k in (6, 9, 12, 13):
                x = splitmix_field(seed * 100 + k, 1 << k)
                y = R.Polynomial.new(F, x).fft()
                assert np.array_equal(y.coefficients, orc.fft(GP, GG, x))
                assert np.array_equal(y.ifft().coefficients, x)
                a = splitmix_field(seed, 200); b = splitmix_field(seed + 1, 300)
                assert np.array_equal((R.Polynomial.new(F, a) * R.Polynomial.new(F, b)).coefficients, orc.poly_mul(GP, a, b))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


def test_linear_divisor_scan_vs_oracle(R, orc):
    """kzg::open shape: poly / (b0 + b1 x) through the parallel-scan kernels, against the oracle's long division"""
    for p in (101, 17, GP):
        F = R.PrimeField(p)
        for d, seed in ((2, 1), (3, 2), (17, 3), (4095, 4), (4096, 5), (4097, 6), (12289, 7)):
            if p != GP and d > 5000:
                continue
            a = splitmix_field(seed, d, p)
            for b in ([int(splitmix_field(seed + 50, 1, p)[0]) or 1, 1],                 # x - z (monic, kzg::open)
                      [int(splitmix_field(seed + 60, 1, p)[0]) or 2, int(splitmix_field(seed + 70, 1, p)[0]) or 3]):
                q, r = R.Polynomial.new(F, a).quotient_and_remainder(R.Polynomial.new(F, b))
                oq, orr = orc.poly_divrem(p, a, b)
                assert np.array_equal(q.coefficients, oq), (p, d, b)
                assert np.array_equal(r.coefficients, orr), (p, d, b)
        if p == GP:
            # full size through a size-independent property: a(x) == q(x) * (b0 + b1 x) + r at random points
            d = 1 << 22
            a = splitmix_field(99, d, p)
            b = [123456789, 987654321]
            q, r = R.Polynomial.new(F, a).quotient_and_remainder(R.Polynomial.new(F, b))
            assert not r.coefficients[1:].any() and q.coefficients[-1] == 0
            for x in (3, 0xDEADBEEF12345, GP - 2):
                lhs = orc.poly_eval(p, a, x)
                rhs = orc.add(p, orc.mul(p, orc.poly_eval(p, q.coefficients, x), orc.add(p, b[0], orc.mul(p, b[1], x))),
                              int(r.coefficients[0]))
                assert lhs == rhs
            # ... and ELEMENT FOR ELEMENT at the benchmarked size (kzg::open, src/kzg/setup.rs:63-78; the long division of
            # src/polynomial/mod.rs:170-225 by b0 + b1 x): with z = -b0 / b1 the quotient of the reference satisfies
            # q[d-1] = 0, q[j-1] = a[j] / b1 + z q[j] for every j, and the remainder is a(z) -- 2^22 - 1 equations that pin
            # every coefficient (the oracle's own O(d) division would say the same; this form needs no 2^22-step C loop per
            # case).  Monic (x - z: what kzg::open divides by) and non-monic, out of place and quotient over the dividend.
            import torch
            from ronkathon_amd import _lib as L
            da = torch.from_numpy(a.view(np.int64)).cuda()
            for b0, b1 in ((orc.neg(p, 0x123456789ABCDEF1 % p), 1), (b[0], b[1]), (0, 5), (GP - 1, GP - 1)):
                z = orc.mul(p, orc.neg(p, b0), orc.inverse(p, b1))
                sc = orc.inverse(p, b1)
                for in_place in (False, True):
                    src = da.clone()
                    dq = src if in_place else torch.full((d,), -1, dtype=torch.int64, device="cuda")
                    dr = torch.zeros(1, dtype=torch.int64, device="cuda")
                    L.check(L.lib.ronk_poly_div_linear_dev(p, src.data_ptr(), d, b0, b1, dq.data_ptr(), dr.data_ptr(), 0))
                    torch.cuda.synchronize()
                    qq = dq.cpu().numpy().view(np.uint64)
                    assert int(dr.cpu().numpy().view(np.uint64)[0]) == orc.poly_eval(p, a, z), (b0, b1, in_place)
                    assert int(qq[d - 1]) == 0
                    rhs = orc.vec_add(p, orc.vec_mul(p, a[1:], np.full(d - 1, sc, dtype=np.uint64)),
                                      orc.vec_mul(p, qq[1:], np.full(d - 1, z, dtype=np.uint64)))
                    assert np.array_equal(qq[:-1], rhs), (b0, b1, in_place)
                    if not in_place:
                        assert np.array_equal(src.cpu().numpy().view(np.uint64), a)     # the dividend is untouched
        # leading zeros in the dividend, zero dividend, x itself as divisor (generic kernel path)
        for a in ([1, 2, 3, 0, 0], [0, 0, 0, 0], [5, 0, 0, 7]):
            for b in ([3, 1], [0, 1], [4, 0]):
                try:
                    oq, orr = orc.poly_divrem(p, a, b)
                except orc.OraclePanic as e:      # e.g. [4, 0]: the reference indexes out of bounds -> both must panic
                    with pytest.raises(R.RonkPanic) as g:
                        R.Polynomial.new(F, a).quotient_and_remainder(R.Polynomial.new(F, b))
                    assert g.value.code == e.code, (p, a, b)
                    continue
                q, r = R.Polynomial.new(F, a).quotient_and_remainder(R.Polynomial.new(F, b))
                assert q.coefficients.tolist() == oq.tolist() and r.coefficients.tolist() == orr.tolist(), (p, a, b)


def test_horner_scan_device_api(R, orc):
    """ronk_poly_eval_dev / ronk_poly_div_linear_dev on device pointers and a side stream: chunk edges, several
    carry segments (> 1024 chunks), a divisor with zero constant term, short dividends"""
    import torch
    from ronkathon_amd import _lib as L
    torch.zeros(1).cuda()            # initialise torch's device context before creating a stream
    side = torch.cuda.Stream()
    for p in (GP, 101, 2):
        for d, seed in ((1, 1), (2, 2), (4096, 3), (4097, 4), (8191, 5), (70001, 6)):
            a = splitmix_field(seed, d, p)
            da = torch.from_numpy(a.view(np.int64)).cuda()
            dq = torch.full((d,), -1, dtype=torch.int64, device="cuda")
            dr = torch.zeros(2, dtype=torch.int64, device="cuda")
            for b0, b1 in ((int(splitmix_field(seed + 9, 1, p)[0]), 1), (0, 1), (5 % p, 7 % p or 1)):
                torch.cuda.synchronize()
                with torch.cuda.stream(side):
                    L.check(L.lib.ronk_poly_div_linear_dev(p, da.data_ptr(), d, b0, b1, dq.data_ptr(), dr.data_ptr(), side.cuda_stream))
                    L.check(L.lib.ronk_poly_eval_dev(p, da.data_ptr(), d, 3 % p, dr.data_ptr() + 8, side.cuda_stream))
                side.synchronize()
                oq, orr = orc.poly_divrem(p, a, [b0, b1]) if d <= 8191 else (None, None)
                q = dq.cpu().numpy().view(np.uint64); r = dr.cpu().numpy().view(np.uint64)
                if oq is not None:
                    assert np.array_equal(q, oq), (p, d, b0, b1)
                    assert int(r[0]) == int(orr[0]) and not orr[1:].any()
                else:  # a == q * (b0 + b1 x) + r at a point, and r == a(-b0/b1)
                    x = 0x1234567 % p
                    rhs = orc.add(p, orc.mul(p, orc.poly_eval(p, q, x), orc.add(p, b0, orc.mul(p, b1, x))), int(r[0]))
                    assert orc.poly_eval(p, a, x) == rhs and q[-1] == 0
                assert int(r[1]) == orc.poly_eval(p, a, 3 % p)
    # 2^22 + 5000 coefficients: 1026 chunks = two carry segments; full-size evaluate against the oracle's Horner
    d = (1 << 22) + 5000
    a = splitmix_field(77, d, GP)
    da = torch.from_numpy(a.view(np.int64)).cuda()
    dq = torch.empty(d, dtype=torch.int64, device="cuda"); dr = torch.zeros(2, dtype=torch.int64, device="cuda")
    z = 0xABCDEF0123456789 % GP
    L.check(L.lib.ronk_poly_div_linear_dev(GP, da.data_ptr(), d, GP - z, 1, dq.data_ptr(), dr.data_ptr(), 0))
    L.check(L.lib.ronk_poly_eval_dev(GP, da.data_ptr(), d, z, dr.data_ptr() + 8, 0))
    torch.cuda.synchronize()
    q = dq.cpu().numpy().view(np.uint64); r = dr.cpu().numpy().view(np.uint64)
    az = orc.poly_eval(GP, a, z)
    assert int(r[0]) == az and int(r[1]) == az and q[-1] == 0
    # synthetic division re-run on the host at the top, at both sides of a segment boundary and at the bottom
    acc = 0
    want = {}
    for j in range(d - 1, d - 3000, -1):
        want[j] = acc
        acc = orc.add(GP, orc.mul(GP, acc, z), int(a[j]))
    for j, v in want.items():
        assert int(q[j]) == v
    for x in (5, GP - 3):
        rhs = orc.add(GP, orc.mul(GP, orc.poly_eval(GP, q, x), orc.sub(GP, x, z)), az)
        assert orc.poly_eval(GP, a, x) == rhs


def test_rs_decode_vs_oracle_and_erasure_roundtrip(R, orc, refvec):
    """Message::decode (codes/reed_solomon.rs:54-106): reference round trips, oracle parity, encode -> erase -> decode"""
    from ronkathon_amd.callers import Message
    d = refvec["rs_decode"]
    F = R.PrimeField(d["p"])
    for msg in d["cases"]:
        xs, ys = Message(F, msg).encode(d["n"])
        assert Message.decode(F, xs, ys, len(msg)).data.tolist() == msg
    rng = np.random.default_rng(5)
    for p, g, n in ((127, 3, 126), (101, 2, 100), (GP, GG, 256)):
        F = R.PrimeField(p)
        for K in (1, 2, 3, 17, 64, 100):
            if K > n:
                continue
            msg = splitmix_field(K + 3, K, p)
            xs, ys = Message(F, msg).encode(n)
            keep = rng.permutation(n)[:K]                     # K surviving coordinates in arbitrary order
            got = Message.decode(F, xs[keep], ys[keep], K).data
            assert np.array_equal(got, orc.rs_decode(p, xs[keep], ys[keep], K)), (p, K)
            assert np.array_equal(got, msg)
            # arbitrary (non-codeword) values at arbitrary distinct nodes
            yy = splitmix_field(K + 99, K, p)
            assert np.array_equal(Message.decode(F, xs[keep], yy, K).data, orc.rs_decode(p, xs[keep], yy, K))
    with pytest.raises(R.RonkPanic) as e:                      # coincident nodes
        Message.decode(R.PrimeField(127), [1, 1, 2], [3, 4, 5], 3)
    assert e.value.code == -2
    with pytest.raises(R.RonkPanic) as e:                      # assert_ge::<M, K>()
        Message.decode(R.PrimeField(127), [1, 2], [3, 4], 3)
    assert e.value.code == -6
    # size the oracle cannot reach: 3000 symbols, 4096-point codeword (NTT path), 1096 erasures
    F = R.GoldilocksField
    K, n = 3000, 4096
    msg = splitmix_field(1234, K)
    xs, ys = Message(F, msg).encode(n)
    keep = np.sort(rng.permutation(n)[:K])
    assert np.array_equal(Message.decode(F, xs[keep], ys[keep], K).data, msg)


def test_rs_encode_batch_dev_vs_oracle(R, orc):
    """ronk_rs_encode_batch_dev: compact [batch][k] messages -> [batch][n] codeword values, vs Message::encode of
    the oracle (implicit zero padding on multi-pass plans, pad kernel on single-pass / generic-prime plans)"""
    import torch
    from ronkathon_amd import _lib as L
    torch.zeros(1).cuda()
    for p, g, k2, batch, K in ((GP, GG, 13, 7, 5000), (GP, GG, 14, 3, 1), (GP, GG, 13, 2, 8192), (GP, GG, 10, 5, 300),
                               (GP, GG, 16, 4, 32768), (257, 3, 8, 3, 100), (GP, GG, 25, 1, 1 << 24)):
        n = 1 << k2
        msgs = splitmix_field(k2 * 100 + batch, batch * K, p)
        dm = torch.from_numpy(msgs.view(np.int64)).cuda()
        dy = torch.full((batch * n,), -1, dtype=torch.int64, device="cuda")
        plan = L.Plan(p, g, k2, batch)
        plan.rs_encode_batch_dev(dm.data_ptr(), K, dy.data_ptr(), 0)
        torch.cuda.synchronize()
        got = dy.cpu().numpy().view(np.uint64)
        for b in range(batch):
            if k2 <= 16:
                want = orc.fft(p, g, orc.poly_from(msgs[b * K:(b + 1) * K], n))
                assert np.array_equal(got[b * n:(b + 1) * n], want), (p, k2, b)
            else:   # too big for the oracle's recursion in test time: the padded transform through the plain entry point
                pad = np.zeros(n, dtype=np.uint64); pad[:K] = msgs[b * K:(b + 1) * K]
                assert np.array_equal(got[b * n:(b + 1) * n], plan.forward(pad))
        plan.close()
    with pytest.raises(R.RonkPanic) as e:                    # assert_ge::<N, K>()
        plan = L.Plan(GP, GG, 13, 1)
        plan.rs_encode_batch_dev(dm.data_ptr(), 8193, dy.data_ptr(), 0)
    assert e.value.code == -6


def test_kzg_commit_msm_vs_reference_vectors_and_oracle(R, orc, refvec):
    """kzg::commit / kzg::open through ronk_curve_msm -- the reference's own vectors
    (src/kzg/tests.rs:92-176), random MSMs against the oracle's repeated-addition restatement, panics"""
    from ronkathon_amd import callers as K
    v = refvec["curve"]
    cv = K.Curve(v["p"], v["nr"], v["a"], v["b"])
    oc = orc.Curve(v["p"], v["nr"], v["a"], v["b"])
    srs = v["g1_srs"]
    for case in refvec["kzg_commit"]["cases"]:
        assert K.kzg_commit(cv, case["coeffs"], srs) == case["commit"]
    o = refvec["kzg_commit"]["opening"]
    assert K.kzg_open(cv, R.PlutoScalarField, o["coeffs"], o["z"], srs) == o["open"]
    assert K.kzg_commit(cv, [], srs) == K.INFINITY
    # single terms = scalar multiplication: every multiple of both generators, scalars beyond the group order too
    for base in (v["g"], v["g2"]):
        order, acc = 1, base
        while acc != orc.INFINITY:                       # order of the point, by the oracle's repeated addition
            acc = orc.curve_add(oc, acc, base); order += 1
        for k in list(range(0, 40)) + [16 * 17 + 5, 2**40 + 3, 2**64 - 1]:
            assert K.kzg_commit(cv, [k], [base]) == orc.curve_mul(oc, base, k % order), (base, k)
    # random MSMs over points of both subgroups (and Infinity), more terms than one workgroup
    rng = np.random.default_rng(11)
    pool = [orc.curve_mul(oc, v["g"], k) for k in range(0, 17)] + [orc.curve_mul(oc, v["g2"], k) for k in range(1, 12)]
    for n in (1, 2, 7, 255, 256, 257, 1000):
        pts = [pool[i] for i in rng.integers(0, len(pool), n)]
        sc = rng.integers(0, 17, n).tolist()
        assert K.kzg_commit(cv, sc, pts) == orc.kzg_commit(oc, sc, pts), n
    with pytest.raises(R.RonkPanic) as e:
        K.kzg_commit(cv, [1, 2], [v["false_point"], v["g"]])
    assert e.value.code == -11
    with pytest.raises(R.RonkPanic) as e:
        K.kzg_commit(cv, [1, 2, 3], [v["g"], v["g"]])
    assert e.value.code == -6


def test_device_resident_callers(R, orc, refvec):
    """the `_dev` forms of the callers (no host round trip between steps): kzg::open entirely on the device -- quotient by
    the linear-divisor scan, then the commitment MSM on the quotient still in HBM (src/kzg/setup.rs:63-78) -- against the
    reference's own opening vector; dft / vec / Lagrange-evaluate / RS-decode / general division `_dev` against the oracle"""
    import ctypes as C
    import torch
    from ronkathon_amd import _lib as L
    from ronkathon_amd import callers as K

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.uint64)).view(np.int64)).cuda()

    def host(t):
        return t.cpu().numpy().view(np.uint64)

    v = refvec["curve"]
    cv = K.Curve(v["p"], v["nr"], v["a"], v["b"])
    o = refvec["kzg_commit"]["opening"]
    q_ord = 17                                                       # scalar field of the reference's curve
    coeffs = dev([c % q_ord for c in o["coeffs"]])
    d = coeffs.numel()
    quot = torch.zeros(d, dtype=torch.int64, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    pts = dev(np.array(v["g1_srs"], dtype=np.uint64).reshape(-1))
    out = torch.zeros(5, dtype=torch.int64, device="cuda")
    z = o["z"] % q_ord
    L.check(L.lib.ronk_poly_div_linear_dev(q_ord, coeffs.data_ptr(), d, (q_ord - z) % q_ord, 1, quot.data_ptr(), None, None))
    L.check(L.lib.ronk_curve_msm_dev(C.byref(cv), pts.data_ptr(), len(v["g1_srs"]), quot.data_ptr(), d, out.data_ptr(),
                                     status.data_ptr(), None))
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and host(out).tolist() == list(o["open"])
    # vec neg / pow / inv
    a = splitmix_field(71, 5000); a[17] = 1
    da = dev(a); dout = torch.empty_like(da)
    L.check(L.lib.ronk_vec_neg_dev(GP, da.data_ptr(), dout.data_ptr(), a.size, None)); torch.cuda.synchronize()
    assert np.array_equal(host(dout), orc.vec_neg(GP, a))
    L.check(L.lib.ronk_vec_pow_dev(GP, da.data_ptr(), 65537, dout.data_ptr(), a.size, None)); torch.cuda.synchronize()
    assert np.array_equal(host(dout), orc.vec_pow(GP, a, 65537))
    status.zero_()
    L.check(L.lib.ronk_vec_inv_dev(GP, da.data_ptr(), dout.data_ptr(), a.size, status.data_ptr(), None)); torch.cuda.synchronize()
    assert int(status.item()) == 0 and np.array_equal(orc.vec_mul(GP, host(dout), a), np.ones(a.size, dtype=np.uint64))
    a0 = a.copy(); a0[3] = 0
    da0 = dev(a0)
    L.check(L.lib.ronk_vec_inv_dev(GP, da0.data_ptr(), dout.data_ptr(), a.size, status.data_ptr(), None)); torch.cuda.synchronize()
    assert int(status.item()) != 0
    # dft: power of two (cached plan), Bluestein, direct kernel
    for n in (1 << 14, 3 * 512, 15):
        x = splitmix_field(72 + n, n); dx = dev(x); dy = torch.empty_like(dx)
        L.check(L.lib.ronk_dft_dev(GP, GG, dx.data_ptr(), dy.data_ptr(), n, None)); torch.cuda.synchronize()
        assert np.array_equal(host(dy), orc.dft(GP, GG, x) if n <= 2048 else orc.fft(GP, GG, x)), n
    # Lagrange evaluate + RS decode on the device
    n = 256
    vals = splitmix_field(73, n); nodes = orc.lagrange_nodes(GP, GG, n)
    res = torch.zeros(1, dtype=torch.int64, device="cuda")
    dvals, dnodes = dev(vals), dev(nodes)          # (kept alive: a temporary's memory would be reused by the next one)
    L.check(L.lib.ronk_lagrange_eval_dev(GP, dvals.data_ptr(), dnodes.data_ptr(), n, 12345, res.data_ptr(), None, None))
    torch.cuda.synchronize()
    assert int(host(res)[0]) == orc.lagrange_eval(GP, vals, nodes, 12345)
    k = 300
    xs = splitmix_field(74, k); ys = splitmix_field(75, k)
    dmsg = torch.zeros(k, dtype=torch.int64, device="cuda"); status.zero_()
    dxs, dys = dev(xs), dev(ys)
    L.check(L.lib.ronk_rs_decode_dev(GP, dxs.data_ptr(), dys.data_ptr(), k, dmsg.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and np.array_equal(host(dmsg), orc.rs_decode(GP, xs, ys, k))
    # general division on the device (long-division kernel) incl. the zero-divisor panic code
    a = splitmix_field(76, 700); b = splitmix_field(77, 33)
    dq = torch.zeros(700, dtype=torch.int64, device="cuda"); dr = torch.zeros(700, dtype=torch.int64, device="cuda")
    da_, db_, dz_ = dev(a), dev(b), dev(np.zeros(4, dtype=np.uint64))
    L.check(L.lib.ronk_poly_divrem_dev(GP, da_.data_ptr(), 700, db_.data_ptr(), 33, dq.data_ptr(), dr.data_ptr(),
                                       status.data_ptr(), None))
    torch.cuda.synchronize()
    oq, o_r = orc.poly_divrem(GP, a, b)
    assert int(status.item()) == 0 and np.array_equal(host(dq), oq) and np.array_equal(host(dr), o_r)
    L.check(L.lib.ronk_poly_divrem_dev(GP, da_.data_ptr(), 700, dz_.data_ptr(), 4, dq.data_ptr(),
                                       dr.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    assert int(status.item()) == -6


def test_dft_non_power_of_two_bluestein(R, orc):
    """Polynomial::dft for n | p-1 that is not a power of two (polynomial/mod.rs:240-258): chirp-z on the NTT path
    for n >= 512, the O(n^2) kernel below; both against the oracle's definition-by-definition dft"""
    from ronkathon_amd import _lib as L
    F = R.GoldilocksField
    for n in (3, 5, 15, 17, 255, 257, 768, 771, 1285, 3855, 4369):
        assert (GP - 1) % n == 0
        x = splitmix_field(n, n)
        out = np.empty(n, dtype=np.uint64)
        L.check(L.lib.ronk_dft(GP, GG, L.ptr(x), L.ptr(out), n))
        assert np.array_equal(out, orc.dft(GP, GG, x)), n
    # sizes the O(n^2) oracle cannot reach: delta at index 1 -> the nodes omega^k; X_0 = sum; linearity
    for n in (65537, 3 * (1 << 20), 65535 * 16):
        assert (GP - 1) % n == 0
        nodes = np.empty(n, dtype=np.uint64)
        L.check(L.lib.ronk_lagrange_nodes(GP, GG, L.ptr(nodes), n))
        d1 = np.zeros(n, dtype=np.uint64); d1[1] = 1
        out = np.empty(n, dtype=np.uint64)
        L.check(L.lib.ronk_dft(GP, GG, L.ptr(d1), L.ptr(out), n))
        assert np.array_equal(out, nodes)
        x = splitmix_field(n + 1, n); y = splitmix_field(n + 2, n)
        X = np.empty(n, dtype=np.uint64); Y = np.empty(n, dtype=np.uint64); Z = np.empty(n, dtype=np.uint64)
        L.check(L.lib.ronk_dft(GP, GG, L.ptr(x), L.ptr(X), n))
        L.check(L.lib.ronk_dft(GP, GG, L.ptr(y), L.ptr(Y), n))
        L.check(L.lib.ronk_dft(GP, GG, L.ptr(F.vec_add(x, y)), L.ptr(Z), n))
        assert np.array_equal(Z, F.vec_add(X, Y))
        assert int(X[0]) == int(np.sum(x.astype(object)) % GP)
        # X_k for one k by the definition: sum_j x_j w^(jk) = evaluate at w^k
        k = 12345 % n
        assert int(X[k]) == orc.poly_eval(GP, x, int(nodes[k]))
    # Reed-Solomon encode with a non-power-of-two codeword length goes through the same path
    from ronkathon_amd.callers import Message
    msg = splitmix_field(5, 1000)
    xs, ys = Message(F, msg).encode(3 * 1024)
    assert int(ys[7]) == orc.poly_eval(GP, msg, int(xs[7])) and int(ys[3071]) == orc.poly_eval(GP, msg, int(xs[3071]))


def test_scan_and_chirp_adversarial_inputs(R, orc):
    """edge values through the Horner scans, the chirp-z dft and the interpolation: all p-1 / all zero / single one
    coefficients, evaluation points and divisors at 0, 1, p-1, 2^32-1 (= 2^64 mod p), 2^32"""
    from ronkathon_amd import _lib as L
    from ronkathon_amd.callers import Message
    F = R.GoldilocksField
    d = 9000
    for a in adversarial(d):
        for z in (0, 1, GP - 1, 0xFFFFFFFF, 1 << 32, GP - 0xFFFFFFFF):
            assert int(R.Polynomial.new(F, a).evaluate(z)) == orc.poly_eval(GP, a, z)
            b = [(GP - z) % GP, 1]                                    # x - z
            q, r = R.Polynomial.new(F, a).quotient_and_remainder(R.Polynomial.new(F, b))
            oq, orr = orc.poly_divrem(GP, a, b)
            assert np.array_equal(q.coefficients, oq) and np.array_equal(r.coefficients, orr), z
        b = [GP - 1, GP - 1]                                           # non-monic, both coefficients p-1
        q, r = R.Polynomial.new(F, a).quotient_and_remainder(R.Polynomial.new(F, b))
        oq, orr = orc.poly_divrem(GP, a, b)
        assert np.array_equal(q.coefficients, oq) and np.array_equal(r.coefficients, orr)
    n = 771                                                            # 3 * 257: chirp-z path
    for a in adversarial(n):
        out = np.empty(n, dtype=np.uint64)
        L.check(L.lib.ronk_dft(GP, GG, L.ptr(a), L.ptr(out), n))
        assert np.array_equal(out, orc.dft(GP, GG, a))
    K = 64
    xs, _ = Message(F, [1] * K).encode(256)
    for y in adversarial(K):                                           # interpolation of edge values at the first K nodes
        assert np.array_equal(Message.decode(F, xs, y, K).data, orc.rs_decode(GP, xs, y, K))


def test_lagrange_evaluate_vs_oracle(R, orc):
    """Polynomial::<Lagrange>::evaluate on the GPU (ronk_lagrange_eval) vs the oracle's step-by-step fold"""
    for p, g, ns in ((101, 2, (1, 2, 4, 5, 10, 20, 25)), (17, 14, (1, 2, 4, 8, 16)), (GP, GG, (1, 3, 8, 15, 64, 96, 1024))):
        F = R.PrimeField(p)
        for n in ns:
            c = splitmix_field(n + 7, n, p)
            lag = R.Polynomial.new_lagrange(F, c)
            nodes = orc.lagrange_nodes(p, g, n)
            assert np.array_equal(lag.basis.nodes, nodes)
            for x in [int(v) for v in splitmix_field(n + 99, 3, p)] + [int(nodes[n // 2]), 0]:
                assert int(lag.evaluate(x)) == orc.lagrange_eval(p, c, nodes, x), (p, n, x)
    # >= 256 nodes: the device picks the O(n) form for omega^i tables and the general formula for anything else
    F = R.PrimeField(GP)
    for n in (256, 510, 1024):
        c = splitmix_field(n + 1, n)
        nodes = orc.lagrange_nodes(GP, GG, n)
        other = np.unique(splitmix_field(n + 2, n + 50))[:n]            # distinct, unstructured
        shuffled = nodes.copy(); shuffled[[3, 9]] = shuffled[[9, 3]]     # the same set, not in omega^i order
        for tab in (nodes, other, shuffled):
            lag = R.Polynomial(F, c, R.Lagrange(tab))
            for x in (int(splitmix_field(n, 1)[0]), int(tab[5])):
                assert int(lag.evaluate(x)) == orc.lagrange_eval(GP, c, tab, x), (n, x)
    dup = orc.lagrange_nodes(GP, GG, 256).copy(); dup[17] = dup[200]
    with pytest.raises(R.RonkPanic) as e:
        R.Polynomial(F, splitmix_field(5, 256), R.Lagrange(dup)).evaluate(3)
    assert e.value.code == -2
    # coincident nodes: the reference panics on ONE.div(ZERO)
    bad = R.Polynomial(R.PlutoBaseField, [1, 2, 3], R.Lagrange([5, 7, 5]))
    with pytest.raises(R.RonkPanic) as e:
        bad.evaluate(3)
    assert e.value.code == -2
    with pytest.raises(orc.OraclePanic):
        orc.lagrange_eval(101, [1, 2, 3], [5, 7, 5], 3)


@pytest.mark.parametrize("k", [23, 24, 25, 26])
def test_large_plans_whole_vector(R, orc, k):
    """three-pass plans (the library default from 2^23 up; 2^23 x 1 runs two passes of 2^12 x 2^11) on ONE GPU: every output
    of the forward AND of the inverse transform against the oracle's restatement of Polynomial::fft / ifft
    (src/polynomial/mod.rs:273-323, :430-484), plus a few outputs against the DEFINITION X[i] = sum_j x[j] w^(i j) and the
    bit-exact round trip.  (The oracle takes ~2 s at 2^24 and ~8 s at 2^26 per direction.)"""
    from ronkathon_amd import _lib as L
    n = 1 << k
    x = splitmix_field(0x5EED0100 + k, n)
    x[0] = GP - 1; x[n - 1] = GP - 1; x[1] = 0
    plan = L.Plan(GP, GG, k)
    assert plan.path() == 1
    y = plan.forward(x)
    assert np.array_equal(y, orc.fft(GP, GG, x)), k
    w = orc.primitive_root_of_unity(GP, GG, n)
    for i in (1, 12345, n - 1):
        assert int(y[i]) == orc.poly_eval(GP, x, orc.pow_(GP, w, i)), (k, i)
    z = plan.inverse(x)
    assert np.array_equal(z, orc.ifft(GP, GG, x)), k
    del z
    assert np.array_equal(plan.inverse(y), x)
    plan.close()


def test_kzg_open_over_bn254(R, orc):
    """kzg::open on a production curve (src/kzg/setup.rs:63-78): `poly.div([-z, ONE])` over BN254's scalar field
    (csrc/fr_scan_kernels.h) followed by `commit(quotient, srs)` (the bucket-method MSM).  The quotient and poly(z) against the
    oracle's synthetic division on Python integers for lengths around every chunk boundary (1024 coefficients per workgroup,
    1024 chunks per carry block), non-canonical inputs included; the opening proof against the oracle's fold; and the KZG
    identity itself with an SRS of known secret: C - v G == (tau - z) pi."""
    import torch
    from oracle import bn254 as ob
    from ronkathon_amd import _lib as L
    from ronkathon_amd import callers
    rng = np.random.default_rng(77)
    torch.zeros(1).cuda()

    def words(vals):
        w = np.zeros((len(vals), 4), dtype=np.uint64)
        for i, v in enumerate(vals):
            w[i] = [(int(v) >> (64 * j)) & (2**64 - 1) for j in range(4)]
        return w

    def ints(w):
        return [sum(int(w[i, j]) << (64 * j) for j in range(4)) for i in range(w.shape[0])]
    for n in (1, 2, 3, 5, 1023, 1024, 1025, 4097, 70001, (1 << 20) + 1025 + 3):
        cs = [int(x) for x in rng.integers(0, 2**62, size=n)]
        cs = [(c * 0x9E3779B97F4A7C15F39CC0605CEDC835 + i) % ob.R for i, c in enumerate(cs)]
        if n > 2:
            cs[0] = ob.R - 1; cs[1] = 0; cs[-1] = 2**256 - 1        # a non-canonical input: taken mod r
        z = (0x123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % ob.R) if n % 2 else ob.R - 1
        dc = torch.from_numpy(words(cs).view(np.int64)).cuda()
        dq = torch.full((n, 4), -1, dtype=torch.int64, device="cuda")
        drem = torch.zeros(4, dtype=torch.int64, device="cuda")
        zw = np.array([(z >> (64 * j)) & (2**64 - 1) for j in range(4)], dtype=np.uint64)
        L.check(L.lib.ronk_poly_div_linear_bn254_dev(dc.data_ptr(), n, L.ptr(zw), dq.data_ptr(), drem.data_ptr(), 0))
        torch.cuda.synchronize()
        q, v = ob.fr_div_linear([c % ob.R for c in cs], z)
        got = ints(dq.cpu().numpy().view(np.uint64))
        assert got == q, n
        assert ints(drem.cpu().numpy().view(np.uint64).reshape(1, 4))[0] == v, n
    # the whole opening against the oracle's fold
    n = 300
    srs = ob.multiples(n)
    cs = [int(x) % ob.R for x in rng.integers(0, 2**63, size=n)]
    cs[7] = ob.R - 1
    z = 0xDEADBEEF12345
    proof, value = callers.kzg_open_bn254(cs, z, srs)
    want_pt, want_v = ob.kzg_open(cs, z, srs)
    assert proof == want_pt and value == want_v
    # the KZG identity with an SRS of known secret tau: commit(p) - p(z) G == (tau - z) * open(p, z)
    tau = 0x1F2E3D4C5B6A79887766554433221100FFEEDDCCBBAA99 % ob.R
    n = 64
    srs = [ob.mul(pow(tau, i, ob.R), ob.G) for i in range(n)]
    cs = [int(x) % ob.R for x in rng.integers(0, 2**63, size=n)]
    proof, value = callers.kzg_open_bn254(cs, z, srs)
    commit = callers.msm_bn254(srs, cs)
    lhs = ob.add(commit, ob.neg(ob.mul(value, ob.G)))
    assert lhs == ob.mul((tau - z) % ob.R, proof)
    assert value == sum(c * pow(z, i, ob.R) for i, c in enumerate(cs)) % ob.R
    with pytest.raises(R.RonkPanic) as e:                       # assert!(g1_srs.len() >= coeffs.len()), setup.rs:53
        callers.kzg_open_bn254(cs, z, srs[:10])
    assert e.value.code == -6


def test_r4_round_structure_opt_in():
    """RONK_R4MID=1 (read once per process: a subprocess): the 2^9 / 2^10-row passes as [16 . 4] . [8 | 16] with a wave-uniform
    shift layer (tile_kernels_r4.hip) -- forward and inverse against the oracle at sizes whose plans contain such passes, one
    transform and batched"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from ronkathon_amd import _lib as L
import oracle as orc
from conftest import splitmix_field
P, G = 0xFFFFFFFF00000001, 7
for k, batch, opts in ((18, 1, {}), (19, 1, {}), (20, 1, {}), (20, 3, {"tile_log2_columns": 2}), (21, 1, {}), (19, 5, {"tile_log2_columns": 4}), (24, 1, {})):
    n = 1 << k
    x = splitmix_field(0xA400 + k, n * batch)
    plan = L.Plan(P, G, k, batch, **opts)
    y, z = plan.forward(x), plan.inverse(x)
    for b in range(batch):
        assert np.array_equal(y[b * n:(b + 1) * n], orc.fft(P, G, x[b * n:(b + 1) * n])), (k, b)
        assert np.array_equal(z[b * n:(b + 1) * n], orc.ifft(P, G, x[b * n:(b + 1) * n])), (k, b)
    plan.close()
print("R4 OK")
""" % (root, root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, RONK_R4MID="1"))
    assert out.returncode == 0 and "R4 OK" in out.stdout, out.stdout[-1000:] + out.stderr[-2000:]


def test_sharded_peer_access_is_reported(R, orc):
    """the outcome of hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess is kept per rank pair and reported
    (ronk_sharded_plan_peer_access), never discarded.  Logical ranks on one device are SAME_DEVICE; with more than one visible
    GPU the real outcome is DIRECT or STAGED; RONK_FORCE_NO_PEER=1 takes the refused branch (a subprocess: the switch is read
    once per process) -- the transform still gives the oracle's result, the report says STAGED, and RONK_REQUIRE_PEER=1 turns
    it into an error at plan creation."""
    import subprocess
    import sys
    from ronkathon_amd import _lib as L
    sp = L.ShardedPlan(16, [0, 0, 0, 0])
    m, staged = sp.peer_access()
    assert staged == 0 and all(v == 0 for row in m for v in row)
    sp.close()
    nd = R.device_count()
    if nd >= 2:
        sp = L.ShardedPlan(16, [0, 1])
        m, staged = sp.peer_access()
        assert m[0][0] == 0 and m[1][1] == 0 and m[0][1] in (1, 2) and m[1][0] in (1, 2)
        assert staged == sum(v == 2 for row in m for v in row)
        sp.close()
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import ronkathon_amd as R
from ronkathon_amd import _lib as L
import oracle as orc
from conftest import splitmix_field
nd = R.device_count()
devs = [g %% nd for g in range(2)]
if os.environ.get("RONK_REQUIRE_PEER"):
    try:
        L.ShardedPlan(16, devs)
        print("CREATED")
    except R.RonkPanic as e:
        print("REFUSED", e.code)
    sys.exit(0)
sp = L.ShardedPlan(16, devs)
m, staged = sp.peer_access()
x = splitmix_field(77, 1 << 16)
ok = np.array_equal(sp.transform(x), orc.fft(0xFFFFFFFF00000001, 7, x))
print("MATRIX", m, "STAGED", staged, "OK", ok)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    # "ranks": on a one-GPU box the two logical ranks share the device; the test switch then marks the pair refused anyway, so
    # the refused branch (report, peer-copy route, RONK_REQUIRE_PEER) runs here too.  With >= 2 GPUs "1" refuses the real pair.
    env = dict(os.environ, RONK_FORCE_NO_PEER="1" if nd >= 2 else "ranks")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("MATRIX")][-1]
    assert "OK True" in line
    assert "STAGED 2" in line and "[[0, 2], [2, 0]]" in line, line
    env2 = dict(env, RONK_REQUIRE_PEER="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env2)
    assert "REFUSED" in out.stdout, out.stdout + out.stderr[-1000:]


# ---------------------------------------------------------------- row N4: bucket-method MSM over BN254 G1 (kzg::commit)
def _bn254_words(points, scalars):
    n = len(points)
    pw = np.zeros((n, 8), dtype=np.uint64); sw = np.zeros((n, 4), dtype=np.uint64)
    m64 = (1 << 64) - 1
    for i, (pt, k) in enumerate(zip(points, scalars)):
        if pt is not None:
            pw[i] = [(pt[0] >> (64 * j)) & m64 for j in range(4)] + [(pt[1] >> (64 * j)) & m64 for j in range(4)]
        sw[i] = [(int(k) >> (64 * j)) & m64 for j in range(4)]
    return pw, sw


def test_msm_bn254_small_vs_oracle(R):
    """kzg::commit's fold (src/kzg/setup.rs:48-60) on BN254: random points and scalars against the oracle's affine fold,
    with the cases the group law distinguishes: zero scalars, scalars >= r and the full 256 bits, the point at infinity, the
    same point twice in one bucket (doubling inside the mixed addition), P and -P in one bucket (sum = infinity)"""
    import random
    from oracle import bn254 as o
    from ronkathon_amd import callers
    rng = random.Random(20)
    base = [o.mul(rng.randrange(1, o.R), o.G) for _ in range(40)]
    assert callers.msm_bn254([], []) is None
    assert callers.msm_bn254([o.G], [0]) is None
    assert callers.msm_bn254([o.G], [1]) == o.G
    assert callers.msm_bn254([o.G], [o.R]) is None
    assert callers.msm_bn254([o.G, o.G], [5, 5]) == o.mul(10, o.G)                 # equal points, equal digits: P + P
    assert callers.msm_bn254([o.G, o.neg(o.G)], [77, 77]) is None                   # P + (-P)
    assert callers.msm_bn254([None, o.G, None], [9, 2, 3]) == o.TWO_G               # infinity operands
    for n in (1, 2, 3, 17, 64, 200):
        pts = [base[rng.randrange(len(base))] if rng.random() < 0.8 else None for _ in range(n)]
        ks = [rng.choice([0, 1, 2, o.R - 1, o.R, 2**256 - 1, rng.randrange(2**256), rng.randrange(o.R), rng.randrange(1 << 20)])
              for _ in range(n)]
        assert callers.msm_bn254(pts, ks) == o.msm(pts, ks), n
    # every window size the planner can pick
    pts = [base[i % len(base)] for i in range(300)]
    ks = [rng.randrange(2**256) for _ in range(300)]
    want = o.msm(pts, ks)
    try:
        for c in range(5, 17):
            os.environ["RONK_MSM_C"] = str(c)
            assert callers.msm_bn254(pts, ks) == want, c
    finally:
        os.environ.pop("RONK_MSM_C", None)


def test_msm_bn254_heavy_buckets(R):
    """every scalar equal (ONE bucket per window holds all n entries), scalars that share their top digit, a single
    point repeated: the runs are cut in tasks and folded by the collect / heavy kernels"""
    import random
    from oracle import bn254 as o
    from ronkathon_amd import callers
    rng = random.Random(5)
    n = 3000
    pts = o.multiples(n)
    tri = n * (n + 1) // 2
    assert callers.msm_bn254(pts, [1] * n) == o.mul(tri % o.R, o.G)
    k = rng.randrange(o.R)
    assert callers.msm_bn254(pts, [k] * n) == o.mul(k * tri % o.R, o.G)
    ks = [(1 << 252) + rng.randrange(1 << 40) for _ in range(n)]                    # same top digits, different low ones
    assert callers.msm_bn254(pts, ks) == o.mul(sum(kk * (i + 1) for i, kk in enumerate(ks)) % o.R, o.G)
    assert callers.msm_bn254([o.TWO_G] * n, ks) == o.mul(2 * sum(ks) % o.R, o.G)    # one point, n times
    try:
        for c in (5, 9, 16):
            os.environ["RONK_MSM_C"] = str(c)
            assert callers.msm_bn254(pts, [k] * n) == o.mul(k * tri % o.R, o.G), c
    finally:
        os.environ.pop("RONK_MSM_C", None)


def test_msm_bn254_rejects_bad_points(R):
    from oracle import bn254 as o
    from ronkathon_amd import _lib as L
    for bad in ((o.G[0], o.G[1] + 1), (o.P, 2), (1, o.P + 2), (0, 1)):
        pw, sw = _bn254_words([o.G, bad, o.TWO_G], [1, 2, 3])
        out = np.zeros(8, dtype=np.uint64)
        assert L.lib.ronk_msm_bn254(L.ptr(pw), L.ptr(sw), 3, L.ptr(out)) == -11      # RONK_ERR_NOT_ON_CURVE


@pytest.mark.parametrize("logn", [12, 16, 20])
def test_msm_bn254_large_structured(R, logn):
    """full-size MSM through the device-pointer entry point: the points are known multiples a_i * G (built by the oracle's
    repeated addition, tiled), so sum k_i P_i = (sum k_i a_i mod r) * G -- one oracle scalar multiplication checks 2^20
    terms bit for bit"""
    import torch
    from oracle import bn254 as o
    from ronkathon_amd import _lib as L
    n = 1 << logn
    m = min(n, 1 << 12)
    mult = o.multiples(m)                                   # (j+1) * G
    rng = np.random.default_rng(logn)
    sw = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    sw[::7, 3] = 0; sw[::11, 2:] = 0; sw[5] = 0                      # short scalars, a zero
    pw1, _ = _bn254_words(mult, [0] * m)
    pw = np.tile(pw1, (n // m, 1))
    acc = 0
    a = np.tile(np.arange(1, m + 1, dtype=object), n // m)
    ks = [sum(int(sw[i, j]) << (64 * j) for j in range(4)) for i in range(n)]
    acc = sum(k * int(ai) for k, ai in zip(ks, a)) % o.R
    want = o.mul(acc, o.G)
    dp = torch.from_numpy(pw.view(np.int64)).cuda(); ds = torch.from_numpy(sw.view(np.int64)).cuda()
    out = np.zeros(8, dtype=np.uint64)
    L.check(L.lib.ronk_msm_bn254_dev(dp.data_ptr(), ds.data_ptr(), n, L.ptr(out), 0))
    got = (sum(int(out[i]) << (64 * i) for i in range(4)), sum(int(out[4 + i]) << (64 * i) for i in range(4)))
    assert got == want


def test_lagrange_evaluate_fast_path_small_sizes(R):
    """the O(n) form (nodes = powers of an order-n element) forced on for sizes the oracle covers, in its own process"""
    import subprocess, sys
    e = dict(os.environ, RONK_LAGRANGE_FAST_MIN="1")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "lagrange_fast_check.py")],
                         capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0 and "lagrange fast check ok" in out.stdout, out.stdout[-500:] + out.stderr[-1500:]


def test_lagrange_evaluate_large(R, orc):
    """more than 2^16 nodes: `poly.dft().evaluate(x) == poly.evaluate(x)` (the reference's own check, src/polynomial/tests.rs:13-44)
    at 2^18 and 3 * 2^16 nodes, 0 at a node, and RONK_ERR_UNSUPPORTED for a table that is not omega^i"""
    from ronkathon_amd import _lib as L
    for n in (1 << 18, 3 << 16):
        c = splitmix_field(n % 1000, n)
        y = np.empty(n, dtype=np.uint64); nodes = np.empty(n, dtype=np.uint64)
        L.check(L.lib.ronk_dft(GP, GG, L.ptr(c), L.ptr(y), n))
        L.check(L.lib.ronk_lagrange_nodes(GP, GG, L.ptr(nodes), n))
        import ctypes as C
        for x in (7, 0x123456789ABCDEF % GP):      # (2 has order 192: it IS a node when 192 | n, where the reference yields 0)
            assert L.out_scalar(L.lib.ronk_lagrange_eval, GP, L.ptr(y), L.ptr(nodes), n, x) == orc.poly_eval(GP, c, x), (n, x)
        if n % 192 == 0:
            assert L.out_scalar(L.lib.ronk_lagrange_eval, GP, L.ptr(y), L.ptr(nodes), n, 2) == 0
        assert L.out_scalar(L.lib.ronk_lagrange_eval, GP, L.ptr(y), L.ptr(nodes), n, int(nodes[12345])) == 0
        bad = nodes.copy(); bad[777] = bad[778]
        out = C.c_uint64(0)
        assert L.lib.ronk_lagrange_eval(GP, L.ptr(y), L.ptr(bad), n, 2, C.byref(out)) == -9


def test_rs_decode_fast_path_small_sizes(R):
    """the O(K log K) decode (x_j = q^j) forced on from K = 2, against the oracle, in its own process"""
    import subprocess, sys
    e = dict(os.environ, RONK_RS_FAST_MIN="2")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "rs_fast_check.py")],
                         capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0 and "rs fast check ok" in out.stdout, out.stdout[-500:] + out.stderr[-1500:]


def test_rs_decode_large_roundtrip(R, orc):
    """Message::decode(encode(msg)) == msg (src/codes/reed_solomon.rs:136-218 round trips) far beyond the O(K^2) kernels:
    K = 2^16 of N = 2^17, K = 3 * 2^14 of N = 3 * 2^15, K = 2^20 of N = 2^21; the default path at K = 2^12; and
    RONK_ERR_UNSUPPORTED for a large node set that is not geometric"""
    from ronkathon_amd import _lib as L
    for k, N in ((1 << 12, 1 << 13), (1 << 16, 1 << 17), (3 << 14, 3 << 15), (1 << 20, 1 << 21)):
        msg = splitmix_field(k % 977, k)
        full = np.empty(N, dtype=np.uint64); nodes = np.empty(N, dtype=np.uint64)
        L.check(L.lib.ronk_dft(GP, GG, L.ptr(np.concatenate([msg, np.zeros(N - k, dtype=np.uint64)])), L.ptr(full), N))
        L.check(L.lib.ronk_lagrange_nodes(GP, GG, L.ptr(nodes), N))
        out = np.zeros(k, dtype=np.uint64)
        L.check(L.lib.ronk_rs_decode(GP, L.ptr(nodes), L.ptr(full), k, L.ptr(out)))
        assert np.array_equal(out, msg), (k, N)
    bad = nodes[: 1 << 15].copy(); bad[[5, 9]] = bad[[9, 5]]
    assert L.lib.ronk_rs_decode(GP, L.ptr(bad), L.ptr(full), 1 << 15, L.ptr(out)) == -9


def test_low_degree_extension_batch(R, orc):
    """row N2's "iNTT -> zero-pad -> NTT": values on {omega_K^i} -> values on shift * {omega_N^i}, batched, against the oracle
    (ifft, then the polynomial evaluated on the extended domain); shift = 1 and a coset; single- and multi-pass plan sizes"""
    import torch
    from ronkathon_amd import _lib as L
    for lk, ln, batch in ((4, 6, 3), (10, 12, 5), (12, 14, 2), (14, 15, 3), (16, 17, 1)):
        K, N = 1 << lk, 1 << ln
        pk = L.Plan(GP, GG, lk, batch); pn = L.Plan(GP, GG, ln, batch)
        ev = splitmix_field(lk * 31 + batch, K * batch)
        d_ev = torch.from_numpy(ev.view(np.int64)).cuda()
        d_co = torch.empty(K * batch, dtype=torch.int64, device="cuda"); d_out = torch.empty(N * batch, dtype=torch.int64, device="cuda")
        for shift in (1, 7, 0x123456789ABCDEF % GP):
            L.check(L.lib.ronk_lde_batch_dev(pk.h, pn.h, d_ev.data_ptr(), d_co.data_ptr(), d_out.data_ptr(), shift, 0))
            torch.cuda.synchronize()
            got = d_out.cpu().numpy().view(np.uint64)
            for b in range(batch):
                co = orc.ifft(GP, GG, ev[b * K:(b + 1) * K])
                sc = np.array([orc.mul(GP, int(co[i]), pow(shift, i, GP)) for i in range(K)], dtype=np.uint64) if shift != 1 else co
                want = orc.fft(GP, GG, np.concatenate([sc, np.zeros(N - K, dtype=np.uint64)]))
                assert np.array_equal(got[b * N:(b + 1) * N], want), (lk, ln, b, shift)
                if shift == 1:   # an extension: the values on the K-point sub-domain are the original ones
                    assert np.array_equal(got[b * N:(b + 1) * N][:: N // K], ev[b * K:(b + 1) * K])
        pk.close(); pn.close()
