"""The tile kernels over generic odd 64-bit primes (Montgomery form, csrc/field_policy.h, tile_kernels_mont.hip) against
the oracle: ronkathon's field and transform are generic over the modulus (src/algebra/field/prime/mod.rs:39-52,
src/polynomial/mod.rs:273-323, :430-484).  Whole vectors, bit-exact.  Needs a real MI355X (-m gpu)."""
import numpy as np
import pytest

from conftest import splitmix_field

pytestmark = pytest.mark.gpu

GP = 0xFFFFFFFF00000001
# (p, primitive element): a prime above 2^63 with 2-adicity 34 (sums carry into bit 64), 29 * 2^57 + 1 (below 2^62:
# no carries at all), 3 * 2^30 + 1 (a 32-bit prime: high limbs mostly zero), Goldilocks itself with a generator other
# than the reference's 7 (its shift-twiddle kernels do not apply: the roots are no powers of two)
PRIMES = [(0xFFFFFFFC00000001, 10), (29 * 2**57 + 1, 3), (3 * 2**30 + 1, 5)]
GL_OTHER = (GP, 7 * 7 * 7)   # 343 = 7^3: an odd power of a non-residue is a non-residue -- omega_{2^k} keeps its exact order


@pytest.fixture(scope="module")
def L():
    from ronkathon_amd import _lib as L
    import ronkathon_amd as R
    assert R.device_count() >= 1
    return L


@pytest.fixture(scope="module")
def orc():
    import oracle
    return oracle


def poly_mul(L, p, g, a, b):
    """ronk_poly_mul through the C ABI (host buffers)"""
    a, b = L.arr(a), L.arr(b)
    out = np.empty(a.size + b.size - 1, dtype=np.uint64)
    L.check(L.lib.ronk_poly_mul(p, g, L.ptr(a), a.size, L.ptr(b), b.size, L.ptr(out)))
    return out


def dft(L, p, g, x):
    x = L.arr(x)
    out = np.empty_like(x)
    L.check(L.lib.ronk_dft(p, g, L.ptr(x), L.ptr(out), x.size))
    return out


def corners(x, p):
    x = x.copy()
    x[0] = p - 1
    x[-1] = p - 1
    if x.size > 2:
        x[1] = 0
    return x


@pytest.mark.parametrize("p,g", PRIMES + [GL_OTHER])
def test_mont_tiled_path_selected(L, p, g):
    for k in (4, 12, 16, 22):
        plan = L.Plan(p, g, k)
        assert plan.path() == 2, "generic primes with a full 2-power subgroup under g run the tile kernels"
        plan.close()
    plan = L.Plan(p, g, 3)       # n < 16: the radix-2 path
    assert plan.path() == 0
    plan.close()
    plan = L.Plan(GP, 7, 16)
    assert plan.path() == 1      # the Goldilocks shift-twiddle kernels are untouched
    plan.close()


@pytest.mark.parametrize("p,g", PRIMES + [GL_OTHER])
@pytest.mark.parametrize("k", [4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_mont_single_pass_sizes(L, orc, p, g, k):
    """n <= 2^12: one pass, the batch is the column axis (ragged last tile, staged I/O for n = 16 / 32)"""
    n, batch = 1 << k, 37
    x = corners(splitmix_field(0x600 + k, n * batch, p), p)
    plan = L.Plan(p, g, k, batch)
    y = plan.forward(x)
    for b in range(batch):
        assert np.array_equal(y[b * n:(b + 1) * n], orc.fft(p, g, x[b * n:(b + 1) * n])), (k, b)
    z = plan.inverse(y)
    assert np.array_equal(z, x)
    for b in (0, batch - 1):
        assert np.array_equal(plan.inverse(x)[b * n:(b + 1) * n], orc.ifft(p, g, x[b * n:(b + 1) * n]))
    plan.close()


@pytest.mark.parametrize("p,g", PRIMES + [GL_OTHER])
@pytest.mark.parametrize("k", [13, 14, 15, 16, 17, 18, 19, 20, 21, 22])
def test_mont_two_pass_whole_vector(L, orc, p, g, k):
    """2^13 .. 2^22 (latency kernels up to 2^18, tile kernels above), forward and inverse against the oracle, every element"""
    n = 1 << k
    x = corners(splitmix_field(0x700 + k, n, p), p)
    plan = L.Plan(p, g, k)
    y = plan.forward(x)
    assert np.array_equal(y, orc.fft(p, g, x)), k
    assert np.array_equal(plan.inverse(x), orc.ifft(p, g, x)), k
    assert np.array_equal(plan.inverse(y), x)
    plan.close()


@pytest.mark.parametrize("p,g", PRIMES[:2])
def test_mont_batched_tile_kernels(L, orc, p, g):
    """batches large enough for the 16-coefficient tile kernels at 2^16 (and their specialised shapes), all rows"""
    n, batch = 1 << 16, 24
    x = splitmix_field(0x7F0, n * batch, p)
    for opts in ({}, {"tile_log2_columns": 4}, {"tile_log2_columns": 2, "twiddle_matrix_log2_max": 0}):
        plan = L.Plan(p, g, 16, batch, **opts)
        y = plan.forward(x)
        for b in range(batch):
            assert np.array_equal(y[b * n:(b + 1) * n], orc.fft(p, g, x[b * n:(b + 1) * n])), (opts, b)
        assert np.array_equal(plan.inverse(y), x)
        plan.close()


@pytest.mark.parametrize("p,g", PRIMES[:1])
@pytest.mark.parametrize("k", [23, 24])
def test_mont_three_pass_whole_vector(L, orc, p, g, k):
    n = 1 << k
    x = corners(splitmix_field(0x800 + k, n, p), p)
    plan = L.Plan(p, g, k)
    assert plan.path() == 2
    y = plan.forward(x)
    assert np.array_equal(y, orc.fft(p, g, x))
    assert np.array_equal(plan.inverse(y), x)
    plan.close()


@pytest.mark.parametrize("p,g", PRIMES + [GL_OTHER])
def test_mont_poly_mul_ntt_path(L, orc, p, g):
    """Polynomial Mul (src/polynomial/arithmetic.rs:97-119) through the Montgomery NTT path: schoolbook oracle at small sizes,
    three oracle transforms at large ones; ragged lengths, the pair plan and its padding limits"""
    for d, d2 in ((40, 30), (100, 29), (257, 1000), (4096, 4096), (5000, 3)):
        a = corners(splitmix_field(d, d, p), p)
        b = corners(splitmix_field(d2 + 7, d2, p), p)
        got = poly_mul(L, p, g, a, b)
        assert got.size == d + d2 - 1
        assert np.array_equal(got, orc.poly_mul(p, a, b)), (d, d2)
    for d, d2 in ((1 << 15, 1 << 15), (300001, 7), (1 << 19, (1 << 19) + 5), (1 << 21, 1 << 21)):
        m = d + d2 - 1
        N = 1 << (m - 1).bit_length()
        if (p - 1) % N:
            continue
        a = corners(splitmix_field(d, d, p), p)
        b = corners(splitmix_field(d2 + 7, d2, p), p)
        got = poly_mul(L, p, g, a, b)
        pa = np.zeros(N, dtype=np.uint64); pa[:d] = a
        pb = np.zeros(N, dtype=np.uint64); pb[:d2] = b
        fa, fb = orc.fft(p, g, pa), orc.fft(p, g, pb)
        want = orc.ifft(p, g, orc.vec_mul(p, fa, fb))
        assert got.size == m and np.array_equal(got, want[:m]), (d, d2)
        assert not want[m:].any()


@pytest.mark.parametrize("p,g", PRIMES[:2])
def test_mont_dft_and_lagrange_nodes(L, orc, p, g):
    """Polynomial::dft == fft on the tiled path (omega_n has order exactly n); the nodes that come with it"""
    n = 1 << 14
    x = splitmix_field(0x900, n, p)
    assert np.array_equal(dft(L, p, g, x), orc.fft(p, g, x))
    plan = L.Plan(p, g, 14)
    y, nodes = plan.forward(x, nodes=True)
    assert np.array_equal(nodes, orc.lagrange_nodes(p, g, n))
    plan.close()


def test_mont_non_generator_keeps_radix2_path(L, orc):
    """a `g` that is a quadratic residue generates no full 2-power subgroup: the reference's recursion is then not the DFT,
    and the plan stays on the radix-2 path that restates it stage by stage"""
    p = 0xFFFFFFFC00000001
    g = 100   # 10^2
    plan = L.Plan(p, g, 10)
    assert plan.path() == 0
    x = splitmix_field(5, 1 << 10, p)
    assert np.array_equal(plan.forward(x), orc.fft(p, g, x))
    plan.close()


@pytest.mark.parametrize("p,g", PRIMES[:2])
def test_mont_rs_encode_batch_and_lde(L, orc, p, g):
    """the batched Reed-Solomon encode (src/codes/reed_solomon.rs:42-52: y_i = message(omega_N^i), compact messages, implicit zero
    padding) and the plain low-degree extension (ifft on the small domain, encode on the large one) over a Montgomery prime:
    multi-pass plans read the messages in place through the padding feature, single-pass plans pad first"""
    import torch
    torch.zeros(1).cuda()
    for k2, batch, K in ((13, 7, 5000), (16, 24, 32768), (10, 5, 300), (14, 3, 1)):
        n = 1 << k2
        msgs = splitmix_field(k2 * 100 + batch, batch * K, p)
        dm = torch.from_numpy(msgs.view(np.int64)).cuda()
        dy = torch.full((batch * n,), -1, dtype=torch.int64, device="cuda")
        plan = L.Plan(p, g, k2, batch)
        assert plan.path() == 2
        plan.rs_encode_batch_dev(dm.data_ptr(), K, dy.data_ptr(), 0)
        torch.cuda.synchronize()
        got = dy.cpu().numpy().view(np.uint64)
        for b in range(batch):
            pad = np.zeros(n, dtype=np.uint64); pad[:K] = msgs[b * K:(b + 1) * K]
            assert np.array_equal(got[b * n:(b + 1) * n], orc.fft(p, g, pad)), (k2, b)
        plan.close()
    # LDE with shift 1: values on the 2^12-point domain -> values on the 2^14-point domain
    kk, kn, batch = 12, 14, 6
    coeffs = splitmix_field(0xE1, batch << kk, p)
    pk, pn = L.Plan(p, g, kk, batch), L.Plan(p, g, kn, batch)
    evals = pk.forward(coeffs)
    de = torch.from_numpy(evals.view(np.int64)).cuda()
    dc = torch.empty_like(de)
    dout = torch.empty(batch << kn, dtype=torch.int64, device="cuda")
    L.check(L.lib.ronk_lde_batch_dev(pk.h, pn.h, de.data_ptr(), dc.data_ptr(), dout.data_ptr(), 1, 0))
    torch.cuda.synchronize()
    assert np.array_equal(dc.cpu().numpy().view(np.uint64), coeffs)
    got = dout.cpu().numpy().view(np.uint64)
    for b in range(batch):
        pad = np.zeros(1 << kn, dtype=np.uint64); pad[:1 << kk] = coeffs[b << kk:(b + 1) << kk]
        assert np.array_equal(got[b << kn:(b + 1) << kn], orc.fft(p, g, pad)), b
    # on a coset: values of the same polynomials at shift * omega_N^i = the transform of c_i * shift^i
    shift = 7
    L.check(L.lib.ronk_lde_batch_dev(pk.h, pn.h, de.data_ptr(), dc.data_ptr(), dout.data_ptr(), shift, 0))
    torch.cuda.synchronize()
    got = dout.cpu().numpy().view(np.uint64)
    pw = np.array([pow(shift, i, p) for i in range(1 << kk)], dtype=np.uint64)
    for b in (0, batch - 1):
        pad = np.zeros(1 << kn, dtype=np.uint64); pad[:1 << kk] = orc.vec_mul(p, coeffs[b << kk:(b + 1) << kk], pw)
        assert np.array_equal(got[b << kn:(b + 1) << kn], orc.fft(p, g, pad)), b
    pk.close(); pn.close()
