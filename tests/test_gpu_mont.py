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


@pytest.mark.parametrize("k", [25, 26])
def test_mont_three_pass_largest_sizes_whole_vector(L, orc, k):
    """2^25 / 2^26 over the prime above 2^63 (three-pass plans on the Montgomery bodies): every output against the oracle,
    forward; the inverse by round trip"""
    p, g = PRIMES[0]
    n = 1 << k
    x = corners(splitmix_field(0x900 + k, n, p), p)
    plan = L.Plan(p, g, k)
    assert plan.path() == 2
    y = plan.forward(x)
    assert np.array_equal(y, orc.fft(p, g, x))
    assert np.array_equal(plan.inverse(y), x)
    plan.close()


@pytest.mark.parametrize("p,g", PRIMES[:2] + [GL_OTHER])
def test_mont_sharded_transform(L, orc, p, g):
    """ronk_sharded_plan_create_p: the four-step transform inside the library over a generic prime (the phases on the Montgomery
    tile bodies), 1 .. 8 logical ranks on the visible GPU(s), chunked and unchunked, forward and inverse, up to 2^24 -- whole
    vectors against the oracle called with (p, g).  Reference: the transform is generic over the modulus
    (src/algebra/field/prime/mod.rs:39-52, src/polynomial/mod.rs:273-323)."""
    import ronkathon_amd as R
    ndev = R.device_count()
    cases = [(12, 1, 1, False), (16, 2, 0, False), (16, 4, 2, True), (18, 8, 1, False), (20, 8, 4, False), (20, 4, 0, True)]
    if (p, g) == PRIMES[0]:
        cases += [(22, 2, 0, True), (24, 8, 2, False)]
    for log2n, W, chunks, inv in cases:
        x = corners(splitmix_field(0xA00 + log2n + W, 1 << log2n, p), p)
        ref = orc.ifft(p, g, x) if inv else orc.fft(p, g, x)
        layouts = [[0] * W] + ([[r % ndev for r in range(W)]] if ndev >= 2 else [])
        for devs in layouts:
            sp = L.ShardedPlan(log2n, devs, inverse=inv, chunks=chunks, p=p, g=g)
            assert np.array_equal(sp.transform(x), ref), (hex(p), log2n, W, chunks, inv, devs)
            if log2n <= 20:
                assert np.array_equal(sp.transform(x), ref)     # second call: buffer reuse behind the event guards
            sp.close()
    # what the reference reports by panicking, and what this path does not cover
    with pytest.raises(L.RonkPanic) as e:
        L.ShardedPlan(60, [0], p=p, g=g)                        # 2^60 does not divide p - 1
    assert e.value.code == L.ERR_NO_ROOT
    with pytest.raises(L.RonkPanic) as e:
        L.ShardedPlan(16, [0, 0], p=p, g=(g * g) % p)           # a residue generates no full 2-power subgroup: no radix-2 fallback here
    assert e.value.code == L.ERR_UNSUPPORTED
    with pytest.raises(L.RonkPanic) as e:
        L.ShardedPlan(16, [0, 0], p=p - 2, g=3)                 # not a prime
    assert e.value.code in (L.ERR_NOT_PRIME, L.ERR_NO_ROOT)


def test_mont_dist_phases_one_process_per_rank_protocol(L, orc):
    """ronk_dist_plan_create_p: the two local phases a rank runs around the caller's all-to-all (one process per GPU), here for
    every rank of a world of 4 in one process with the exchange done by numpy -- generic prime, chunked send layout"""
    import ctypes as C
    from ronkathon_amd import dist as D
    from test_gpu_parity import _DevArr
    p, g = PRIMES[0]
    for log2n, W, chunks in ((16, 4, 1), (20, 4, 2), (20, 2, 4)):
        n = 1 << log2n
        x = corners(splitmix_field(0xB00 + log2n + W, n, p), p)
        Rr, Cc, Rw, Cw = D.shape(log2n, W)
        Cwc = Cw // chunks
        send = []
        for r in range(W):
            eng = D.HipEngine(log2n, False, r, W, chunks=chunks, p=p, g=g)
            din, dsend = _DevArr(D.scatter_input(x, r, W)), _DevArr(n=n // W)
            eng.phase1(din.ptr, dsend.ptr)
            send.append(dsend.get())
            eng.close()
        out = np.empty(n, dtype=np.uint64)
        for h in range(W):
            recv = np.empty(n // W, dtype=np.uint64)
            blk = Rw * Cwc
            for r in range(W):
                for j in range(chunks):
                    recv[(r * chunks + j) * blk:(r * chunks + j + 1) * blk] = send[r][j * Rr * Cwc + h * blk: j * Rr * Cwc + (h + 1) * blk]
            eng = D.HipEngine(log2n, False, h, W, chunks=chunks, p=p, g=g)
            drecv, dout = _DevArr(recv), _DevArr(n=n // W)
            eng.phase2(drecv.ptr, dout.ptr)
            D.place_output(out, dout.get(), h, W)
            eng.close()
        assert np.array_equal(out, orc.fft(p, g, x)), (log2n, W, chunks)


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


def test_newton_division_over_generic_primes(L, orc):
    """quotient_and_remainder (reference src/polynomial/mod.rs:170-225) in O(n log n) for EVERY prime with enough 2-adicity, not
    only Goldilocks (round 6): ronk_poly_divrem and ronk_poly_divrem_dev take the Newton form on the NTT path over Montgomery
    arithmetic; the oracle is the statement-by-statement long division called with p.  Whole quotients and remainders; 2^20 by
    2^19 for the prime above 2^63 (the one-workgroup long division needs minutes there) via the identity a = q b + r."""
    import torch

    def divrem(p, a, b):
        a, b = L.arr(a), L.arr(b)
        q, r = np.empty_like(a), np.empty_like(a)
        L.check(L.lib.ronk_poly_divrem(p, L.ptr(a), a.size, L.ptr(b), b.size, L.ptr(q), L.ptr(r)))
        return q, r

    def z(v, k):
        return np.concatenate([v, np.zeros(k, dtype=np.uint64)])

    for p, g in PRIMES[:2] + [(GP, 7)]:
        cases = [(splitmix_field(1, 40000, p), splitmix_field(2, 9000, p)),
                 (z(splitmix_field(5, 20000, p), 1234), splitmix_field(6, 700, p)),      # dividend with leading zeros
                 (splitmix_field(3, 30000, p), z(splitmix_field(4, 3000, p), 500)),      # ragged divisor: the reference's early stop / panic
                 (splitmix_field(9, 8192, p), splitmix_field(10, 4096, p))]
        for a, b in cases:
            try:
                oq, o_r = orc.poly_divrem(p, a, b)
            except orc.OraclePanic as e:
                with pytest.raises(L.RonkPanic) as e2:
                    divrem(p, a, b)
                assert e2.value.code == e.code
                continue
            q, r = divrem(p, a, b)
            assert np.array_equal(q, oq) and np.array_equal(r, o_r), (hex(p), a.size, b.size)
        # device-resident forms: the probing one and the fully asynchronous one for full-length operands
        a, b = cases[0]
        oq, o_r = orc.poly_divrem(p, a, b)
        da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
        for fn in (L.lib.ronk_poly_divrem_dev, L.lib.ronk_poly_divrem_full_dev):
            dq = torch.full((a.size,), -1, dtype=torch.int64, device="cuda"); dr = torch.full((a.size,), -1, dtype=torch.int64, device="cuda")
            status = torch.full((1,), 77, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            L.check(fn(p, da.data_ptr(), a.size, db.data_ptr(), b.size, dq.data_ptr(), dr.data_ptr(), status.data_ptr(), None))
            torch.cuda.synchronize()
            assert int(status.item()) == 0
            assert np.array_equal(dq.cpu().numpy().view(np.uint64), oq) and np.array_equal(dr.cpu().numpy().view(np.uint64), o_r), hex(p)
        # the promise of the full-length form is checked on the device
        a0 = a.copy(); a0[-1] = 0
        da0 = torch.from_numpy(a0.view(np.int64)).cuda()
        L.check(L.lib.ronk_poly_divrem_full_dev(p, da0.data_ptr(), a.size, db.data_ptr(), b.size, dq.data_ptr(), dr.data_ptr(), status.data_ptr(), None))
        torch.cuda.synchronize()
        assert int(status.item()) == L.ERR_INVALID
    p, g = PRIMES[0]
    a, b = splitmix_field(21, 1 << 20, p), splitmix_field(22, 1 << 19, p)
    q, r = divrem(p, a, b)
    assert not r[b.size - 1:].any()
    for pt in (5, 0xABCDEF0123456789 % p):
        lhs = orc.poly_eval(p, a, pt)
        rhs = orc.add(p, orc.mul(p, orc.poly_eval(p, q, pt), orc.poly_eval(p, b, pt)), orc.poly_eval(p, r, pt))
        assert lhs == rhs
    assert int(q[a.size - b.size]) == orc.div(p, int(a[-1]), int(b[-1])) and not q[a.size - b.size + 1:].any()
    # a prime without the 2-adicity (F_101: 4 | p - 1 only) keeps the long division -- same window, same results
    a, b = splitmix_field(31, 3000, 101), splitmix_field(32, 100, 101)
    a[-1] = 1; b[-1] = 1
    q, r = divrem(101, a, b)
    oq, o_r = orc.poly_divrem(101, a, b)
    assert np.array_equal(q, oq) and np.array_equal(r, o_r)


def test_full_length_division_is_capturable(L, orc):
    """ronk_poly_divrem_full_dev inside a hipGraph: nothing is read back, so a size the long division could never serve (2^17 by
    2^16) is captured and replayed; values against the oracle on a smaller pair, identity a = q b + r on the large one"""
    import torch
    for p in (GP, PRIMES[0][0]):
        d, d2 = 6000, 1500
        a, b = splitmix_field(41, d, p), splitmix_field(42, d2, p)
        da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
        dq = torch.zeros(d, dtype=torch.int64, device="cuda"); dr = torch.zeros(d, dtype=torch.int64, device="cuda")
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        s = torch.cuda.Stream()
        # warm the workspace pool and the plan cache outside the capture
        L.check(L.lib.ronk_poly_divrem_full_dev(p, da.data_ptr(), d, db.data_ptr(), d2, dq.data_ptr(), dr.data_ptr(), status.data_ptr(), s.cuda_stream))
        s.synchronize()
        oq, o_r = orc.poly_divrem(p, a, b)
        assert int(status.item()) == 0 and np.array_equal(dq.cpu().numpy().view(np.uint64), oq) and np.array_equal(dr.cpu().numpy().view(np.uint64), o_r)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            st = torch.cuda.current_stream().cuda_stream
            L.check(L.lib.ronk_poly_divrem_full_dev(p, da.data_ptr(), d, db.data_ptr(), d2, dq.data_ptr(), dr.data_ptr(), status.data_ptr(), st))
        for rep in range(2):
            a = splitmix_field(43 + rep, d, p)
            da.copy_(torch.from_numpy(a.view(np.int64)))
            oq, o_r = orc.poly_divrem(p, a, b)
            dq.fill_(-1); dr.fill_(-1); status.fill_(9)
            gr.replay()
            torch.cuda.synchronize()
            assert int(status.item()) == 0
            assert np.array_equal(dq.cpu().numpy().view(np.uint64), oq) and np.array_equal(dr.cpu().numpy().view(np.uint64), o_r), (hex(p), rep)
        del gr
    # the probing form refuses a capture it could only serve with minutes of long division
    d, d2 = 1 << 17, 1 << 16
    a, b = splitmix_field(51, d), splitmix_field(52, d2)
    da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
    dq = torch.zeros(d, dtype=torch.int64, device="cuda"); dr = torch.zeros(d, dtype=torch.int64, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        st = torch.cuda.current_stream().cuda_stream
        rc = L.lib.ronk_poly_divrem_dev(GP, da.data_ptr(), d, db.data_ptr(), d2, dq.data_ptr(), dr.data_ptr(), status.data_ptr(), st)
    assert rc == L.ERR_UNSUPPORTED
    del gr
