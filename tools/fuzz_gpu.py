#!/usr/bin/env python3
"""tools/fuzz_gpu.py -- time-bounded randomised parity sweep of the C-ABI on a GPU box against the oracle (development aid;
the fixed cases live in tests/).  usage: FUZZ_SEED=n FUZZ_SECONDS=s python tools/fuzz_gpu.py [section ...]

Sections (all by default, each gets an equal share of the time): ntt (plans of random size / batch / direction / planner
options, host and device forms, several arrays per call), mul (products of random lengths, incl. the fused N = 2^21 / 2^22
path), generic (transforms, products, dft over small primes and non-power-of-two sizes), mont (random NTT-friendly primes of
33 .. 64 bits on the Montgomery tile path: transforms, products, dft), divrem (general divisors incl.
trailing zeros and the Newton path), lindiv (division by a linear divisor + evaluate, whole-vector recurrence), codes
(Reed-Solomon encode / decode / LDE), vec (element-wise operators incl. zero inverses), lagrange (barycentric evaluate), msm (BN254 G1 against known multiples of G),
sharded (the in-library four-step plan with 1 .. 8 logical ranks on this GPU), threads (four host threads, null stream and own
streams mixed).
Exit status 1 on any mismatch; every mismatch prints the arguments that reproduce it."""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C  # noqa: E402

import torch  # noqa: E402

import oracle as orc  # noqa: E402
import ronkathon_amd as R  # noqa: E402
from conftest import splitmix_field  # noqa: E402
from ronkathon_amd import _lib as L  # noqa: E402

GP, GG = R.GOLDILOCKS_P, R.GOLDILOCKS_G
SEED = int(os.environ.get("FUZZ_SEED", "1"))
SECONDS = float(os.environ.get("FUZZ_SECONDS", "120"))
rng = random.Random(SEED)
bad = 0
counts = {}


def report(section, what, *args):
    global bad
    bad += 1
    print("MISMATCH [%s] %s %r" % (section, what, args), flush=True)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint64)


def edge_values(p, n, it):
    """random residues with the edge values of the carry paths sprinkled in"""
    x = splitmix_field(SEED * 100003 + it, n, p)
    if n and rng.random() < 0.5:
        pool = [0, 1, p - 1, p - 2, (p - 1) // 2, 0xFFFFFFFF % p, 0x100000000 % p, (p - 0xFFFFFFFF) % p]
        for _ in range(min(n, 16)):
            x[rng.randrange(n)] = rng.choice(pool)
    return x


def sec_ntt(deadline):
    it = 0
    while time.time() < deadline:
        it += 1
        k = rng.choice([rng.randrange(0, 12), rng.randrange(12, 19), rng.randrange(19, 24)])
        n = 1 << k
        batch = rng.choice([1, 1, 2, 3, 5, 8, 17, 64]) if k <= 16 else (rng.choice([1, 2, 3]) if k <= 20 else 1)
        opts = {}
        if rng.random() < 0.5:
            opts = dict(tile_log2_columns=rng.choice([-1, 2, 3, 4]), twiddle_matrix_log2_max=rng.choice([-1, 0, 18, 25]),
                        in_flight=rng.choice([-1, 1, 2, 3]))
        x = edge_values(GP, n * batch, it)
        try:
            plan = L.Plan(GP, GG, k, batch, **opts)
        except L.RonkPanic as e:
            if opts:
                continue   # an option combination the planner refuses is not a parity matter
            report("ntt", "plan_create", k, batch, str(e)); continue
        want = np.concatenate([orc.fft(GP, GG, x[b * n:(b + 1) * n]) for b in range(batch)]) if k <= 20 or rng.random() < 0.3 else None
        y = plan.forward(x)
        if want is not None and not np.array_equal(y, want):
            report("ntt", "forward", k, batch, opts)
        if not np.array_equal(plan.inverse(y), x):
            report("ntt", "inverse(forward)", k, batch, opts)
        # device forms: out of place, in place, several arrays per call
        dx = dev(x); dy = torch.empty_like(dx)
        plan.forward_dev(dx.data_ptr(), dy.data_ptr())
        torch.cuda.synchronize()
        if not np.array_equal(host(dy), y):
            report("ntt", "forward_dev", k, batch, opts)
        plan.inverse_dev(dy.data_ptr(), dy.data_ptr())
        torch.cuda.synchronize()
        if not np.array_equal(host(dy), x):
            report("ntt", "inverse_dev in place", k, batch, opts)
        if k <= 22:
            cnt = rng.randrange(1, 6)
            ins = [dev(np.roll(x, i)) for i in range(cnt)]
            outs = [torch.empty_like(dx) for _ in range(cnt)]
            plan.forward_many_dev([t.data_ptr() for t in ins], [t.data_ptr() for t in outs])
            torch.cuda.synchronize()
            for i in range(cnt):
                single = plan.forward(np.roll(x, i))
                if not np.array_equal(host(outs[i]), single):
                    report("ntt", "forward_many_dev", k, batch, opts, cnt, i)
        plan.close()
    counts["ntt"] = it


def sec_mul(deadline):
    it = 0
    while time.time() < deadline:
        it += 1
        cls = rng.random()
        if cls < 0.5:
            d1, d2 = rng.randrange(1, 3000), rng.randrange(1, 3000)
        elif cls < 0.8:
            d1, d2 = rng.randrange(1, 300000), rng.randrange(1, 300000)
        else:   # products whose transform length is 2^21 or 2^22: the fused middle
            tot = rng.randrange((1 << 20) + 2, (1 << 22) + 2)
            d1 = rng.randrange(1, tot - 1); d2 = tot - d1
        a = edge_values(GP, d1, 2 * it); b = edge_values(GP, d2, 2 * it + 1)
        out = np.empty(d1 + d2 - 1, dtype=np.uint64)
        if rng.random() < 0.5:
            L.check(L.lib.ronk_poly_mul(GP, GG, L.ptr(a), d1, L.ptr(b), d2, L.ptr(out)))
        else:
            da, db = dev(a), dev(b)
            do = torch.empty(d1 + d2 - 1, dtype=torch.int64, device="cuda")
            L.check(L.lib.ronk_poly_mul_dev(GP, GG, da.data_ptr(), d1, db.data_ptr(), d2, do.data_ptr(), 0))
            torch.cuda.synchronize()
            out = host(do)
        if d1 * d2 <= 9_000_000:
            if not np.array_equal(out, orc.poly_mul(GP, a, b)):
                report("mul", "schoolbook", d1, d2)
        else:
            for t in (2, 0x1234567, rng.randrange(GP)):
                if orc.poly_eval(GP, out, t) != orc.mul(GP, orc.poly_eval(GP, a, t), orc.poly_eval(GP, b, t)):
                    report("mul", "homomorphism", d1, d2, t); break
            # the ends of the product are one-term / few-term sums: exact
            if int(out[0]) != orc.mul(GP, int(a[0]), int(b[0])) or int(out[-1]) != orc.mul(GP, int(a[-1]), int(b[-1])):
                report("mul", "end coefficients", d1, d2)
    counts["mul"] = it


SMALL = [(17, [2, 4, 8, 16]), (97, [2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 96]), (101, [2, 4, 5, 10, 20, 25, 50]),
         (12289, [3, 4, 256, 1024, 4096, 12, 768]), (65537, [2, 64, 4096, 65536]), (998244353, [7, 17, 119, 1 << 12, 1 << 20]),
         (0xFFFFFFFF00000001, [3, 5, 15, 17, 255, 257, 51, 85, 4369])]


def sec_generic(deadline):
    it = 0
    while time.time() < deadline:
        it += 1
        p, sizes = rng.choice(SMALL)
        n = rng.choice(sizes)
        g = GG if p == GP else orc.find_primitive_element(p)
        x = edge_values(p, n, it)
        out = np.empty(n, dtype=np.uint64)
        if n <= 65536:
            L.check(L.lib.ronk_dft(p, g, L.ptr(x), L.ptr(out), n))
            want = orc.dft(p, g, x) if n <= 4096 else (orc.fft(p, g, x) if n & (n - 1) == 0 else None)
            if want is not None and not np.array_equal(out, want):
                report("generic", "dft", p, n)
        if n & (n - 1) == 0:
            nodes = np.empty(n, dtype=np.uint64)
            L.check(L.lib.ronk_fft(p, g, L.ptr(x), L.ptr(out), L.ptr(nodes), n))
            if not np.array_equal(out, orc.fft(p, g, x)) or not np.array_equal(nodes, orc.lagrange_nodes(p, g, n)):
                report("generic", "fft", p, n)
            back = np.empty(n, dtype=np.uint64)
            L.check(L.lib.ronk_ifft(p, g, L.ptr(out), L.ptr(back), n))
            if not np.array_equal(back, x):
                report("generic", "ifft", p, n)
        d1, d2 = rng.randrange(1, 300), rng.randrange(1, 300)
        a = edge_values(p, d1, 3 * it); b = edge_values(p, d2, 3 * it + 1)
        prod = np.empty(d1 + d2 - 1, dtype=np.uint64)
        rc = L.lib.ronk_poly_mul(p, g, L.ptr(a), d1, L.ptr(b), d2, L.ptr(prod))
        if rc == 0:
            if not np.array_equal(prod, orc.poly_mul(p, a, b)):
                report("generic", "poly_mul", p, d1, d2)
        elif (p - 1) % (1 << (d1 + d2 - 2).bit_length()) == 0:
            report("generic", "poly_mul rc", p, d1, d2, rc)
    counts["generic"] = it


def _is_prime(n):
    if n < 2:
        return False
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % a == 0:
            return n == a
    d, s_ = n - 1, 0
    while d % 2 == 0:
        d //= 2; s_ += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s_ - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def random_ntt_prime(min_adicity):
    """a random odd prime c * 2^t + 1 below 2^64 with t >= min_adicity, of a random bit length (33 .. 64), and a quadratic
    non-residue g: every such (p, g) runs the tile kernels over Montgomery arithmetic (ronk_plan_path() == 2)"""
    while True:
        bits = rng.randrange(max(33, min_adicity + 2), 65)
        t = rng.randrange(min_adicity, min(bits - 1, 40))
        c = rng.randrange(1 << (bits - t - 1), 1 << (bits - t)) | 1
        p = c * (1 << t) + 1
        if p < (1 << 64) and p != GP and _is_prime(p):
            g = next(x for x in range(2, 200) if pow(x, (p - 1) // 2, p) == p - 1)
            return p, g


def sec_mont(deadline):
    """random NTT-friendly primes of every bit length on the Montgomery tile path: transforms of random size / batch / planner
    options (host and device forms), products through the NTT path against the schoolbook oracle, dft, round trips"""
    it = 0
    while time.time() < deadline:
        it += 1
        k = rng.choice([rng.randrange(4, 13), rng.randrange(13, 19), rng.randrange(19, 23)])
        p, g = random_ntt_prime(max(k, 12))
        n = 1 << k
        batch = rng.choice([1, 1, 2, 3, 5, 17, 40]) if k <= 16 else (rng.choice([1, 2, 3]) if k <= 19 else 1)
        opts = {}
        if rng.random() < 0.5:
            opts = dict(tile_log2_columns=rng.choice([-1, 2, 3, 4]), twiddle_matrix_log2_max=rng.choice([-1, 0, 18, 25]),
                        in_flight=rng.choice([-1, 1, 2]))
        x = edge_values(p, n * batch, it)
        try:
            plan = L.Plan(p, g, k, batch, **opts)
        except L.RonkPanic as e:
            if opts:
                continue
            report("mont", "plan_create", hex(p), g, k, batch, str(e)); continue
        if plan.path() != 2:
            report("mont", "path", hex(p), g, k, plan.path())
        want = np.concatenate([orc.fft(p, g, x[b * n:(b + 1) * n]) for b in range(batch)])
        y = plan.forward(x)
        if not np.array_equal(y, want):
            report("mont", "forward", hex(p), g, k, batch, opts)
        wanti = np.concatenate([orc.ifft(p, g, x[b * n:(b + 1) * n]) for b in range(batch)])
        if not np.array_equal(plan.inverse(x), wanti):
            report("mont", "inverse", hex(p), g, k, batch, opts)
        dx = dev(x); dy = torch.empty_like(dx)
        plan.forward_dev(dx.data_ptr(), dy.data_ptr())
        plan.inverse_dev(dy.data_ptr(), dy.data_ptr())
        torch.cuda.synchronize()
        if not np.array_equal(host(dy), x):
            report("mont", "inverse_dev(forward_dev) in place", hex(p), g, k, batch, opts)
        plan.close()
        # products: short ones against the schoolbook definition, longer ones against three oracle transforms
        d1, d2 = (rng.randrange(40, 3000), rng.randrange(30, 3000)) if rng.random() < 0.6 else (rng.randrange(1, 1 << 17), rng.randrange(1, 1 << 17))
        m = d1 + d2 - 1
        N = 1 << max(4, (m - 1).bit_length())
        if (p - 1) % N == 0:
            a = edge_values(p, d1, 5 * it); b = edge_values(p, d2, 5 * it + 1)
            prod = np.empty(m, dtype=np.uint64)
            L.check(L.lib.ronk_poly_mul(p, g, L.ptr(a), d1, L.ptr(b), d2, L.ptr(prod)))
            if d1 * d2 <= 9_000_000:
                ok = np.array_equal(prod, orc.poly_mul(p, a, b))
            else:
                pa = np.zeros(N, dtype=np.uint64); pa[:d1] = a
                pb = np.zeros(N, dtype=np.uint64); pb[:d2] = b
                ok = np.array_equal(prod, orc.ifft(p, g, orc.vec_mul(p, orc.fft(p, g, pa), orc.fft(p, g, pb)))[:m])
            if not ok:
                report("mont", "poly_mul", hex(p), g, d1, d2)
        if k <= 14:
            out = np.empty(n, dtype=np.uint64)
            L.check(L.lib.ronk_dft(p, g, L.ptr(x[:n]), L.ptr(out), n))
            if not np.array_equal(out, want[:n]):
                report("mont", "dft", hex(p), g, n)
    counts["mont"] = it


def sec_divrem(deadline):
    it = 0
    while time.time() < deadline:
        it += 1
        p = rng.choice([GP, GP, 101, 17, 0xFFFFFFFFFFFFFFC5, 0xFFFFFFFC00000001, 29 * 2**57 + 1])   # (the last two: Newton over Montgomery)
        big = p in (GP, 0xFFFFFFFC00000001, 29 * 2**57 + 1) and rng.random() < 0.25
        d = rng.randrange(1, 40000 if big else 700)
        d2 = rng.randrange(1, d + 3)
        a = edge_values(p, d, 2 * it); b = edge_values(p, d2, 2 * it + 1)
        shape = rng.random()
        if shape < 0.25:
            b[d2 - rng.randrange(1, d2 + 1):] = 0        # trailing zeros: leading coefficient search of the reference
        elif shape < 0.35:
            a[d - rng.randrange(1, d + 1):] = 0
        if not b.any():
            b[0] = 1
        want = None
        try:
            want = orc.poly_divrem(p, a, b)
        except Exception as e:   # the reference panics (index / inverse of zero): the library must refuse too
            want = e
        q = np.empty(d, dtype=np.uint64); r = np.empty(d, dtype=np.uint64)
        rc = L.lib.ronk_poly_divrem(p, L.ptr(a), d, L.ptr(b), d2, L.ptr(q), L.ptr(r))
        if isinstance(want, Exception):
            if rc == 0:
                report("divrem", "accepted what the oracle refuses", p, d, d2, shape)
        elif rc != 0:
            report("divrem", "rc", p, d, d2, rc)
        elif not (np.array_equal(q, want[0]) and np.array_equal(r, want[1])):
            report("divrem", "values", p, d, d2, shape)
        # the device-resident entry on the same operands (long division, or Newton after the degree probe), now and then on a
        # caller stream and with the remainder written over the dividend
        if d >= d2:
            import torch
            da = torch.from_numpy(a.view(np.int64)).cuda(); db = torch.from_numpy(b.view(np.int64)).cuda()
            alias = rng.random() < 0.3
            dq = torch.full((d,), -1, dtype=torch.int64, device="cuda")
            dr = da if alias else torch.full((d,), -1, dtype=torch.int64, device="cuda")
            dst = torch.full((1,), 55, dtype=torch.int32, device="cuda")
            st = torch.cuda.Stream() if rng.random() < 0.5 else None
            torch.cuda.synchronize()
            rc2 = L.lib.ronk_poly_divrem_dev(p, da.data_ptr(), d, db.data_ptr(), d2, dq.data_ptr(), dr.data_ptr(), dst.data_ptr(),
                                             st.cuda_stream if st is not None else None)
            torch.cuda.synchronize()
            code = int(dst.item())
            if rc2 != 0:
                report("divrem", "dev rc", p, d, d2, rc2)
            elif isinstance(want, Exception):
                if code == 0:
                    report("divrem", "dev accepted what the oracle refuses", p, d, d2, shape)
            elif code != 0:
                report("divrem", "dev status", p, d, d2, code)
            elif not (np.array_equal(dq.cpu().numpy().view(np.uint64), want[0]) and np.array_equal(dr.cpu().numpy().view(np.uint64), want[1])):
                report("divrem", "dev values", p, d, d2, shape, alias)
            # the fully asynchronous form for full-length operands (ronk_poly_divrem_full_dev): values when the promise holds,
            # RONK_ERR_INVALID in the status word when a top coefficient is ZERO, RONK_ERR_UNSUPPORTED for fields without the roots
            if not alias:
                dq.fill_(-1); dr.fill_(-1); dst.fill_(55)
                torch.cuda.synchronize()
                rc3 = L.lib.ronk_poly_divrem_full_dev(p, da.data_ptr(), d, db.data_ptr(), d2, dq.data_ptr(), dr.data_ptr(), dst.data_ptr(),
                                                      st.cuda_stream if st is not None else None)
                torch.cuda.synchronize()
                full = int(a[-1]) != 0 and int(b[-1]) != 0
                if rc3 == L.ERR_UNSUPPORTED:
                    if p in (GP, 0xFFFFFFFC00000001, 29 * 2**57 + 1):
                        report("divrem", "full: unsupported over an NTT-friendly prime", p, d, d2)
                elif rc3 != 0:
                    report("divrem", "full rc", p, d, d2, rc3)
                elif not full:
                    if int(dst.item()) != L.ERR_INVALID:
                        report("divrem", "full: broken promise not reported", p, d, d2, int(dst.item()))
                elif isinstance(want, Exception) or int(dst.item()) != 0:
                    report("divrem", "full status", p, d, d2, int(dst.item()))
                elif not (np.array_equal(dq.cpu().numpy().view(np.uint64), want[0]) and np.array_equal(dr.cpu().numpy().view(np.uint64), want[1])):
                    report("divrem", "full values", p, d, d2)
    counts["divrem"] = it


def sec_lindiv(deadline):
    it = 0
    while time.time() < deadline:
        it += 1
        p = rng.choice([GP, GP, GP, 101, 2, 0xFFFFFFFFFFFFFFC5])
        d = rng.choice([rng.randrange(1, 64), rng.randrange(1, 5000), rng.randrange(1, 400000),
                        2048 * rng.randrange(1, 40) + rng.choice([-1, 0, 1]), rng.randrange(1, 1 << 23),
                        8192 * rng.randrange(129, 513) + rng.choice([-1, 0, 1])])   # (the one-launch form: 129 .. 512 chunks of 8192)
        a = edge_values(p, d, it)
        z = rng.choice([0, 1, p - 1, rng.randrange(p)]) % p
        b1 = rng.choice([1, 1, rng.randrange(1, p) if p > 2 else 1])
        b0 = orc.mul(p, orc.neg(p, z), b1)
        inplace = rng.random() < 0.3
        misalign = rng.random() < 0.3
        buf = torch.zeros(d + 1, dtype=torch.int64, device="cuda")
        da = buf[1:] if misalign else buf[:d]
        da.copy_(torch.from_numpy(a.view(np.int64)))
        dq = da if inplace else torch.full((d,), -1, dtype=torch.int64, device="cuda")
        dr = torch.zeros(2, dtype=torch.int64, device="cuda")
        L.check(L.lib.ronk_poly_eval_dev(p, da.data_ptr(), d, z, dr.data_ptr() + 8, 0))
        L.check(L.lib.ronk_poly_div_linear_dev(p, da.data_ptr(), d, b0, b1, dq.data_ptr(), dr.data_ptr(), 0))
        torch.cuda.synchronize()
        q = host(dq); r = host(dr)
        val = orc.poly_eval(p, a, z); sc = orc.inverse(p, b1)
        ok = int(r[0]) == val and int(r[1]) == val and int(q[d - 1]) == 0
        if ok and d > 1:
            rhs = orc.vec_add(p, orc.vec_mul(p, a[1:], np.full(d - 1, sc, dtype=np.uint64)),
                              orc.vec_mul(p, q[1:], np.full(d - 1, z, dtype=np.uint64)))
            ok = np.array_equal(q[:-1], rhs)
        if not ok:
            report("lindiv", "div / eval", p, d, z, b1, inplace, misalign)
    counts["lindiv"] = it


def sec_codes(deadline):
    it = 0
    while time.time() < deadline:
        it += 1
        p = rng.choice([GP, GP, 101, 12289])
        g = GG if p == GP else orc.find_primitive_element(p)
        if p == GP:
            logn = rng.randrange(1, 15 if rng.random() < 0.8 else 21)
        else:
            logn = rng.randrange(1, {101: 3, 12289: 13}[p])
        n = 1 << logn
        k = rng.randrange(1, n + 1)
        if p != GP or rng.random() < 0.5:
            k = min(k, 600)
        msg = edge_values(p, k, it)
        xs = np.empty(n, dtype=np.uint64); ys = np.empty(n, dtype=np.uint64)
        L.check(L.lib.ronk_rs_encode(p, g, L.ptr(msg), k, n, L.ptr(xs), L.ptr(ys)))
        if n * k <= 1 << 26:
            wx, wy = orc.rs_encode(p, g, msg, n)     # the reference's form: N evaluations of a K-term polynomial
        else:                                        # the same values as one transform of the zero-padded message
            wx = orc.lagrange_nodes(p, g, n)
            wy = orc.fft(p, g, np.concatenate([msg, np.zeros(n - k, dtype=np.uint64)]))
        if not (np.array_equal(xs, wx) and np.array_equal(ys, wy)):
            report("codes", "encode", p, k, n); continue
        # decode from the first k coordinates, and (small k) from a random k-subset = erasures elsewhere.  The reference's
        # generator heuristic can return a non-generator (12289 -> 12287, a square): omega_N then has a smaller order, nodes
        # repeat, and decode panics on the zero denominator -- the oracle says which.
        def decode_case(what, sx, sy):
            out = np.empty(k, dtype=np.uint64)
            rc = L.lib.ronk_rs_decode(p, L.ptr(sx), L.ptr(sy), k, L.ptr(out))
            if k <= 1500:
                try:
                    want = orc.rs_decode(p, sx, sy, k)
                except orc.OraclePanic:
                    want = None
            else:
                want = msg
            if want is None:
                if rc == 0:
                    report("codes", what + ": accepted what the oracle refuses", p, k, n)
            elif rc != 0 or not np.array_equal(out, want):
                report("codes", what, p, k, n, rc)
        decode_case("decode prefix", xs, ys)
        if k <= 512 and n > k:
            sel = np.array(sorted(rng.sample(range(n), k)))
            decode_case("decode subset", np.ascontiguousarray(xs[sel]), np.ascontiguousarray(ys[sel]))
        if p == GP and logn <= 16:
            cnt = rng.randrange(1, 9)
            plan = L.Plan(GP, GG, logn, cnt)
            msgs = edge_values(GP, cnt * k, 7 * it)
            dm = dev(msgs); dyv = torch.empty(cnt * n, dtype=torch.int64, device="cuda")
            plan.rs_encode_batch_dev(dm.data_ptr(), k, dyv.data_ptr())
            torch.cuda.synchronize()
            got = host(dyv)
            for i in range(cnt):
                if not np.array_equal(got[i * n:(i + 1) * n], orc.fft(GP, GG, np.concatenate([msgs[i * k:(i + 1) * k], np.zeros(n - k, dtype=np.uint64)]))):
                    report("codes", "encode_batch", k, n, cnt, i); break
            plan.close()
    counts["codes"] = it


def sec_vec(deadline):
    it = 0
    while time.time() < deadline:
        it += 1
        p = rng.choice([GP, GP, 101, 2, 3, 0xFFFFFFFFFFFFFFC5, 0x7FFFFFFF])
        n = rng.choice([rng.randrange(1, 100), rng.randrange(1, 100000)])
        a = edge_values(p, n, 2 * it); b = edge_values(p, n, 2 * it + 1)
        out = np.empty(n, dtype=np.uint64)
        for name, fn, ofn in (("add", L.lib.ronk_vec_add, orc.vec_add), ("sub", L.lib.ronk_vec_sub, orc.vec_sub),
                              ("mul", L.lib.ronk_vec_mul, orc.vec_mul)):
            L.check(fn(p, L.ptr(a), L.ptr(b), L.ptr(out), n))
            if not np.array_equal(out, ofn(p, a, b)):
                report("vec", name, p, n)
        L.check(L.lib.ronk_vec_neg(p, L.ptr(a), L.ptr(out), n))
        if not np.array_equal(out, orc.vec_neg(p, a)):
            report("vec", "neg", p, n)
        e = rng.choice([0, 1, 2, p - 2, p - 1, rng.randrange(1 << 64)])
        L.check(L.lib.ronk_vec_pow(p, L.ptr(a), e, L.ptr(out), n))
        if not np.array_equal(out, orc.vec_pow(p, a, e)):
            report("vec", "pow", p, n, e)
        rc = L.lib.ronk_vec_inv(p, L.ptr(a), L.ptr(out), n)
        if (a == 0).any():
            if rc != L.ERR_ZERO_INVERSE:
                report("vec", "inverse of zero accepted", p, n, rc)
        elif rc != 0 or not np.array_equal(out, orc.vec_inv(p, a)):
            report("vec", "inv", p, n, rc)
        # FieldExt (prime/mod.rs:142-226): euler_criterion of everything, sqrt of the squares; a non-residue is the reference's assert
        if p > 2:
            L.check(L.lib.ronk_vec_euler(p, L.ptr(a), L.ptr(out), n))
            if not np.array_equal(out, orc.vec_euler(p, a)):
                report("vec", "euler", p, n)
            sq = orc.vec_mul(p, a, a)
            r0 = np.empty(n, dtype=np.uint64); r1 = np.empty(n, dtype=np.uint64)
            rc = L.lib.ronk_vec_sqrt(p, L.ptr(sq), L.ptr(r0), L.ptr(r1), n)
            o0, o1 = orc.vec_sqrt(p, sq)
            if rc != 0 or not (np.array_equal(r0, o0) and np.array_equal(r1, o1)):
                report("vec", "sqrt", p, n, rc)
            nonres = a[(orc.vec_euler(p, a) == 0) & (a != 0)]
            if nonres.size:
                t = sq.copy(); t[rng.randrange(n)] = nonres[0]
                if L.lib.ronk_vec_sqrt(p, L.ptr(t), L.ptr(r0), L.ptr(r1), n) != L.ERR_NOT_RESIDUE:
                    report("vec", "sqrt of a non-residue accepted", p, n)
    counts["vec"] = it


def sec_lagrange(deadline):
    it = 0
    while time.time() < deadline:
        it += 1
        p = rng.choice([GP, GP, 101, 12289])
        g = GG if p == GP else orc.find_primitive_element(p)
        if rng.random() < 0.6:   # Lagrange::new's node table, any n | p - 1
            n = rng.choice({GP: [2, 3, 5, 15, 16, 17, 256, 4096, 65536, 1 << 18, 3 << 16], 101: [2, 4, 5, 10, 25, 100],
                            12289: [3, 4, 12, 256, 4096]}[p])
            nodes = orc.lagrange_nodes(p, g, n) if n & (n - 1) == 0 else None
            if nodes is None:
                w = orc.primitive_root_of_unity(p, g, n)
                nodes = np.empty(n, dtype=np.uint64); acc = 1
                for i in range(n):
                    nodes[i] = acc; acc = orc.mul(p, acc, w)
        else:                      # arbitrary distinct nodes
            n = rng.randrange(1, min(p - 1, 400))
            nodes = np.array(rng.sample(range(min(p, 1 << 62)), n), dtype=np.uint64) if p > 1 << 20 else \
                np.array(rng.sample(range(p), n), dtype=np.uint64)
        c = edge_values(p, n, it)
        x = rng.choice([rng.randrange(p), int(nodes[rng.randrange(n)]), 0, 1])
        o = C.c_uint64(0)
        rc = L.lib.ronk_lagrange_eval(p, L.ptr(c), L.ptr(nodes), n, x, C.byref(o))
        if n <= 4096:
            try:
                want = orc.lagrange_eval(p, c, nodes, x)
            except Exception:
                want = None
            if want is None:
                if rc == 0:
                    report("lagrange", "accepted what the oracle refuses", p, n, x)
            elif rc != 0 or o.value != want:
                report("lagrange", "eval", p, n, x, rc)
        elif rc != 0:
            report("lagrange", "rc", p, n, x, rc)
        else:   # large omega tables: through the coefficient form (ifft, then Horner) when n is a power of two
            if n & (n - 1) == 0:
                coef = orc.ifft(p, g, c)
                is_node = orc.pow_(p, x, n) == 1
                want = 0 if is_node else orc.poly_eval(p, coef, x)
                if o.value != want:
                    report("lagrange", "eval (omega table)", p, n, x)
    counts["lagrange"] = it


def sec_msm(deadline):
    from oracle import bn254 as ob
    from ronkathon_amd import callers
    mult = ob.multiples(1 << 10)
    it = 0
    while time.time() < deadline:
        it += 1
        n = rng.choice([rng.randrange(1, 40), rng.randrange(1, 5000), rng.randrange(1, 70000)])
        idx = [rng.randrange(len(mult)) for _ in range(n)]
        pts = [mult[i] for i in idx]
        ks = [rng.choice([rng.randrange(2**256), rng.randrange(ob.R), rng.randrange(1 << 64), 0, 1, ob.R - 1, ob.R]) for _ in range(n)]
        if rng.random() < 0.3:      # points at infinity, repeated points with opposite scalars (bucket sums through infinity)
            for _ in range(min(n, 8)):
                j = rng.randrange(n); pts[j] = None; idx[j] = -1
        want_k = sum(k * (i + 1) for k, i in zip(ks, idx) if i >= 0) % ob.R
        want = ob.mul(want_k, ob.G) if want_k else None
        if callers.msm_bn254(pts, ks) != want:
            report("msm", "bn254", n, it)
        # kzg::open over the same curve: the quotient of a random polynomial over F_r by (x - z), committed against multiples of G
        # -- with SRS[i] = (idx_i + 1) G the proof is (sum_i q_i (idx_i + 1)) G and the value poly(z), both from Python integers
        if n <= 20000 and all(i >= 0 for i in idx):
            cs = [rng.choice([rng.randrange(ob.R), rng.randrange(2**256), 0, ob.R - 1]) for _ in range(n)]
            z = rng.choice([rng.randrange(ob.R), 0, 1, ob.R - 1, rng.randrange(2**256)])
            proof, value = callers.kzg_open_bn254(cs, z, pts)
            q, v = ob.fr_div_linear([c % ob.R for c in cs], z % ob.R)
            wk = sum(qi * (i + 1) for qi, i in zip(q, idx)) % ob.R
            if value != v or proof != (ob.mul(wk, ob.G) if wk else None):
                report("msm", "kzg_open_bn254", n, it)
    counts["msm"] = it


def sec_sharded(deadline):
    it = 0
    while time.time() < deadline:
        it += 1
        w = rng.choice([1, 2, 4, 8])
        k = rng.randrange(max(8, 2 * w.bit_length()), 23)
        inverse = rng.random() < 0.4
        chunks = rng.choice([0, 1, 2, 4])
        p, g = rng.choice([(GP, GG), (GP, GG), (0xFFFFFFFC00000001, 10), (29 * 2**57 + 1, 3)])   # (the four-step phases over Montgomery too)
        x = edge_values(p, 1 << k, it)
        try:
            sp = L.ShardedPlan(k, [0] * w, inverse=inverse, chunks=chunks) if p == GP else L.ShardedPlan(k, [0] * w, inverse=inverse, chunks=chunks, p=p, g=g)
        except L.RonkPanic as e:
            if e.code in (L.ERR_INVALID, L.ERR_UNSUPPORTED):
                continue    # a split this size cannot take (too few columns per rank / chunk)
            report("sharded", "create", k, w, chunks, str(e)); continue
        y = sp.transform(x)
        want = orc.ifft(p, g, x) if inverse else orc.fft(p, g, x)
        if not np.array_equal(y, want):
            report("sharded", "transform", hex(p), k, w, inverse, chunks)
        sp.close()
    counts["sharded"] = it


def sec_threads(deadline):
    """four host threads at once (the reference's `cargo test` runs tests on parallel threads): one-shot transforms, products up
    to the fused sizes, linear and general divisions, evaluations -- half of the threads on the null stream, half on their own"""
    import threading
    total = [0]
    lock = threading.Lock()

    def worker(tid):
        r = random.Random(SEED * 1000 + tid)
        st = torch.cuda.Stream() if tid & 1 else None
        sp = st.cuda_stream if st else 0
        it = 0
        while time.time() < deadline:
            it += 1
            op = r.choice(["fft", "mul", "mulbig", "lindiv", "divrem", "eval"])
            try:
                if op == "fft":
                    k = r.randrange(4, 21); n = 1 << k
                    x = splitmix_field(tid * 7919 + it, n)
                    out = np.empty(n, dtype=np.uint64)
                    L.check(L.lib.ronk_fft(GP, GG, L.ptr(x), L.ptr(out), None, n))
                    if not np.array_equal(out, orc.fft(GP, GG, x)):
                        report("threads", "fft", tid, k)
                elif op in ("mul", "mulbig"):
                    if op == "mul":
                        d1, d2 = r.randrange(1, 100000), r.randrange(1, 100000)
                    else:
                        tot = r.randrange((1 << 20) + 2, (1 << 22) + 2); d1 = r.randrange(1, tot - 1); d2 = tot - d1
                    a = splitmix_field(tid * 7919 + 2 * it, d1); b = splitmix_field(tid * 7919 + 2 * it + 1, d2)
                    with torch.cuda.stream(st) if st else torch.cuda.stream(torch.cuda.default_stream()):
                        da, db = dev(a), dev(b)
                        do = torch.empty(d1 + d2 - 1, dtype=torch.int64, device="cuda")
                        torch.cuda.current_stream().synchronize()
                    L.check(L.lib.ronk_poly_mul_dev(GP, GG, da.data_ptr(), d1, db.data_ptr(), d2, do.data_ptr(), sp))
                    (st or torch.cuda.default_stream()).synchronize()
                    out = host(do)
                    t = 0x1234567 + it
                    if orc.poly_eval(GP, out, t) != orc.mul(GP, orc.poly_eval(GP, a, t), orc.poly_eval(GP, b, t)):
                        report("threads", op, tid, d1, d2)
                elif op in ("lindiv", "eval"):
                    d = r.choice([r.randrange(1, 5000), r.randrange(1, 3000000)])
                    a = splitmix_field(tid * 7919 + it, d)
                    z = r.randrange(GP)
                    with torch.cuda.stream(st) if st else torch.cuda.stream(torch.cuda.default_stream()):
                        da = dev(a); dq = torch.empty(d, dtype=torch.int64, device="cuda"); dr = torch.zeros(2, dtype=torch.int64, device="cuda")
                        torch.cuda.current_stream().synchronize()
                    L.check(L.lib.ronk_poly_eval_dev(GP, da.data_ptr(), d, z, dr.data_ptr() + 8, sp))
                    L.check(L.lib.ronk_poly_div_linear_dev(GP, da.data_ptr(), d, orc.neg(GP, z), 1, dq.data_ptr(), dr.data_ptr(), sp))
                    (st or torch.cuda.default_stream()).synchronize()
                    q = host(dq); rr = host(dr)
                    val = orc.poly_eval(GP, a, z)
                    ok = int(rr[0]) == val and int(rr[1]) == val
                    if ok and d > 1:
                        ok = np.array_equal(q[:-1], orc.vec_add(GP, a[1:], orc.vec_mul(GP, q[1:], np.full(d - 1, z, dtype=np.uint64))))
                    if not ok:
                        report("threads", op, tid, d, z)
                else:
                    d = r.randrange(2, 30000); d2 = r.randrange(1, d + 1)
                    a = splitmix_field(tid * 7919 + 2 * it, d); b = splitmix_field(tid * 7919 + 2 * it + 1, d2)
                    if not b[-1]:
                        b[-1] = 1
                    q = np.empty(d, dtype=np.uint64); rm = np.empty(d, dtype=np.uint64)
                    L.check(L.lib.ronk_poly_divrem(GP, L.ptr(a), d, L.ptr(b), d2, L.ptr(q), L.ptr(rm)))
                    wq, wr = orc.poly_divrem(GP, a, b)
                    if not (np.array_equal(q, wq) and np.array_equal(rm, wr)):
                        report("threads", "divrem", tid, d, d2)
            except Exception as e:   # noqa: BLE001
                report("threads", "exception in " + op, tid, repr(e))
        with lock:
            total[0] += it

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    counts["threads"] = total[0]


SECTIONS = dict(ntt=sec_ntt, mul=sec_mul, generic=sec_generic, mont=sec_mont, divrem=sec_divrem, lindiv=sec_lindiv, codes=sec_codes, vec=sec_vec,
                lagrange=sec_lagrange, msm=sec_msm, sharded=sec_sharded, threads=sec_threads)


def main():
    names = sys.argv[1:] or list(SECTIONS)
    share = SECONDS / len(names)
    for nm in names:
        t0 = time.time()
        try:
            SECTIONS[nm](t0 + share)
        except Exception as e:   # an unexpected refusal is a finding too
            import traceback
            traceback.print_exc()
            report(nm, "exception", repr(e))
        print("section %-9s %5d cases in %.0f s, mismatches so far %d" % (nm, counts.get(nm, 0), time.time() - t0, bad), flush=True)
    print("fuzz seed %d done, mismatches: %d" % (SEED, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
