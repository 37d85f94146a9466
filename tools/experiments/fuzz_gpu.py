"""randomised parity sweep on the GPU (development aid; the fixed cases live in tests/): transforms of random size / batch /
direction against the oracle, fused multiplies of random lengths, MSMs of random length against known multiples of G,
divisions by a linear divisor / evaluations of random length, root and modulus"""
import os, sys, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc
from oracle import bn254 as ob
from conftest import splitmix_field
import ronkathon_amd as R
from ronkathon_amd import _lib as L, callers
GP, GG = R.GOLDILOCKS_P, R.GOLDILOCKS_G
rng = random.Random(int(os.environ.get("FUZZ_SEED", "1")))
bad = 0
for it in range(60):
    k = rng.randrange(4, 21); batch = rng.choice([1, 1, 2, 3, 5, 8, 17]) if k <= 17 else 1
    n = 1 << k
    x = splitmix_field(it * 7 + 1, n * batch)
    plan = L.Plan(GP, GG, k, batch)
    y = plan.forward(x)
    ok = all(np.array_equal(y[b * n:(b + 1) * n], orc.fft(GP, GG, x[b * n:(b + 1) * n])) for b in range(batch))
    ok = ok and np.array_equal(plan.inverse(y), x)
    plan.close()
    if not ok: bad += 1; print("NTT MISMATCH", k, batch)
for it in range(25):
    d1 = rng.randrange(1, 20000); d2 = rng.randrange(1, 20000)
    a = splitmix_field(1000 + it, d1); b = splitmix_field(2000 + it, d2)
    out = np.empty(d1 + d2 - 1, dtype=np.uint64)
    L.check(L.lib.ronk_poly_mul(GP, GG, L.ptr(a), d1, L.ptr(b), d2, L.ptr(out)))
    if d1 * d2 <= 4_000_000:
        if not np.array_equal(out, orc.poly_mul(GP, a, b)): bad += 1; print("MUL MISMATCH", d1, d2)
    else:
        t = 0x1234567 % GP
        if orc.poly_eval(GP, out, t) != orc.mul(GP, orc.poly_eval(GP, a, t), orc.poly_eval(GP, b, t)): bad += 1; print("MUL MISMATCH", d1, d2)
mult = ob.multiples(1 << 10)
for it in range(12):
    n = rng.randrange(1, 5000)
    idx = [rng.randrange(len(mult)) for _ in range(n)]
    ks = [rng.choice([rng.randrange(2**256), rng.randrange(ob.R), rng.randrange(1 << 64), 0, 1]) for _ in range(n)]
    want = ob.mul(sum(k * (i + 1) for k, i in zip(ks, idx)) % ob.R, ob.G)
    if callers.msm_bn254([mult[i] for i in idx], ks) != want: bad += 1; print("MSM MISMATCH", n)
# division by a linear divisor and evaluate (device entry points): random lengths, roots, leading coefficients, moduli;
# the recurrence q[j-1] = c[j]/b1 + z q[j] at every j pins the whole quotient, the remainder is c(z)
import torch
for it in range(40):
    p = rng.choice([GP, GP, GP, 101, 2, 0xFFFFFFFFFFFFFFC5])
    d = rng.choice([rng.randrange(1, 64), rng.randrange(1, 5000), rng.randrange(1, 400000), 2048 * rng.randrange(1, 40) + rng.choice([-1, 0, 1])])
    a = splitmix_field(5000 + it, d, p)
    z = rng.choice([0, 1, p - 1, rng.randrange(p)]) % p
    b1 = rng.choice([1, 1, rng.randrange(1, p) if p > 2 else 1])
    b0 = orc.mul(p, orc.neg(p, z), b1)
    da = torch.from_numpy(a.view(np.int64)).cuda(); dq = torch.full((d,), -1, dtype=torch.int64, device="cuda")
    dr = torch.zeros(2, dtype=torch.int64, device="cuda")
    L.check(L.lib.ronk_poly_div_linear_dev(p, da.data_ptr(), d, b0, b1, dq.data_ptr(), dr.data_ptr(), 0))
    L.check(L.lib.ronk_poly_eval_dev(p, da.data_ptr(), d, z, dr.data_ptr() + 8, 0))
    torch.cuda.synchronize()
    q = dq.cpu().numpy().view(np.uint64); r = dr.cpu().numpy().view(np.uint64)
    val = orc.poly_eval(p, a, z); sc = orc.inverse(p, b1)
    ok = int(r[0]) == val and int(r[1]) == val and int(q[d - 1]) == 0
    if ok and d > 1:
        rhs = orc.vec_add(p, orc.vec_mul(p, a[1:], np.full(d - 1, sc, dtype=np.uint64)), orc.vec_mul(p, q[1:], np.full(d - 1, z, dtype=np.uint64)))
        ok = np.array_equal(q[:-1], rhs)
    if not ok: bad += 1; print("DIV / EVAL MISMATCH", p, d, z, b1)
print("fuzz done, mismatches:", bad)
sys.exit(1 if bad else 0)
