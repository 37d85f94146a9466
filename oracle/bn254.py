"""CPU oracle for the BN254 G1 multi-scalar multiplication (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).

Restates, on Python integers, what the reference computes for `kzg::commit` (src/kzg/setup.rs:48-60):

    g1_srs.into_iter().zip(coeffs).map(|(g1, coeff)| g1 * coeff).sum()

with `AffinePoint`'s group law (src/curve/mod.rs:183-211: Infinity is the identity, equal x and opposite y give
Infinity, the tangent slope is (3x^2 + a) / 2y, the chord slope (y2 - y1) / (x2 - x1)) and `Mul<ScalarField>` as k-fold
addition (src/curve/mod.rs:157-172; computed here by double-and-add, the same group element) -- for the curve
y^2 = x^3 + 3 over F_p with the 254-bit BN prime instead of the reference's GF(101^2) toy curve.

PINNED BY: the BN parametrisation (p, r from x = 4965661367192848881), r*G = Infinity, 2G = the EIP-196 test vector,
the curve equation for every produced point (tests/test_oracle_golden.py::test_bn254_oracle_pins).  The reference holds
no vector for a 254-bit curve, so these are first-principles / public-constant pins ("derived").
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

X_BN = 4965661367192848881
P = 36 * X_BN**4 + 36 * X_BN**3 + 24 * X_BN**2 + 6 * X_BN + 1
R = 36 * X_BN**4 + 36 * X_BN**3 + 18 * X_BN**2 + 6 * X_BN + 1
B = 3
G = (1, 2)
# EIP-196 / go-ethereum bn256 test vector: 2 * (1, 2)
TWO_G = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
         9918110051302171585080402603319702774565515993150576347155970296011118125764)


def on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return 0 <= x < P and 0 <= y < P and (y * y - x * x * x - B) % P == 0


def neg(pt):
    return None if pt is None else (pt[0], (-pt[1]) % P)


def add(p1, p2):
    """AffinePoint::add (src/curve/mod.rs:183-211), None = AffinePoint::Infinity"""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def mul(k, pt):
    """pt * k (src/curve/mod.rs:157-172 adds pt k times; same element)"""
    acc = None
    while k:
        if k & 1:
            acc = add(acc, pt)
        pt = add(pt, pt)
        k >>= 1
    return acc


def msm(points, scalars):
    """kzg::commit's fold (src/kzg/setup.rs:54-59)"""
    assert len(points) >= len(scalars)
    acc = None
    for pt, k in zip(points, scalars):
        acc = add(acc, mul(int(k), pt))
    return acc


def multiples(n, start=G):
    """start, 2*start, ..., n*start by repeated addition"""
    out, cur = [], start
    for _ in range(n):
        out.append(cur)
        cur = add(cur, start)
    return out


def fr_div_linear(coeffs, z):
    """`poly.div([-z, ONE])` over the scalar field F_r (src/kzg/setup.rs:70-74 through quotient_and_remainder,
    src/polynomial/mod.rs:170-225, with a monic linear divisor): the D-long quotient (top entry ZERO) and the remainder's
    constant term poly(z).  Synthetic division: q[j-1] = c[j] + z q[j]."""
    n = len(coeffs)
    q = [0] * n
    acc = 0
    for j in range(n - 1, 0, -1):
        acc = (int(coeffs[j]) + z * acc) % R
        q[j - 1] = acc
    return q, (int(coeffs[0]) + z * acc) % R if n else 0


def kzg_open(coeffs, z, srs):
    """`kzg::open` (src/kzg/setup.rs:63-78): commit(quotient of poly by (x - z), srs)"""
    assert len(srs) >= len(coeffs)
    q, value = fr_div_linear(coeffs, z % R)
    return msm(srs, q), value
