#!/usr/bin/env python3
"""Generate tests/golden/prime64_derived.json from first principles: the generic odd 64-bit primes the Montgomery tile path
is tested on (round 5), same definitions as make_goldilocks_vectors.py (natural-order DFT with omega = g^((p-1)/n),
src/polynomial/mod.rs:240-258; schoolbook product, src/polynomial/arithmetic.rs:97-119; Euclidean division,
src/polynomial/mod.rs:170-225 for divisors with a non-zero last coefficient), plain Python big-integer arithmetic.  Uses NEITHER
the oracle NOR the HIP library, so it pins both: the oracle in the CPU suite, the library (through the oracle and directly on
these small cases) in the GPU suite.  Deterministic (SplitMix64 stream).
"""
import json
import os

M64 = (1 << 64) - 1
# (p, g): g a quadratic non-residue (checked below; whether it generates all of F_p* is recorded per field)
PRIMES = [
    (0xFFFFFFFC00000001, 10),      # 2^64 - 2^34 + 1 = 2^34 * (2^30 - 1) + 1
    (0x3A00000000000001, 3),       # 29 * 2^57 + 1
    (0xC0000001, 5),               # 3 * 2^30 + 1
    (0xFFFFFFFF00000001, 343),     # Goldilocks with another generator (7^3)
    (0xFFFFFFFF00000001, 7),       # Goldilocks itself: the callers' vectors below (evaluate, Lagrange evaluate, Reed-Solomon)
]


def prime_factors(m):
    out = []
    f = 2
    while f * f <= m:
        if m % f == 0:
            out.append(f)
            while m % f == 0:
                m //= f
        f += 1 if f == 2 else 2
    if m > 1:
        out.append(m)
    return out


def splitmix64(state):
    while True:
        state = (state + 0x9E3779B97F4A7C15) & M64
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        yield z ^ (z >> 31)


def take(gen, n, p):
    out = []
    while len(out) < n:
        x = next(gen)
        if p < (1 << 63):
            x %= p            # (a bias of 2^-30 or less: these are test inputs, not samples)
            out.append(x)
        elif x < p:
            out.append(x)
    return out


def dft(x, p, g):
    n = len(x)
    w = pow(g, (p - 1) // n, p)
    return [sum(x[j] * pow(w, i * j, p) for j in range(n)) % p for i in range(n)]


def mul(a, b, p):
    c = [0] * (len(a) + len(b) - 1)
    for i, ai in enumerate(a):
        for j, bj in enumerate(b):
            c[i + j] = (c[i + j] + ai * bj) % p
    return c


def divrem(a, b, p):
    """both results with len(a) entries (the reference's D-long quotient and remainder); b[-1] != 0"""
    r = list(a)
    q = [0] * len(a)
    inv = pow(b[-1], p - 2, p)
    for k in range(len(a) - len(b), -1, -1):
        s = r[k + len(b) - 1] * inv % p
        q[k] = s
        for i, bi in enumerate(b):
            r[k + i] = (r[k + i] - s * bi) % p
    return q, r


def main():
    out = {"_comment": "derived (see make_prime64_vectors.py): generic odd 64-bit primes of the Montgomery tile path", "fields": []}
    for p, g in PRIMES:
        assert all(p % f for f in range(3, 1 << 16, 2)) and pow(2, p - 1, p) == 1, hex(p)      # (a sanity check, not a proof)
        factors = prime_factors(p - 1)
        # what the power-of-two transforms need of g: omega = g^((p-1)/2^k) of order exactly 2^k, i.e. g a quadratic non-residue
        # (7^3 over Goldilocks is one without generating the whole group: 3 | p - 1)
        assert pow(g, (p - 1) // 2, p) == p - 1, (hex(p), g, "g is a square")
        full = all(pow(g, (p - 1) // f, p) != 1 for f in factors)
        two = (p - 1 & -(p - 1)).bit_length() - 1
        gen = splitmix64(0x5EED1000 + (p & 0xFFFF))
        e = {"p": p, "g": g, "two_adicity": two, "p_minus_1_prime_factors": factors, "g_generates_whole_group": full}
        e["roots"] = {str(k): pow(g, (p - 1) >> k, p) for k in (1, 2, 4, 16, min(22, two), two)}
        e["n_inverse"] = {str(k): pow(1 << k, p - 2, p) for k in (2, 16, 22)}
        e["dft_1234"] = dft([1, 2, 3, 4], p, g)
        e["dft_random"] = []
        for n in (1, 2, 8, 64, 256):
            x = take(gen, n, p)
            e["dft_random"].append({"in": x, "out": dft(x, p, g)})
        e["mul_random"] = []
        for d, d2 in ((1, 1), (4, 5), (17, 17), (33, 7)):
            a, b = take(gen, d, p), take(gen, d2, p)
            e["mul_random"].append({"a": a, "b": b, "out": mul(a, b, p)})
        e["divrem_random"] = []
        for d, d2 in ((5, 2), (9, 4), (33, 33), (40, 7)):
            a, b = take(gen, d, p), take(gen, d2, p)
            if b[-1] == 0:
                b[-1] = 1
            q, r = divrem(a, b, p)
            e["divrem_random"].append({"a": a, "b": b, "quot": q, "rem": r})
        # the callers on the same fields, from their definitions: evaluate = sum c_i x^i (src/polynomial/mod.rs:133-139);
        # Lagrange evaluate = the value at x of THE polynomial of degree < n through (nodes[j], c[j]) (mod.rs:382-415 computes it
        # barycentrically; off the nodes the value is the same field element), here by solving for it with the inverse DFT;
        # Message::encode::<N> = values at omega_N^i of the zero-padded message (src/codes/reed_solomon.rs:42-52), decode = the
        # message back from its first K coordinates (:54-106)
        e["evaluate"] = []
        for d in (1, 2, 9, 40):
            c = take(gen, d, p)
            for x in (0, 1, p - 1, take(gen, 1, p)[0]):
                e["evaluate"].append({"c": c, "x": x, "out": sum(ci * pow(x, i, p) for i, ci in enumerate(c)) % p})
        e["lagrange_evaluate"] = []
        for n in (2, 4, 16):
            w = pow(g, (p - 1) // n, p)
            nodes = [pow(w, i, p) for i in range(n)]
            vals = take(gen, n, p)
            # coefficients of the interpolant: c = (1/n) * DFT with omega^-1 of the values
            winv, ninv = pow(w, p - 2, p), pow(n, p - 2, p)
            coef = [sum(vals[j] * pow(winv, i * j, p) for j in range(n)) * ninv % p for i in range(n)]
            for x in take(gen, 3, p):
                if x in nodes:
                    continue
                e["lagrange_evaluate"].append({"values": vals, "nodes": nodes, "x": x,
                                               "out": sum(ci * pow(x, i, p) for i, ci in enumerate(coef)) % p})
        e["reed_solomon"] = []
        for k, n in ((1, 2), (3, 8), (5, 8), (16, 64)):
            msg = take(gen, k, p)
            w = pow(g, (p - 1) // n, p)
            xs = [pow(w, i, p) for i in range(n)]
            ys = [sum(mi * pow(x, i, p) for i, mi in enumerate(msg)) % p for x in xs]
            e["reed_solomon"].append({"msg": msg, "n": n, "xs": xs, "ys": ys})
        edge = sorted({0, 1, 2, p - 1, p - 2, p >> 1, (p >> 1) + 1, 0xFFFFFFFF % p, 0x100000000 % p, (1 << 63) % p})
        e["field_edge"] = {
            "values": edge,
            "add": [[(a + b) % p for b in edge] for a in edge],
            "sub": [[(a - b) % p for b in edge] for a in edge],
            "mul": [[(a * b) % p for b in edge] for a in edge],
            "inv": [pow(a, p - 2, p) if a else None for a in edge],
        }
        out["fields"].append(e)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "prime64_derived.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path)


if __name__ == "__main__":
    main()
