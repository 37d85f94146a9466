#!/usr/bin/env python3
"""Generate tests/golden/goldilocks_derived.json from first principles.

Pure-Python big-integer arithmetic, straight from the definitions the reference states
(natural-order DFT with omega = g^((p-1)/n), src/polynomial/mod.rs:240-258; schoolbook
product, src/polynomial/arithmetic.rs:97-119).  Uses NEITHER the oracle NOR the HIP
library, so it is an independent pin for both.  Deterministic (SplitMix64 stream).
"""
import json
import os

P = 0xFFFFFFFF00000001
G = 7
M64 = (1 << 64) - 1


def splitmix64(state):
    while True:
        state = (state + 0x9E3779B97F4A7C15) & M64
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        yield z ^ (z >> 31)


def field_stream(seed):
    for x in splitmix64(seed):
        if x < P:
            yield x


def take(gen, n):
    return [next(gen) for _ in range(n)]


def dft(x):
    n = len(x)
    w = pow(G, (P - 1) // n, P)
    return [sum(x[j] * pow(w, i * j, P) for j in range(n)) % P for i in range(n)]


def mul(a, b):
    c = [0] * (len(a) + len(b) - 1)
    for i, ai in enumerate(a):
        for j, bj in enumerate(b):
            c[i + j] = (c[i + j] + ai * bj) % P
    return c


def main():
    out = {"_comment": "derived (see make_goldilocks_vectors.py); p = 2^64-2^32+1, g = 7",
           "p": P, "g": G}
    out["roots"] = {str(k): pow(G, (P - 1) >> k, P) for k in (1, 2, 3, 6, 16, 22, 26, 32)}
    out["n_inverse"] = {str(k): pow(1 << k, P - 2, P) for k in (2, 16, 22)}
    out["dft_1234"] = dft([1, 2, 3, 4])
    gen = field_stream(0x5EED0000)
    dfts = []
    for n in (1, 2, 8, 64, 256):
        x = take(gen, n)
        dfts.append({"in": x, "out": dft(x)})
    out["dft_random"] = dfts
    # non power of two n | p-1 (the reference's dft() accepts any such n)
    odd = []
    for n in (3, 5, 15, 17, 96):
        x = take(gen, n)
        odd.append({"in": x, "out": dft(x)})
    out["dft_non_pow2"] = odd
    muls = []
    for d, d2 in ((1, 1), (4, 5), (17, 17), (33, 7)):
        a, b = take(gen, d), take(gen, d2)
        muls.append({"a": a, "b": b, "out": mul(a, b)})
    out["mul_random"] = muls
    edge = [0, 1, 2, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 1 << 63, (1 << 32) - 2]
    out["field_edge"] = {
        "values": edge,
        "add": [[(a + b) % P for b in edge] for a in edge],
        "sub": [[(a - b) % P for b in edge] for a in edge],
        "mul": [[(a * b) % P for b in edge] for a in edge],
        "inv": [pow(a, P - 2, P) if a else None for a in edge],
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "goldilocks_derived.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path)


if __name__ == "__main__":
    main()
