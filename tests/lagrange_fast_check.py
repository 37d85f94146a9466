"""Run by tests/test_gpu_parity.py::test_lagrange_evaluate_fast_path_small_sizes with RONK_LAGRANGE_FAST_MIN=1: the O(n)
root-of-unity form of Lagrange evaluate on sizes the oracle's O(n^2) restatement of the reference can check."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle as orc
    from conftest import splitmix_field
    from ronkathon_amd import _lib as L
    GP, GG = 0xFFFFFFFF00000001, 7
    for p, g, sizes in ((GP, GG, (1, 2, 4, 8, 96, 255, 1024)), (101, 2, (4, 5, 10, 20)), (17, 14, (4, 8, 16))):
        for n in sizes:
            nodes = np.empty(n, dtype=np.uint64)
            L.check(L.lib.ronk_lagrange_nodes(p, g, L.ptr(nodes), n))
            c = splitmix_field(n + 3, n, p)
            for x in (2 % p, int(nodes[n // 2]), int(splitmix_field(n, 1, p)[0])):      # a node among them: the value is 0
                got = L.out_scalar(L.lib.ronk_lagrange_eval, p, L.ptr(c), L.ptr(nodes), n, x)
                assert got == orc.lagrange_eval(p, c, nodes, x), (p, n, x)
    # not the powers of an order-n element: refused in this mode (the general formula is the default below 2^16 nodes)
    nodes = np.array([1, 5, 25, 124], dtype=np.uint64)
    import ctypes as C
    out = C.c_uint64(0)
    assert L.lib.ronk_lagrange_eval(GP, L.ptr(nodes), L.ptr(nodes), 4, 3, C.byref(out)) == -9
    print("lagrange fast check ok")


if __name__ == "__main__":
    main()
