"""Run by tests/test_gpu_parity.py::test_rs_decode_fast_path_small_sizes with RONK_RS_FAST_MIN=2: the O(K log K) decode for
geometric node sequences (x_j = q^j) on sizes the oracle's restatement of Message::decode (src/codes/reed_solomon.rs:54-106)
can check; other node sets must take the general kernels."""
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
    for N, ks in ((8, (2, 3, 5, 7)), (96, (2, 17, 64, 95)), (1024, (3, 100, 511, 1000)), (4096, (1500,))):
        nodes = orc.lagrange_nodes(GP, GG, N)                       # omega_N^j
        for k in ks:
            xs = np.ascontiguousarray(nodes[:k])
            for ys in (splitmix_field(N + k, k), orc.dft(GP, GG, np.concatenate([splitmix_field(k, k), np.zeros(N - k, dtype=np.uint64)]))[:k]):
                out = np.zeros(k, dtype=np.uint64)
                L.check(L.lib.ronk_rs_decode(GP, L.ptr(xs), L.ptr(ys), k, L.ptr(out)))
                assert np.array_equal(out, orc.rs_decode(GP, xs, ys, k)), (N, k)
    # any other node set of the same size: the device must pick the general kernels
    for k in (5, 64, 300):
        xs = np.unique(splitmix_field(900 + k, k + 20))[:k]
        ys = splitmix_field(901 + k, k)
        out = np.zeros(k, dtype=np.uint64)
        L.check(L.lib.ronk_rs_decode(GP, L.ptr(xs), L.ptr(ys), k, L.ptr(out)))
        assert np.array_equal(out, orc.rs_decode(GP, xs, ys, k)), k
        geo = orc.lagrange_nodes(GP, GG, 1024)[:k].copy(); geo[[1, 3]] = geo[[3, 1]]          # geometric set, wrong order
        L.check(L.lib.ronk_rs_decode(GP, L.ptr(geo), L.ptr(ys), k, L.ptr(out)))
        assert np.array_equal(out, orc.rs_decode(GP, geo, ys, k)), k
    print("rs fast check ok")


if __name__ == "__main__":
    main()
