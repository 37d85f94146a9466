#!/usr/bin/env python3
"""Developer probe: device time per call of the callers that ride on the NTT path -- general division (Newton inversion on the
NTT path, reached through ronk_poly_divrem_dev for large Goldilocks operands) and the O(K log K) Reed-Solomon decode (ronk_rs_decode_dev) -- device-resident data.
usage: python tools/callers_time.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle as orc  # noqa: E402
from conftest import splitmix_field  # noqa: E402
from ronkathon_amd import _lib as L  # noqa: E402

P, G = 0xFFFFFFFF00000001, 7


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def timed(f, reps):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    for lg, lg2 in ((20, 19), (22, 21), (22, 12)):
        d, d2 = 1 << lg, (1 << lg2) + 1
        a, b = splitmix_field(lg, d), splitmix_field(lg2 + 50, d2)
        b[-1] = 1
        da, db = dev(a), dev(b)
        dq, dr = torch.empty_like(da), torch.empty_like(da)
        st = torch.zeros(2, dtype=torch.int32, device="cuda")

        def f():
            L.check(L.lib.ronk_poly_divrem_dev(P, da.data_ptr(), d, db.data_ptr(), d2, dq.data_ptr(), dr.data_ptr(), st.data_ptr(), 0))
        print("divrem 2^%d by degree 2^%d (Newton on the NTT path): %.3f ms per call" % (lg, lg2, timed(f, 5)), flush=True)
    for lk in (12, 16, 20):
        K = 1 << lk
        N = 2 * K
        xs = orc.lagrange_nodes(P, G, N)[:K].copy()
        ys = splitmix_field(lk + 7, K)
        dx, dy = dev(xs), dev(ys)
        do = torch.empty_like(dx)
        st = torch.zeros(2, dtype=torch.int32, device="cuda")

        def g():
            L.check(L.lib.ronk_rs_decode_dev(P, dx.data_ptr(), dy.data_ptr(), K, do.data_ptr(), st.data_ptr(), 0))
        print("rs decode K = 2^%d (geometric nodes, O(K log K)): %.3f ms per call" % (lk, timed(g, 5)), flush=True)


if __name__ == "__main__":
    main()
