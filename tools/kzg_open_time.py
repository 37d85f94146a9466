#!/usr/bin/env python3
"""Developer probe: kzg::open over BN254 on device-resident data -- the division over the scalar field alone and the whole
opening (division + MSM of the quotient), per call, for n = 2^16 .. 2^20.  usage: python tools/kzg_open_time.py [log2n ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bn254 as ob  # noqa: E402  (points of the SRS: multiples of G, outside every timed region)
from ronkathon_amd import _lib as L  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16, 18, 20]
    m = 1 << 11
    mult = ob.multiples(m)
    m64 = (1 << 64) - 1
    pw1 = np.array([[(pt[0] >> (64 * j)) & m64 for j in range(4)] + [(pt[1] >> (64 * j)) & m64 for j in range(4)] for pt in mult], dtype=np.uint64)
    rng = np.random.default_rng(9)
    for lg in sizes:
        n = 1 << lg
        srs = torch.from_numpy(np.tile(pw1, (n // m, 1)).view(np.int64)).cuda()
        cw = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
        cw[:, 3] &= np.uint64((1 << 61) - 1)
        dc = torch.from_numpy(cw.view(np.int64)).cuda()
        dq = torch.empty_like(dc)
        drem = torch.zeros(4, dtype=torch.int64, device="cuda")
        z = np.array([0x123456789ABCDEF, 0x1111, 0x2222, 0x3], dtype=np.uint64)
        out = np.zeros(8, dtype=np.uint64); val = np.zeros(4, dtype=np.uint64)

        def div():
            L.check(L.lib.ronk_poly_div_linear_bn254_dev(dc.data_ptr(), n, L.ptr(z), dq.data_ptr(), drem.data_ptr(), 0))

        def full():
            L.check(L.lib.ronk_kzg_open_bn254_dev(dc.data_ptr(), n, L.ptr(z), srs.data_ptr(), dq.data_ptr(), L.ptr(out), L.ptr(val), 0))
        for f, name, reps in ((div, "division over F_r", 20), (full, "kzg::open (division + MSM)", 5)):
            f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
            torch.cuda.synchronize()
            print("n = 2^%d  %-28s %.3f ms per call" % (lg, name, (time.perf_counter() - t0) / reps * 1e3), flush=True)


if __name__ == "__main__":
    main()
