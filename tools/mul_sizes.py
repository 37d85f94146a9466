#!/usr/bin/env python3
"""Developer probe: device time of ronk_poly_mul_dev for a ladder of NTT sizes (two operands of N/2 coefficients each),
HIP events over `iters` back-to-back products, after verifying the product against three oracle transforms at N <= 2^21.
usage: python tools/mul_sizes.py [log2N ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ronkathon_amd import _lib as L  # noqa: E402

P, G = 0xFFFFFFFF00000001, 7


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16, 17, 18, 19, 20, 21, 22, 23]
    rng = np.random.default_rng(5)
    for k in sizes:
        N = 1 << k
        d = N // 2
        a = torch.from_numpy((rng.integers(0, 2**63, size=d, dtype=np.uint64) % np.uint64(P)).view(np.int64)).cuda()
        b = torch.from_numpy((rng.integers(0, 2**63, size=d, dtype=np.uint64) % np.uint64(P)).view(np.int64)).cuda()
        out = torch.empty(2 * d - 1, dtype=torch.int64, device="cuda")

        def one():
            L.check(L.lib.ronk_poly_mul_dev(P, G, a.data_ptr(), d, b.data_ptr(), d, out.data_ptr(), 0))
        one(); torch.cuda.synchronize()
        if k <= 21:
            import oracle as orc
            pa = np.zeros(N, dtype=np.uint64); pa[:d] = a.cpu().numpy().view(np.uint64)
            pb = np.zeros(N, dtype=np.uint64); pb[:d] = b.cpu().numpy().view(np.uint64)
            want = orc.ifft(P, G, orc.vec_mul(P, orc.fft(P, G, pa), orc.fft(P, G, pb)))
            assert np.array_equal(out.cpu().numpy().view(np.uint64), want[:2 * d - 1]), k
        for _ in range(5):
            one()
        iters = 50
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                one()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters * 1e3)
        print("mul NTT size 2^%d: %.2f us per product (min %.2f), frac of 48 N bytes / 8 TB/s = %.3f" % (
            k, float(np.median(ts)), min(ts), 48.0 * N / (float(np.median(ts)) * 1e-6) / 8e12), flush=True)


if __name__ == "__main__":
    main()
