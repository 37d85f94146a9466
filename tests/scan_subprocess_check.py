"""Run by tests/test_gpu_parity.py::test_scan_onepass_variants in a fresh process (the library reads its RONK_* knobs
once): evaluate and division by a linear factor through the device API against the oracle, repeated calls with
different sizes (the one-launch forms alternate between two look-back arrays that the PREVIOUS call cleared)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import oracle as orc
    from ronkathon_amd import _lib as L
    from conftest import splitmix_field
    GP = 0xFFFFFFFF00000001

    torch.zeros(1).cuda()
    side = torch.cuda.Stream()
    sizes = [(1 << 20) + 77, 5, 2048, 2049, (1 << 22), 40000, (1 << 23) - 3, 4096 * 3 + 5, 1 << 21]
    for p in (GP, 101):
        for rep, d in enumerate(sizes if p == GP else sizes[:4] + [70001]):
            a = splitmix_field(100 + rep, d, p)
            da = torch.from_numpy(a.view(np.int64)).cuda()
            dq = torch.full((d,), -1, dtype=torch.int64, device="cuda")
            dr = torch.zeros(2, dtype=torch.int64, device="cuda")
            z = int(splitmix_field(7 + rep, 1, p)[0])
            st = side.cuda_stream if rep % 2 else 0
            torch.cuda.synchronize()
            L.check(L.lib.ronk_poly_div_linear_dev(p, da.data_ptr(), d, (p - z) % p, 1, dq.data_ptr(), dr.data_ptr(), st))
            L.check(L.lib.ronk_poly_eval_dev(p, da.data_ptr(), d, z, dr.data_ptr() + 8, st))
            torch.cuda.synchronize()
            q = dq.cpu().numpy().view(np.uint64)
            r = dr.cpu().numpy().view(np.uint64)
            val = orc.poly_eval(p, a, z)
            assert int(r[0]) == val and int(r[1]) == val, (p, d, "remainder / evaluate")
            assert int(q[d - 1]) == 0
            t = 0xFEEDFACE12345 % p
            assert orc.poly_eval(p, a, t) == orc.add(p, orc.mul(p, orc.poly_eval(p, q, t), orc.sub(p, t, z)), val), (p, d)
            if d <= 5000:
                assert np.array_equal(q, orc.kzg_open_quotient(p, a, z)), (p, d)
            # the recurrence q[j-1] = c[j] + z q[j] at EVERY j (with q[d-1] = 0 above it pins the whole quotient)
            if d > 1:
                assert np.array_equal(q[:-1], orc.vec_add(p, a[1:], orc.vec_mul(p, q[1:], np.full(d - 1, z, dtype=np.uint64)))), (p, d)
            # a non-monic divisor b0 + b1 x (quotient scaled by 1/b1), operands 8 bytes off a 16-byte boundary (the
            # 16-byte access form must not be chosen), and the quotient written over the dividend
            if rep < 6:
                b1 = 5 % p or 1
                b0 = orc.mul(p, orc.neg(p, z), b1)
                db = torch.zeros(d + 3, dtype=torch.int64, device="cuda")
                off = 1 if (db.data_ptr() % 16 == 0) else 0                      # -> address = 8 mod 16
                db[off:off + d] = da
                torch.cuda.synchronize()
                L.check(L.lib.ronk_poly_div_linear_dev(p, db.data_ptr() + 8 * off, d, b0, b1, db.data_ptr() + 8 * off, dr.data_ptr(), st))
                torch.cuda.synchronize()
                q2 = db.cpu().numpy().view(np.uint64)[off:off + d]
                assert np.array_equal(q2, orc.vec_mul(p, q, np.full(d, orc.inverse(p, b1), dtype=np.uint64))), (p, d, "scaled, unaligned, in place")
                assert int(dr.cpu().numpy().view(np.uint64)[0]) == val
                assert not db.cpu().numpy()[off + d:].any() and (off == 0 or int(db[0]) == 0), "wrote outside the operand"
                # the same divisor, aligned and out of place (the one-launch form up to 2^23 coefficients), twice in a row on the
                # same stream (the two look-back arrays alternate)
                for _ in range(2):
                    dq.fill_(-1)
                    L.check(L.lib.ronk_poly_div_linear_dev(p, da.data_ptr(), d, b0, b1, dq.data_ptr(), dr.data_ptr(), st))
                    torch.cuda.synchronize()
                    assert np.array_equal(dq.cpu().numpy().view(np.uint64), q2), (p, d, "scaled, aligned, out of place")
                    assert int(dr.cpu().numpy().view(np.uint64)[0]) == val
    print("scan check ok")


if __name__ == "__main__":
    main()
