"""host-pointer entry points (what a Rust caller's `poly.fft()` goes through): time per call incl. both PCIe copies"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ronkathon_amd as R
from ronkathon_amd import _lib as L
P, G = R.GOLDILOCKS_P, R.GOLDILOCKS_G
for k in (16, 20, 22, 24):
    n = 1 << k
    x = (np.random.default_rng(k).integers(0, 2**62, size=n, dtype=np.uint64))
    y = np.empty_like(x)
    for _ in range(3):
        L.check(L.lib.ronk_fft(P, G, L.ptr(x), L.ptr(y), None, n))
    reps = 20 if k <= 22 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        L.check(L.lib.ronk_fft(P, G, L.ptr(x), L.ptr(y), None, n))
    dt = (time.perf_counter() - t0) / reps
    print("ronk_fft 2^%d: %.3f ms per call, %.2f GB/s of host traffic (16 B/coefficient)" % (k, dt * 1e3, 16.0 * n / dt / 1e9))
