"""time of Message::decode (device-resident) for geometric node sequences: the O(K log K) form"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ronkathon_amd as R
from ronkathon_amd import _lib as L
GP, GG = R.GOLDILOCKS_P, R.GOLDILOCKS_G
for lk in (12, 14, 16, 18, 20):
    k = 1 << lk; N = 2 * k
    nodes = np.empty(N, dtype=np.uint64)
    L.check(L.lib.ronk_lagrange_nodes(GP, GG, L.ptr(nodes), N))
    ys = np.random.default_rng(lk).integers(0, 2**62, size=k, dtype=np.uint64)
    dx = torch.from_numpy(nodes[:k].copy().view(np.int64)).cuda(); dy = torch.from_numpy(ys.view(np.int64)).cuda()
    do = torch.empty(k, dtype=torch.int64, device="cuda"); st = torch.zeros(2, dtype=torch.int32, device="cuda")
    for _ in range(2):
        L.check(L.lib.ronk_rs_decode_dev(GP, dx.data_ptr(), dy.data_ptr(), k, do.data_ptr(), st.data_ptr(), 0))
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        L.check(L.lib.ronk_rs_decode_dev(GP, dx.data_ptr(), dy.data_ptr(), k, do.data_ptr(), st.data_ptr(), 0))
    torch.cuda.synchronize()
    print("rs decode K = 2^%d: %.3f ms" % (lk, (time.perf_counter() - t0) / reps * 1e3), "status", st.cpu().tolist())
