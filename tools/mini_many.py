#!/usr/bin/env python3
"""mini_many.py [iters] [arrays] [log2n] -- the two-lane throughput regime with nothing around it (for profilers): ONE plan handle
with in_flight = 2, `arrays` device arrays (HBM-cold rotation when arrays * 32 MiB * 2 > 256 MiB) through ronk_ntt_forward_many_dev,
`iters` calls.  Prints the wall time per transform.  Developer tool."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ronkathon_amd import _lib as L

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
k = int(sys.argv[3]) if len(sys.argv) > 3 else 22
lanes = int(os.environ.get("MINI_LANES", "2"))
n = 1 << k
P, G = L.GOLDILOCKS_P, L.GOLDILOCKS_G
plan = L.Plan(P, G, k, 1, in_flight=lanes)
xs = [torch.randint(0, 2**62, (n,), dtype=torch.int64, device="cuda") for _ in range(K)]
ys = [torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(K)]
xp, yp = [x.data_ptr() for x in xs], [y.data_ptr() for y in ys]
plan.forward_many_dev(xp, yp, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    plan.forward_many_dev(xp, yp, 0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("mini_many: %d x %d transforms of 2^%d, %d lanes: %.2f us per transform" % (iters, K, k, lanes, dt / (iters * K) * 1e6))
