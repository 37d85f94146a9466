"""hipGraph capture of the _dev entry points (torch.cuda.CUDAGraph on a torch stream): correctness + replay time."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ronkathon_amd as R
from ronkathon_amd import _lib as L
from conftest import splitmix_field
P, G = R.GOLDILOCKS_P, R.GOLDILOCKS_G
for log2n, reps in ((16, 2000), (12, 2000), (22, 300)):
    n = 1 << log2n
    x = torch.from_numpy(splitmix_field(7, n).view(np.int64)).cuda()
    y = torch.empty_like(x); z = torch.empty_like(x)
    plan = L.Plan(P, G, log2n, 1, 0)
    s = torch.cuda.Stream()
    def body(st):
        plan.forward_dev(x.data_ptr(), y.data_ptr(), st)
        plan.inverse_dev(y.data_ptr(), z.data_ptr(), st)
    with torch.cuda.stream(s):
        for _ in range(20): body(s.cuda_stream)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): body(s.cuda_stream)
        s.synchronize()
        direct = (time.perf_counter() - t0) / reps * 1e6
    g = torch.cuda.CUDAGraph()
    z.zero_()
    with torch.cuda.graph(g, stream=s):
        body(torch.cuda.current_stream().cuda_stream)
    g.replay(); torch.cuda.synchronize()
    ok = bool(torch.equal(z, x))
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / reps * 1e6
    # 8 round trips per graph
    g8 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g8, stream=s):
        for _ in range(8): body(torch.cuda.current_stream().cuda_stream)
    g8.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps // 8): g8.replay()
    torch.cuda.synchronize()
    graph8 = (time.perf_counter() - t0) / (reps // 8) / 8 * 1e6
    print("2^%d fwd+inv: direct %.1f us, graph %.1f us, graph(8 per replay) %.1f us, roundtrip ok %s" % (log2n, direct, graph, graph8, ok))
