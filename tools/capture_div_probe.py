import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle as orc
from conftest import splitmix_field
from ronkathon_amd import _lib as L
GP = 0xFFFFFFFF00000001
for d, d2 in ((2300, 100), (700, 33)):
    a, b = splitmix_field(771, d), splitmix_field(772, d2)
    da = torch.from_numpy(a.view(np.int64)).cuda(); db = torch.from_numpy(b.view(np.int64)).cuda()
    dq = torch.zeros(d, dtype=torch.int64, device="cuda"); dr = torch.zeros(d, dtype=torch.int64, device="cuda")
    status = torch.zeros(8, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    oq, o_r = orc.poly_divrem(GP, a, b)
    rc = L.lib.ronk_poly_divrem_dev(GP, da.data_ptr(), d, db.data_ptr(), d2, dq.data_ptr(), dr.data_ptr(), status.data_ptr(), s.cuda_stream)
    s.synchronize()
    print(d, d2, "direct rc", rc, "status", status.tolist(), "q ok", np.array_equal(dq.cpu().numpy().view(np.uint64), oq), "r ok", np.array_equal(dr.cpu().numpy().view(np.uint64), o_r))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        st = torch.cuda.current_stream().cuda_stream
        rc = L.lib.ronk_poly_divrem_dev(GP, da.data_ptr(), d, db.data_ptr(), d2, dq.data_ptr(), dr.data_ptr(), status.data_ptr(), st)
    print("capture rc", rc)
    for rep in range(3):
        dq.fill_(-1); dr.fill_(-1); status.fill_(9)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        print(" replay", rep, "status", [hex(v & 0xFFFFFFFF) for v in status.tolist()], "q ok", np.array_equal(dq.cpu().numpy().view(np.uint64), oq),
              "r ok", np.array_equal(dr.cpu().numpy().view(np.uint64), o_r), "q[:3]", [hex(v) for v in dq.cpu().numpy().view(np.uint64)[:3]])
    del g
