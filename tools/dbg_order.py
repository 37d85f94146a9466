import sys, os; sys.path.insert(0,'.')
mode = sys.argv[1]
def torch_init():
    import torch
    try:
        t = torch.zeros(1).cuda(); torch.cuda.synchronize(); return "torch ok"
    except Exception as e:
        return "torch FAIL %r" % (str(e)[:60],)
def lib_count():
    import ronkathon_amd as R
    return "lib count %d" % R.device_count()
def lib_work():
    import numpy as np
    import ronkathon_amd as R
    from ronkathon_amd import _lib as L
    a = L.arr([1,2,3,4]); 
    return "lib eval %d" % L.out_scalar(L.lib.ronk_poly_eval, 101, L.ptr(a), 4, 2)
steps = {"A": [lib_count, torch_init], "B": [torch_init, lib_count, lib_work], "C": [lib_count, lib_work, torch_init, lib_work],
         "D": [lambda: __import__("torch") and "import torch", lib_count, torch_init, lib_work]}[mode]
print(mode, [f() for f in steps])
