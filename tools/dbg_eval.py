import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import ronkathon_amd as R, oracle as orc
from ronkathon_amd import _lib as L
from conftest import splitmix_field
GP=0xFFFFFFFF00000001
bad=0; n=0
for it in range(1500):
  for p in (101, GP):
    d = 2 + (it*7) % 40
    a = splitmix_field(it+1, d, p)
    b0 = int(splitmix_field(it+5000,1,p)[0]); b1 = int(splitmix_field(it+9000,1,p)[0]) or 1
    q = np.empty(d,dtype=np.uint64); r=np.empty(d,dtype=np.uint64)
    L.check(L.lib.ronk_poly_divrem(p, L.ptr(a), d, L.ptr(L.arr([b0,b1])), 2, L.ptr(q), L.ptr(r)))
    z = orc.mul(p, orc.sub(p,0,b0), orc.inverse(p,b1))
    want = orc.poly_eval(p,a,z)
    got2 = L.out_scalar(L.lib.ronk_poly_eval,p,L.ptr(a),d,z)
    n+=1
    if int(r[0])!=want or got2!=want:
        bad+=1
        if bad<6: print("MISMATCH it",it,p,d,int(r[0]),got2,want)
print("calls",n,"bad",bad)
