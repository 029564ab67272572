import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, ctypes as ct
from __graft_entry__ import load_package; load_package()
import jrl_qp_amd.capi as capi, jrl_qp_amd.problems as P
from oracle_lib import get_oracle
o = get_oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pb = P.generate_batch(1, P.Characteristics(nVar=n, nObj=n, nIneq=2 * n, nStrongActIneq=n // 4), first_index=930)
ret, Lref = o.llt(pb.G[0])
lib = capi.load_library()
lib.jrlqp_dense_bind.restype = ct.c_int
lib.jrlqp_dense_bind.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_void_p, ct.c_int] + [ct.c_void_p] * 4
lib.jrlqp_dense_init.restype = ct.c_int
lib.jrlqp_dense_init.argtypes = [ct.c_void_p, ct.POINTER(ct.c_int), ct.POINTER(ct.c_double), ct.c_void_p]
s = capi.DenseSolver(n, 2 * n, False)
G = torch.from_numpy(np.ascontiguousarray(pb.G[0])).cuda(); a = torch.from_numpy(pb.a[0].copy()).cuda()
C = torch.from_numpy(np.ascontiguousarray(pb.C[0])).cuda(); bl = torch.from_numpy(pb.bl[0].copy()).cuda(); bu = torch.from_numpy(pb.bu[0].copy()).cuda()
p = lambda t: ct.c_void_p(t.data_ptr())
print("bind", lib.jrlqp_dense_bind(s.h, p(G), n, p(a), p(C), n, p(bl), p(bu), None, None))
info = ct.c_int(-7); f0 = ct.c_double(0)
print("init", lib.jrlqp_dense_init(s.h, ct.byref(info), ct.byref(f0), None), "info", info.value, "oracle ret", ret)
torch.cuda.synchronize()
L = G.cpu().numpy().T
d = np.tril(L) != np.tril(Lref)
print("mismatches in lower triangle:", int(d.sum()))
if d.any():
    cols = np.nonzero(d.any(axis=0))[0]; print("first bad columns", cols[:10])
    c = cols[0]; rows = np.nonzero(d[:, c])[0]; print("col", c, "bad rows", rows[:20], "max abs diff", np.abs(L - Lref)[rows, c].max())
    print("gpu", L[rows[:4], c], "ref", Lref[rows[:4], c])
