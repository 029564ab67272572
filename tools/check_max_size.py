"""One-off parity check of Mode A at its size limit (n = 4096 / 4095; the oracle needs a minute per problem, so not in the suite).
   python tools/check_max_size.py [n ...]      JRLQP_DENSE_SYNC=events|tags to force the hand-off mode"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from __graft_entry__ import load_package; load_package()
import jrl_qp_amd.capi as capi, jrl_qp_amd.problems as P
from oracle_lib import get_oracle
from test_gpu_dense import solve_gpu, check
sizes = [int(a) for a in sys.argv[1:]] or [4096, 4095]
ORACLE = os.environ.get("SKIP_ORACLE") is None
for n in sizes:
    m, ma = 96, 30
    ch = P.benchmark_characteristics(n, m, ma)
    t = time.time(); pb = P.generate_batch(1, ch, first_index=123); print("gen", n, round(time.time() - t, 1), flush=True)
    t = time.time()
    try:
        gpu = solve_gpu(capi, pb, 0, False, max_iter=80); print("gpu", n, round(time.time() - t, 2), gpu["term"], gpu["iters"], flush=True)
    except Exception as e:
        print("gpu", n, "FAILED", e, flush=True); continue
    if ORACLE:
        t = time.time(); ref = get_oracle().solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], max_iter=80); print("cpu", round(time.time() - t, 1), ref.term, ref.iters, flush=True)
        check(gpu, ref, f"n={n}")
        print("n =", n, "bit-identical", flush=True)
