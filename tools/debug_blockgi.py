"""Debug aid: find the first iteration at which the GPU structured GI solver and the oracle part ways."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from __graft_entry__ import load_package
load_package()
import jrl_qp_amd.capi as capi, jrl_qp_amd.structured_problems as SP
from oracle_lib import get_oracle
from test_gpu_blockgi import gpu_solve, oracle_solve
o = get_oracle()
kind = sys.argv[1] if len(sys.argv) > 1 else "arrow_up"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
qp = SP.random_structured_qp(kind, [3, 5, 2, 3], [3, 3, 3, 3], seed)
for mi in range(0, 12):
    g = gpu_solve(capi, [qp], max_iter=max(mi, 1) if mi else 1)
    if mi == 0:
        # max_iter >= 1 is required by the plan: compare the factor only
        Gs = qp.G.copy(order="F"); d, f = SP.block_views(Gs, qp.bs, qp.kind)
        r = o.blockgi_solve(qp.kind, d, f, qp.a, qp.Cblocks, qp.bl, qp.bu, max_iter=1)
        L = g["L"][0].T
        mask = np.tril(np.ones_like(Gs)) > 0
        print("factor equal:", np.array_equal(np.tril(L)[np.tril(Gs) != 0], np.tril(Gs)[np.tril(Gs) != 0]))
        continue
    r, _ = oracle_solve(o, SP, qp, max_iter=mi)
    print(mi, "term", g["term"][0], r.term, "iters", g["iters"][0], r.iters, "x eq", np.array_equal(g["x"][0], r.x), "dx %.2e" % np.abs(g["x"][0] - r.x).max(),
          "u eq", np.array_equal(g["u"][0], r.u), "f eq", g["f"][0] == r.f, "status eq", np.array_equal(g["status"][0], r.status))
