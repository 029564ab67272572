"""BASELINE config c5: block-arrow LLT/solves, 100 stages x 30 (+30 head), batch of independent problems."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import load_package; load_package()
import jrl_qp_amd.capi as capi
from oracle_lib import get_oracle
from test_gpu_structured import spd, views, to_dev, rhs_dev
o = get_oracle()
rng = np.random.default_rng(5)
for kind, nblk in (("arrow_down", 101), ("tridiag", 100)):
    bs = [30] * nblk
    for nprob in (1, 64):
        Hs = [spd(rng, bs, kind) for _ in range(min(nprob, 4))]
        Hs = (Hs * ((nprob + len(Hs) - 1) // len(Hs)))[:nprob]
        T0 = to_dev(Hs)
        n = sum(bs)
        B = rhs_dev([np.asfortranarray(rng.standard_normal((n, 1))) for _ in range(nprob)])
        ts_f, ts_s = [], []
        for rep in range(4):
            T = T0.clone(); M = B.clone()
            plan = capi.StructuredBlocks(kind, bs, T, max_nrhs=1)
            torch.cuda.synchronize(); t = time.perf_counter(); ok = plan.llt(); torch.cuda.synchronize(); ts_f.append(time.perf_counter() - t)
            t = time.perf_counter(); plan.solve(M, False); plan.solve(M, True); torch.cuda.synchronize(); ts_s.append(time.perf_counter() - t)
        Hc = np.asfortranarray(Hs[0].copy()); D, S = views(Hc, bs, kind)
        t = time.perf_counter(); (o.block_arrow_llt(D, S, False) if kind == "arrow_down" else o.tri_block_diag_llt(D, S)); tc = time.perf_counter() - t
        print(f"{kind:10s} {nblk}x30  batch {nprob:3d}: GPU factor {min(ts_f)*1e6:8.0f} us  L+L^T solve {min(ts_s)*1e6:8.0f} us   (per problem {min(ts_f)/nprob*1e6:7.1f} / {min(ts_s)/nprob*1e6:7.1f} us);  CPU oracle factor {tc*1e6:.0f} us/problem; all ok {all(ok)}", flush=True)
