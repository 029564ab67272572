"""Throughput of the batched kernel against the batch size at the c3 shape (n=50, m=200): how much of a launch is its tail
(4096 workgroups over 256 CUs x 3 resident workgroups = 5.33 rounds).  Not the judged bench; prints a table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
load_package()
import jrl_qp_amd.capi as capi, jrl_qp_amd.problems as P
n, m = 50, 200
base = P.generate_batch(512, P.benchmark_characteristics(n, m))
for B in [int(a) for a in sys.argv[1:]] or [768, 1536, 3072, 3840, 4096, 4608, 6144, 8192, 16384]:
    rep = (B + base.B - 1) // base.B
    tile = lambda a: np.ascontiguousarray(np.concatenate([a] * rep)[:B])
    pb = P.PackedBatch(n=n, m=m, nb=0, G=tile(base.G), a=tile(base.a), C=tile(base.C), bl=tile(base.bl), bu=tile(base.bu),
                       xl=tile(base.xl), xu=tile(base.xu), x_star=tile(base.x_star), lam_star=tile(base.lam_star))
    dev = capi.to_device(pb)
    s = capi.BatchSolver(n, m, False, B)
    for _ in range(3):
        r = s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(15):
        e0.record(); r = s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"]); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    it = r.iters.cpu().numpy()
    print(f"B={B:6d}  rounds={B / 768:6.2f}  min {min(ts):7.3f} ms  median {np.median(ts):7.3f} ms  -> {B / min(ts) * 1e3:9.0f} solves/s   iters mean {it.mean():.1f} min {it.min()} max {it.max()} std {it.std():.1f}", flush=True)
