"""Timing + phase profile of the large-n batched kernel (gi_batch_big.hip): python tools/time_big.py [B] [n] [m]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
load_package()
import jrl_qp_amd.capi as capi, jrl_qp_amd.problems as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n, m = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100, 200)
base = P.generate_batch(min(B, 256), P.benchmark_characteristics(n, m))
rep = (B + base.B - 1) // base.B
tile = lambda a: np.ascontiguousarray(np.concatenate([a] * rep)[:B])
pb = P.PackedBatch(n=base.n, m=base.m, nb=0, G=tile(base.G), a=tile(base.a), C=tile(base.C), bl=tile(base.bl), bu=tile(base.bu), xl=tile(base.xl),
                   xu=tile(base.xu), x_star=tile(base.x_star), lam_star=tile(base.lam_star))
dev = capi.to_device(pb)
s = capi.BatchSolver(n, m, False, B)
for _ in range(2):
    r = s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(8):
    e0.record(); r = s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"]); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"B={B} n={n} m={m}: {min(ts):.3f} ms -> {B / min(ts) * 1e3:.0f} solves/s; mean iters {r.iters.double().mean().item():.2f}")
prof = torch.zeros((B, 32), dtype=torch.int64, device="cuda")
capi.set_profile(capi.PROFILE_GI_BATCH, prof)
s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"]); torch.cuda.synchronize()
capi.set_profile(capi.PROFILE_GI_BATCH, None)
pr = prof.cpu().numpy().astype(np.float64)
names = ["load", "LLT", "x0 / J", "scan + selection", "d = J^T n+", "z || R^-1 d || chain", "scalar products", "step", "addition (sweep)", "removal"]
tot = pr[:, :10].sum(axis=1).mean()
print(f"cycles per solve {tot:.0f}; additions {pr[:, 10].mean():.1f}, removals {pr[:, 11].mean():.1f}")
for i, nm in enumerate(names):
    print(f"  {nm:22s} {pr[:, i].mean():12.0f} ({100 * pr[:, i].mean() / tot:5.1f} %)")
