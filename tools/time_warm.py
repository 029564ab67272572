"""Timing probe of the warm-started batched kernel (not the judged bench): c3 shape, B QPs, HIP-event timing.
Guesses: the optimal active set (a control tick re-solving the same QP: 0 iterations), a stale one (some entries
dropped / flipped / added), and none (equalities only => the cold iteration count)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
import jrl_qp_amd.capi as capi, jrl_qp_amd.problems as P
from test_gpu_warm import perturbed_guess
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n, m = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (50, 200)
base = P.generate_batch(min(B, 512), P.benchmark_characteristics(n, m))
rep = (B + base.B - 1) // base.B
tile = lambda a: np.ascontiguousarray(np.concatenate([a] * rep)[:B])
pb = P.PackedBatch(n=base.n, m=base.m, nb=0, G=tile(base.G), a=tile(base.a), C=tile(base.C), bl=tile(base.bl), bu=tile(base.bu),
                   xl=tile(base.xl), xu=tile(base.xu), x_star=tile(base.x_star), lam_star=tile(base.lam_star))
dev = capi.to_device(pb)
s = capi.BatchSolver(n, m, False, B)
args = (dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"])
def timed(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts), r
ms, r = timed(lambda: s.solve(*args))
cold_status = r.status.clone()
x_cold = r.x.clone()
print(f"cold          : {ms:.3f} ms -> {B / ms * 1e3:9.0f} solves/s  mean iters {r.iters.double().mean().item():.2f}")
stale = torch.from_numpy(perturbed_guess(np.random.default_rng(1), cold_status.cpu().numpy(), m, flip=0.2, add=0.02)).cuda()
for name, g in (("exact guess", cold_status), ("stale guess", stale), ("no guess", None)):
    ms, r = timed(lambda: s.solve_warm(*args, guess=g))
    err = (r.x - x_cold).abs().max().item()
    print(f"warm, {name:11s}: {ms:.3f} ms -> {B / ms * 1e3:9.0f} solves/s  mean iters {r.iters.double().mean().item():.2f}  ok {int((r.term == 0).sum())}/{B}  max|x - x_cold| {err:.2e}")
