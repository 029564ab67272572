"""Timing + phase profile of the large-n batched kernel on a row of the reference's benchmark families (tools/bench_families.py):
python tools/time_big_family.py [test1|test3] [n] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
load_package()
import jrl_qp_amd.capi as capi, jrl_qp_amd.problems as P
fam = sys.argv[1] if len(sys.argv) > 1 else "test3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
warm = fam.startswith("warm")  # warm1 / warm3: the family re-solved from its own optimal active set (0 iterations: the warm init alone); needs a build with -DJRLQP_BIG_WARM_PROF for the phases
if warm:
    fam = "test" + fam[4:]
if fam == "testF":  # (the warm family's shape: 10 n inequalities and a box -- too many for the register-resident instantiations)
    ch = P.Characteristics(nVar=n, nObj=n, nIneq=10 * n, nStrongActIneq=int(0.35 * n), bounds=True, nStrongActBounds=int(0.1 * n), doubleSidedIneq=True)
elif fam == "test1":
    ch = P.Characteristics(nVar=n, nObj=n, nEq=int(0.4 * n))
else:
    ch = P.Characteristics(nVar=n, nObj=n, nEq=int(0.2 * n), nIneq=n, nStrongActIneq=int(0.3 * n), bounds=True, nStrongActBounds=int(0.1 * n), doubleSidedIneq=True)
base = P.generate_batch(min(B, 128), ch, first_index=50000)
rep = (B + base.B - 1) // base.B
tile = lambda a: np.ascontiguousarray(np.concatenate([a] * rep)[:B])
hb = ch.bounds
pb = P.PackedBatch(n=base.n, m=base.m, nb=base.nb, G=tile(base.G), a=tile(base.a), C=tile(base.C), bl=tile(base.bl), bu=tile(base.bu), xl=tile(base.xl),
                   xu=tile(base.xu), x_star=tile(base.x_star), lam_star=tile(base.lam_star))
dev = capi.to_device(pb)
s = capi.BatchSolver(pb.n, pb.m, hb, B)
args = (dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"]) + ((dev["xl"], dev["xu"]) if hb else ())
for _ in range(2):
    r = s.solve(*args)
torch.cuda.synchronize()
if warm:
    guess = r.status.clone()
    cold_solve = s.solve
    s.solve = lambda *a: s.solve_warm(*a, guess=guess)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(8):
    e0.record(); r = s.solve(*args); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print(f"{fam} B={B} n={pb.n} m={pb.m} nb={pb.nb}: {min(ts):.3f} ms -> {B / min(ts) * 1e3:.0f} solves/s; mean iters {r.iters.double().mean().item():.2f}; ok {int((r.term == 0).sum())}")
if os.environ.get("JRLQP_NOPROF"):
    sys.exit(0)
prof = torch.zeros((B, 32), dtype=torch.int64, device="cuda")
capi.set_profile(capi.PROFILE_GI_BATCH, prof)
s.solve(*args); torch.cuda.synchronize()
capi.set_profile(capi.PROFILE_GI_BATCH, None)
pr = prof.cpu().numpy().astype(np.float64)
if os.environ.get("JRLQP_RAW_PROF"):
    print("raw slots: " + ", ".join(f"{i}: {pr[:, i].mean():.0f}" for i in range(32)))
names = ["load", "LLT", "x0 / J / initActiveSet", "scan + selection", "d = J^T n+", "z || R^-1 d || chain", "scalar products", "step", "addition (sweep)", "removal"]
tot = pr[:, :10].sum(axis=1).mean()
print(f"cycles per solve {tot:.0f}; additions {pr[:, 10].mean():.1f}, removals {pr[:, 11].mean():.1f}")
for i, nm in enumerate(names):
    print(f"  {nm:26s} {pr[:, i].mean():12.0f} ({100 * pr[:, i].mean() / tot:5.1f} %)")
if warm:
    wn = ["processInitialActiveSet", "N, b_act", "B = L^-1 N", "QR", "J = L^-T", "J = J Q, R", "start point + purge"]
    for i, nm in enumerate(wn):
        print(f"  warm: {nm:24s} {pr[:, 16 + i].mean():12.0f}")
print("  of 'z || R^-1 d || chain', per wave: " + ", ".join(f"w{w} {pr[:, 12 + w].mean():.0f}" for w in range(4)) + "  (0: substitution + ratio test, 1: coefficient chain, 2-3: z)")
