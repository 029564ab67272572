"""First timing probe of the batched kernel (not the judged bench): c3 workload, B QPs, HIP-event timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
import jrl_qp_amd.capi as capi, jrl_qp_amd.problems as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n, m = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (50, 200)
t = time.time(); base = P.generate_batch(min(B, 512), P.benchmark_characteristics(n, m)); print("gen", time.time() - t, flush=True)
rep = (B + base.B - 1) // base.B
tile = lambda a: np.ascontiguousarray(np.concatenate([a] * rep)[:B])
pb = P.PackedBatch(n=base.n, m=base.m, nb=0, G=tile(base.G), a=tile(base.a), C=tile(base.C), bl=tile(base.bl), bu=tile(base.bu),
                   xl=tile(base.xl), xu=tile(base.xu), x_star=tile(base.x_star), lam_star=tile(base.lam_star))
dev = capi.to_device(pb)
s = capi.BatchSolver(n, m, False, B)
print("lds bytes", s.lds_bytes, "C placement", s.c_placement)
for _ in range(2):
    s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    e0.record(); r = s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"]); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = min(ts)
it = r.iters.cpu().numpy()
print(f"B={B} n={n} m={m}: {ms:.3f} ms  -> {B / ms * 1e3:.0f} solves/s ; mean iters {it.mean():.2f}; term ok {int((r.term == 0).sum())}/{B}")
print("all times ms", [round(t, 3) for t in sorted(ts)][:8])
# ---- phase profile (diagnostic build)
prof = torch.zeros((B, 32), dtype=torch.int64, device="cuda")
capi.set_profile(capi.PROFILE_GI_BATCH, prof)
s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"]); torch.cuda.synchronize()
capi.set_profile(capi.PROFILE_GI_BATCH, None)
pr = prof.cpu().numpy().astype(np.float64)
# role-specialised main loop (WS): wave 0 (chain wave) stamps slots 3-7, the MAIN wave (wave 3) slots 23-31
names = ["load", "llt", "Jinv/x0", "w0: wait for d (idle)", "w0: coefficient chain", "w0: (c,s) + post", "w0: wait for the decision", "w0: scan + publish + post", "", "store",
         "#adds", "#drops", "realtime(100MHz)", "seeded-chain fallbacks", "", "", "w2: everything but its scans", "w2: scan + publish + post", "llt chain", "llt pivot", "", "", "MAIN: sweep of J",
         "MAIN: drop (+ wait chain wave)", "MAIN: init + first selection", "MAIN: d = J^T n+ (+ post)", "MAIN: trsv + ratio", "MAIN: wait z / dots", "MAIN: step length + update",
         "MAIN: wait (c,s)", "MAIN: R column + bookkeeping", "MAIN: wait scans + combine"]
tot = pr[:, 21:32].sum(axis=1)  # the MAIN wave's time line
print("mean cycles per solve", tot.mean(), " mean realtime us", pr[:, 12].mean() / 100.0, " => clock GHz", tot.mean() / (pr[:, 12].mean() / 100.0) / 1e3)
for i, nm in enumerate(names):
    if nm == '': continue
    print(f"  {nm:18s} {pr[:, i].mean():12.1f}  ({100 * pr[:, i].mean() / tot.mean():5.1f} %)")

# ---- where the hardware put the MAIN wave (slot 14) and wave 0 (slot 15) of every workgroup: SIMD collisions of the serial waves?
hw3, hw0 = prof[:, 14].cpu().numpy().astype(np.uint64), prof[:, 15].cpu().numpy().astype(np.uint64)
simd3, simd0 = (hw3 >> np.uint64(4)) & np.uint64(3), (hw0 >> np.uint64(4)) & np.uint64(3)
cu = ((hw3 >> np.uint64(8)) & np.uint64(15)) | (((hw3 >> np.uint64(12)) & np.uint64(15)) << np.uint64(4)) | ((hw3 >> np.uint64(32)) << np.uint64(8))
print("MAIN wave SIMD histogram", np.bincount(simd3.astype(np.int64), minlength=4).tolist(), " wave 0 SIMD histogram", np.bincount(simd0.astype(np.int64), minlength=4).tolist())
print("(MAIN simd - wave0 simd) mod 4 histogram", np.bincount(((simd3.astype(np.int64) - simd0.astype(np.int64)) % 4), minlength=4).tolist())
print("distinct CU ids seen", len(np.unique(cu)), " wave slots of MAIN", np.bincount((hw3 & np.uint64(15)).astype(np.int64), minlength=16).tolist())
slot3, slot0 = (hw3 & np.uint64(15)).astype(np.int64), (hw0 & np.uint64(15)).astype(np.int64)
print("workgroups whose MAIN wave and wave 0 sit in the same wave slot:", int((slot3 == slot0).sum()), "of", len(slot3),
      "; MAIN on SIMD == slot mod 4:", int((simd3.astype(np.int64) == (slot3 & 3)).sum()))
