"""Timing of the structured GI solver (jrlqp_blockgi_*): BASELINE config c5 (block arrow, 100 stages x 30 + head, 10
constraints per stage) single problem and batches, and the reference's two MultiIK fixtures; the CPU oracle
(structured and dense) timed beside it on one host core.  HIP-event timing of factorisation + loop (one solve call)."""
import ctypes as ct, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import load_package; load_package()
import jrl_qp_amd.capi as capi, jrl_qp_amd.structured_problems as SP
from oracle_lib import get_oracle
from ikdata import read_arrow, read_tri_block_diag
from test_gpu_blockgi import KIND
o = get_oracle()
PROFILE_LINES = []  # printed by the caller behind the timing line they belong to
FORMS = [0]  # form of the last plan (0: orthonormal sequence, 1: contract C9)
FORM_NAME = {0: "sequence", 1: "C9"}


def flush_profile():
    while PROFILE_LINES:
        print(PROFILE_LINES.pop(0), flush=True)


def gpu_time(qps, max_iter, reps=5):
    q0 = qps[0]
    B, n, m = len(qps), q0.n, q0.m
    ub = q0.xl is not None
    H0 = torch.tensor(np.stack([np.ascontiguousarray(q.G.T) for q in qps]), device="cuda")
    Cb = torch.tensor(np.stack([np.concatenate([c.ravel(order="F") for c in q.Cblocks]) for q in qps]), device="cuda")
    a = torch.tensor(np.stack([q.a for q in qps]), device="cuda")
    bl = torch.tensor(np.stack([q.bl for q in qps]), device="cuda"); bu = torch.tensor(np.stack([q.bu for q in qps]), device="cuda")
    xl = torch.tensor(np.stack([q.xl for q in qps]), device="cuda") if ub else None
    xu = torch.tensor(np.stack([q.xu for q in qps]), device="cuda") if ub else None
    s = capi.BlockGISolver(KIND[q0.kind], q0.bs, [c.shape for c in q0.Cblocks], ub, B, max_iter=max_iter, form=os.environ.get("BGI_FORM", "auto"))
    FORMS[0] = s.form
    if os.environ.get("BGI_PACKED", "1") != "0":  # the blocks back to back (ld = block rows) instead of views into N x N matrices
        H0 = s.pack(H0)
    H = H0.clone()
    ts = []
    for _ in range(reps):
        H.copy_(H0)
        s._bind(H, Cb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = s.solve(H, a, Cb, bl, bu, xl, xu); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    if os.environ.get("BGI_PROF"):
        prof = torch.zeros((B, capi.PROFILE_SLOTS), dtype=torch.int64, device="cuda")
        capi.set_profile(capi.PROFILE_BLOCKGI, prof)
        H.copy_(H0); s._bind(H, Cb); s.solve(H, a, Cb, bl, bu, xl, xu); torch.cuda.synchronize()
        capi.set_profile(capi.PROFILE_BLOCKGI, None)
        pr = prof.cpu().numpy().astype(np.float64)[0] / 100.0  # us
        names = ["init", "scan", "L^-1 n+", "Q^T d", "Q z", "L^-T z", "R solve", "step length", "update", "add", "drop", "(Q: reflectors", "Q: Givens)"]
        if s.form == 1:
            names = ["init", "scan", "L^-1 n+", "B^T v", "y -= B dl", "|y|, L^-T y", "S^T g", "step length", "update", "add", "drop", "S d1", "r += dl"]
        PROFILE_LINES.append("      phase profile of the line above, problem 0 (us total): " + "; ".join(f"{nm} {pr[i]:.0f}" for i, nm in enumerate(names))
              + f"  | sum {pr[:13].sum() if s.form == 1 else pr[:11].sum():.0f} us over {int(r.iters[0])} iterations; core clock {pr[13] / max(pr[14], 1e-9) * 0.1:.2f} GHz")
    return min(ts), r.iters.cpu().numpy(), r.term.cpu().numpy(), s.workspace_bytes


def cpu_time(qp, max_iter, dense=False, c9=False):
    Gs = qp.G.copy(order="F")
    t = time.perf_counter()
    if dense:
        r = o.solve(Gs, qp.a, qp.dense_C(), qp.bl, qp.bu, qp.xl, qp.xu, max_iter=max_iter)
    else:
        d, f = SP.block_views(Gs, qp.bs, qp.kind)
        r = o.blockgi_solve(qp.kind, d, f, qp.a, qp.Cblocks, qp.bl, qp.bu, qp.xl, qp.xu, max_iter=max_iter, c9=c9)
    return (time.perf_counter() - t) * 1e3, r.iters


def fixtures():
    d = read_tri_block_diag(); nd, C = d["block"], d["C"]
    first = np.array([int(np.flatnonzero(C[:, i])[0]) for i in range(C.shape[1])]); blk = first // nd
    Cb = [np.asfortranarray(C[i * nd:(i + 1) * nd, blk == i]) for i in range(d["nblk"])]
    tri = SP.StructuredQP("tri", [nd] * d["nblk"], np.asfortranarray(d["G"]), d["a"], Cb, d["bl"], d["bu"])
    d = read_arrow(); nd, k, C = d["block"], d["nblk"], d["C"]
    Cb = [np.asfortranarray(C[i * nd:(i + 1) * nd, 5 * i:5 * (i + 1)]) for i in range(k)]
    arr = SP.StructuredQP("arrow_up", [nd] * k, np.asfortranarray(d["G"]), d["a"], Cb, d["bl"], d["bu"], d["xl"], d["xu"])
    return tri, arr


what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("all", "ik"):
    tri, arr = fixtures()
    for name, qp in (("MultiIK tri-block 9x43 m=1621", tri), ("MultiIK arrow-up 5x42 m=25+box", arr)):
        cs, it = cpu_time(qp, 500); cd, _ = cpu_time(qp, 500, dense=True)
        for B in (1, 256, 1024):
            ms, iters, term, ws = gpu_time([qp] * B, 64)
            print(f"{name}: batch {B:4d} [{FORM_NAME[FORMS[0]]}]: GPU {ms:8.3f} ms ({ms / B * 1e3:8.1f} us/problem, {B / ms * 1e3:9.0f} solves/s) iters {iters[0]} term {term[0]};  "
                  f"CPU oracle structured {cs:.3f} ms, dense {cd:.3f} ms (1 core)", flush=True)
            flush_profile()
if what in ("all", "c5", "c5b64", "c5b512"):
    for sa in (1.0,):
        base = [SP.mpc_arrow_qp(100, 30, 10, seed=s, scale_a=sa) for s in range(2)]
        cs, it = cpu_time(base[0], 1000)
        c9s, _ = cpu_time(base[0], 1000, c9=True)
        print(f"c5 (n=3030, m=1000, scale_a={sa}): CPU oracle structured, sequence form {cs:.1f} ms, C9 {c9s:.1f} ms, {it} iterations (dense oracle: ~50 s)", flush=True)
        for B in ((64,) if what == "c5b64" else (512,) if what == "c5b512" else (1, 16, 64)):
            qps = (base * B)[:B]
            ms, iters, term, ws = gpu_time(qps, 1000, reps=3)
            print(f"   batch {B:3d} [{FORM_NAME[FORMS[0]]}]: GPU {ms:9.2f} ms  ({ms / B:8.2f} ms/problem, {B / ms * 1e3:7.1f} solves/s)  iters {iters[:2]} term {term[:2]}  "
                  f"{ms / iters.mean() * 1e3:7.1f} us/iteration/batch  workspace {ws / 2**20:.0f} MiB", flush=True)
            flush_profile()
