"""The reference's own benchmark families (benchmarks/Solvers.cpp:613-639: test1 .. test7, 100 random problems per
size from randomProblem(ProblemCharacteristics(...))) through the batched HIP path, with the CPU
oracle timed beside it on the same problems and the results compared bit for bit.  Sizes above n = 64 (the families go
to n = 100) run through the large-n batched kernel (gi_batch_big.hip).  Not the judged bench; writes a table to stdout.

  python tools/bench_families.py [problems-per-size (default 100)] [batch replication (default 20)]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from __graft_entry__ import load_package

load_package()
import jrl_qp_amd.capi as capi
import jrl_qp_amd.problems as P
from oracle_lib import get_oracle

NPROB = int(sys.argv[1]) if len(sys.argv) > 1 else 100
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 20
NMAX = 128


def families():
    fam = []
    for n in range(10, 101, 10):  # test1: 40 % equalities, no inequalities
        fam.append(("test1", f"n={n}, {int(0.4 * n)} eq", dict(nVar=n, nObj=n, nEq=int(0.4 * n))))
    for frac in (0.0, 0.2, 0.4, 0.6, 0.8, 1.0):  # test2: n=50, m=80, active fraction of min(n, m)
        fam.append(("test2", f"n=50, m=80, {int(frac * 50)} active", dict(nVar=50, nObj=50, nIneq=80, nStrongActIneq=int(frac * 50))))
    for n in range(10, 101, 10):  # test3: 20 % eq, m = n double-sided 30 % active, bounds 10 % active
        fam.append(("test3", f"n={n}, eq+ineq+bounds", dict(nVar=n, nObj=n, nEq=int(0.2 * n), nIneq=n, nStrongActIneq=int(0.3 * n), bounds=True,
                                                            nStrongActBounds=int(0.1 * n), doubleSidedIneq=True)))
    for me in (0, 10, 20, 30, 40):  # test4: n=50, varying equalities
        fam.append(("test4", f"n=50, {me} eq", dict(nVar=50, nObj=50, nEq=me)))
    for act in (0, 10, 20, 30, 40, 50):  # test5 / test6: 50 single- / double-sided inequalities, varying active
        fam.append(("test5", f"n=50, m=50 single, {act} active", dict(nVar=50, nObj=50, nIneq=50, nStrongActIneq=act)))
    for act in (0, 10, 20, 30, 40, 50):
        fam.append(("test6", f"n=50, m=50 double, {act} active", dict(nVar=50, nObj=50, nIneq=50, nStrongActIneq=act, doubleSidedIneq=True)))
    for act in (0, 10, 20, 30, 40, 50):  # test7: bounds only
        fam.append(("test7", f"n=50, bounds, {act} active", dict(nVar=50, nObj=50, bounds=True, nStrongActBounds=act)))
    return fam


def main():
    oracle = get_oracle()
    print(f"# {NPROB} problems per size, GPU batch = {NPROB} x {REP} (the same problems replicated); CPU oracle: one thread, us per solve")
    print(f"{'family':6s} {'problem':34s} {'iters':>6s} {'GPU solves/s':>13s} {'GPU us/batch':>13s} {'CPU us/solve':>13s}  parity")
    for k, (name, label, chd) in enumerate(families()):
        if chd["nVar"] > NMAX:
            print(f"{name:6s} {label:34s}   (n > {NMAX}: outside the batched kernel)")
            continue
        ch = P.Characteristics(**chd)
        pb = P.generate_batch(NPROB, ch, first_index=50000 + 1000 * k)
        hb = ch.bounds
        args = (pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl if hb else None, pb.xu if hb else None)
        t0 = time.perf_counter()
        ref = oracle.solve_batch(*args, nthreads=1)
        cpu1 = (time.perf_counter() - t0) / NPROB * 1e6
        # the same problems, replicated to a batch that fills the GPU (and gives every host thread enough to do)
        tile = lambda a: np.ascontiguousarray(np.concatenate([a] * REP))
        big = P.PackedBatch(n=pb.n, m=pb.m, nb=pb.nb, G=tile(pb.G), a=tile(pb.a), C=tile(pb.C), bl=tile(pb.bl), bu=tile(pb.bu), xl=tile(pb.xl), xu=tile(pb.xu),
                            x_star=tile(pb.x_star), lam_star=tile(pb.lam_star))
        dev = capi.to_device(big)
        s = capi.BatchSolver(big.n, big.m, hb, big.B)
        call = lambda: s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"], dev["xl"] if hb else None, dev["xu"] if hb else None)
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(10):
            e0.record()
            r = call()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = min(ts)
        same = all(np.array_equal(getattr(r, f).cpu().numpy()[:NPROB], ref[f]) for f in ("x", "u", "f", "status", "term", "iters"))
        ok = bool(np.all(ref["term"] == 0))
        print(f"{name:6s} {label:34s} {ref['iters'].mean():6.1f} {big.B / ms * 1e3:13.0f} {ms * 1e3:13.1f} {cpu1:13.1f}  "
              f"{'bit-identical' if same else 'DIFFERENT'}{'' if ok else ' (not all SUCCESS)'}", flush=True)


if __name__ == "__main__":
    main()
