"""Extended fuzz of the batched kernel against the oracle (not part of the test suite: minutes, not seconds): every
n in 1 .. 64 (or the given range, up to 128) several times with random constraint mixes, cold and warm, including the embedded sizes and the LDS-staged
column passes.  Exits non-zero on the first difference.   python tools/fuzz_batch.py [trials (default 400)] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from __graft_entry__ import load_package

load_package()
import jrl_qp_amd.capi as capi
import jrl_qp_amd.problems as P
from oracle_lib import get_oracle
from test_gpu_batch import assert_bit_identical, run_gpu
from test_gpu_warm import oracle_warm, perturbed_guess, run_warm

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
NLO = int(sys.argv[3]) if len(sys.argv) > 3 else 1    # range of n (default 1 .. 64; e.g. 36 64 for the role-specialised loop,
NHI = int(sys.argv[4]) if len(sys.argv) > 4 else 64   # 65 256 for the large-n kernel; warm starts up to n = 128)
rng = np.random.default_rng(seed)
oracle = get_oracle()
terms = {}
for trial in range(trials):
    n = int(rng.integers(NLO, NHI + 1))
    nEq = int(rng.integers(0, max(1, n // 4) + 1)) if rng.random() < 0.4 else 0
    bounds = bool(rng.random() < 0.4)
    nIneq = int(rng.integers(0, (8 if rng.random() < 0.15 else 4) * n + 8))
    room = n - nEq
    nSA = int(rng.integers(0, min(nIneq, room) + 1))
    nSB = int(rng.integers(0, max(0, min(n, room - nSA)) + 1)) if bounds else 0
    nWA = int(rng.integers(0, min(3, nIneq - nSA) + 1)) if rng.random() < 0.3 else 0
    nWB = int(rng.integers(0, min(2, n - nSB) + 1)) if (bounds and rng.random() < 0.3) else 0
    ch = P.Characteristics(nVar=n, nObj=n, nEq=nEq, nIneq=nIneq, nStrongActIneq=nSA, nWeakActIneq=nWA, bounds=bounds, nStrongActBounds=nSB,
                           nWeakActBounds=nWB, doubleSidedIneq=bool(rng.random() < 0.5))
    pb = P.generate_batch(16, ch, first_index=700000 + 40 * trial)
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl if bounds else None, pb.xu if bounds else None, nthreads=8)
    gpu = run_gpu(capi, pb, use_bounds=bounds)
    assert_bit_identical(gpu, ref, f"trial {trial}: {ch}")
    if trial % 4 == 0:  # every fourth shape also warm-started, from a stale guess (purge loop + hand-over to the regular loop)
        g = perturbed_guess(rng, ref["status"], pb.m)
        assert_bit_identical(run_warm(capi, pb, g, use_bounds=bounds), oracle_warm(oracle, pb, g, bounds), f"trial {trial} warm: {ch}")
    for t in ref["term"].tolist():
        terms[t] = terms.get(t, 0) + 1
    if trial % 50 == 49:
        print(f"{trial + 1} shapes ok; termination codes so far {terms}", flush=True)
print(f"all {trials} shapes bit-identical; termination codes {terms}")
