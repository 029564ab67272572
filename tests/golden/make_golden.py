"""Generates tests/golden/gi_golden.npz: seeded inputs are NOT stored (they are regenerated from the seeds by
jrl-qp_amd/problems.py); the file holds the expected outputs of the CPU oracle for them, so that (a) an oracle
regression shows up as a CPU-test failure and (b) the GPU parity tests can check the HIP path against committed
vectors as well as against the live oracle.

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402

load_package()
import jrl_qp_amd.problems as P  # noqa: E402
from oracle_lib import get_oracle  # noqa: E402

SETS = {
    "c1": dict(ch=dict(nVar=10, nObj=10, nIneq=20, nStrongActIneq=3), count=64, first=0),
    "c3": dict(ch=dict(nVar=50, nObj=50, nIneq=200, nStrongActIneq=15), count=32, first=0),
    "mixed": dict(ch=dict(nVar=10, nObj=10, nEq=3, nIneq=20, nStrongActIneq=3, bounds=True, nStrongActBounds=2, doubleSidedIneq=True), count=32, first=7000),
}


def main():
    o = get_oracle()
    out = {}
    for name, spec in SETS.items():
        ch = P.Characteristics(**spec["ch"])
        pb = P.generate_batch(spec["count"], ch, first_index=spec["first"])
        hb = ch.bounds
        r = o.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl if hb else None, pb.xu if hb else None)
        for k in ("x", "u", "status", "term", "iters", "f"):
            out[f"{name}_{k}"] = r[k]
        # fingerprint of the regenerated inputs: catches generator drift (numpy / scipy version changes)
        hsh = hashlib.sha256()
        for arr in (pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl, pb.xu):
            hsh.update(np.ascontiguousarray(arr).tobytes())
        out[f"{name}_inputs_sha256"] = np.frombuffer(hsh.digest(), dtype=np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "gi_golden.npz"), **out)
    print("wrote tests/golden/gi_golden.npz", {k: v.shape for k, v in out.items() if k.endswith("_x")})


if __name__ == "__main__":
    main()
