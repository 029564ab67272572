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


# structured GI loop (oracle/blockgi_oracle.c): seeded structured problems, expected outputs per problem
STRUCTURED = {
    "sg_tri": dict(kind="tri", bs=[7, 4, 9, 6, 5, 8], mc=[4, 6, 3, 5, 7, 2], seeds=range(100, 108), ridge=0.05),
    "sg_arrow_up": dict(kind="arrow_up", bs=[6, 9, 4, 7, 5], mc=[5, 3, 6, 4, 2], seeds=range(100, 108), ridge=0.05),
    "sg_arrow_down_box": dict(kind="arrow_down", bs=[6, 9, 4, 7, 5], mc=[5, 7, 6], seeds=range(300, 308), ridge=0.05, cbs=[10, 8, 13], box=1.0),
}


def structured_problems(spec):
    import jrl_qp_amd.structured_problems as SP

    kw = {k: spec[k] for k in ("ridge", "cbs", "box") if k in spec}
    return [SP.random_structured_qp(spec["kind"], spec["bs"], spec["mc"], seed, **kw) for seed in spec["seeds"]]


def structured_digest(qps):
    hsh = hashlib.sha256()
    for q in qps:
        for arr in [q.G, q.a, q.bl, q.bu] + list(q.Cblocks) + ([q.xl, q.xu] if q.xl is not None else []):
            hsh.update(np.ascontiguousarray(arr).tobytes())
    return np.frombuffer(hsh.digest(), dtype=np.uint8)


def structured_expected(o, qps):
    import jrl_qp_amd.structured_problems as SP

    res = dict(x=[], u=[], status=[], term=[], iters=[], f=[])
    for q in qps:
        Gs = q.G.copy(order="F")
        diag, off = SP.block_views(Gs, q.bs, q.kind)
        r = o.blockgi_solve(q.kind, diag, off, q.a, q.Cblocks, q.bl, q.bu, q.xl, q.xu)
        res["x"].append(r.x); res["u"].append(r.u); res["status"].append(r.status)
        res["term"].append(r.term); res["iters"].append(r.iters); res["f"].append(r.f)
    return {k: np.array(v) for k, v in res.items()}


def main():
    o = get_oracle()
    out = {}
    for name, spec in STRUCTURED.items():
        qps = structured_problems(spec)
        for k, v in structured_expected(o, qps).items():
            out[f"{name}_{k}"] = v
        out[f"{name}_inputs_sha256"] = structured_digest(qps)
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
