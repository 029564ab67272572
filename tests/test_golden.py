"""Committed golden vectors (tests/golden/gi_golden.npz, made by tests/golden/make_golden.py): the oracle must
reproduce them bit for bit on CPU; on the GPU box the batched HIP path must too."""
import hashlib
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gi_golden.npz")


def _sets():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(GOLD), "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.SETS


def _mod():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(GOLD), "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _inputs(P, spec):
    ch = P.Characteristics(**spec["ch"])
    pb = P.generate_batch(spec["count"], ch, first_index=spec["first"])
    h = hashlib.sha256()
    for arr in (pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl, pb.xu):
        h.update(np.ascontiguousarray(arr).tobytes())
    return ch, pb, np.frombuffer(h.digest(), dtype=np.uint8)


@pytest.mark.parametrize("name", ["c1", "c3", "mixed"])
def test_oracle_reproduces_golden(pkg, oracle, name):
    import jrl_qp_amd.problems as P

    gold = np.load(GOLD)
    ch, pb, digest = _inputs(P, _sets()[name])
    if not np.array_equal(digest, gold[f"{name}_inputs_sha256"]):
        pytest.skip("generator inputs differ on this numpy/scipy build; golden outputs do not apply")
    hb = ch.bounds
    r = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl if hb else None, pb.xu if hb else None)
    for k in ("x", "u", "status", "term", "iters", "f"):
        assert np.array_equal(r[k], gold[f"{name}_{k}"]), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c1", "c3", "mixed"])
def test_hip_reproduces_golden(pkg, name):
    import torch
    import jrl_qp_amd.capi as capi
    import jrl_qp_amd.problems as P

    gold = np.load(GOLD)
    ch, pb, digest = _inputs(P, _sets()[name])
    if not np.array_equal(digest, gold[f"{name}_inputs_sha256"]):
        pytest.skip("generator inputs differ on this numpy/scipy build; golden outputs do not apply")
    hb = ch.bounds
    dev = capi.to_device(pb, "cuda:0")
    s = capi.BatchSolver(pb.n, pb.m, hb, pb.B, device="cuda:0")
    r = s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"], dev["xl"] if hb else None, dev["xu"] if hb else None)
    torch.cuda.synchronize()
    got = dict(x=r.x, u=r.u, status=r.status, term=r.term, iters=r.iters, f=r.f)
    for k, t in got.items():
        assert np.array_equal(t.cpu().numpy(), gold[f"{name}_{k}"]), k


SG = ["sg_tri", "sg_arrow_up", "sg_arrow_down_box"]


@pytest.mark.parametrize("name", SG)
def test_structured_oracle_reproduces_golden(pkg, oracle, name):
    mod = _mod()
    gold = np.load(GOLD)
    qps = mod.structured_problems(mod.STRUCTURED[name])
    if not np.array_equal(mod.structured_digest(qps), gold[f"{name}_inputs_sha256"]):
        pytest.skip("generator inputs differ on this numpy/scipy build; golden outputs do not apply")
    exp = mod.structured_expected(oracle, qps)
    for k, v in exp.items():
        assert np.array_equal(v, gold[f"{name}_{k}"]), k
    assert (gold[f"{name}_term"] == 0).all() and gold[f"{name}_iters"].sum() > 50


@pytest.mark.gpu
@pytest.mark.parametrize("name", SG)
def test_structured_hip_reproduces_golden(pkg, name):
    from test_gpu_blockgi import gpu_solve
    import jrl_qp_amd.capi as capi

    mod = _mod()
    gold = np.load(GOLD)
    qps = mod.structured_problems(mod.STRUCTURED[name])
    if not np.array_equal(mod.structured_digest(qps), gold[f"{name}_inputs_sha256"]):
        pytest.skip("generator inputs differ on this numpy/scipy build; golden outputs do not apply")
    g = gpu_solve(capi, qps)
    for k in ("x", "u", "status", "term", "iters", "f"):
        assert np.array_equal(g[k], gold[f"{name}_{k}"]), k
