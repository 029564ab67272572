"""The Givens radius recurrence of an addition runs SEEDED on the GPU (device_common.h givens_seed /
givens_seeded_step): one correction step per division and per square root from seeds prepared in parallel, every step
checked afterwards against the plain formula, the plain recurrence as fallback.  These tests drive the three debug
modes of that machinery (jrlqp_debug_set_chain_mode) through the batched kernel and the host-driven dense path and
require BIT-identical results in all of them:
  0  production: seeded + verified
  1  plain recurrence only
  2  seeds deliberately spoiled: every sweep fails its check and is redone by the fallback."""
import numpy as np
import pytest

from test_gpu_batch import assert_bit_identical, run_gpu
from test_gpu_dense import check, solve_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi(pkg):
    import jrl_qp_amd.capi as capi

    return capi


@pytest.fixture(scope="module")
def P(pkg):
    import jrl_qp_amd.problems as problems

    return problems


@pytest.fixture()
def chain_mode(capi):
    lib = capi.load_library()

    def set_mode(mode):
        assert lib.jrlqp_debug_set_chain_mode(mode) == 0

    yield set_mode
    lib.jrlqp_debug_set_chain_mode(0)


BATCH_CASES = [
    dict(nVar=50, nObj=50, nIneq=200, nStrongActIneq=15),  # register-resident C, chain sliced on wave 3
    dict(nVar=64, nObj=64, nIneq=128, nStrongActIneq=19),
    dict(nVar=30, nObj=30, nEq=3, nIneq=60, nStrongActIneq=8, bounds=True, nStrongActBounds=4),  # generic kernel, bounds: zero tails of d
]


@pytest.mark.parametrize("ci", range(len(BATCH_CASES)))
def test_batched_kernel_chain_modes(capi, P, oracle, chain_mode, ci):
    ch = P.Characteristics(**BATCH_CASES[ci])
    pb = P.generate_batch(192, ch, first_index=31000 + 1000 * ci)
    hb = ch.bounds
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl if hb else None, pb.xu if hb else None, nthreads=8)
    for mode in (0, 1, 2):
        chain_mode(mode)
        gpu = run_gpu(capi, pb, use_bounds=hb)
        assert_bit_identical(gpu, ref, f"case {ci}, chain mode {mode}")


def test_spoiled_seeds_take_the_fallback(capi, P, chain_mode):
    """Mode 2 must really go through the fallback (the phase-profile build counts them), mode 0 must not need it."""
    import torch

    pb = P.generate_batch(128, P.benchmark_characteristics(50, 200), first_index=777)
    lib = capi.load_library()
    counts = {}
    for mode in (0, 2):
        chain_mode(mode)
        prof = torch.zeros((pb.B, 32), dtype=torch.int64, device="cuda")
        capi.set_profile(capi.PROFILE_GI_BATCH, prof)
        try:
            run_gpu(capi, pb)
        finally:
            capi.set_profile(capi.PROFILE_GI_BATCH, None)
        pr = prof.cpu().numpy()
        counts[mode] = (int(pr[:, 13].sum()), int(pr[:, 10].sum()))  # (fallbacks, additions)
    assert counts[0][0] == 0, counts
    assert counts[2][0] > 0.9 * counts[2][1], counts  # (an addition at q = n - 1 has no rotation to verify)


def test_dense_path_chain_modes(capi, P, oracle, chain_mode):
    ch = P.benchmark_characteristics(200, 300)
    pb = P.generate_batch(1, ch, first_index=5151)
    ref = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], max_iter=3000)
    assert ref.term == 0
    for mode in (0, 1, 2):
        chain_mode(mode)
        gpu = solve_gpu(capi, pb, 0, False, max_iter=3000)
        check(gpu, ref, f"chain mode {mode}")
