"""RCCL on the one-GPU box: the HIP batched solver under torch.distributed with backend "nccl" (= RCCL) and world size 1
-- communicator creation, the epilogue all-reduce of counters and the all-gather of the device-resident results of
jrl-qp_amd/dist.py, on device tensors the kernel has just written.  (The N = 2 partitioning itself is covered on CPU by
tests/test_dist_gloo.py; N > 1 on GPUs is the driver's scaling run.)"""
import ctypes as ct
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1(pkg):
    import torch
    import torch.distributed as dist

    import jrl_qp_amd.dist as jdist

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    old = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    rank, local_rank, world = jdist.init_process_group("nccl", force=True)
    assert (rank, local_rank, world) == (0, 0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    yield jdist
    torch.cuda.synchronize()
    dist.destroy_process_group()
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_hip_solver_under_rccl_world1(pkg, oracle, nccl_world1):
    import torch
    import torch.distributed as dist

    import jrl_qp_amd.capi as capi
    import jrl_qp_amd.problems as P

    jdist = nccl_world1
    total = 96
    lo, hi = jdist.shard_range(total, dist.get_rank(), dist.get_world_size())
    assert (lo, hi) == (0, total)
    pb = P.generate_batch(hi - lo, P.benchmark_characteristics(50, 200), first_index=lo)
    dev = capi.to_device(pb, "cuda:0")
    solver = capi.BatchSolver(pb.n, pb.m, False, pb.B, device="cuda:0")
    r = solver.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"])
    # epilogue collectives on DEVICE tensors (RCCL): counters / max time, then the result assembly
    n_ok, sum_it, tmax = jdist.reduce_stats(int((r.term == 0).sum().item()), int(r.iters.sum().item()), 0.125, device="cuda:0")
    x_all = jdist.all_gather_results(r.x, total)
    it_all = jdist.all_gather_results(r.iters.to(torch.int64)[:, None], total)
    assert x_all.is_cuda and x_all.shape == (total, pb.n)
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu)
    assert np.array_equal(x_all.cpu().numpy(), ref["x"])
    assert np.array_equal(it_all.cpu().numpy()[:, 0], ref["iters"])
    assert n_ok == total == int((ref["term"] == 0).sum()) and sum_it == int(ref["iters"].sum()) and tmax == 0.125
    # one more collective the bench epilogue uses: all_gather of per-rank timings
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
    out = [torch.empty_like(t)]
    dist.all_gather(out, t)
    assert float(out[0].item()) == 1.5


def test_multi_device_entry_with_one_device(pkg, oracle):
    """jrlqp_gi_solve_batch_multi / jrlqp_gi_gather_x_multi of the C ABI (one process, a list of devices; here the list is [0] -- a
    box with more GPUs runs the same code with more shards): results bit-identical to the oracle, for a shape without a workspace
    (n = 50) and one with (n = 100), and to the single-device entry point."""
    import torch

    import jrl_qp_amd.capi as capi
    import jrl_qp_amd.problems as P

    for ch, total in ((P.benchmark_characteristics(50, 200), 96), (P.Characteristics(nVar=100, nObj=100, nIneq=120, nStrongActIneq=20), 24)):
        pb = P.generate_batch(total, ch, first_index=600)
        dev = capi.to_device(pb, "cuda:0")
        ms = capi.MultiDeviceBatchSolver(pb.n, pb.m, False, total, ["cuda:0"])
        out = ms.solve([dev])
        x_all = ms.gather_x(0)
        ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, nthreads=8)
        for k in ("x", "u", "status", "term", "iters", "f"):
            assert np.array_equal(out[0][k].cpu().numpy(), ref[k]), k
        assert np.array_equal(x_all.cpu().numpy(), ref["x"])
        single = capi.BatchSolver(pb.n, pb.m, False, total, device="cuda:0").solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"])
        torch.cuda.synchronize()
        assert torch.equal(single.x, out[0]["x"]) and torch.equal(single.u, out[0]["u"])
    # a device id the process does not see, and a device listed twice
    lib = capi.load_library()
    sh = (capi.ShardBuffers * 2)()
    assert lib.jrlqp_gi_solve_batch_multi(0, 50, 200, 0, 8, 1, (ct.c_int * 1)(torch.cuda.device_count()), sh, None, 1) == -1
    assert lib.jrlqp_gi_solve_batch_multi(0, 50, 200, 0, 8, 2, (ct.c_int * 2)(0, 0), sh, None, 1) == -1


def test_two_shards_rehearsed_on_one_device(pkg, oracle):
    """Round 5: jrlqp_gi_solve_batch_multi with TWO shards on the one GPU this box has (debug switch: the device list [0, 0]) -- two
    shard offsets, two streams, two workspaces, jrlqp_gi_gather_x_multi assembling x: equal to the single call and to the oracle,
    with an uneven split (97 = 49 + 48) and a shape that needs a workspace.  Then an injected failure in shard 1: the call must
    come back with the error AFTER shard 0's stream has drained (shard 0's results complete), and say which shard failed.
    No scaling meaning: both shards share one device."""
    import torch

    import jrl_qp_amd.capi as capi
    import jrl_qp_amd.problems as P

    lib = capi.load_library()
    lib.jrlqp_debug_multi_allow_shared_device(1)
    try:
        for ch, total in ((P.benchmark_characteristics(50, 200), 97), (P.Characteristics(nVar=100, nObj=100, nIneq=120, nStrongActIneq=20), 25)):
            pb = P.generate_batch(total, ch, first_index=700)
            dev = capi.to_device(pb, "cuda:0")
            cuts = [capi.shard_range(total, 2, r) for r in range(2)]
            assert cuts[0][1] + cuts[1][1] == total and cuts[1][0] == cuts[0][1]
            shards = [{k: (v[f : f + c].contiguous() if v is not None else None) for k, v in dev.items()} for f, c in cuts]
            streams = [torch.cuda.Stream(device="cuda:0") for _ in range(2)]
            torch.cuda.synchronize()
            ms = capi.MultiDeviceBatchSolver(pb.n, pb.m, False, total, ["cuda:0", "cuda:0"], streams=streams)
            out = ms.solve(shards)
            x_all = ms.gather_x(1)
            ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, nthreads=8)
            for r, (f, c) in enumerate(cuts):
                for k in ("x", "u", "status", "term", "iters", "f"):
                    assert np.array_equal(out[r][k].cpu().numpy(), ref[k][f : f + c]), (r, k)
            assert np.array_equal(x_all.cpu().numpy(), ref["x"])
            single = capi.BatchSolver(pb.n, pb.m, False, total, device="cuda:0").solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"])
            torch.cuda.synchronize()
            assert torch.equal(single.x, x_all)
            assert lib.jrlqp_gi_multi_failed_shard() == -1
            # an injected failure in shard 1: error returned, shard 0 finished and correct, shard 1 untouched
            for r in range(2):
                out[r]["x"].fill_(float("nan"))
            torch.cuda.synchronize()
            lib.jrlqp_debug_multi_inject_failure(1)
            try:
                rc = lib.jrlqp_gi_solve_batch_multi(0, pb.n, pb.m, 0, total, 2, ms.ids, ms._shards(shards), ct.byref(ms.opts), 1)
            finally:
                lib.jrlqp_debug_multi_inject_failure(-1)
            assert rc == -1 and lib.jrlqp_gi_multi_failed_shard() == 1
            # NO synchronisation here: the call itself must have drained shard 0's stream
            x0 = out[0]["x"].cpu().numpy()
            assert np.array_equal(x0, ref["x"][: cuts[0][1]])
            assert np.isnan(out[1]["x"].cpu().numpy()).all()
    finally:
        lib.jrlqp_debug_multi_allow_shared_device(0)
    # without the switch a device listed twice is still refused
    sh = (capi.ShardBuffers * 2)()
    assert lib.jrlqp_gi_solve_batch_multi(0, 50, 200, 0, 8, 2, (ct.c_int * 2)(0, 0), sh, None, 1) == -1
    assert lib.jrlqp_gi_gather_x_multi(50, 8, 2, (ct.c_int * 2)(0, 0), sh, 0, ct.c_void_p(8), 1) == -1
