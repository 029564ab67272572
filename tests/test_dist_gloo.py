"""The N > 1 path on CPU: world_size-2 `gloo` processes exercise the batch sharding and the epilogue collectives
of jrl-qp_amd/dist.py (shard_range, reduce_stats, all_gather_results).  The local solver inside each rank is the
CPU oracle here (test infrastructure standing in for the GPU kernel, which cannot run on this box); on the GPU
box bench.py runs the same code path with the HIP solver and backend nccl (= RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, outdir, use_hip=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from __graft_entry__ import load_package

    load_package()
    import jrl_qp_amd.dist as jdist
    import jrl_qp_amd.problems as P
    from oracle_lib import get_oracle

    r, lr, w = jdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    lo, hi = jdist.shard_range(total, rank, world)
    pb = P.generate_batch(hi - lo, P.benchmark_characteristics(10, 20), first_index=lo)  # only this rank's shard
    if use_hip:  # the product solver: both ranks share the one GPU of the box, collectives stay on gloo / host tensors
        import jrl_qp_amd.capi as capi

        dev = capi.to_device(pb, "cuda:0")
        r = capi.BatchSolver(pb.n, pb.m, False, pb.B, device="cuda:0").solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"])
        res = {k: getattr(r, k).cpu().numpy() for k in ("x", "term", "iters")}
    else:
        res = get_oracle().solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu)
    n_ok, sum_it, tmax = jdist.reduce_stats(int((res["term"] == 0).sum()), int(res["iters"].sum()), 0.25 * (rank + 1))
    x_all = jdist.all_gather_results(torch.from_numpy(res["x"]), total)
    it_all = jdist.all_gather_results(torch.from_numpy(res["iters"].astype(np.int64))[:, None], total)
    if rank == 0:
        np.savez(os.path.join(outdir, "gathered.npz"), x=x_all.numpy(), iters=it_all.numpy()[:, 0], n_ok=n_ok, sum_it=sum_it, tmax=tmax)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly(pkg):
    import jrl_qp_amd.dist as jdist

    for total in (0, 1, 7, 8, 4096, 32768, 1001):
        for world in (1, 2, 3, 8):
            rs = [jdist.shard_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in rs]
            assert max(sizes) - min(sizes) <= 1
    assert [jdist.shard_range(32768, r, 8) for r in range(8)] == [(4096 * r, 4096 * (r + 1)) for r in range(8)]  # BASELINE c4


def _two_ranks(oracle, tmp_path, total, use_hip):
    import socket

    import jrl_qp_amd.problems as P

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, total, str(tmp_path), use_hip), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    pb = P.generate_batch(total, P.benchmark_characteristics(10, 20), first_index=0)
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu)
    assert np.array_equal(got["x"], ref["x"])  # sharded + gathered == unsharded, bit for bit
    assert np.array_equal(got["iters"], ref["iters"])
    assert int(got["n_ok"]) == int((ref["term"] == 0).sum()) == total
    assert int(got["sum_it"]) == int(ref["iters"].sum())
    assert float(got["tmax"]) == 0.5  # MAX over ranks of the per-rank elapsed time


@pytest.mark.parametrize("total", [37, 64])
def test_two_rank_gloo_matches_single_process(pkg, oracle, tmp_path, total):
    _two_ranks(oracle, tmp_path, total, use_hip=False)


@pytest.mark.gpu
def test_two_rank_hip_solver_under_torch_distributed(pkg, oracle, tmp_path):
    """The same N = 2 path with the HIP solver inside each rank (two rank processes on the box's one GPU)."""
    _two_ranks(oracle, tmp_path, 37, use_hip=True)
