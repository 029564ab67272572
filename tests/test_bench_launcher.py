"""`python bench.py --gpus N` is its own launcher: the parent starts N fresh rank processes and must fail loudly -- never
fall back to one GPU -- when the node has fewer than N devices, when the launcher's WORLD_SIZE disagrees with --gpus, or
when any rank fails."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


def _run(args, env=None):
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=900, env=env or _clean_env())


def test_more_gpus_than_visible_is_refused():
    import torch

    n = torch.cuda.device_count() + 1  # counting devices does not initialise the GPU
    r = _run(["--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "refusing to run on fewer" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
def test_dry_gpus2_on_the_one_gpu_box_fails_loudly():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has several GPUs")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "refusing to run on fewer" in r.stderr and not r.stdout.strip()


def test_world_size_mismatch_is_refused():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = _run(["--gpus", "2"], env)
    assert r.returncode == 2 and "WORLD_SIZE=4 but --gpus 2" in r.stderr


def test_a_failing_rank_fails_the_launch(monkeypatch):
    """With the device count check bypassed on a GPU-less box every rank stops at "needs a GPU": the launcher must
    report that as a failure of the whole run."""
    import torch

    if torch.cuda.device_count() > 0:
        pytest.skip("needs a box without GPUs")
    sys.path.insert(0, ROOT)
    import bench

    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 2)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    rc = bench.launch_ranks(2, ["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert rc != 0


@pytest.mark.gpu
def test_two_rank_rehearsal_on_one_gpu():
    """The whole N = 2 path of `python bench.py --gpus 2` on the one-GPU box: launcher -> two fresh rank processes -> rendezvous ->
    each rank generates and solves ITS shard with the HIP solver -> barrier-to-barrier timing, MAX over ranks -> epilogue
    collectives -> one JSON line from rank 0.  (Both ranks share cuda:0 and talk over gloo: JRLQP_BENCH_SHARED_GPU=1, a rehearsal
    hook; the line says so and is no scaling measurement.  RCCL itself is exercised by tests/test_gpu_dist.py.)"""
    import json

    env = dict(_clean_env(), JRLQP_BENCH_SHARED_GPU="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "512"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["requested_gpus"] == 2 and d["steps"] == 3
    assert d["config"]["total_qps_per_step"] == 1024 and d["config"]["solved"] == 1024
    rk = d["ranks"]
    assert rk["rccl_world_size"] == 2 and rk["backend"] == "gloo" and rk["shared_gpu_rehearsal"] is True
    assert len(rk["per_rank_solves_per_s"]) == 2 and all(v > 0 for v in rk["per_rank_solves_per_s"])
    assert rk["allgather_x_bytes"] == 1024 * 50 * 8 and rk["allgather_x_ms"] > 0
    assert d["value"] > 0 and "cpu_baseline" not in d  # the CPU baseline belongs to N = 1 only


def test_ranks_inherit_the_ipc_mode_and_get_dmabuf_by_default(monkeypatch, tmp_path):
    """launch_ranks gives every rank HSA_ENABLE_IPC_MODE_LEGACY: the caller's value when set, "0" (dmabuf IPC, the only mode
    the pool's host driver supports -- RCCL fails with hipIpcGetMemHandle otherwise) when absent; and RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1."""
    sys.path.insert(0, ROOT)
    import bench

    seen = []

    class FakeProc:
        def __init__(self, cmd, env=None):
            seen.append(env)

        def poll(self):
            return 0

    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 2)
    monkeypatch.setattr("subprocess.Popen", FakeProc)
    for preset, want in ((None, "0"), ("1", "1")):
        seen.clear()
        if preset is None:
            monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
        else:
            monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", preset)
        assert bench.launch_ranks(2, ["--gpus", "2"]) == 0
        assert [e["RANK"] for e in seen] == ["0", "1"] and all(e["WORLD_SIZE"] == "2" and e["MASTER_ADDR"] == "127.0.0.1" for e in seen)
        assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == want for e in seen)
        assert len({e["MASTER_PORT"] for e in seen}) == 1


def test_numa_binding_helper(pkg, monkeypatch, tmp_path):
    """bind_rank_to_gpu_numa: reads <sysfs>/bus/pci/devices/<bdf>/numa_node and node<N>/cpulist, intersects with the mask the
    process has, never widens it, never raises."""
    import types

    import torch

    import jrl_qp_amd.dist as jdist

    assert jdist._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    props = types.SimpleNamespace(pci_domain_id=0, pci_bus_id=0x75, pci_device_id=0)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: props)
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:75:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    have = sorted(os.sched_getaffinity(0))
    (node / "cpulist").write_text(f"{have[0]}-{have[0]}\n")
    try:
        msg = jdist.bind_rank_to_gpu_numa(0, 8, sysfs=str(tmp_path))
        assert msg.startswith("bound to 1 NUMA-local") and os.sched_getaffinity(0) == {have[0]}
    finally:
        os.sched_setaffinity(0, have)
    (dev / "numa_node").write_text("-1\n")
    assert jdist.bind_rank_to_gpu_numa(0, 8, sysfs=str(tmp_path)).startswith("not bound") and os.sched_getaffinity(0) == set(have)
    (dev / "numa_node").write_text("1\n")
    (node / "cpulist").write_text("100000\n")
    assert "none of the" in jdist.bind_rank_to_gpu_numa(0, 8, sysfs=str(tmp_path)) and os.sched_getaffinity(0) == set(have)
