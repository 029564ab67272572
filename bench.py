#!/usr/bin/env python3
"""bench.py -- dense Goldfarb-Idnani QP solves/sec on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: B = 4096 independent QPs (n = 50, m_ineq = 200, 15 strongly
active, the reference's benchmark recipe benchmarks/Solvers.cpp:257-263) per GPU, solved by ONE launch of the
device-resident batched kernel (jrlqp_gi_solve_batch).  Inputs are resident in HBM before the timed region.
N > 1: one process per GPU (torchrun), each rank owns its own 4096-QP shard (weak scaling; BASELINE config c4 =
32768 QPs on 8 GPUs); no data-path collective, only the epilogue all-reduce of counters / max time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_VAR, M_INEQ, BATCH = 50, 200, 4096
# Rehearsal hook for one-GPU boxes (tests/test_bench_launcher.py): JRLQP_BENCH_SHARED_GPU=1 lets `--gpus N` start its N ranks on
# cuda:0 with the gloo backend, so that the launcher, the rendezvous, the sharding, the timing protocol and the epilogue
# collectives of the N > 1 path run end to end.  The line it prints says so ("shared_gpu_rehearsal": true): it is not a
# scaling number and the driver never sets the variable.
SHARED_GPU_TEST = os.environ.get("JRLQP_BENCH_SHARED_GPU") == "1"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_solve(n, m, nb):
    """compulsory traffic per dense solve: inputs G,a,C,bl,bu,xl,xu + outputs x,u,status,{term,iters},f."""
    return 8 * (n * n + n + n * m + 2 * m + 2 * nb) + 8 * (n + m + nb) + 4 * (m + nb) + 16


def kernel_source_hash():
    """sha256[:16] over the kernel sources (jrl-qp_amd/csrc).  tools/collect_profiles.py stores it next to every PMC
    traffic figure, and a figure is quoted only while the sources it was measured on are the sources of this tree."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "jrl-qp_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "jrl-qp_amd", "csrc", "*.h"))
                    + glob.glob(os.path.join(ROOT, "jrl-qp_amd", "csrc", "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(pattern, key):
    """HBM bytes per launch from the separate rocprofv3 --pmc passes of tools/profile_round.sh (newest profiles/ file
    matching `pattern`).  Returns (bytes or None, provenance string): None when no file exists or when the file was
    measured on other kernel sources than this tree's (stale numbers are never quoted beside fresh timings)."""
    import glob

    import re

    def round_no(path):  # profiles/r<N>_...: newest = highest round NUMBER (a lexicographic sort puts r10 before r2)
        m = re.match(r"r(\d+)_", os.path.basename(path))
        return (int(m.group(1)) if m else -1, os.path.getmtime(path))

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=round_no)
    if not files:
        return None, "no PMC pass on record (profiles/" + pattern + ")"
    path = files[-1]
    try:
        d = json.load(open(path))
        rel = os.path.relpath(path, ROOT)
        if d.get("kernel_source_sha16") != kernel_source_hash():
            return None, f"{rel} was measured on other kernel sources (sha16 {d.get('kernel_source_sha16')}); not quoted"
        return float(d[key]), f"{rel}: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, commit {d.get('commit')}"
    except (OSError, ValueError, KeyError) as e:
        return None, f"unreadable {path}: {e}"


def oracle_build_flags():
    """The compiler flags the CPU baseline was built with (oracle/Makefile), quoted in the bench line."""
    try:
        for line in open(os.path.join(ROOT, "oracle", "Makefile")):
            if line.startswith("CFLAGS"):
                return "gcc " + line.split("=", 1)[1].strip()
    except OSError:
        pass
    return None


def cgroup_cpu_quota():
    """CPUs the container may use according to its cgroup (v2 cpu.max, v1 cfs quota); None = unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(pb, budget_s=12.0):
    """The CPU restatement (oracle/, kind "port") timed on this box's host cores on a bounded sample of the same workload.
    The timed loop is the reference's own benchmark loop (benchmarks/Solvers.cpp:447-453, :508-520) and runs entirely in C
    (oracle orc_gi_bench_rounds): one workspace and one G buffer per OpenMP thread, G restored by a per-thread memcpy INSIDE
    the loop before every solve, outputs preallocated, the clock between two barriers inside the parallel region -- no Python,
    no serial copy and no allocation inside the timed rounds.  Reported baseline, not the optimisation target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import get_oracle  # cpu_baseline leg only

    o = get_oracle()
    try:
        cores = len(os.sched_getaffinity(0))  # the cores this process may actually use
    except AttributeError:
        cores = os.cpu_count() or 1
    sample = pb.B
    run, _ = o.bench_rounds_fn(pb.G[:sample], pb.a[:sample], pb.C[:sample], pb.bl[:sample], pb.bu[:sample])
    np_ = min(256, pb.B)
    probe_run, _ = o.bench_rounds_fn(pb.G[:np_], pb.a[:np_], pb.C[:np_], pb.bl[:np_], pb.bu[:np_])
    probe_run(1, 1)
    per = min(probe_run(1, 1) for _ in range(3)) / np_  # single-thread seconds per solve (same loop, same restore)
    # The affinity mask can overstate what the container may really use: a cgroup CPU quota throttles a pool that is wider than
    # the quota, but only once the quota of a scheduling period (100 ms) is spent -- a probe of a few milliseconds runs at the
    # full width of the host and says nothing about a sustained run.  So every width is probed for >= 0.6 s (several periods),
    # widths stop at the quota, and the fastest SUSTAINED width is used; the short-burst rates are reported beside it
    # (`burst_probe_solves_per_s`: what the host's cores do when the quota does not bind -- an upper bound for a dedicated host).
    quota = cgroup_cpu_quota()
    # (no width beyond the quota: threads the cgroup cannot run only add context switches, and a 0.6 s probe of such a width is
    # noisy enough to win now and then -- a 32-thread pick under a 16-cpu quota cost 10 % of the sustained rate)
    widest = cores if quota is None else max(1, min(cores, int(quota + 0.5)))
    best_t, best_rate, t, probed, burst = 1, 0.0, 1, {}, {}
    while True:
        run(1, t)  # the first call at a new width pays for the OpenMP pool and page faults
        b = sample / run(1, t)
        burst[t] = b
        k = max(1, int(0.6 * b / sample + 0.5))
        rate = sample * k / run(k, t)
        probed[t] = rate
        if rate > best_rate:
            best_t, best_rate = t, rate
        if t >= widest:
            break
        t = min(widest, t * 2)
    for t in sorted({min(cores, w) for w in (32, 64, 128)} - set(burst)):  # information only: one-round bursts beyond the quota
        run(1, t)
        burst[t] = max(sample / run(1, t) for _ in range(2))
    cores_avail, cores = cores, best_t
    rounds_per_call = max(1, int(2.0 * best_rate / sample))  # about two seconds per C call
    rounds, dt = 0, 0.0
    while dt < budget_s and rounds < 100000:
        dt += run(rounds_per_call, cores)
        rounds += rounds_per_call
    value = sample * rounds / dt
    return {
        "value": value,
        "unit": "solves/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{rounds} rounds over the first {sample} QPs of the same workload ({dt:.1f} s inside the C timing loop), {cores} OpenMP threads "
                  f"(fastest SUSTAINED width of {sorted(probed)}, each probed for >= 0.6 s; affinity mask has {cores_avail} cpus, cgroup quota "
                  f"{quota}), one workspace + one G buffer per thread, G restored by memcpy inside the loop before every solve "
                  f"(benchmarks/Solvers.cpp:451); single-thread probe {per * 1e6:.1f} us/solve",
        "single_thread_us_per_solve": per * 1e6,
        "parallel_efficiency": value * per / cores,  # value / (threads x single-thread rate)
        "thread_probe_solves_per_s": {str(k): v for k, v in sorted(probed.items())},
        "burst_probe_solves_per_s": {str(k): v for k, v in sorted(burst.items())},
        "cgroup_cpu_quota": quota,
        "affinity_cpus": cores_avail,
        "build_flags": oracle_build_flags(),
    }


def visible_gpu_count():
    """Number of GPUs the rank processes will see, WITHOUT initialising the GPU in this process: counted by a short-lived
    child (torch.cuda.device_count() honours HIP/ROCR/CUDA_VISIBLE_DEVICES)."""
    import subprocess

    out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
    try:
        return int(out.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def launch_ranks(n, argv):
    """Parent of `python bench.py --gpus N`: start N fresh rank processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, exactly what torch.distributed.run would set), wait for them, and fail loudly -- no
    silent single-GPU run -- when fewer than N devices are visible or any rank fails.  Rank 0 prints the JSON line."""
    import socket
    import subprocess

    have = visible_gpu_count()
    if SHARED_GPU_TEST and have >= 1:
        have = n  # rehearsal on a one-GPU box: every rank uses cuda:0, collectives over gloo (never a scaling measurement)
    if have < n:
        print(f"bench.py: --gpus {n} requested but only {have} GPU(s) visible on this node; refusing to run on fewer", file=sys.stderr)
        return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf IPC; without it RCCL's cross-process
        # buffer registration fails with "hipIpcGetMemHandle: invalid argument".  The image exports it already -- the ranks
        # inherit whatever the caller set, and get 0 only when the variable is absent (tests/test_bench_launcher.py).
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {procs.index(p)} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in live:
                    q.terminate()  # exactly the PIDs started above
        time.sleep(0.05)
    return rc


def rank_report(jdist, torch, device, local_solves, elapsed_local, x_local, total):
    """Epilogue of an N-rank run: the RCCL world size this rank saw, every rank's own solves/s (all-gather of the local
    elapsed times) and the all-gather of x -- the optional result assembly of SURVEY 8(e) -- timed on its own, outside
    the solve timing."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return {"rccl_world_size": 1, "backend": None, "per_rank_solves_per_s": [local_solves / elapsed_local], "allgather_x_ms": None,
                "allgather_x_bytes": 0, "process_group_initialized": False}
    world = dist.get_world_size()
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    ts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(ts, t)
    jdist.all_gather_results(x_local, total)  # warm the communicator / buffers
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    dist.barrier()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        xa = jdist.all_gather_results(x_local, total)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    ag_ms = (time.perf_counter() - t0) / reps * 1e3
    assert xa.shape[0] == total
    return {"rccl_world_size": world, "backend": dist.get_backend(), "per_rank_solves_per_s": [local_solves / float(v.item()) for v in ts],
            "allgather_x_ms": ag_ms, "allgather_x_bytes": int(xa.numel() * 8), "process_group_initialized": True}


def bench_ik(args, rank, world, device, jdist):
    """Secondary workload: a batch of copies of one of the reference's MultiIK fixtures (tests/MultiIK.zip of the reference,
    committed as tests/golden/MultiIK.zip: the tri-block "Sequential IK" problem, n = 387, m = 1621, or the arrow
    "Simultaneous IK" problem, n = 210, 25 inequalities + box) through jrlqp_blockgi_solve.  The reference prints
    "Vanilla vs Structured" milliseconds for exactly these two problems (tests/BlockGISolverTest.in.cpp:252-268, :332-348)."""
    import numpy as np
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ikdata  # reader of the fixture files (data, not reference code)

    import jrl_qp_amd.capi as capi

    tri = args.workload == "ik-tri"
    B = args.batch if args.batch != BATCH else 1024
    lo, hi = jdist.shard_range(B * world, rank, world)
    nloc = hi - lo
    if tri:
        d = ikdata.read_tri_block_diag()
        nd, C = d["block"], d["C"]
        first = np.array([int(np.flatnonzero(C[:, i])[0]) for i in range(C.shape[1])])
        blk = first // nd
        Cb = [np.asfortranarray(C[i * nd:(i + 1) * nd, blk == i]) for i in range(d["nblk"])]
        kind, xl, xu = "tridiag", None, None
    else:
        d = ikdata.read_arrow()
        nd = d["block"]
        Cb = [np.asfortranarray(d["C"][i * nd:(i + 1) * nd, 5 * i:5 * (i + 1)]) for i in range(d["nblk"])]
        kind, xl, xu = "arrow_up", d["xl"], d["xu"]
    n, m = d["G"].shape[0], sum(c.shape[1] for c in Cb)
    rep = lambda v: torch.tensor(np.ascontiguousarray(v), device=device).unsqueeze(0).expand(nloc, *np.shape(v)).contiguous()
    H0 = rep(d["G"].T)
    Cbt = rep(np.concatenate([c.ravel(order="F") for c in Cb]))
    a, bl, bu = rep(d["a"]), rep(d["bl"]), rep(d["bu"])
    xlt, xut = (rep(xl), rep(xu)) if xl is not None else (None, None)
    solver = capi.BlockGISolver(kind, [nd] * d["nblk"], [c.shape for c in Cb], xl is not None, nloc, max_iter=64, device=device)
    H = H0.clone()
    stream = torch.cuda.current_stream(device)
    for _ in range(max(args.warmup, 1)):
        H.copy_(H0)
        solver._bind(H, Cbt)
        r = solver.solve(H, a, Cbt, bl, bu, xlt, xut)
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    kern_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        H.copy_(H0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r = solver.solve(H, a, Cbt, bl, bu, xlt, xut)
        e1.record(stream)
        torch.cuda.synchronize(device)
        kern_ms.append(e0.elapsed_time(e1))
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    n_ok, sum_it, elapsed = jdist.reduce_stats(int((r.term == 0).sum().item()), int(r.iters.sum().item()), elapsed, device=device)
    if rank == 0:
        total = B * world
        gbytes = 8.0 * (sum(c.size for c in Cb) + (2 * d["nblk"] - 1) * nd * nd + n + 2 * m + (2 * n if xl is not None else 0))
        alg = nloc * (gbytes + 8.0 * (n + m) + 4.0 * m + 16)
        avg_s = float(np.mean(kern_ms)) * 1e-3
        out = {"metric": f"structured GI QP solves/sec, MultiIK {'tri-block 9x43 (m=1621)' if tri else 'arrow 5x42 (25 ineq + box)'} fixture, batch per GPU",
               "value": total * args.steps / elapsed, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "reference fixture (tests/golden/MultiIK.zip), replicated",
               "config": {"workload": f"{B} copies per GPU of the reference's MultiIK fixture through jrlqp_blockgi_solve (in-place block LLT + structured GI "
                                      f"loop); wall time includes restoring G before each solve", "batch_per_gpu": B,
                          "mean_iterations": sum_it / total, "solved": n_ok},
               "roofline": {"bound": "hbm", "achieved": alg / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / avg_s / 1e9 / HBM_PEAK_GBS,
                            "traffic": None, "kernel": "tri_llt / arrow_* + blockgi_kernel<256,1>", "algorithmic_bytes_per_launch": alg,
                            "avg_launch_ms": avg_s * 1e3,
                            "note": "compulsory bytes only (G blocks, C, vectors, results once); these few-iteration solves are bound by the serial "
                                    "substitution chains of the block solves (DESIGN.md 7.3)"}}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def bench_single_large(args, rank, world, device, jdist, emit=True):
    """Secondary workload (BASELINE config c2): one dense QP, n = 1024, m = 2048, 307 strongly active, through Mode A
    (jrlqp_dense_solve: state in HBM, host keeps the active-set state machine, one kernel per hook).  The path does not
    shard: at N > 1 every rank solves its own replica.  A step = one solve; G is restored before each
    solve (the solver factorises it in place, like the reference).  Roofline: HBM against the streamed-state model of
    SURVEY.md 8(d) -- per iteration 8(nm + n^2 + n(n-q) + q^2/2) + 16 n(n-q) bytes, q ~ iterations/2 -- the state
    (32 MiB) stays in the Infinity Cache, so this is an upper bound on HBM bytes, not a measurement."""
    import numpy as np
    import torch

    import jrl_qp_amd.capi as capi
    import jrl_qp_amd.problems as P

    n, m = 1024, 2048
    pb = P.generate_batch(1, P.benchmark_characteristics(n, m), first_index=rank)
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(device)
    G0, a, C, bl, bu = dev(pb.G[0]), dev(pb.a[0]), dev(pb.C[0]), dev(pb.bl[0]), dev(pb.bu[0])
    s = capi.DenseSolver(n, m, False, max_iter=100000)
    G = G0.clone()
    steps = min(args.steps, 5)
    for _ in range(max(1, min(args.warmup, 2))):
        G.copy_(G0)
        r = s.solve(G, a, C, bl, bu)
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    per_solve = []
    t0 = time.perf_counter()
    for _ in range(steps):
        G.copy_(G0)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        r = s.solve(G, a, C, bl, bu)
        torch.cuda.synchronize(device)
        per_solve.append(time.perf_counter() - t1)
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    n_ok, sum_it, elapsed = jdist.reduce_stats(int(r["term"] == 0), int(r["iters"]), elapsed, device=device)
    # init_ alone (LLT of G, J = L^-T, x0; SURVEY 8(d) asks for it next to the per-iteration time): through the exported hooks
    init_ms = None
    if rank == 0 and not getattr(args, "no_secondary", False):  # (counter passes: only whole solves, see tools/profile_round.sh)
        import ctypes as ct

        lib = capi.load_library()
        ptr = lambda t: ct.c_void_p(t.data_ptr())
        stream = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
        info, f0 = ct.c_int(0), ct.c_double(0.0)
        best = None
        for _ in range(3):
            G.copy_(G0)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            ok = lib.jrlqp_dense_bind(s.h, ptr(G), n, ptr(a), ptr(C), n, ptr(bl), ptr(bu), None, None) == 0
            ok = ok and lib.jrlqp_dense_init(s.h, ct.byref(info), ct.byref(f0), stream) == 0  # (synchronises: it returns the LLT info)
            dt = (time.perf_counter() - t1) * 1e3
            if ok and info.value < 0:
                best = dt if best is None else min(best, dt)
        init_ms = best
    if rank == 0:
        it = r["iters"]
        q = it / 2.0
        per_iter = 8.0 * (n * m + n * n + n * (n - q) + q * q / 2) + 16.0 * n * (n - q)
        alg = 8.0 * (n * n + n + n * m + 2 * m) + it * per_iter
        sec = elapsed / steps
        traffic, traffic_source = pmc_traffic("r*_c2_dense_traffic.json", "traffic_bytes_per_solve")
        out = {"metric": "dense GI, one large QP (n=1024, m=2048): solves/sec per GPU (replicas)", "value": world * steps / elapsed, "unit": "solves/s",
               "n_gpus": world, "steps": steps, "warmup": max(1, min(args.warmup, 2)), "ms_per_step": sec * 1e3,
               "ms_per_step_min": min(per_solve) * 1e3, "ms_per_step_median": float(np.median(per_solve)) * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE configs[1] (c2): one dense QP n=1024, m_ineq=2048, 307 strongly active, Mode A (host-driven hooks); "
                                      "replicas only at N > 1", "iterations": it, "solved": n_ok, "us_per_iteration": sec / max(it, 1) * 1e6, "init_ms": init_ms},
               "roofline": {"bound": "hbm", "achieved": alg / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / sec / 1e9 / HBM_PEAK_GBS,
                            "traffic": traffic, "traffic_source": traffic_source,
                            "kernel": "per-hook kernels (select / jtn / z / step_tail / add_chain / rot_apply), all launches of one solve",
                            "algorithmic_bytes_per_launch": alg,
                            "note": "streamed-state model (upper bound: J, R, C = 32 MiB stay in the Infinity Cache); the iteration is bound by "
                                    "the serial Givens coefficient chain and the blocked back substitution, not by bandwidth (DESIGN.md 7.1)"}}
        if not emit:
            return out
        print(json.dumps(out), flush=True)
    if not emit:
        return None
    if world > 1:
        torch.distributed.destroy_process_group()


def bench_structured(args, rank, world, device, jdist, emit=True):
    """Secondary workload (BASELINE config c5): a batch of structured QPs -- block-arrow Hessian (100 stage blocks of 30 +
    head), block-diagonal C with 10 one-sided constraints per stage -- through jrlqp_blockgi_solve (factorisation + the
    device-resident structured GI loop, contract C9: DESIGN.md 7.3).  A step = one batch; each rank owns its own shard (weak scaling).
    32 distinct seeded problems per GPU, tiled to the batch; the blocks in packed storage (1.43 MB per problem).  Beside the batch
    rate the record carries what SURVEY 8(d) asks of c5: single-problem latency, block factorisation us, block solve us."""
    import numpy as np
    import torch

    import jrl_qp_amd.capi as capi
    import jrl_qp_amd.structured_problems as SP

    B = args.batch if args.batch != BATCH else 512
    lo, hi = jdist.shard_range(B * world, rank, world)
    nloc = hi - lo
    ND = min(32, nloc)  # distinct problems
    gen = [SP.mpc_arrow_packed(100, 30, 10, seed=s, scale_a=1.0) for s in range(ND)]
    bs, cshapes = gen[0][5], gen[0][6]
    n, m = sum(bs), sum(c for _, c in cshapes)
    idx = torch.tensor([(lo + i) % ND for i in range(nloc)], device=device)
    dev = lambda k: torch.tensor(np.stack([g[k] for g in gen]), device=device)[idx].contiguous()
    H0, a, Cb, bl, bu = dev(0), dev(1), dev(2), dev(3), dev(4)
    solver = capi.BlockGISolver("arrow_down", bs, cshapes, False, nloc, max_iter=1000, device=device)
    H = H0.clone()
    stream = torch.cuda.current_stream(device)

    def step():
        H.copy_(H0)  # the solver factorises G in place (like the reference): restored outside the kernel timing events
        solver._bind(H, Cb)
        return solver.solve(H, a, Cb, bl, bu)

    for _ in range(max(args.warmup, 1)):
        r = step()
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
    steps = min(args.steps, 5)
    kern_ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        H.copy_(H0)
        solver._bind(H, Cb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r = solver.solve(H, a, Cb, bl, bu)
        e1.record(stream)
        torch.cuda.synchronize(device)
        kern_ms.append(e0.elapsed_time(e1))
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    iters = r.iters.cpu().numpy()
    term = r.term.cpu().numpy()
    n_ok, sum_it, elapsed = jdist.reduce_stats(int((term == 0).sum()), int(iters.sum()), elapsed, device=device)
    if rank == 0:
        total = B * world
        # ---- one problem alone (latency layout, a plan of its own), and the factorisation / the two block solves alone ----
        # (--no-secondary: skipped, so that a profiler's per-kernel averages see the batch launches only)
        extras = not getattr(args, "no_secondary", False)
        single = capi.BlockGISolver("arrow_down", bs, cshapes, False, 1, max_iter=1000, device=device) if extras else None
        H1 = H0[:1].clone()
        one_ms = []
        r1 = None
        for rep in range(4 if extras else 0):
            H1.copy_(H0[:1])
            single._bind(H1, Cb[:1])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            r1 = single.solve(H1, a[:1], Cb[:1], bl[:1], bu[:1])
            e1.record(stream)
            torch.cuda.synchronize(device)
            if rep:
                one_ms.append(e0.elapsed_time(e1))
        fact = structured_factor_times(capi, SP, device) if extras else {"factor_us": None, "solve_us": None, "what": "skipped (--no-secondary)"}
        # Byte models, named as such:
        #  * compulsory: every input read once, every output written once -- the factor blocks (201 x 900 doubles, in
        #    place: read + written), C, a, bl, bu; x, u, status, {term, iters}, f.  This is `roofline.achieved`: the path's own floor.
        #  * implementation model of contract C9: per pass of an iteration with q active constraints the two triangular products read S
        #    twice (q^2 / 2 doubles each), the sparse columns of B are read twice (60 doubles per column), and L is read once by L^-T
        #    (201 x 900 doubles) -- two passes per iteration.  Implementation traffic, not a property of the problem.
        blocks = 201 * 900
        compulsory = nloc * (8.0 * (2 * blocks + n + 1000 * 30 + 2 * m) + 8 * (n + m) + 4 * m + 16)
        qsum2 = float(sum(2 * (0.92 * k) ** 2 for it in iters for k in range(0, int(it), 8)) * 8)  # sum over iterations of 2 q^2 (q ~ 0.92 k), sampled every 8th
        model = compulsory + 8.0 * (qsum2 + float(sum(int(i) for i in iters)) * (blocks + 4 * 60 * 250))
        avg_s = float(np.mean(kern_ms)) * 1e-3
        traffic, traffic_source = pmc_traffic(f"r*_blockgi_c5_batch{nloc}_traffic.json", "traffic_bytes_per_launch_fetch_x2_plus_write")
        out = {
            "metric": "structured GI QP solves/sec, block arrow 100x30+30 (n=3030, m=1000), batch per GPU",
            "value": total * steps / elapsed, "unit": "solves/s", "n_gpus": world, "steps": steps, "warmup": max(args.warmup, 1),
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[4] (c5) as a batch: {B} structured QPs per GPU ({ND} distinct seeded problems, tiled), block arrow (down) 100 "
                                   f"stage blocks of 30 + head, 10 one-sided constraints per stage, blocks in packed storage, through jrlqp_blockgi_solve "
                                   f"(in-place block LLT + device-resident structured GI loop, contract C9); wall time includes restoring G before each solve",
                       "batch_per_gpu": B, "distinct_problems": ND, "mean_iterations": sum_it / total, "min_iterations": int(iters.min()),
                       "max_iterations": int(iters.max()), "solved": n_ok, "form": "C9" if solver.form == 1 else "sequence"},
            "single_problem_ms": float(np.median(one_ms)) if one_ms else None, "single_problem_iterations": int(r1.iters[0]) if r1 is not None else None,
            "single_problem_term": int(r1.term[0]) if r1 is not None else None,
            "factor_us": fact["factor_us"], "solve_us": fact["solve_us"], "factor_solve_what": fact["what"],
            "roofline": {"bound": "hbm", "achieved": compulsory / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": compulsory / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "blockgi_c9_kernel", "algorithmic_bytes_per_launch": compulsory, "avg_launch_ms": avg_s * 1e3,
                         "implementation_model": {"bytes_per_launch": model, "achieved": model / avg_s / 1e9, "frac": model / avg_s / 1e9 / HBM_PEAK_GBS,
                                                  "what": "compulsory bytes + what contract C9 streams per iteration: S twice per pass (q^2 / 2 doubles each), L once "
                                                          "(L^-T), the sparse columns of B; one problem's working set (L 1.4 MB, S up to 2 x 1.9 MB, B 0.25 MB) is "
                                                          "L2-resident, a batch of 512 is not (2.6 GB): these are HBM / Infinity Cache bytes only in a batch"},
                         "note": "frac is against the COMPULSORY bytes of SURVEY 8(d): the kernel is latency- and issue-bound (one workgroup per problem; the "
                                 "phase profile of tools/time_blockgi.py: L^-T y 38 %, y -= B dl 18 %, the triangular products 22 %), not bandwidth-bound; rounds "
                                 "1-4 kept the reference's orthonormal sequence and streamed 7.9 GB per solve (0.57 of the HBM peak at batch 512, 560 solves/s)"},
        }
        out["ms_per_step_min"], out["ms_per_step_median"] = float(np.min(kern_ms)), float(np.median(kern_ms))
        if traffic is not None:
            out["roofline"]["traffic_over_compulsory"] = traffic / compulsory
            out["roofline"]["traffic_over_implementation_model"] = traffic / model
        if not emit:
            return out
        print(json.dumps(out), flush=True)
    if not emit:
        return None
    if world > 1:
        torch.distributed.destroy_process_group()


def structured_factor_times(capi, SP, device):
    """blockArrowLLT of ONE 101 x 30 arrow (jrlqp_blocks_llt) and one L solve + one L^T solve of a vector
    (jrlqp_blocks_solve), HIP-event microseconds, median of 5 after a warm-up."""
    import numpy as np
    import torch

    bs = [30] * 101
    qp_blocks = SP.mpc_arrow_packed(100, 30, 10, seed=0, scale_a=1.0)[0]
    n = sum(bs)
    # jrlqp_blocks_* take the blocks as views into a full matrix: assemble the arrow once (one problem: 73 MB)
    G = np.zeros((n, n))
    o = 0
    for i in range(101):
        G[30 * i:30 * i + 30, 30 * i:30 * i + 30] = qp_blocks[o:o + 900].reshape(30, 30, order="F")
        o += 900
    for i in range(100):
        G[3000:, 30 * i:30 * i + 30] = qp_blocks[o:o + 900].reshape(30, 30, order="F")
        o += 900
    T0 = torch.tensor(np.ascontiguousarray(np.tril(G).T)[None], device=device)  # T[0].T = lower triangle of G
    rhs = torch.tensor(np.random.default_rng(5).standard_normal((1, 1, n)), device=device)
    stream = torch.cuda.current_stream(device)
    tf, ts = [], []
    for rep in range(6):
        T = T0.clone()
        M = rhs.clone()
        plan = capi.StructuredBlocks("arrow_down", bs, T, max_nrhs=1)
        torch.cuda.synchronize(device)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(stream)
        plan.llt(fetch=False)
        e1.record(stream)
        plan.solve(M, False)
        plan.solve(M, True)
        e2.record(stream)
        torch.cuda.synchronize(device)
        if rep:
            tf.append(e0.elapsed_time(e1) * 1e3)
            ts.append(e1.elapsed_time(e2) * 1e3)
    return {"factor_us": float(np.median(tf)), "solve_us": float(np.median(ts)),
            "what": "one problem: jrlqp_blocks_llt of the 101 x 30 arrow; jrlqp_blocks_solve L then L^T on one vector (HIP events)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="QPs per GPU per step")
    ap.add_argument("--n", type=int, default=N_VAR)
    ap.add_argument("--m", type=int, default=M_INEQ)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline only: skip the ma in {0, 25} rates, the c2 / c5 secondary records and the sustained loop (they run AFTER the "
                         "timed region and never enter `value`); the profiling scripts use it to keep counter passes short")
    ap.add_argument("--sustain", type=float, default=2.5, help="seconds of the labelled sustained loop after the timed region (N = 1)")
    ap.add_argument("--workload", default="c3", choices=["c3", "c5", "c2", "ik-tri", "ik-arrow"],
                    help="c3 (default, the BASELINE metric): batch of dense QPs; c5: batch of structured block-arrow QPs "
                         "(100 stages x 30 + head, 10 constraints per stage) through the structured GI solver; "
                         "c2: ONE large dense QP (n=1024, m=2048) through the host-driven hooks (replicas only at N > 1); "
                         "ik-tri / ik-arrow: batches of the reference's MultiIK fixtures (tests/golden/MultiIK.zip) through the "
                         "structured GI solver")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from separate rocprofv3 --pmc passes (default: profiles/r1_pmc_traffic.json if it matches)")
    args = ap.parse_args()

    env_world = int(os.environ.get("WORLD_SIZE", "1")) if "RANK" in os.environ else None
    if args.gpus > 1 and env_world is None:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It never touches the GPU and
        # never loads the HIP library; it starts N fresh rank processes and exits with their verdict.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and env_world != args.gpus:
        print(f"bench.py: launched with WORLD_SIZE={env_world} but --gpus {args.gpus}; run either `python bench.py --gpus N` (spawns its own N "
              f"ranks) or `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N`", file=sys.stderr)
        raise SystemExit(2)

    import numpy as np
    import torch

    from __graft_entry__ import load_package

    load_package()
    import jrl_qp_amd.capi as capi
    import jrl_qp_amd.dist as jdist
    import jrl_qp_amd.problems as P

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    if torch.cuda.device_count() < args.gpus and not SHARED_GPU_TEST:
        print(f"bench.py: --gpus {args.gpus} requested but only {torch.cuda.device_count()} device(s) visible", file=sys.stderr)
        raise SystemExit(3)
    rank, local_rank, world = jdist.init_process_group("gloo" if SHARED_GPU_TEST else None)
    if world != args.gpus:
        raise SystemExit(f"bench.py: process group has {world} rank(s) but --gpus {args.gpus}")
    device = torch.device("cuda", local_rank if (world > 1 and not SHARED_GPU_TEST) else 0)
    torch.cuda.set_device(device)
    # one rank per GPU: keep the rank's host threads on the cores next to that GPU (N = 1: the CPU baseline must keep every core)
    cpu_binding = jdist.bind_rank_to_gpu_numa(device.index, world) if world > 1 else "not bound (single rank: the CPU baseline uses every core)"

    if args.workload == "c5":
        return bench_structured(args, rank, world, device, jdist)
    if args.workload == "c2":
        return bench_single_large(args, rank, world, device, jdist)
    if args.workload.startswith("ik-"):
        return bench_ik(args, rank, world, device, jdist)

    n, m, B = args.n, args.m, args.batch
    total = B * world
    lo, hi = jdist.shard_range(total, rank, world)
    ch = P.benchmark_characteristics(n, m)
    pb = P.generate_batch(hi - lo, ch, first_index=lo)  # each rank generates only its own shard
    dev = capi.to_device(pb, device)
    solver = capi.BatchSolver(n, m, False, hi - lo, device=device)
    stream = torch.cuda.current_stream(device)

    def step():
        return solver.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"], stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize(device)
    # HIP events on the launch stream around every launch (kernel duration for the roofline) ...
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record(stream)
        r = step()
        e1.record(stream)
    torch.cuda.synchronize(device)
    elapsed_local = time.perf_counter() - t0  # this rank's own K steps (per-rank solves/s of the N > 1 report)
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0  # ... and the wall clock around EXACTLY `steps` steps, barrier to barrier
    kern_ms = [a.elapsed_time(b) for a, b in evs]

    term = r.term.cpu().numpy()
    iters = r.iters.cpu().numpy()
    cdev = "cpu" if SHARED_GPU_TEST else device  # (gloo rehearsal: collectives on host tensors)
    n_ok, sum_it, elapsed = jdist.reduce_stats(int((term == 0).sum()), int(iters.sum()), elapsed, device=cdev)
    ranks = rank_report(jdist, torch, cdev, (hi - lo) * args.steps, elapsed_local, r.x.cpu() if SHARED_GPU_TEST else r.x, total)
    if SHARED_GPU_TEST:
        ranks["shared_gpu_rehearsal"] = True
    ranks["rank0_cpu_binding"] = cpu_binding
    ranks["hsa_enable_ipc_mode_legacy"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total * args.steps / elapsed
        bytes_per_launch = algorithmic_bytes_per_solve(n, m, 0) * (hi - lo)
        traffic, traffic_source = args.traffic_bytes, "--traffic-bytes"
        if traffic is None:
            traffic, traffic_source = None, "not measured for this shape"
            if (n, m, hi - lo) == (N_VAR, M_INEQ, BATCH):
                traffic, traffic_source = pmc_traffic("r*_pmc_traffic.json", "traffic_bytes_per_launch")
        avg_kernel_s = float(np.mean(kern_ms)) * 1e-3
        achieved = bytes_per_launch / avg_kernel_s / 1e9
        out = {
            "metric": "dense GI QP solves/sec, (n=50,m=200) batch=4096",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "requested_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"batch of {B} dense QPs per GPU, n={n}, m_ineq={m}, {ch.nStrongActIneq} strongly active, planted solution "
                            f"(reference recipe benchmarks/Solvers.cpp:257-263), seeds 0x6A726C71+index; BASELINE configs[2] (c3) per GPU, "
                            f"configs[3] (c4) at 8 GPUs",
                "batch_per_gpu": B,
                "total_qps_per_step": total,
                "mean_iterations": sum_it / total,
                "solved": n_ok,
                "parallelism": f"batch-sharded x{world}, no data-path collective",
            },
            "ranks": ranks,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": "gi_batch_kernel",
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": avg_kernel_s * 1e3,
                "note": "latency/LDS-bound kernel (serial Givens/TRSV chains); HBM fraction is low by construction, see DESIGN.md §6",
            },
        }
        # "min/median of >= 10 repetitions after warm-up" -- per-launch HIP-event times of the K timed steps
        out["ms_per_step_min"], out["ms_per_step_median"] = float(np.min(kern_ms)), float(np.median(kern_ms))
        out["value_at_median_step"] = (hi - lo) / (float(np.median(kern_ms)) * 1e-3) * world
        out["value_at_min_step"] = (hi - lo) / (float(np.min(kern_ms)) * 1e-3) * world
        if world == 1 and not args.no_secondary:
            # Everything below runs AFTER the timed region of the headline and never enters `value`.
            # ---- "also report ma in {0, floor(0.5 n)}" -- the same shape with 0 / 25 strongly active constraints
            other = {}
            for ma in (0, n // 2):
                pb2 = P.generate_batch(hi - lo, P.benchmark_characteristics(n, m, ma), first_index=lo)
                d2 = capi.to_device(pb2, device)
                for _ in range(2):
                    r2 = solver.solve(d2["G"], d2["a"], d2["C"], d2["bl"], d2["bu"], stream=stream)
                ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
                for e0, e1 in ev2:
                    e0.record(stream)
                    r2 = solver.solve(d2["G"], d2["a"], d2["C"], d2["bl"], d2["bu"], stream=stream)
                    e1.record(stream)
                torch.cuda.synchronize(device)
                k2 = [a.elapsed_time(b) for a, b in ev2]
                other[f"ma={ma}"] = {"value": (hi - lo) / (float(np.median(k2)) * 1e-3), "unit": "solves/s", "ms_per_step_median": float(np.median(k2)),
                                     "ms_per_step_min": float(np.min(k2)), "steps": 10, "mean_iterations": float(r2.iters.double().mean().item()),
                                     "solved": int((r2.term == 0).sum().item())}
                del d2, pb2
            out["other_active_fractions"] = other
            # ---- how much of a launch is its first and last round: the same problems tiled to 2 B and 4 B (4096 workgroups are 5.33 rounds
            # of the 768 resident ones; the marginal rate between two batch sizes is what the kernel sustains once the machine is full)
            try:
                scal = {"B": [], "ms_per_launch_min": []}
                for mult in (1, 2, 4):
                    dm = {k: (torch.cat([v] * mult) if mult > 1 else v) for k, v in dev.items() if k in ("G", "a", "C", "bl", "bu")}
                    sm = capi.BatchSolver(n, m, False, (hi - lo) * mult, device=device)
                    for _ in range(2):
                        sm.solve(dm["G"], dm["a"], dm["C"], dm["bl"], dm["bu"], stream=stream)
                    evm = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
                    for e0, e1 in evm:
                        e0.record(stream)
                        sm.solve(dm["G"], dm["a"], dm["C"], dm["bl"], dm["bu"], stream=stream)
                        e1.record(stream)
                    torch.cuda.synchronize(device)
                    scal["B"].append((hi - lo) * mult)
                    scal["ms_per_launch_min"].append(float(min(a.elapsed_time(b) for a, b in evm)))
                    del dm, sm
                tms, bs = scal["ms_per_launch_min"], scal["B"]
                scal["solves_per_s"] = [b_ / (t_ * 1e-3) for b_, t_ in zip(bs, tms)]
                scal["marginal_solves_per_s"] = (bs[2] - bs[0]) / ((tms[2] - tms[0]) * 1e-3)
                scal["launch_overhead_ms"] = tms[0] - bs[0] / scal["marginal_solves_per_s"] * 1e3
                scal["what"] = ("the headline problems tiled to 2 B and 4 B per launch, best of 8 launches each; marginal = (4 B - B) / (t(4 B) - t(B)): "
                                "the rate once every CU holds its three workgroups; not part of `value`")
                out["batch_scaling"] = scal
            except Exception as e:  # (an extra: never let it take the bench line down)
                out["batch_scaling"] = {"error": repr(e)}
            # ---- a clearly labelled SUSTAINED loop: the K timed steps are ~30 ms of kernels, too short for a utilisation sampler to see
            if args.sustain > 0:
                torch.cuda.synchronize(device)
                ts, k = time.perf_counter(), 0
                while time.perf_counter() - ts < args.sustain:
                    for _ in range(50):
                        step()
                    torch.cuda.synchronize(device)
                    k += 50
                dts = time.perf_counter() - ts
                out["sustained"] = {"what": "back-to-back launches of the headline workload for GPU-utilisation samplers; not part of `value`",
                                    "seconds": dts, "steps": k, "value": (hi - lo) * k / dts, "unit": "solves/s"}
            # ---- secondary workloads of BASELINE.json, each with a roofline of its own (c2: Mode A; c5: structured GI batch)
            sec_args = argparse.Namespace(**vars(args))
            sec_args.steps, sec_args.warmup, sec_args.batch = 3, 1, BATCH
            del dev
            torch.cuda.empty_cache()
            secondary = {}
            for name, fn in (("c2", bench_single_large), ("c5", bench_structured)):
                try:
                    secondary[name] = fn(sec_args, 0, 1, device, jdist, emit=False)
                except Exception as e:  # a secondary record must never cost the headline line
                    secondary[name] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
            out["secondary"] = secondary
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pb)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
