"""Condenses gpurun_out/prof_round/ (written by tools/profile_round.sh on the GPU box) into the tracked summaries under
profiles/: bench lines, rocprofv3 kernel stats, PMC counter summaries and the per-launch HBM traffic figures that
bench.py quotes next to the algorithmic bytes.  Usage: python tools/collect_profiles.py [round-tag]   (default r1)"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_round")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r5"
sys.path.insert(0, ROOT)
import subprocess

from bench import kernel_source_hash  # noqa: E402

SHA = kernel_source_hash()  # run this script on the tree the profile was taken on
try:
    COMMIT = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except OSError:
    COMMIT = None


def first_json_line(path):
    for line in open(path):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    raise RuntimeError(f"no JSON line in {path}")


def kernel_stats(path, match):
    for r in csv.DictReader(open(path)):
        if match in r["Name"]:
            return r
    raise RuntimeError(f"{match} not in {path}")


def pmc_per_launch(path, counter, match):
    """mean over launches of the counter summed over its dimensions (one row per (dispatch, dimension instance))"""
    per = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or match not in r["Kernel_Name"]:
            continue
        per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    vals = list(per.values())
    return sum(vals) / len(vals), len(vals)


def main():
    os.makedirs(DST, exist_ok=True)
    # ---- headline (c3)
    bench = first_json_line(os.path.join(SRC, "bench.json"))
    json.dump(bench, open(os.path.join(DST, f"{TAG}_bench_line.json"), "w"), indent=1)
    shutil.copy(os.path.join(SRC, "p_stats", "bench_kernel_stats.csv"), os.path.join(DST, f"{TAG}_bench_kernel_stats.csv"))
    with open(os.path.join(SRC, "p_stats", "bench_kernel_trace.csv")) as f, open(os.path.join(DST, f"{TAG}_bench_kernel_trace_head.csv"), "w") as g:
        for i, line in enumerate(f):
            if i > 40:
                break
            g.write(line)
    shutil.copy(os.path.join(SRC, "p_fetch", "fetch_counter_collection.csv"), os.path.join(DST, f"{TAG}_pmc_fetch_size.csv"))
    shutil.copy(os.path.join(SRC, "p_write", "write_counter_collection.csv"), os.path.join(DST, f"{TAG}_pmc_write_size.csv"))
    ks = kernel_stats(os.path.join(SRC, "p_stats", "bench_kernel_stats.csv"), "gi_batch_kernel")
    fetch, nf = pmc_per_launch(os.path.join(SRC, "p_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE", "gi_batch_kernel")
    write, nw = pmc_per_launch(os.path.join(SRC, "p_write", "write_counter_collection.csv"), "WRITE_SIZE", "gi_batch_kernel")
    traffic = {
        "kernel": "gi_batch_kernel<50,2,false,false>",
        "kernel_source_sha16": SHA,
        "commit": COMMIT,
        "workload": bench["config"]["workload"] if "config" in bench else "c3",
        "FETCH_SIZE_KB_per_launch": round(fetch, 1),
        "WRITE_SIZE_KB_per_launch": round(write, 1),
        "launches_averaged": [nf, nw],
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of streaming reads -> x2; WRITE_SIZE exact; both in KiB; "
                      "collected in two separate rocprofv3 --pmc passes (tools/profile_round.sh)",
        "traffic_bytes_per_launch": int(round((2.0 * fetch + write) * 1024)),
        "algorithmic_bytes_per_launch": bench["roofline"].get("algorithmic_bytes_per_launch"),
        "rocprofv3_avg_launch_ns": float(ks["AverageNs"]),
    }
    json.dump(traffic, open(os.path.join(DST, f"{TAG}_pmc_traffic.json"), "w"), indent=1)
    # ---- c5 (structured GI, batch)
    b5 = first_json_line(os.path.join(SRC, "bench_c5.json"))
    json.dump(b5, open(os.path.join(DST, f"{TAG}_bench_line_c5.json"), "w"), indent=1)
    nb = b5["config"]["batch_per_gpu"]
    shutil.copy(os.path.join(SRC, "p_bgi", "bgi_kernel_stats.csv"), os.path.join(DST, f"{TAG}_blockgi_c5_batch{nb}_kernel_stats.csv"))
    shutil.copy(os.path.join(SRC, "p_bgi_fetch", "fetch_counter_collection.csv"), os.path.join(DST, f"{TAG}_blockgi_c5_batch{nb}_pmc_fetch_size.csv"))
    shutil.copy(os.path.join(SRC, "p_bgi_write", "write_counter_collection.csv"), os.path.join(DST, f"{TAG}_blockgi_c5_batch{nb}_pmc_write_size.csv"))
    BGK = "blockgi_c9_kernel" if "blockgi_c9_kernel" in open(os.path.join(SRC, "p_bgi", "bgi_kernel_stats.csv")).read() else "blockgi_kernel"
    ks5 = kernel_stats(os.path.join(SRC, "p_bgi", "bgi_kernel_stats.csv"), BGK)
    f5, _ = pmc_per_launch(os.path.join(SRC, "p_bgi_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE", BGK)
    w5, _ = pmc_per_launch(os.path.join(SRC, "p_bgi_write", "write_counter_collection.csv"), "WRITE_SIZE", BGK)
    json.dump({
        "kernel": BGK, "batch": nb, "kernel_source_sha16": SHA, "commit": COMMIT,
        "FETCH_SIZE_KB_per_launch": round(f5, 1), "WRITE_SIZE_KB_per_launch": round(w5, 1),
        "traffic_bytes_per_launch_fetch_x2_plus_write": int(round((2.0 * f5 + w5) * 1024)),
        "algorithmic_bytes_per_launch": b5["roofline"].get("algorithmic_bytes_per_launch"),
        "rocprofv3_avg_launch_ns": float(ks5["AverageNs"]),
    }, open(os.path.join(DST, f"{TAG}_blockgi_c5_batch{nb}_traffic.json"), "w"), indent=1)
    # ---- c2 (Mode A): every launch of a solve counts; the pass holds two solves (one warm-up, one step)
    c2_fetch = os.path.join(SRC, "p_c2_fetch", "fetch_counter_collection.csv")
    if os.path.exists(c2_fetch):
        def pmc_total(path, counter):
            tot, disp = 0.0, set()
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] == counter and "jrlqp" in r["Kernel_Name"]:
                    tot += float(r["Counter_Value"])
                    disp.add(r["Dispatch_Id"])
            return tot, len(disp)
        solves = 2
        f2, nd = pmc_total(c2_fetch, "FETCH_SIZE")
        w2, _ = pmc_total(os.path.join(SRC, "p_c2_write", "write_counter_collection.csv"), "WRITE_SIZE")
        shutil.copy(os.path.join(SRC, "p_c2", "c2_kernel_stats.csv"), os.path.join(DST, f"{TAG}_c2_dense_kernel_stats.csv"))
        json.dump({
            "kernels": "all jrlqp:: kernels of jrlqp_dense_solve (Mode A)", "kernel_source_sha16": SHA, "commit": COMMIT, "solves_in_pass": solves,
            "dispatches_per_solve": nd / solves,
            "FETCH_SIZE_KB_per_solve": round(f2 / solves, 1), "WRITE_SIZE_KB_per_solve": round(w2 / solves, 1),
            "correction": "FETCH_SIZE x2 (gfx950 streaming-read under-report, MI355X_MICROARCH.md) + WRITE_SIZE, KiB, separate --pmc passes",
            "traffic_bytes_per_solve": int(round((2.0 * f2 + w2) * 1024 / solves)),
        }, open(os.path.join(DST, f"{TAG}_c2_dense_traffic.json"), "w"), indent=1)
    json.dump(first_json_line(os.path.join(SRC, "bench_c2.json")), open(os.path.join(DST, f"{TAG}_bench_line_c2.json"), "w"), indent=1)
    sq = os.path.join(ROOT, "gpurun_out", "pmc_sq", "summary.json")
    if os.path.exists(sq):
        d = json.load(open(sq))
        d["kernel_source_sha16"], d["commit"] = SHA, COMMIT
        json.dump(d, open(os.path.join(DST, f"{TAG}_pmc_sq_c3.json"), "w"), indent=1)
    if os.path.exists(os.path.join(SRC, "families.txt")):
        shutil.copy(os.path.join(SRC, "families.txt"), os.path.join(DST, f"{TAG}_reference_families.txt"))
    if os.path.exists(os.path.join(SRC, "warm_family.txt")):
        shutil.copy(os.path.join(SRC, "warm_family.txt"), os.path.join(DST, f"{TAG}_reference_warm_family.txt"))
    for src, dst in (("c2.log", "c2_dense_timing.txt"), ("time_blockgi.log", "blockgi_timing.txt"), ("time_structured.log", "c5_structured_timing.txt"),
                     ("quick.log", "phase_profile.txt")):
        if os.path.exists(os.path.join(SRC, src)):
            shutil.copy(os.path.join(SRC, src), os.path.join(DST, f"{TAG}_{dst}"))
    print(json.dumps({"c3": bench["value"], "c3_avg_launch_ms": float(ks["AverageNs"]) / 1e6, "c3_traffic_MB": traffic["traffic_bytes_per_launch"] / 1e6,
                      "c5": b5["value"], "c5_traffic": (2 * f5 + w5) * 1024}, indent=1))


if __name__ == "__main__":
    main()
