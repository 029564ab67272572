# Round profile: bench line, rocprofv3 kernel stats, PMC traffic (FETCH_SIZE / WRITE_SIZE in SEPARATE passes, no other trace
# domain with --pmc), for the headline batched kernel (c3) and for the structured GI kernel (c5), plus the timing logs of
# the secondary workloads.  Run on the GPU box:  gpurun -- bash tools/profile_round.sh ; then copy summaries to profiles/.
set -e
export TMPDIR=/tmp
O=gpurun_out/prof_round; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err   # the driver's command: headline + ma variants + sustained loop + secondary c2 / c5 + cpu_baseline
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_stats -o bench -- python3 bench.py --no-cpu-baseline --no-secondary > $O/bench_prof.json 2> $O/p_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -o fetch -- python3 bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 1 > /dev/null 2> $O/p_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_write -o write -- python3 bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 1 > /dev/null 2> $O/p_write.err
python bench.py --workload c5 --steps 2 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_bgi -o bgi -- python3 bench.py --workload c5 --steps 2 --warmup 1 --no-secondary > /dev/null 2> $O/p_bgi.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_bgi_fetch -o fetch -- python3 bench.py --workload c5 --steps 1 --warmup 1 --no-secondary > /dev/null 2> $O/p_bgi_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_bgi_write -o write -- python3 bench.py --workload c5 --steps 1 --warmup 1 --no-secondary > /dev/null 2> $O/p_bgi_write.err
python bench.py --workload c2 --steps 3 > $O/bench_c2.json 2> $O/bench_c2.err
# Mode A (c2): kernel stats and HBM traffic of ALL launches of a solve (two solves per pass: one warm-up + one step; --no-secondary
# also skips the init-only probes, so every dispatch of the run belongs to one of the two solves)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_c2 -o c2 -- python3 bench.py --workload c2 --steps 1 --warmup 1 --no-secondary > /dev/null 2> $O/p_c2.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_c2_fetch -o fetch -- python3 bench.py --workload c2 --steps 1 --warmup 1 --no-secondary > /dev/null 2> $O/p_c2_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_c2_write -o write -- python3 bench.py --workload c2 --steps 1 --warmup 1 --no-secondary > /dev/null 2> $O/p_c2_write.err
python tools/time_dense.py 1024 2048 > $O/c2.log 2>&1
python tools/time_blockgi.py > $O/time_blockgi.log 2>&1
python tools/time_structured.py > $O/time_structured.log 2>&1
python tools/quick_time.py > $O/quick.log 2>&1
bash tools/pmc_sq.sh --no-secondary > $O/pmc_sq.log 2>&1   # SQ counters of the headline kernel: issue / wait split, LDS bank conflicts, instruction mix
python tools/bench_families.py 64 8 > $O/families.txt 2>&1
python tools/bench_warm_family.py 128 10 > $O/warm_family.txt 2>&1   # the reference's warm-start benchmark family (benchmarks/SolversWarmStart.cpp), n = 10 .. 100
cut -c1-250 $O/bench.json; cut -c1-250 $O/bench_c5.json
