set -e
export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_stats -o bench -- python3 bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/p_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -o fetch -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2> $O/p_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_write -o write -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2> $O/p_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_bgi -o bgi -- python3 tools/time_blockgi.py c5b64 > $O/bgi_c5b64.log 2> $O/p_bgi.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_bgi_fetch -o fetch -- python3 tools/time_blockgi.py c5b64 > /dev/null 2> $O/p_bgi_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_bgi_write -o write -- python3 tools/time_blockgi.py c5b64 > /dev/null 2> $O/p_bgi_write.err
python tools/time_dense.py 1024 2048 > $O/c2.log 2>&1
find $O -name "*.csv" | head -30
cat $O/bench.json
