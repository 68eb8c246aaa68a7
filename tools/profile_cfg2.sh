set -e
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -20 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log

export TMPDIR=/tmp
rm -rf gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_cfg2.json
cat gpurun_out/bench_cfg2.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt --output-format csv -- python bench.py --steps 20 --warmup 3 --no-kernel-events > gpurun_out/prof_kt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o f --output-format csv -- python bench.py --steps 3 --warmup 1 --no-kernel-events > gpurun_out/prof_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -o w --output-format csv -- python bench.py --steps 3 --warmup 1 --no-kernel-events > gpurun_out/prof_write.log 2>&1
python tools/pmc_traffic.py gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/pmc_traffic.json
cat gpurun_out/pmc_traffic.json
find gpurun_out/prof_kt -name "*kernel_stats.csv" | head -2
