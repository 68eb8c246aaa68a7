set -e
O=gpurun_out/r03c
mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_cyclic_reduction.py tests/test_gpu_parity.py tests/test_gpu_two_ranks.py -m gpu -x -q --timeout 200 --timeout-method thread > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for W in cfg3 cfg4; do
  A="--steps 5 --warmup 1"; [ $W = cfg4 ] && A="--steps 1 --warmup 0 --max-iters 100"
  timeout -k 10 200 python bench.py --workload $W --large none --no-cpu-baseline $A > $O/$W.json 2> $O/$W.err
  python tools/kt.py $O/$W.json
done
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py --workload cfg3 --large none --no-cpu-baseline --no-kernel-events --steps 3 --warmup 1 > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/cfg3_kernel_stats.csv
rm -rf $O/kt
cut -d, -f1-4 $O/cfg3_kernel_stats.csv | cut -c1-120
