# Round-3 profiles (run on the GPU box from the repo root): bench line, rocprofv3 kernel statistics and the two PMC
# passes (FETCH_SIZE, WRITE_SIZE; a run each) for configs 2, 3, 4, and kernel statistics of a 1024-window batched run.
# Summaries land in gpurun_out/r03/; the ones to keep are copied into profiles/ by hand.
set -e
export TMPDIR=/tmp
O=gpurun_out/r03
rm -rf $O && mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench done" 
for W in cfg2 cfg3 cfg4; do
  case $W in cfg2) A="--steps 20 --warmup 3";; cfg3) A="--steps 3 --warmup 1";; cfg4) A="--steps 1 --warmup 0 --max-iters 60";; esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_$W -o kt --output-format csv -- python bench.py --workload $W --large none --no-cpu-baseline --no-like-for-like --no-kernel-events $A > $O/kt_$W.log 2>&1
  echo "kernel trace $W done"
  case $W in cfg2) A="--steps 3 --warmup 1";; cfg3) A="--steps 1 --warmup 0 --max-iters 12";; cfg4) A="--steps 1 --warmup 0 --max-iters 8";; esac
  timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$W -o f --output-format csv -- python bench.py --workload $W --large none --no-cpu-baseline --no-like-for-like --no-kernel-events $A > $O/fetch_$W.log 2>&1
  timeout -s KILL 280 rocprofv3 --pmc WRITE_SIZE -d $O/write_$W -o w --output-format csv -- python bench.py --workload $W --large none --no-cpu-baseline --no-like-for-like --no-kernel-events $A > $O/write_$W.log 2>&1
  python tools/pmc_traffic.py $O/fetch_$W $O/write_$W $O/pmc_traffic_$W.json > /dev/null
  echo "pmc $W done"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_batched -o kt --output-format csv -- python tools/run_batched.py --B 1024 --reps 3 > $O/kt_batched.log 2>&1
echo "batched done"
for W in cfg2 cfg3 cfg4 batched; do cp $(find $O/kt_$W -name "*kernel_stats.csv" | head -1) $O/${W}_kernel_stats.csv; done
rm -rf $O/kt_* $O/fetch_* $O/write_*  # the raw traces are large; the summaries stay
ls -la $O
