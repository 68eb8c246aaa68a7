# A/B of the round-3 linearisation (rows, no atomics) against round 2's kernels on the GPU box.
set -e
O=gpurun_out/r03b
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for W in cfg2 cfg3; do
  for R in 1 0; do
    BA_LIN_ROWS=$R timeout -k 10 200 python bench.py --workload $W --large none --no-cpu-baseline --steps 10 --warmup 2 > $O/${W}_rows$R.json 2> $O/${W}_rows$R.err
    python tools/kt.py $O/${W}_rows$R.json
  done
done
for R in 1 0; do
  BA_LIN_ROWS=$R timeout -k 10 200 python bench.py --workload cfg4 --large none --no-cpu-baseline --steps 1 --warmup 0 > $O/cfg4_rows$R.json 2> $O/cfg4_rows$R.err
  python tools/kt.py $O/cfg4_rows$R.json
done
# the same solve again: the iteration count and the cost must repeat bit for bit
timeout -k 10 200 python bench.py --workload cfg4 --large none --no-cpu-baseline --steps 1 --warmup 0 --no-kernel-events > $O/cfg4_again.json 2> $O/cfg4_again.err
python tools/kt.py $O/cfg4_again.json
