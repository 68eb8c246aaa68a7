set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_two_ranks.py tests/test_gpu_robust.py tests/test_gpu_batched.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread -m gpu 2>&1 | tail -n 5
for f in 1 0; do
  echo "== FUSE_DECIDE $f"
  BA_FUSE_DECIDE=$f timeout -k 10 120 python bench.py --workload cfg2 --large none --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d['roofline']['per_kernel']
print(d['ms_per_step'], d['final_cost'], {k:round(v['avg_us'],2) for k,v in pk.items()})"
done
