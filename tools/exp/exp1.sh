for b in 8 4 2; do
  echo "== BLOCKS_PER_CU $b"
  BA_SCHUR_BLOCKS_PER_CU=$b timeout -k 10 120 python bench.py --workload cfg2 --large none --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d['roofline']['per_kernel']
print(d['ms_per_step'], {k:round(v['avg_us'],2) for k,v in pk.items()})" || exit 1
done
