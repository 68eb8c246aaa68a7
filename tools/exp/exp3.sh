set -e
for f in 1 0; do for ce in 1 2 3 5; do
  echo "== FUSE_DECIDE $f check_every $ce"
  BA_FUSE_DECIDE=$f timeout -k 10 120 python bench.py --workload cfg2 --large none --no-cpu-baseline --steps 20 --warmup 3 --lm-check-every $ce 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pk=d['roofline']['per_kernel']
print(d['ms_per_step'], d['final_cost'], d['lm_iterations'])"
done; done
