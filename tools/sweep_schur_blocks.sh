# BA_SCHUR_BLOCKS_PER_CU sweep (k_schur + k_gather_S per iteration) at configs 2 and 3; run on the GPU box
set -e
mkdir -p gpurun_out/sweep
for W in cfg2 cfg3; do
  for B in 4 6 8 10 12 16; do
    BA_SCHUR_BLOCKS_PER_CU=$B timeout -k 10 200 python bench.py --workload $W --large none --no-cpu-baseline --no-like-for-like --steps 10 --warmup 2 > gpurun_out/sweep/$W.$B.json
    python - <<PY
import json
d=json.loads(open("gpurun_out/sweep/$W.$B.json").read().strip().splitlines()[-1])
pk=d["roofline"]["per_kernel"]
print("$W blocks/CU $B: iteration %.1f us  schur %.1f  gather %.1f  its %d" % (d["roofline"]["whole_iteration"]["us"], pk["k_schur"]["avg_us"], pk["k_gather_S"]["avg_us"], d["lm_iterations"]))
PY
  done
done
