# A / B of an environment switch on the per-iteration time: bash tools/ab_env.sh VAR=VALUE [workloads...]
set -e
KV=$1; shift
mkdir -p gpurun_out/ab
for W in ${@:-cfg2 cfg3 cfg4}; do
  case $W in cfg4) A="--steps 1 --warmup 0 --max-iters 80";; cfg3) A="--steps 3 --warmup 1";; *) A="--steps 10 --warmup 2";; esac
  for MODE in base "$KV"; do
    if [ "$MODE" = base ]; then env -u ${KV%%=*} timeout -k 10 300 python bench.py --workload $W --large none --no-cpu-baseline --no-like-for-like $A > gpurun_out/ab/$W.base.json
    else env $KV timeout -k 10 300 python bench.py --workload $W --large none --no-cpu-baseline --no-like-for-like $A > gpurun_out/ab/$W.alt.json; fi
  done
  python - <<PY
import json
for m in ("base", "alt"):
    d=json.loads(open("gpurun_out/ab/$W.%s.json" % m).read().strip().splitlines()[-1])
    pk=d["roofline"]["per_kernel"]
    print("$W %-5s iteration %.1f us  its %d  cost %r  " % (m, d["roofline"]["whole_iteration"]["us"], d["lm_iterations"], d["final_cost"]) + " ".join("%s %.1f" % (k, v["avg_us"]) for k, v in pk.items()))
PY
done
