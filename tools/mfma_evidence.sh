# the fp64 MFMA evidence, tracked.  Run on the GPU box from the repo root; writes gpurun_out/r03m/.
#  1. the issue-rate microbenchmark (tools/mfma_f64_bench.hip, built beforehand into tools/bin/);
#  2. kernel durations of the two MFMA users: k_dense_update (dense Cholesky, 300 cameras with random visibility) and
#     k_bcr_level at m = 30 (998-camera ring: the equal-split layout, whose block products run on the matrix unit;
#     at m = 24 they run on the vector unit since round 3);
#  3. one PMC pass each with the MFMA / VALU counters.
set -e
export TMPDIR=/tmp
O=gpurun_out/r03m
rm -rf $O && mkdir -p $O
timeout -k 10 60 tools/bin/mfma_f64_bench > $O/mfma_f64_bench.txt
cat $O/mfma_f64_bench.txt
DENSE="tools/run_scene.py --C 300 --L 30000 --d 4 --visibility random --verbose 0"
BCR="tools/run_scene.py --C 998 --L 99800 --d 5 --verbose 0"
for W in dense bcr; do
  if [ $W = dense ]; then A="$DENSE"; else A="$BCR"; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kt_$W -o kt --output-format csv -- python $A > $O/kt_$W.log 2>&1
  cp $(find $O/kt_$W -name "*kernel_stats.csv" | head -1) $O/${W}_kernel_stats.csv
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES -d $O/pmc_$W -o p --output-format csv -- python $A > $O/pmc_$W.log 2>&1 || { tail -5 $O/pmc_$W.log; exit 1; }
  echo "$W done"
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for W in ("dense", "bcr"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r03m/pmc_{W}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0].split("<")[0].split()[-1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if not any(t in k for t in ("k_dense", "k_bcr")): continue
        out[f"{W}:{k}"] = {"launches": len(next(iter(v.values()))), **{c: sum(x) / len(x) for c, x in v.items()}}
json.dump(out, open("gpurun_out/r03m/mfma_pmc.json", "w"), indent=1)
for k, v in out.items(): print(k, {c: round(x, 1) for c, x in v.items()})
PY
rm -rf $O/kt_* $O/pmc_dense $O/pmc_bcr
